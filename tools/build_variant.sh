#!/bin/bash
# Experimental build of the library with extra flags for ONE source, beside the product library:
#   tools/build_variant.sh <name> <source.hip> [flags...]   ->  shi-vae_amd/_exp/lib_<name>.so  (use: SHIVAE_LIB_PATH=...)
set -e
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; SRC="$2"; shift 2
mkdir -p "$R/shi-vae_amd/_exp"
O="$R/shi-vae_amd/_exp/${NAME}_$(basename "$SRC" .hip).o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed \
    "$@" -c "$R/shi-vae_amd/csrc/$SRC" -o "$O"
OBJS=""
for f in "$R"/shi-vae_amd/csrc/_obj/*.o; do
  [ "$(basename "$f")" = "$(basename "$SRC" .hip).o" ] || OBJS="$OBJS $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/shi-vae_amd/_exp/lib_${NAME}.so" $OBJS "$O"
rm -f "$O"
echo "built shi-vae_amd/_exp/lib_${NAME}.so"
