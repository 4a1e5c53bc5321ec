#!/bin/bash
# Build libshivae_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/libshivae_hip.so"
SRC="$HERE/csrc/gemm.hip $HERE/csrc/gemm_split.hip $HERE/csrc/loss.hip $HERE/csrc/scan.hip $HERE/csrc/optim.hip $HERE/csrc/plan.hip $HERE/csrc/prep.hip $HERE/csrc/heads.hip"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OBJ="$HERE/csrc/_obj"
mkdir -p "$OBJ"
pids=()
for f in $SRC; do
  o="$OBJ/$(basename "$f" .hip).o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/csrc/common.h" -nt "$o" ] || [ "$HERE/csrc/cell_math.h" -nt "$o" ] || [ "$HERE/csrc/noise.h" -nt "$o" ] \
     || [ "$HERE/../include/shivae_hip.h" -nt "$o" ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed \
        -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
