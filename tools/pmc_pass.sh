#!/bin/bash
# One rocprofv3 --pmc pass over bench.py (program directly after `--`), summed per kernel by tools/pmc_sum.py.
#   tools/pmc_pass.sh <COUNTER> <tag> [bench.py args...]      (run from the repo root on the GPU box)
set -e
CTR="$1"; TAG="$2"; shift 2
R="$PWD"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/pmc_$TAG"
timeout -s KILL 900 rocprofv3 --pmc "$CTR" --output-format csv -d "/tmp/pmc_$TAG" -o f -- python3 "$R/bench.py" "$@" \
    > "$R/gpurun_out/pmc_${TAG}_bench.json" 2> "$R/gpurun_out/pmc_${TAG}.err"
python3 "$R/tools/pmc_sum.py" "$(find "/tmp/pmc_$TAG" -name '*counter_collection.csv' | head -1)" "$R/gpurun_out/pmc_${TAG}.json" > /dev/null
echo "pmc pass $TAG ok"
