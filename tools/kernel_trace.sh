#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py (program directly after `--`); the per-kernel summary goes to
# gpurun_out/<tag>_kernel_stats.csv, the bench line to gpurun_out/<tag>_bench.json.
#   tools/kernel_trace.sh <tag> [bench.py args...]      (run from the repo root on the GPU box)
set -e
TAG="$1"; shift
R="$PWD"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/kt_$TAG"
timeout -s KILL 1000 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/kt_$TAG" -o f -- python3 "$R/bench.py" "$@" \
    > "$R/gpurun_out/${TAG}_bench.json" 2> "$R/gpurun_out/${TAG}.err"
cp "$(find "/tmp/kt_$TAG" -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/${TAG}_kernel_stats.csv"
echo "kernel trace $TAG ok"
