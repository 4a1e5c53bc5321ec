#!/bin/bash
# rocprofv3 --pmc passes over one GEMM shape (tools/gemm_bench.py one <layout> <M> <N> <K>), program directly after `--`.
#   tools/pmc_gemm.sh <tag> "<counters>" <layout> <M> <N> <K>      (run from the repo root on the GPU box)
set -e
TAG="$1"; CTR="$2"; shift 2
R="$PWD"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/pg_$TAG"
timeout -s KILL 90 rocprofv3 --pmc $CTR --output-format csv -d "/tmp/pg_$TAG" -o f -- python3 "$R/tools/gemm_bench.py" one "$@" \
    > "$R/gpurun_out/pg_${TAG}.out" 2> "$R/gpurun_out/pg_${TAG}.err"
python3 "$R/tools/pmc_sum.py" "$(find "/tmp/pg_$TAG" -name '*counter_collection.csv' | head -1)" "$R/gpurun_out/pg_${TAG}.json" > /dev/null
echo "pmc gemm pass $TAG ok"
