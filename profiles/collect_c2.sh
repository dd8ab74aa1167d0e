#!/bin/bash
# BASELINE configs[1] (64 rows x 10 Mb): the bench line and rocprofv3 --kernel-trace --stats of the same command.
# usage (on the GPU box): profiles/collect_c2.sh <tag>   -> gpurun_out/bench_<tag>_c2.json, gpurun_out/prof_<tag>_c2/
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload c2 --steps 20 --warmup 3 > $R/gpurun_out/bench_${TAG}_c2.json 2> $R/gpurun_out/bench_${TAG}_c2.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_c2 -o c2 --output-format csv -- python3 $R/bench.py --workload c2 --steps 20 --warmup 3 --cpu-baseline-mb 0 --verify 0 > $R/gpurun_out/prof_${TAG}_c2.log 2>&1
echo "c2 stats rc=$?"
cut -c1-200 $R/gpurun_out/bench_${TAG}_c2.json
