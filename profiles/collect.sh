#!/bin/bash
# rocprofv3 evidence for the default bench workload: kernel stats, then FETCH_SIZE, WRITE_SIZE and the SQ wave counters in passes of their own.
# usage (on the GPU box): profiles/collect.sh <tag>      -> gpurun_out/prof_<tag>, pmc_<tag>_fetch, pmc_<tag>_write
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-baseline-mb 0 --verify 0 --aligned-probe 0 > $R/gpurun_out/prof_$TAG.log 2>&1
echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_fetch -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-mb 0 --verify 0 --aligned-probe 0 > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_write -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-mb 0 --verify 0 --aligned-probe 0 > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_${TAG}_sq -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-mb 0 --verify 0 --aligned-probe 0 > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
echo "sq rc=$?"
tail -1 $R/gpurun_out/prof_$TAG.log | cut -c1-300
