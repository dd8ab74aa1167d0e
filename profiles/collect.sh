#!/bin/bash
# rocprofv3 evidence for the default bench workload: kernel stats, then FETCH_SIZE and WRITE_SIZE in passes of their own
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v5 -o v5 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-baseline-mb 0 > $R/gpurun_out/prof_v5.log 2>&1
echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_v5_fetch -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-mb 0 > $R/gpurun_out/pmc_v5_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_v5_write -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline-mb 0 > $R/gpurun_out/pmc_v5_write.log 2>&1
echo "write rc=$?"
tail -1 $R/gpurun_out/prof_v5.log | cut -c1-300
