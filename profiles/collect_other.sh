#!/bin/bash
# rocprofv3 kernel stats of the merge and VCF device kernels (profiles/other_paths.py)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_other -o other --output-format csv -- python3 $R/profiles/other_paths.py > $R/gpurun_out/prof_other.log 2>&1
echo "rc=$?"
tail -4 $R/gpurun_out/prof_other.log
