#!/bin/bash
# rocprofv3 --kernel-trace --stats of the eager per-kernel workload (scripts/pmc_workload.py): DiT score calls + decode,
# and the NCSN++ score calls; CSV summaries under gpurun_out/trace/ (copy the *_kernel_stats.csv into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace; mkdir -p $out
PART=all SCORE=dit timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dit -o dit -- python3 scripts/pmc_workload.py > $out/dit.log 2>&1 || exit 1
PART=score SCORE=ncsnpp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ncsn -o ncsn -- python3 scripts/pmc_workload.py > $out/ncsn.log 2>&1 || exit 1
find $out -name "*kernel_stats.csv"
echo done
