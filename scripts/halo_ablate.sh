#!/bin/bash
# development: rocprof kernel stats of the NCSN++ trace workload with the ablation builds of igemm.hip
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h3
for m in ${MODES:-0 1 2 3}; do
  if [ $m = 0 ]; then unset DSN_LIB; else export DSN_LIB=$GRAFT_REPO_ROOT/ditsep_amd/libdbg$m.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h3/m$m -o n -- python3 scripts/ncsn_trace_workload.py > gpurun_out/h3/m$m.log 2>&1 || exit 1
done
echo done
