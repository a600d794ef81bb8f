#!/bin/bash
# L2 (TCC) hit / miss counts per kernel of one DiT score call at C2 (separate --pmc passes, no other tracing);
# summarise with scripts/pmc_l2_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_l2; mkdir -p $out
export PART=score SCORE=dit
for c in TCC_HIT_sum TCC_MISS_sum; do
  echo "== $c" >> $out/log.txt
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out/$c -o score -- python3 scripts/pmc_workload.py >> $out/log.txt 2>&1 || exit 1
done
echo done
