#!/bin/bash
# Collect the HBM-traffic counters of one DiT score call, one NCSN++ score call and one decode (separate --pmc passes,
# no other tracing), into gpurun_out/pmc/<part>_<counter>/ ; summarise with scripts/pmc_summary.py afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc; mkdir -p $out
run() {  # name counter
  echo "== $1 $2" >> $out/log.txt
  timeout -k 10 400 rocprofv3 --pmc $2 --output-format csv -d $out/$1_$2 -o $1 -- python3 scripts/pmc_workload.py >> $out/log.txt 2>&1 || exit 1
}
export PART=score SCORE=dit;    run score FETCH_SIZE && run score WRITE_SIZE || exit 1
export PART=decode SCORE=dit;   run decode FETCH_SIZE && run decode WRITE_SIZE || exit 1
export PART=score SCORE=ncsnpp; run ncsn FETCH_SIZE && run ncsn WRITE_SIZE || exit 1
echo done
