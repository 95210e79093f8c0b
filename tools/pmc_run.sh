#!/bin/bash
# developer tool (GPU box): one rocprofv3 --pmc pass per argument (a quoted counter list), summarised per kernel
# usage: tools/pmc_run.sh tag "SQ_WAVES SQ_INSTS_VALU ..." "..."
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
n=0
for set in "$@"; do
  n=$((n+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n
  rm -rf $out
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --strata 4 4 --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_EXTRA} > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_${n}_summary.txt
done
