#!/bin/bash
# developer tool (GPU box): short rocprofv3 --pmc passes (one per quoted counter list) over one 16-spp frame of a config, summarised per
# kernel into gpurun_out/pmc_<tag>_<n>_summary.txt.   usage: CFG=c4 tools/pmc_quick.sh tag "CTR_A CTR_B" "CTR_C" ...
tag=$1; shift
cfg=${CFG:-c4}
cd /tmp && export TMPDIR=/tmp
n=0
for set in "$@"; do
  n=$((n+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n
  rm -rf $out
  timeout -k 5 ${PASS_TIMEOUT:-100} rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/tools/profile_frame.py --config $cfg --strata 4 4 --no-warm > $out.log 2>&1 || { echo "pass $n failed"; grep -m2 -i 'error\|exceeds' $out.log; continue; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_${n}_summary.txt
  rm -rf $out
done
