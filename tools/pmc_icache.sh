#!/bin/bash
# developer tool (GPU box): instruction-cache counters per kernel for one config.  usage: tools/pmc_icache.sh cfg [tag]
cfg=${1:-c3}; tag=${2:-x}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache_${cfg}_$tag
rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/raw -o p -- python3 $GRAFT_REPO_ROOT/tools/profile_frame.py --config $cfg --strata 8 8 > $out/log.txt 2>&1 || tail -5 $out/log.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/raw > $out/summary.txt; rm -rf $out/raw; grep -A9 "k_shade\|k_extend<false" $out/summary.txt | head -40
