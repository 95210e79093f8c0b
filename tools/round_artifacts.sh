#!/bin/bash
# Runs on the GPU box (through gpurun): the round's measured artifacts under gpurun_out/artifacts_<tag>/, per config —
#   kernel_stats_<cfg>.csv           rocprofv3 --kernel-trace --stats of `python3 bench.py --config <cfg> --also ""` (for c4: the default
#                                    workload of `python3 bench.py` without its C2 / C3 extras, so that per-kernel averages are of ONE workload)
#   bench_under_rocprof_<cfg>.json   the JSON line of that same run
#   pmc_<set>_<cfg>.txt              per-kernel counter sums of separate --pmc passes (no trace domains besides --kernel-trace): L2 memory-side
#                                    bytes (1-4), the SQ's instruction and wave-cycle counts (5, 6), the L1's accesses and the busy cycles of
#                                    the texture address / data units (7, 8)
#   traffic_<cfg>.json               HBM bytes per launch per kernel (tools/traffic_from_pmc.py)
#   bench.json                       the plain default bench line (no profiler), last
#   PBRS_SKIP_PLAIN_BENCH=1 skips the closing plain bench (a call that collects one config's counters and nothing else)
# usage: tools/round_artifacts.sh TAG [cfg ...]      (default: c4 c2 c3).  Copy what should be judged into profiles/ afterwards.
set -o pipefail
tag=${1:-r02}; shift
cfgs=${@:-c4 c2 c3}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/artifacts_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in $cfgs; do
  extra=""; [ $cfg = c5 ] && extra="--strata 8 8"   # C5 rides along as a 64-spp slice of its 4096 spp (bench.py ALSO_STRATA): the same slice here
  echo "== $cfg: kernel trace"; date +%T
  rm -rf $out/trace_$cfg
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$cfg -o t -- python3 $R/bench.py --config $cfg $extra --also "" --steps 3 --warmup 1 --no-cpu-baseline > $out/trace_$cfg.log 2>&1 || { tail -5 $out/trace_$cfg.log; exit 1; }
  grep '^{"metric"' $out/trace_$cfg.log > $out/bench_under_rocprof_$cfg.json
  cp $(find $out/trace_$cfg -name '*kernel_stats.csv' | head -1) $out/kernel_stats_$cfg.csv
  rm -rf $out/trace_$cfg
  n=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
             "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
             "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"; do
    n=$((n+1)); name=$(echo $set | cut -d' ' -f1 | tr A-Z a-z)
    echo "== $cfg: pmc pass $n ($name)"; date +%T
    rm -rf $out/pmc_tmp
    timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_tmp -o p -- python3 $R/tools/profile_frame.py --config $cfg $extra > $out/pmc_${n}_$cfg.log 2>&1 || { echo "pass $n ($name) failed:"; tail -5 $out/pmc_${n}_$cfg.log; [ $n -le 2 ] && exit 1; continue; }
    python3 $R/tools/pmc_summary.py $out/pmc_tmp > $out/pmc_${n}_${name}_$cfg.txt
    grep '^{"config"' $out/pmc_${n}_$cfg.log > $out/geometry_$cfg.json
    rm -rf $out/pmc_tmp
  done
  python3 $R/tools/traffic_from_pmc.py $out/geometry_$cfg.json $out/pmc_1_fetch_size_$cfg.txt $out/pmc_2_write_size_$cfg.txt $out/pmc_3_tcc_ea0_rdreq_sum_$cfg.txt $out/pmc_4_tcc_ea0_wrreq_sum_$cfg.txt $out/pmc_5_sq_wave_cycles_$cfg.txt $out/pmc_7_tcp_total_cache_accesses_sum_$cfg.txt $out/pmc_8_ta_ta_busy_sum_$cfg.txt $out/pmc_6_sq_waves_$cfg.txt > $out/traffic_$cfg.json || exit 1
done
if [ -z "$PBRS_SKIP_PLAIN_BENCH" ]; then
  echo "== plain default bench"; date +%T
  cd $R && timeout -k 10 600 python3 bench.py > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/bench.json
fi
ls -la $out
