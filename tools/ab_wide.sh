#!/bin/bash
# developer tool (GPU box): full-size bench lines of library variants built with -DPBRS_DEV_OVERRIDES, with and without the
# wide-walk kernels (PBRS_WIDE=0 selects the binary-walk kernels of the same binary).  usage: CFGS="c4 c2" tools/ab_wide.sh w5 w4
cfgs=${CFGS:-c4 c2 c3}
run() {  # label, env, lib
  line=$(env $2 PBRS_GPU_LIB=$PWD/pbrs_amd/lib/abl_$3.so timeout -k 10 300 python bench.py --config $c --also "" --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-parity-window $BENCH_EXTRA 2>&1 | grep '^{"metric"')
  echo "$c $1 $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Msamples/s %.0f Mrays/s" % (d["value"], d["mrays_per_s"]), {k: round(v,1) for k,v in d["stages_ms_per_step"].items() if k.startswith("ms_")})' 2>&1 | tail -1)"
}
for c in $cfgs; do
  run "binary($1)" "PBRS_WIDE=0" $1
  for v in "$@"; do run "wide($v)" "PBRS_WIDE=3" $v; done
done
