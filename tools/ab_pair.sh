#!/bin/bash
# developer tool (GPU box): full-size bench lines of library variants built with -DPBRS_DEV_OVERRIDES under PBRS_PAIR / PBRS_WIDE
# settings (bit 0 k_extend, bit 1 k_shadow), parity window on.   usage: CFGS="c4" tools/ab_pair.sh lib "PBRS_GRID=3" "PBRS_CNODE=3" "PBRS_PAIR=3" "PBRS_WIDE=2" ...
cfgs=${CFGS:-c4}
lib=$1; shift
for c in $cfgs; do for v in "$@"; do
  line=$(env $v PBRS_GPU_LIB=$PWD/pbrs_amd/lib/abl_$lib.so timeout -k 10 300 python bench.py --config $c --also "" --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>&1 | grep '^{"metric"')
  echo "$c $lib [$v] $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Msamples/s" % d["value"], "parity", d.get("parity_window",{}).get("bit_exact"), {k: round(v,1) for k,v in d["stages_ms_per_step"].items() if k.startswith("ms_")})' 2>&1 | tail -1)"
done; done
