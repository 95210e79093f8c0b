#!/bin/bash
# developer tool (GPU box): full-size bench lines of the built library under different environments, one after the other on the
# same GPU.   usage: CFGS="c3 c5" tools/ab_env.sh "PBRS_SHADE_SPEC=7" ""
cfgs=${CFGS:-c4 c2 c3}
for c in $cfgs; do for v in "$@"; do
  line=$(env $v timeout -k 10 300 python bench.py --config $c --also "" --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>&1 | grep '^{"metric"')
  echo "$c [$v] $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Msamples/s %.0f Mrays/s" % (d["value"], d["mrays_per_s"]), {k: round(v,1) for k,v in d["stages_ms_per_step"].items() if k.startswith("ms_")})' 2>&1 | tail -1)"
done; done
