#!/bin/bash
# developer tool (GPU box): full-size bench lines of the given library variants (pbrs_amd/lib/abl_<name>.so; "new" = the built
# libpbrs_gpu.so) on the given configs, one after the other on the same GPU.   usage: CFGS="c4 c2 c3" tools/ab_full.sh base new
cfgs=${CFGS:-c4 c2 c3}
for c in $cfgs; do for v in "$@"; do
  lib=$PWD/pbrs_amd/lib/abl_$v.so; [ "$v" = new ] && lib=$PWD/pbrs_amd/lib/libpbrs_gpu.so
  line=$(PBRS_GPU_LIB=$lib timeout -k 10 300 python bench.py --config $c --also "" --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>&1 | grep '^{"metric"')
  echo "$c $v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Msamples/s %.0f Mrays/s" % (d["value"], d["mrays_per_s"]), {k: round(v,1) for k,v in d["stages_ms_per_step"].items() if k.startswith("ms_")})' 2>&1 | tail -1)"
done; done
