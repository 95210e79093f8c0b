#!/bin/bash
# developer tool (GPU box): A/B of library variants built by tools/ablate.sh (pbrs_amd/lib/abl_<name>.so), one bench run each, on ONE box.
#   usage: tools/ab_libs.sh "<bench args>" name ...      ("base" = the shipped library).  Prints value and stage times per variant.
args=$1; shift
for v in "$@"; do
  lib=$PWD/pbrs_amd/lib/abl_$v.so; [ $v = base ] && lib=$PWD/pbrs_amd/lib/libpbrs_gpu.so
  PBRS_GPU_LIB=$lib timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-parity-window > gpurun_out/ab_$v.log 2>&1 || { echo "== $v FAILED"; tail -3 gpurun_out/ab_$v.log; exit 1; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
for l in open(f"gpurun_out/ab_{v}.log"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        rows = [(d["config"]["scene"], d)] + list(d.get("other_configs", {}).items())
        print("== %-8s" % v, " | ".join("%s %.1f (x %.1f s %.1f sh %.1f)" % (n, r["value"], r["stages_ms_per_step"]["ms_extend"], r["stages_ms_per_step"]["ms_shade"],
                                                                        r["stages_ms_per_step"]["ms_shadow"]) for n, r in rows), flush=True)
PY
done
