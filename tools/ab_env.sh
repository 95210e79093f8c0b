#!/bin/bash
# developer tool (GPU box): A/B of developer overrides (environment variables a -DPBRS_DEV_OVERRIDES build reads) on ONE box.
#   usage: tools/ab_env.sh LIBNAME "<bench args>" "VAR=val [VAR=val]" ...     (LIBNAME: pbrs_amd/lib/abl_<LIBNAME>.so; "" = no override)
lib=$PWD/pbrs_amd/lib/abl_$1.so; args=$2; shift 2
for v in "$@"; do
  env $v PBRS_GPU_LIB=$lib timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-parity-window > gpurun_out/abenv.log 2>&1 || { echo "== $v FAILED"; tail -3 gpurun_out/abenv.log; exit 1; }
  python - "$v" <<'PY'
import json, sys
for l in open("gpurun_out/abenv.log"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        rows = [(d["config"]["scene"], d)] + list(d.get("other_configs", {}).items())
        print("== %-28s" % sys.argv[1], " | ".join("%s %.1f (%.1f ms)" % (n, r["value"], r["ms_per_step"]) for n, r in rows), flush=True)
PY
done
