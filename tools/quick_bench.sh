#!/bin/bash
# developer helper: GPU parity tests, then 16-spp bench lines for the given configs (default c2 c4)
set -o pipefail
tag=${1:-q}; shift
cfgs=${@:-c2 c4}
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
for c in $cfgs; do
  timeout -k 10 300 python bench.py --config $c --strata 4 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${tag}_$c.log 2>&1
  echo $c; grep -o "\"value\": [0-9.]*\|\"mrays_per_s\": [0-9.]*\|\"stages_ms_per_step\": {[^}]*}" gpurun_out/bench_${tag}_$c.log || tail -5 gpurun_out/bench_${tag}_$c.log
done
