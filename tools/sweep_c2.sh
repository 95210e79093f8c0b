#!/bin/bash
# developer tool (GPU box): C2 extend/shade/shadow time against depth and samples-per-pass
for spp in 4 16; do for depth in 1 2 3 4 8; do
  echo "spp/pass=$spp depth=$depth $(timeout -k 10 200 python bench.py --config c2 --strata 4 4 --depth $depth --samples-per-pass $spp --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"rays_per_step": [0-9.]*\|"stages_ms_per_step": {[^}]*}' | tr '\n' ' ')"
done; done
