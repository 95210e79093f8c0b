#!/bin/bash
# developer tool (runs on the GPU box): times each pbrs_amd/lib/abl_<name>.so given on the command line
cfgs=${CFGS:-c2}
for v in "$@"; do for c in $cfgs; do echo "$v $c $(PBRS_GPU_LIB=$PWD/pbrs_amd/lib/abl_$v.so timeout -k 10 200 python bench.py --config $c --strata 4 4 --steps 2 --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>&1 | grep -o '"stages_ms_per_step": {[^}]*}')"; done; done
