#!/bin/bash
# Runs on the GPU box (through gpurun): produces the round's measured artifacts under gpurun_out/artifacts_<tag>/ —
#   kernel_stats.csv            rocprofv3 --kernel-trace --stats of the default `python3 bench.py` command
#   bench_under_rocprof.json    the JSON line of that same run
#   pmc_fetch_size.txt / pmc_write_size.txt   per-kernel FETCH_SIZE / WRITE_SIZE (separate --pmc passes, no trace domains)
#   traffic.json                HBM bytes per launch per kernel (tools/traffic_from_pmc.py)
#   bench.json                  the plain default bench line (no profiler)
# Copy what should be judged into profiles/ afterwards.
set -o pipefail
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/artifacts_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $R/bench.py --no-cpu-baseline > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
grep '^{"metric"' $out/trace.log > $out/bench_under_rocprof.json
cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
  python3 $R/tools/pmc_summary.py $out/pmc_$c > $out/pmc_$(echo $c | tr A-Z a-z).txt
done
# default C2 bench: 1024x1024 pixels, 128 samples per pass
python3 $R/tools/traffic_from_pmc.py $out/pmc_fetch_size.txt $out/pmc_write_size.txt $((1024*1024*128)) $((1024*1024)) > $out/traffic.json || exit 1
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
cd $R && timeout -k 10 600 python3 bench.py > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/bench.json
ls -la $out
