cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_c5
rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $out/raw -o p -- python3 $GRAFT_REPO_ROOT/tools/profile_frame.py --config c5 --strata 8 8 > $out/log.txt 2>&1 || tail -5 $out/log.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/raw > $out/summary.txt; rm -rf $out/raw; head -40 $out/summary.txt; grep '^{"config"' $out/log.txt
