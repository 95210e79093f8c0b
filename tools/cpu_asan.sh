#!/bin/bash
# CPU sanitizer pass (GPU sanitizers are not available on this pool and are not attempted): builds the host library and
# the oracle with -fsanitize=address,undefined and runs the CPU test suite against those builds.
#   tools/cpu_asan.sh [pytest args]         exit code = pytest's
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/oracle asan
make -C $R/pbrs_amd/csrc host-asan
export PBRS_ORACLE_LIB=$R/oracle/libpbrs_oracle_asan.so PBRS_HOST_LIB=$R/pbrs_amd/lib/libpbrs_host_asan.so
# python itself is not instrumented: preload the runtimes; leak checking would report the interpreter's own allocations
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd $R && exec python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
