#!/bin/bash
# developer tool (GPU box): the whole GPU parity suite through the experimental walks of a developer build
# (pbrs_amd/lib/abl_<name>.so built with -DPBRS_DEV_OVERRIDES: tools/ablate.sh "dev:-DPBRS_DEV_OVERRIDES"), one run per setting.
#   usage: tools/dev_parity.sh dev "PBRS_WIDE=3" "PBRS_WIDE=1"      (round 4: the four-wide closest walk of device/experimental/ is the one developer walk left)
lib=$1; shift
for v in "$@"; do
  echo "== $v"
  env $v PBRS_GPU_LIB=$PWD/pbrs_amd/lib/abl_$lib.so timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
done
