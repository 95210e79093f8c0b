#!/bin/bash
# developer tool (GPU box): the whole GPU parity suite through the experimental walks of a developer build
# (pbrs_amd/lib/abl_<name>.so built with -DPBRS_DEV_OVERRIDES: tools/ablate.sh "dev:-DPBRS_DEV_OVERRIDES"), one run per setting.
#   usage: tools/dev_parity.sh dev "PBRS_WIDE=3" "PBRS_WIDE=1"      (round 4: the four-wide closest walk of device/experimental/ is the one developer walk left)
# (Scenes with a ParallelQuad next to a mesh render through the exact-extent walk in every build; the ray harness of a developer build
#  still sends their closest-hit queries through the four-wide walk when PBRS_WIDE has bit 0 set, which does not follow the extent:
#  tests/test_gpu_fuzz.py::test_a_raised_extent_reaches_a_mirrored_quad_hit checks its ray only where the harness took the binary walk.)
# (PBRS_WIDE without bit 1 switches k_shadow's four-wide walk off: the tests that assert which walk the PRODUCT takes — wide_any, feature bit 4 —
#  fail by design under it; profiles/r04z_dev_parity.log.)
lib=$1; shift
for v in "$@"; do
  echo "== $v"
  env $v PBRS_GPU_LIB=$PWD/pbrs_amd/lib/abl_$lib.so timeout -k 10 400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "^FAILED|passed|failed"
done
