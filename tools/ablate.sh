#!/bin/bash
# developer tool: builds variants of libpbrs_gpu.so for in-session A/B timing: tools/ablate.sh name:flags ...
cd "$(dirname "$0")/../pbrs_amd/csrc" || exit 1
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $flags -Rpass-analysis=kernel-resource-usage -o ../lib/abl_$name.so pbrs_gpu.hip 2> /tmp/abl_$name.log &
done
wait
for v in "$@"; do name=${v%%:*}; echo "== $name"; grep -E "Function Name|VGPRs:|ScratchSize" /tmp/abl_$name.log | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - | grep -E "shade|extendILb0|shadowILb0" | sed 's/Function Name: //'; done
