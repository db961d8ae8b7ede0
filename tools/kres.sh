#!/bin/bash
# tools/kres.sh <pattern> [-D flags...]: registers / spills / scratch of the kernels in spdy_kernels.hip whose mangled name matches
pat=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" \
  -c $(dirname $0)/../speedy.f90_amd/csrc/spdy_kernels.hip -o /dev/null 2>&1 \
 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|SGPRs:" | sed -e 's/.*remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste - - - - - | grep -E "$pat" | sed 's/  */ /g'
