#!/bin/bash
# usage: tools/screen_sweep.sh MODE [-DRRLW_... flags]  -> VGPRs / spills / occupancy of the four k_sweep<MODE, NQ, false> kernels
mode=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c --cuda-device-only -DSCREEN_MODE=$mode "$@" \
    -Rpass-analysis=kernel-resource-usage exp/sweep_regs.hip -o /tmp/sweep_regs.o 2>&1 | \
    grep -E "Function Name|VGPRs:|Spill|Occupancy|ScratchSize|LDS Size" | sed 's/.*remark: //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
