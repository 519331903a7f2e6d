#!/bin/bash
# usage: tools/screen_sweepc.sh PHASE IDRV(false|true) [-D flags] -> VGPRs / spills / occupancy of k_sweepc<NQ, PHASE, IDRV>
ph=$1; idrv=$2; shift; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c --cuda-device-only -DSCREEN_PHASE=$ph -DSCREEN_IDRV=$idrv "$@" \
    -Rpass-analysis=kernel-resource-usage exp/sweepc_regs.hip -o /tmp/sweepc_regs.o 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|Occupancy|ScratchSize" | \
    sed 's/.*remark: //' | paste - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/Function Name: _ZN4rrlw8k_sweepc//; s/EEEvNS_9DevTablesENS_9WorkspaceENS_9SweepArgsE//' | grep -E "^ILi|error"
