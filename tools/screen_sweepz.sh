#!/bin/bash
# usage: tools/screen_sweepz.sh MODE(1..4) IDRV(false|true) [-D flags] -> VGPRs / spills of the four k_sweepz<NQ, MODE, IDRV> kernels
mode=$1; idrv=$2; shift; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c --cuda-device-only -DSCREEN_MODE=$mode -DSCREEN_IDRV=$idrv "$@" \
    -Rpass-analysis=kernel-resource-usage exp/sweepz_regs.hip -o /tmp/sweepz_regs.o 2>&1 | grep -E "error|k_sweepz|VGPRs:|Spill|Scratch" | \
    sed 's/.*remark: //; s/\[-Rpass.*//' | grep -A4 "sweepz\|error" | paste - - - - - | cut -c1-200
