#!/bin/bash
export NAGP_DEVELOPER=1      # developer tool: libnagp.so reads its switches only with this set
# Register / spill report of one instantiation unit:  tools/kernel_regs.sh inst_gf_rest [filter-regex]
cd "$(dirname "$0")/../nonstationary-audio-gp_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -c "$1.hip" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Name:|VGPRs:|AGPRs:|Spill|ScratchSize" | sed 's/.*remark: //; s/\[-Rpass.*//; s/^[^ ]*: *//' | paste - - - - - - |
  sed 's/Function Name: _ZN4nagpL\?[0-9]*//; s/EvNS_.*E\t/ \t/; s/  */ /g' | grep -E "${2:-.}"
rm -f /tmp/kr_$$.o
