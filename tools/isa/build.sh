#!/bin/bash
# Developer tool: ISA + resource usage of the headline kernel only (same flags as synthesis_amd/csrc/Makefile)
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -Wno-unused-function -Wno-unused-value -S --cuda-device-only -o pc_only.s pc_only.hip -Rpass-analysis=kernel-resource-usage 2> pc_only.txt
grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|SGPRs:" pc_only.txt
