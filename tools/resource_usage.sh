#!/bin/bash
# Register / scratch / occupancy table of every shipped kernel instantiation, straight from the compiler (hipcc -Rpass-analysis=kernel-resource-usage
# on each device translation unit with the library's own flags).   usage: tools/resource_usage.sh > profiles/rNN_resource_usage.txt
cd "$(dirname "$0")/../synthesis_amd/csrc" || exit 1
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math --cuda-device-only -S -o /dev/null -Rpass-analysis=kernel-resource-usage"
echo "# hipcc $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version.*')  flags: $FLAGS"
echo "# columns: VGPRs AGPRs SGPRs scratch[B/lane] occupancy[waves/SIMD] spilledSGPRs spilledVGPRs  kernel"
for tu in engine.hip engine_lanes_fast.hip engine_lanes_ref.hip engine_lanes_gen.hip engine_lanes_f16.hip engine_lanes_f16_gen.hip engine_conv.hip engine_free.hip engine_pool.hip engine_pool_f16.hip; do
  echo "## $tu"
  /opt/rocm/bin/hipcc $FLAGS $tu 2>&1 | awk '
    /Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
    /    VGPRs:/ {v=$(NF-1)} /    AGPRs:/ {a=$(NF-1)} /    SGPRs:/ {s=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)}
    /SGPRs Spill:/ {ss=$(NF-1)} /VGPRs Spill:/ {vs=$(NF-1)}
    /LDS Size/ {printf "%4s %4s %4s %6s %3s %4s %4s  %s\n", v,a,s,sc,o,ss,vs,name}' | while read -r line; do
      n=$(echo "$line" | awk '{print $NF}'); echo "$(echo "$line" | awk '{$NF=""; print}') $(echo "$n" | /opt/rocm/lib/llvm/bin/llvm-cxxfilt | sed 's/(syn::EngineParams)//')"; done
done
