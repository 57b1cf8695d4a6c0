#!/bin/bash
# Runs on the GPU box (via gpurun): round 6's small measurements beside tools/collect_profiles.sh — phase stamps of the lane-per-tree
# kernel on the four (arithmetic, network) pairs AND on the reference's own self-play configuration (family 2: trained checkpoint +
# PolicyWithCache + Fpu::Func), the 4,096-game kernels' stamps, the stand-alone kernel rooflines — into gpurun_out/r06_extras/.
OUT=gpurun_out/r06_extras; mkdir -p $OUT
export SYN_DEBUG=1
{
echo "== lane-per-tree kernel, 12 waves x 768 trees per CU, 393,216 games, f32 / f16x2, random-init / trained (SYN_PROFILE=1: instrumented build)"
for a in "" f16x2; do for w in "" trained; do
  echo "-- arithmetic ${a:-f32}, network ${w:-random-init}"
  SYN_PROFILE=1 python3 tools/lane_sweep.py $a $w 196608:12:393216 2>&1 | grep -E "profile lanes|games/s" | tail -3 | cut -c1-1700
done; done
echo "== the reference's own self-play configuration (configuration family 2: trained checkpoint, PolicyWithCache 2^28, Fpu::Func(Normal(1.0, 0.1))), f32, 12 waves"
SYN_PROFILE=1 python3 tools/lane_sweep.py reference 196608:12:393216:28 2>&1 | grep -E "profile lanes|games/s" | tail -3 | cut -c1-1700
echo "== 4,096 concurrent games, 16,384 games: f32 (selfplay_kernel<WPS=1>) and f16x2 (selfplay_kernel_free)"
SYN_PROFILE=1 python3 tools/run4096.py 2>&1 | grep -E "profile|games/s" | tail -6 | cut -c1-900
echo "== the same, production kernels"
python3 tools/run4096.py 2>&1 | grep -E "games/s"
} > $OUT/r06_phase_stamps.txt 2>&1
unset SYN_DEBUG
python3 tools/small_kernel_rooflines.py > $OUT/small_kernels.log 2>&1; cp gpurun_out/small_kernels.json $OUT/r06_small_kernels.json
ls -la $OUT
