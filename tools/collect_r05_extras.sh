#!/bin/bash
# Runs on the GPU box (via gpurun): the round's small measurements beside tools/collect_profiles.sh — phase stamps of the shapes the
# notes quote, stand-alone kernel rooflines, the f16 split program — into gpurun_out/r05_extras/ for copying into profiles/.
OUT=gpurun_out/r05_extras; mkdir -p $OUT
export SYN_DEBUG=1
{
echo "== lane-per-tree kernel, 12 waves x 768 trees per CU, 393,216 games, f32 / f16x2, random-init / trained (SYN_PROFILE=1: instrumented build)"
for a in "" f16x2; do for w in "" trained; do
  echo "-- arithmetic ${a:-f32}, network ${w:-random-init}"
  SYN_PROFILE=1 python3 tools/lane_sweep.py $a $w 196608:12:393216 2>&1 | grep -E "profile lanes|games/s" | tail -3 | cut -c1-1700
done; done
echo "== 4,096 concurrent games, 16,384 games: f32 (selfplay_kernel<WPS=1>) and f16x2 (selfplay_kernel_free)"
SYN_PROFILE=1 python3 tools/run4096.py 2>&1 | grep -E "profile|games/s" | tail -6 | cut -c1-900
echo "== the same, production kernels"
python3 tools/run4096.py 2>&1 | grep -E "games/s"
} > $OUT/r05_phase_stamps.txt 2>&1
unset SYN_DEBUG
python3 tools/small_kernel_rooflines.py > $OUT/small_kernels.log 2>&1; cp gpurun_out/small_kernels.json $OUT/r05_small_kernels.json
tools/ubench/mfma_f16_split tests/golden gpurun_out/f16split > $OUT/r05_f16_split.txt 2>&1
ls -la $OUT
