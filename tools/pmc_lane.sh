#!/bin/bash
# Developer tool (GPU box): counter passes over one lane_sweep configuration, e.g. tools/pmc_lane.sh 131072:8 "P1 P2 M1 M2"
set -u
CFG=${1:-131072:8}
PASSES=${2:-"P1 P2 M1 M2 M3"}
R=$PWD
export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcl; mkdir -p $R/gpurun_out/pmcl
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
P3="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"
M1="FETCH_SIZE"
M2="WRITE_SIZE"
M3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
M4="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum"
T1="TA_TA_BUSY_sum TA_BUSY_avr"
T2="TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
T3="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
T4="TD_TD_BUSY_sum TD_TC_STALL_sum"
T5="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
T6="TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
T7="GRBM_GUI_ACTIVE"
M5="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
for P in $PASSES; do
  timeout -k 5 150 rocprofv3 --pmc ${!P} --output-format csv -d $R/gpurun_out/pmcl/$P -- python3 tools/lane_sweep.py $CFG > $R/gpurun_out/pmcl/$P.log 2>&1
  tail -2 $R/gpurun_out/pmcl/$P.log
done
python3 - <<'PY'
import csv, glob, collections
merged = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmcl/*/*/*_counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "selfplay" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"] + " grid=" + r["Grid_Size"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        g = k.split("grid=")[1]
        for n, x in v.items():
            merged[g][n] = max(merged[g].get(n, 0), x)
for g, v in merged.items():
    print("grid", g)
    for n in sorted(v):
        print(f"  {n:34s} {v[n]:.5g}")
PY
