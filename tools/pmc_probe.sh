#!/bin/bash
# Developer tool (GPU box): SQ counter passes over one quad_sweep configuration, e.g. tools/pmc_probe.sh 12288:3
set -u
CFG=${1:-12288:3}
PASSES=${2:-"P1 P2 P3 P4"}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
P3="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"
P4="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"
i=0
for P in $PASSES; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc ${!P} --output-format csv -d $R/gpurun_out/pmc/p$i -- python3 tools/quad_sweep.py $CFG > $R/gpurun_out/pmc/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "selfplay" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"] + ":" + r["Kernel_Name"][:60] + " grid=" + r["Grid_Size"]][r["Counter_Name"]] += float(r["Counter_Value"])
# print the largest dispatch of each pass merged by counter name (dispatch ids differ per pass: merge by grid)
merged = collections.defaultdict(dict)
for k, v in acc.items():
    g = k.split("grid=")[1]
    for n, x in v.items():
        merged[g][n] = max(merged[g].get(n, 0), x)
for g, v in merged.items():
    print("grid", g)
    for n in sorted(v):
        print(f"  {n:28s} {v[n]:.4g}")
PY
