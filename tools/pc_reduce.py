#!/usr/bin/env python3
"""Reduces rocprofv3 PC-sampling output (csv) to counts per instruction; writes <out>/pc_counts_<name>.csv and a head of the raw
file. Column names differ between rocprofv3 versions, so the reduction keys on whatever of (Code_Object_Id, Code_Object_Offset,
Instruction, Instruction_Comment, Stall_Reason, Wave_Issued, Instruction_Type ...) is present."""
import collections, csv, glob, os, sys
out = sys.argv[-1]
for d in sys.argv[1:-1]:
    name = os.path.basename(d.rstrip("/"))
    for f in glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True):
        print("file", f, os.path.getsize(f))
        with open(f, newline="") as fh:
            rd = csv.reader(fh)
            hdr = next(rd)
            print("header", hdr)
            drop = ("exec_mask", "wave_in_group", "timestamp", "sample_timestamp", "dispatch_id", "correlation_id", "workgroup_id_x",
                    "workgroup_id_y", "workgroup_id_z", "chiplet", "wave_id", "hw_id", "wave_count", "sample_id")
            key_cols = [i for i, h in enumerate(hdr) if h.lower() not in drop]
            cnt = collections.Counter()
            n = 0
            head = []
            for row in rd:
                n += 1
                if n <= 200: head.append(row)
                cnt[tuple(row[i] for i in key_cols)] += 1
        tag = os.path.basename(f).replace(".csv", "")
        with open(os.path.join(out, f"raw_head_{name}_{tag}.csv"), "w", newline="") as fh:
            w = csv.writer(fh); w.writerow(hdr); w.writerows(head)
        with open(os.path.join(out, f"pc_counts_{name}_{tag}.csv"), "w", newline="") as fh:
            w = csv.writer(fh); w.writerow([hdr[i] for i in key_cols] + ["samples"])
            for k, v in cnt.most_common(200000):
                w.writerow(list(k) + [v])
        print("samples", n, "distinct", len(cnt))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        os.system(f"cp {f} {out}/kernel_trace_{name}.csv")
