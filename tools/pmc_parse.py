#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: per-dispatch averages of each counter for the de_* kernels."""
import collections, csv, glob, os, sys
root = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "de_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")}
print(meta)
for k in sorted(agg):
    print(f"{k:28s} {sum(agg[k])/len(agg[k]):.4g}")
