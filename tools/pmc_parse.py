#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: per-launch averages of each counter over the FULL-SIZE launches of the dominant
de_* kernel (the one-sample constant-folding launch of the same kernel and the handler-table kernels are excluded by
grid size).  usage: pmc_parse.py <dir> [out.txt]"""
import collections, csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    rows += [r for r in csv.DictReader(open(f)) if "de_" in r["Kernel_Name"]]
gmax = max(int(r["Grid_Size"]) for r in rows)
rows = [r for r in rows if int(r["Grid_Size"]) == gmax]
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
meta = {k: rows[0][k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")}
lines = [rows[0]["Kernel_Name"][:150], str(meta)] + [f"{k:28s} {sum(v) / len(v):.4g}   ({len(v)} launches)" for k, v in sorted(agg.items())]
def ratio(a, b):
    return sum(agg[a]) / len(agg[a]) / (sum(agg[b]) / len(agg[b])) if agg.get(a) and agg.get(b) else float("nan")
lines += [f"SALU/VALU instructions        {ratio('SQ_INSTS_SALU', 'SQ_INSTS_VALU'):.3f}",
          f"VALU active / busy cycles     {ratio('SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES'):.3f}  (of 8: 4 SIMDs x 2, see DESIGN.md)",
          f"scalar active / busy cycles   {ratio('SQ_ACTIVE_INST_SCA', 'SQ_BUSY_CYCLES'):.3f}"]
print("\n".join(lines))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
