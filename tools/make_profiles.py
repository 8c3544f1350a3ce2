#!/usr/bin/env python3
"""Condense gpurun_out/profiles_<tag>/ (tools/profile_round.sh) into the committed profiles/ files.
usage: python tools/make_profiles.py <tag> <round-prefix> [bench dir under gpurun_out]   e.g.  r2 r2 r2g"""
import csv, json, os, shutil, sys
tag, pre = sys.argv[1], sys.argv[2]
src = f"gpurun_out/profiles_{tag}"
os.makedirs("profiles", exist_ok=True)

def kstats(path, out):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader()
        for r in rows[:6]:
            r = dict(r); r["Name"] = r["Name"][:140]; w.writerow(r)
    return [r for r in rows if "de::" in r["Name"]]

k = kstats(f"{src}/stats/eval_kernel_stats.csv", f"profiles/{pre}_headline_kernel_stats.csv")
g = kstats(f"{src}/stats_C3/grad_kernel_stats.csv", f"profiles/{pre}_C3_grad_kernel_stats.csv")

def big_launches(trace, needle):
    """Durations (us) of the full-size launches of the kernel whose name contains `needle` (the one-sample
    constant-folding launch of the same kernel and the handler-table kernel are excluded by grid size)."""
    rows = [r for r in csv.DictReader(open(trace)) if needle in r["Kernel_Name"]]
    gmax = max(int(r["Grid_Size_X"]) for r in rows)
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if int(r["Grid_Size_X"]) == gmax]
    return d, gmax
ev, gmax = big_launches(f"{src}/stats/eval_kernel_trace.csv", "de_eval_")
def per_step(trace, needle):
    """The gradient runs as one launch per width/samples-per-lane bucket: time of ALL full-size launches per step."""
    rows = [r for r in csv.DictReader(open(trace)) if needle in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 256 * 64]
    gmax = max(int(r["Grid_Size_X"]) for r in rows)
    steps = sum(1 for r in rows if int(r["Grid_Size_X"]) == gmax)
    total = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
    return [total / steps] * steps
gr = per_step(f"{src}/stats_C3/grad_kernel_trace.csv", "de_grad_")

def pmc(path, name, out):
    rows = [r for r in csv.DictReader(open(path)) if "de_eval_" in r["Kernel_Name"] and r["Counter_Name"] == name]
    gm = max(int(r["Grid_Size"]) for r in rows)
    rows = [r for r in rows if int(r["Grid_Size"]) == gm]  # full-size launches only
    with open(out, "w") as fh:
        f = ["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size"]
        w = csv.DictWriter(fh, fieldnames=f); w.writeheader()
        for r in rows: w.writerow({q: r[q] for q in f})
    vals = [float(r["Counter_Value"]) for r in rows]
    return sum(vals) / len(vals), len(vals), rows[0]["Kernel_Name"]

fetch, n1, kn = pmc(f"{src}/pmc_fetch/eval_counter_collection.csv", "FETCH_SIZE", f"profiles/{pre}_pmc_fetch.csv")
write, n2, _ = pmc(f"{src}/pmc_write/eval_counter_collection.csv", "WRITE_SIZE", f"profiles/{pre}_pmc_write.csv")
summ = {"headline": {
    "kernel": kn, "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
    "fetch_correction": "x2: on gfx950 rocprofv3 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read "
                        "(MI355X_MICROARCH.md §HBM); WRITE_SIZE equals the 40.0 GB output exactly, i.e. needs no correction",
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "launches_averaged": min(n1, n2),
    "avg_kernel_us_rocprof": sum(ev) / len(ev), "kernel_launches_in_trace": len(ev), "grid_size": gmax,
    "grad_C3_avg_kernel_us_rocprof": sum(gr) / len(gr),
    "source": f"tools/profile_round.sh {tag}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
              "`python bench.py --steps 2 --warmup 1` (headline workload); kernel time from `rocprofv3 --kernel-trace --stats`"}}
def pmc_per_step(path, name, needle, steps):
    """Counter summed over ALL full-size launches of the kernels containing `needle`, per step (the gradient runs one
    launch per bucket)."""
    rows = [r for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"] and r["Counter_Name"] == name and int(r["Grid_Size"]) > 256 * 64]
    return sum(float(r["Counter_Value"]) for r in rows) / steps, len(rows)
fc3, wc3 = f"{src}/pmc_fetch_C3/grad_counter_collection.csv", f"{src}/pmc_write_C3/grad_counter_collection.csv"
if os.path.exists(fc3) and os.path.exists(wc3):
    steps = 3  # bench.py --steps 2 --warmup 1
    f3, nf = pmc_per_step(fc3, "FETCH_SIZE", "de_grad_", steps)
    w3, nw = pmc_per_step(wc3, "WRITE_SIZE", "de_grad_", steps)
    summ["C3"] = {"kernel": "de_grad_threaded_kernel (all bucket launches of a step)", "FETCH_SIZE_KiB_per_step": f3,
                  "WRITE_SIZE_KiB_per_step": w3, "hbm_bytes_per_launch": (2 * f3 + w3) * 1024, "launches_counted": min(nf, nw),
                  "fetch_correction": "x2, as for the headline",
                  "source": f"tools/profile_round.sh {tag}: separate --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench.py --workload C3 --steps 2 --warmup 1`"}
json.dump(summ, open("profiles/pmc_summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1))
print("eval kernel avg us:", sum(ev) / len(ev), " grad:", sum(gr) / len(gr))
if os.path.exists(f"{src}/stats_turbo/eval_kernel_stats.csv"):
    kstats(f"{src}/stats_turbo/eval_kernel_stats.csv", f"profiles/{pre}_headline_turbo_kernel_stats.csv")
    tv, _ = big_launches(f"{src}/stats_turbo/eval_kernel_trace.csv", "de_eval_")
    summ["headline"]["turbo_avg_kernel_us_rocprof"] = sum(tv) / len(tv)
    json.dump(summ, open("profiles/pmc_summary.json", "w"), indent=1)
    print("turbo eval kernel avg us:", sum(tv) / len(tv))
# bench lines of the same round: gpurun_out/<bench dir>/bench_<workload>.json (tools/gpu_check.sh)
bdir = sys.argv[3] if len(sys.argv) > 3 else None
if bdir:
    for f in sorted(os.listdir(f"gpurun_out/{bdir}")):
        if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(f"gpurun_out/{bdir}/{f}") > 0:
            shutil.copy(f"gpurun_out/{bdir}/{f}", f"profiles/{pre}_{f}")
