#!/usr/bin/env python3
"""Condense gpurun_out/profiles_<tag>/ (tools/profile_round.sh) into the committed profiles/ files.

    python tools/make_profiles.py <tag> <round-prefix> [bench dir under gpurun_out]      e.g.  r3p r3 r3b

Per workload W: profiles/<prefix>_<W>_kernel_stats.csv (the rocprofv3 --stats table, top rows) and an entry of
profiles/pmc_summary.json: per-step duration of every de_* kernel of the step (rocprofv3 kernel trace), HBM bytes per step
from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction: FETCH_SIZE x2, MI355X_MICROARCH.md §HBM), and the hash of the
kernel sources the numbers belong to — bench.py refuses a summary whose hash is not that of the library it runs
(`roofline.traffic` = null then)."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha256 over the sources libde_hip.so is built from (csrc/ and include/): what `kernel_source_hash` in pmc_summary.json and
    bench.py's check mean."""
    h = hashlib.sha256()
    files = []
    for pat in ("dynamicexpressions.jl_amd/csrc/*.hip", "dynamicexpressions.jl_amd/csrc/*.h", "dynamicexpressions.jl_amd/csrc/*.cpp",
                "dynamicexpressions.jl_amd/csrc/*.py", "dynamicexpressions.jl_amd/csrc/build.sh", "include/*.h"):
        files += glob.glob(os.path.join(ROOT, pat))
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def one(path_glob):
    f = sorted(glob.glob(path_glob, recursive=True))
    return f[0] if f else None


def _last_per_kernel(rows, n):
    """The rows of the last n launches (dispatch ids) of every kernel, all counters of those launches."""
    ids = collections.defaultdict(list)
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d not in ids[r["Kernel_Name"]]:
            ids[r["Kernel_Name"]].append(d)
    keep = {k: set(sorted(v)[-n:]) for k, v in ids.items()}
    return [r for r in rows if int(r["Dispatch_Id"]) in keep[r["Kernel_Name"]]]


def main():
    tag, pre = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    summ = {"kernel_source_hash": kernel_source_hash(),
            "fetch_correction": "x2: on gfx950 rocprofv3 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read "
                                "(MI355X_MICROARCH.md §HBM); WRITE_SIZE needs none (it equalled the 40.0 GB output exactly in round 2)",
            "source": f"tools/profile_round.sh {tag}: per workload `rocprofv3 --kernel-trace --stats -- python bench.py --workload W --steps 5 "
                      "--warmup 1` and one `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass each over `--steps 2 --warmup 1`"}
    for wdir in sorted(glob.glob(os.path.join(src, "*"))):
        wl = os.path.basename(wdir)
        if not os.path.isdir(wdir):
            continue
        st = one(os.path.join(wdir, "stats", "**", "*kernel_stats.csv"))
        tr = one(os.path.join(wdir, "stats", "**", "*kernel_trace.csv"))
        if not st or not tr:
            print(f"{wl}: no kernel stats")
            continue
        rows = list(csv.DictReader(open(st)))
        with open(os.path.join(ROOT, "profiles", f"{pre}_{wl}_kernel_stats.csv"), "w") as fh:
            w = csv.DictWriter(fh, fieldnames=rows[0].keys())
            w.writeheader()
            for r in rows[:8]:
                r = dict(r)
                r["Name"] = r["Name"][:140]
                w.writerow(r)
        # per-step time of the de_* kernels: full-size launches only (the one-sample constant-folding launches and the
        # handler-table kernels are excluded by grid size); steps = launches of the largest grid / launches of it per step
        t = [r for r in csv.DictReader(open(tr)) if "de_" in r["Kernel_Name"] and "fill_handlers" not in r["Kernel_Name"]]
        gmax = collections.defaultdict(int)
        for r in t:
            k = r["Kernel_Name"].split("(")[0][-70:]
            gmax[k] = max(gmax[k], int(r["Grid_Size_X"]))
        by = collections.defaultdict(list)
        for r in t:
            k = r["Kernel_Name"].split("(")[0][-70:]
            if 50 * int(r["Grid_Size_X"]) >= gmax[k] and gmax[k] > 64 * 256:  # (>= 2 % of the kernel's largest grid: the launch groups of the reverse / gradient kernels differ in size)
                by[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        steps = 6  # --steps 5 --warmup 1
        if wl == "complete":  # the rejection sampling in front of the steps launches the same kernels (3 batches): only the steps count
            by = collections.defaultdict(list, {k: v[-steps:] for k, v in by.items()})
        e = {"kernels_us_per_step": {k: sum(v) / steps for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))},
             "launches_per_step": {k: len(v) / steps for k, v in by.items()},
             "us_per_step_all_de_kernels": sum(sum(v) for v in by.values()) / steps}
        dom = max(by.items(), key=lambda kv: sum(kv[1])) if by else None
        if dom:
            e["dominant_kernel"] = dom[0]
            e["dominant_kernel_avg_launch_us"] = sum(dom[1]) / len(dom[1])
            e["dominant_kernel_launches"] = len(dom[1])
            if len(dom[1]) > steps:  # several launches per step (gradient buckets): nothing to say per launch beyond the average
                pass
            elif len(dom[1]) > 1:
                e["dominant_kernel_avg_launch_us_without_first"] = sum(dom[1][1:]) / (len(dom[1]) - 1)  # the first launch of a process runs 10-20 % longer
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = one(os.path.join(wdir, f"pmc_{c}", "**", "*counter_collection.csv"))
            if not f:
                continue
            rr = [r for r in csv.DictReader(open(f)) if "de_" in r["Kernel_Name"] and r["Counter_Name"] == c and "fill_handlers" not in r["Kernel_Name"]]
            gm = collections.defaultdict(int)
            for r in rr:
                gm[r["Kernel_Name"]] = max(gm[r["Kernel_Name"]], int(r["Grid_Size"]))
            rr = [r for r in rr if 50 * int(r["Grid_Size"]) >= gm[r["Kernel_Name"]] and gm[r["Kernel_Name"]] > 64 * 256]
            if wl == "complete":  # (see above: the last 3 launches of every kernel are the steps)
                rr = _last_per_kernel(rr, 3)
            tot[c] = sum(float(r["Counter_Value"]) for r in rr) / 3  # --steps 2 --warmup 1
            e[f"{c}_KiB_per_step"] = tot[c]
        if len(tot) == 2:
            e["hbm_bytes_per_launch"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024  # per STEP (= per launch for the plain eval)
        f = one(os.path.join(wdir, "pmc_SQ", "**", "*counter_collection.csv"))
        if f:  # executed instruction counts per step, summed over the de_* kernels (full-size launches)
            rr = [r for r in csv.DictReader(open(f)) if "de_" in r["Kernel_Name"] and "fill_handlers" not in r["Kernel_Name"]]
            gm = collections.defaultdict(int)
            for r in rr:
                gm[r["Kernel_Name"]] = max(gm[r["Kernel_Name"]], int(r["Grid_Size"]))
            rr = [r for r in rr if 50 * int(r["Grid_Size"]) >= gm[r["Kernel_Name"]] and gm[r["Kernel_Name"]] > 64 * 256]
            if wl == "complete":
                rr = _last_per_kernel(rr, 3)
            sq = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in rr:
                sq[r["Kernel_Name"].split("(")[0][-70:]][r["Counter_Name"]] += float(r["Counter_Value"]) / 3  # --steps 2 --warmup 1
            e["sq_per_step_by_kernel"] = {k: dict(v) for k, v in sq.items()}
        b = os.path.join(wdir, "bench_under_rocprof.json")
        if os.path.exists(b) and os.path.getsize(b):
            try:
                d = json.loads(open(b).read().strip().splitlines()[-1])
                e["bench_under_rocprof"] = {"ms_per_step": d["ms_per_step"], "kernel_ms_avg_hipevents": d["roofline"]["kernel_ms_avg"]}
            except Exception as ex:  # noqa: BLE001
                e["bench_under_rocprof"] = str(ex)
        summ[wl] = e
        print(wl, json.dumps({k: v for k, v in e.items() if k not in ("kernels_us_per_step", "launches_per_step")}))
    if not any(isinstance(v, dict) for v in summ.values()):
        sys.exit(f"make_profiles: nothing under gpurun_out/profiles_{tag} (did the profile run happen?) — profiles/pmc_summary.json left as it is")
    json.dump(summ, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)
    shutil.copy(os.path.join(ROOT, "profiles", "pmc_summary.json"), os.path.join(ROOT, "profiles", f"{pre}_pmc_summary.json"))
    bdir = sys.argv[3] if len(sys.argv) > 3 else None
    if bdir:
        for f in sorted(os.listdir(os.path.join(ROOT, "gpurun_out", bdir))):
            p = os.path.join(ROOT, "gpurun_out", bdir, f)
            if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(p) > 0:
                shutil.copy(p, os.path.join(ROOT, "profiles", f"{pre}_{f}"))


if __name__ == "__main__":
    main()
