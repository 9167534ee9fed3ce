#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/gpu_profile.sh (gpurun_out/prof_<tag>/) into the committed evidence under
profiles/: a kernel-stats summary restricted to this project's kernels and the per-launch HBM traffic of every hd:: kernel.

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes (1024 B); on gfx950 FETCH_SIZE tallies the 128-B requests of wide streaming reads at 64 B, so it is doubled
(MI355X_MICROARCH.md, "HBM").  WRITE_SIZE is taken as is.  The stage-1 decimator's figure goes to profiles/traffic.json,
which bench.py reports as roofline.traffic.
"""
import csv, json, sys, re, collections, pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from habdec_amd.build import source_id
KB = 1024.0


def short(name):
    m = re.search(r"hd::(?:exact::|(fast)::)?(k_\w+)(<[^>]*>)?", name)
    if m: return m.group(2) + (m.group(3) or "").replace(" ", "") + ("[fast]" if m.group(1) else "")
    if "fft_rtc" in name: return name.split("(")[0][:60]
    return None


def pmc(path, counter):
    acc = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter: continue
            k = short(r["Kernel_Name"])
            if k: acc[k].append(float(r["Counter_Value"]))
    return acc


def main():
    tag, rnd, workload = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "cfg4")
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    dst = ROOT / "profiles"
    # 1. kernel stats summaries: rocprofv3's own (every launch, the loop's ramp included) and the steady-state one (tools/steady_stats.py on the GPU box:
    # launches behind the first 150 of the dominant kernel), which is what bench.py's roofline.frac_rocprof refers to
    rows = []
    with open(next(p for p in (src / "stats").glob("*kernel_stats.csv") if "steady" not in p.name), newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            k = short(r["Name"])
            if k: rows.append([k, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    out = dst / f"{rnd}_kernel_stats_all_launches.csv"
    with open(out, "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"]); w.writerows(rows)
    steady = src / "stats" / "steady_kernel_stats.csv"
    out = dst / f"{rnd}_kernel_stats.csv"
    if steady.exists():
        out.write_text(steady.read_text())
        with open(steady, newline="") as f:
            rows = [[r["Kernel"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]] for r in csv.DictReader(f)]
    else:
        out.write_text((dst / f"{rnd}_kernel_stats_all_launches.csv").read_text())
    # 2. PMC traffic per launch (steady-state launches: drop the warm-up launches with tiny grids by taking the median)
    fetch = pmc(next((src / "fetch").glob("*counter_collection.csv")), "FETCH_SIZE")
    write = pmc(next((src / "write").glob("*counter_collection.csv")), "WRITE_SIZE")
    import statistics
    table = {}
    for k in sorted(set(fetch) | set(write)):
        fr = statistics.median(fetch.get(k, [0.0])) * KB * 2.0   # gfx950 correction: x2
        wr = statistics.median(write.get(k, [0.0])) * KB
        table[k] = {"launches": len(fetch.get(k, [])), "fetch_bytes": round(fr), "write_bytes": round(wr), "hbm_bytes": round(fr + wr)}
    (dst / f"{rnd}_pmc_traffic.json").write_text(json.dumps(table, indent=1) + "\n")
    front = max((k for k in table if k.startswith("k_decimate") or k.startswith("k_step") or k.startswith("k_stage1")), key=lambda k: table[k]["hbm_bytes"] * max(table[k]["launches"], 1))
    tf = dst / "traffic.json"
    cur = json.loads(tf.read_text()) if tf.exists() else {}
    st = next((r for r in rows if r[0] == front), None)      # the same kernel in the --kernel-trace --stats pass: bench.py reports it beside its live figure
    cur[workload] = {"front_kernel": front, "front_kernel_hbm_bytes_per_launch": table[front]["hbm_bytes"],
                     "fetch_bytes_x2_corrected": table[front]["fetch_bytes"], "write_bytes": table[front]["write_bytes"],
                     "source": f"profiles/{rnd}_pmc_traffic.json",
                     "rocprof_avg_launch_ms": round(float(st[3]) / 1e6, 5) if st else None, "rocprof_launches": int(st[1]) if st else None,
                     "stats_source": f"profiles/{rnd}_kernel_stats.csv",
                     "csrc_id": source_id()}      # the sources this was measured on: bench.py quotes the figures only while habdec_amd.build.source_id() still says so
    tf.write_text(json.dumps(cur, indent=1) + "\n")
    for b in ("stats", "fetch", "write"):
        p = src / f"{b}_bench.json"
        if p.exists(): (dst / f"{rnd}_{b}_bench_line.json").write_text(p.read_text())
    print(out); print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
