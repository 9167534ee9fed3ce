#!/usr/bin/env python3
"""Steady-state kernel statistics from a rocprofv3 --kernel-trace csv (run ON the GPU box, before the trace is deleted): the launches of every hd:: kernel
that START after the `skip`-th launch of the most time-consuming hd:: kernel has started -- i.e. without the step loop's first launches, which run before
the power controller has settled -- in the column layout of rocprofv3's own *_kernel_stats.csv plus the number of launches left out.

    steady_stats.py <kernel_trace.csv> <out.csv> [skip=150]
"""
import csv, re, sys, collections, statistics


def short(name):
    m = re.search(r"hd::(?:exact::|(fast)::)?(k_\w+)(<[^>]*>)?", name)
    if m: return m.group(2) + (m.group(3) or "").replace(" ", "") + ("[fast]" if m.group(1) else "")
    if "fft_rtc" in name: return name.split("(")[0][:60]
    return None


def main():
    src, dst, skip = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 150
    runs = collections.defaultdict(list)
    for r in csv.DictReader(open(src, newline="")):
        k = short(r["Kernel_Name"])
        if k: runs[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not runs:
        raise SystemExit("no hd:: kernels in " + src)
    front = max(runs, key=lambda k: sum(e - s for s, e in runs[k]))
    starts = sorted(s for s, _ in runs[front])
    t0 = starts[min(skip, len(starts) - 1)] if len(starts) > skip else starts[len(starts) // 2]
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "LaunchesLeftOut", "Note"])
        for k in sorted(runs, key=lambda k: -sum(e - s for s, e in runs[k])):
            d = [e - s for s, e in runs[k] if s >= t0]
            if not d: continue
            w.writerow([k, len(d), sum(d), round(sum(d) / len(d), 1), min(d), max(d), round(statistics.pstdev(d), 1), len(runs[k]) - len(d),
                        f"launches that start after launch {skip} of {front}"])
    print(dst, front, "steady launches", sum(1 for s in starts if s >= t0), "of", len(starts))


if __name__ == "__main__":
    main()
