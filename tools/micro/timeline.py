"""Kernel timeline of a bench run from a rocprofv3 --kernel-trace csv: the hd:: kernels of the last few steps with start / end relative to the first one shown,
their queue, and the gap to the previous kernel on the same queue.  Usage: timeline.py <kernel_trace.csv> [n_last=40] [name filter]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
flt = sys.argv[3] if len(sys.argv) > 3 else ""
ks = []
for r in rows:
    m = re.search(r"hd::(?:exact::|(fast)::)?(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
    name = (m.group(2) + (m.group(3) or "") + ("[fast]" if m.group(1) else "")) if m else ("rocfft" if "fft" in r["Kernel_Name"] else None)
    if name and flt in name:
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
ks.sort()
ks = ks[-n:]
t0 = ks[0][0]
last = {}
for s, e, name, q in ks:
    gap = (s - last[q]) / 1e3 if q in last else float("nan")
    print(f"q{q:>3s} {name:28s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap on queue {gap:7.1f}")
    last[q] = e
