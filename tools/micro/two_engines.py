"""Is a shape bound by its launch chain or by the chip?  One engine with S streams against two engines with S/2 each on two host threads
(four queues instead of two): aggregate samples per second.  WL=cfg2|cfg3|cfg5."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, habdec_amd
w = dict(bench.WORKLOADS[os.environ.get("WL", "cfg2")]); C = w["C"]; K = int(os.environ.get("STEPS", "40"))
dev = torch.device("cuda", 0)

def make(S, seed):
    ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, seed)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], pipeline=2)
    return eng, ring, rc

def run(eng, ring, rc, S, n):
    for i in range(n):
        eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
    eng.flush()

for parts in (1, 2):
    S = w["S"] // parts
    es = [make(S, 1234 + j) for j in range(parts)]
    for e, r, rc in es: run(e, r, rc, S, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(e, r, rc, S, K)) for e, r, rc in es]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{parts} engine(s) x {S} streams: {w['S'] * C * K / dt / 1e9:.1f} GS/s, {dt / K * 1e3:.4f} ms per step of {w['S']} streams")
    for e, _, _ in es: e.close()
    del es
    torch.cuda.empty_cache()
