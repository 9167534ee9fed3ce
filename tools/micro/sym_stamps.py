import sys, ctypes, numpy as np, json, subprocess
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, bench, habdec_amd
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant
w = dict(bench.WORKLOADS[os.environ.get("WL", "cfg4")]); S = w["S"]; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                        lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], pipeline=int(os.environ.get("PIPE", "0")))
L = habdec_amd.lib(); f = L.hd_debug_sym_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
tot_p = []; ph_mean = np.zeros(5); ph_max = np.zeros(5); n = 0; span = []
NC = int(os.environ.get('NCALLS', '70'))
for i in range(NC):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
    if i >= 10:
        st = np.zeros(S * 8, np.uint64); f(st.ctypes.data, S * 8); st = st.reshape(S, 8).astype(np.int64)
        d = np.diff(st[:, :6], axis=1).astype(np.float64)
        tot = (st[:, 5] - st[:, 0]).astype(np.float64)
        tot_p.append(np.percentile(tot, [50, 90, 99, 100]))
        ph_mean += d.mean(axis=0); ph_max = np.maximum(ph_max, d.max(axis=0)); n += 1
        span.append((st[:, 5].max() - st[:, 0].min()))
        if i in (NC // 3, NC - 2):
            for j in np.argsort(-tot)[:4]: print('call', i, 'stream', int(j), 'total', int(tot[j]), 'phases', d[j].astype(int).tolist(), 'nflips', int(st[j, 6]), 'last flip', int(st[j, 7]))
            print('   nflips histogram over streams:', np.bincount(st[:, 6].astype(int))[:8].tolist())
if os.environ.get("SWEEP"): print("sweep parts: staging, window sums (cycles, mean over streams of the last call):", st[:, 6].mean().round(0), st[:, 7].mean().round(0))
print("phases [A0 mask image, A1 windows, B search, C run sums, D]: mean cycles", (ph_mean / n).round(0).tolist(), " max", ph_max.tolist())
print("per-stream total cycles p50/p90/p99/max (mean over calls):", np.mean(tot_p, axis=0).round(0).tolist())
print("first-start to last-end span, cycles (mean over calls):", np.mean(span).round(0))
