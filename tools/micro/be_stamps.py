import sys, ctypes, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, bench, habdec_amd
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"])
L = habdec_amd.lib(); f = L.hd_debug_be_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
acc = np.zeros(6); mx = np.zeros(6); n = 0; spans = []
for i in range(40):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
    if i >= 10:
        st = np.zeros(S * 8, np.uint64); f(st.ctypes.data, S * 8); st = st.reshape(S, 8).astype(np.int64)
        d = np.diff(st[:, :7], axis=1).astype(np.float64)
        acc += d.mean(axis=0); mx = np.maximum(mx, d.max(axis=0)); n += 1
        spans.append([np.percentile(st[:, 6] - st[:, 0], q) for q in (50, 100)])
print("phases [staging trip, stage 2, carry+slide, low-pass pass 0, discriminator pass 0, rest (pass 1)]: mean cycles", (acc / n).round(0).tolist(), "max", mx.tolist())
print("per-stream total cycles p50/max:", np.mean(spans, axis=0).round(0).tolist())
