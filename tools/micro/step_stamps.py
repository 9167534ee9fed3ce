"""Stage-1 workgroups inside the step kernel (diagnostic build: HD_EXTRA_FLAGS=-DHD_STAMP_DEC): when they run and what a tile costs them,
early in the launch (beside the stream tails) and late (alone)."""
import sys, ctypes, os, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, bench, habdec_amd
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import _variant
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], pipeline=True)
eng.set_timing(1)
L = habdec_amd.lib(); f = L.hd_debug_dec_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for i in range(30):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
eng.flush()
st = np.zeros(4096 * 8, np.uint64); f(st.ctypes.data, 4096 * 8); st = st.reshape(4096, 8).astype(np.float64)
st = st[st[:, 7] > 0]
st = st[st[:, 0] >= st[:, 0].max() - 100000]      # the last launch only (100 MHz ticks: 1 ms): with drawn runs a workgroup that finds no run writes nothing
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 2] - t0) / 100.0
print("stage-1 workgroups:", len(st), "tiles per WG:", st[0, 7], " step kernel ms:", eng.timing()["ms_front"])
print("start us p0/25/50/75/100:", np.percentile(start, [0, 25, 50, 75, 100]).round(1).tolist())
print("end   us p0/25/50/75/100:", np.percentile(end, [0, 25, 50, 75, 100]).round(1).tolist())
dur = end - start
per = st[:, 3:7] / st[:, 7:8]
for lo, hi in [(0, 20), (20, 60), (60, 100), (100, 140), (140, 400)]:
    m = (start >= lo) & (start < hi)
    if m.sum():
        print(f"WGs starting in [{lo},{hi}) us: n={int(m.sum()):5d}  duration {dur[m].mean():6.1f} us = {dur[m].mean() / st[m, 7].mean():5.2f} us/tile;  cycles/tile [wait, LDS store, out-store+issue, compute] {per[m].mean(axis=0).round(0).tolist()}")
