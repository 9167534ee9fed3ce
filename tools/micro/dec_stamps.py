import sys, ctypes, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, bench, habdec_amd
import os
w = dict(bench.WORKLOADS[os.environ.get("WL", "cfg4")]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring = torch.randn((8, S, C, 2), device=dev) * 0.3
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"])
eng.set_timing(1)
L = habdec_amd.lib(); f = L.hd_debug_dec_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for i in range(12):
    eng.process_device(ring.data_ptr() + (i % 8) * S * C * 8, C, C)
st = np.zeros(4096 * 8, np.uint64); f(st.ctypes.data, 4096 * 8); st = st.reshape(4096, 8).astype(np.float64)
st = st[st[:, 7] > 0]
t0 = st[:, 0].min()
print("workgroups:", len(st), "tiles per WG:", st[0, 7], " kernel ms_front:", eng.timing()["ms_front"])
print("100MHz timeline (us): start  p0/50/100 = %s ; first tile staged p0/50/100 = %s ; end p0/50/100 = %s" % tuple(
    np.percentile((st[:, k] - t0) / 100.0, [0, 50, 100]).round(1).tolist() for k in (0, 1, 2)))
per = st[:, 3:7] / st[:, 7:8]
print("cycles per tile per wave [wait prefetched tile, stage to LDS+barrier, store+issue next loads, compute]:", per.mean(axis=0).round(0).tolist(), "sum", per.sum(axis=1).mean().round(0))
first = st[0::2]; second = st[1::2]
print("stream-first WGs: end %.1f us ; second WGs: end %.1f us" % (np.median(first[:, 2] - t0) / 100.0, np.median(second[:, 2] - t0) / 100.0))
# does the finishing time depend on where the workgroup ran?  (workgroups are dealt round-robin to the 8 XCDs)
w_all = np.arange(4096)[:len(st)]
end = (st[:, 2] - t0) / 100.0
print("mean end (us) by workgroup id mod 8:", [round(float(end[w_all % 8 == j].mean()), 1) for j in range(8)])
print("std of end within an XCD class:", [round(float(end[w_all % 8 == j].std()), 1) for j in range(8)])
print("mean end by (id // 8) mod 4:", [round(float(end[(w_all // 8) % 4 == j].mean()), 1) for j in range(4)])
print("mean end by id // 256 (dispatch order octiles):", [round(float(end[w_all // 256 == j].mean()), 1) for j in range(8)])
