import sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo')
import torch, bench, habdec_amd
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring = torch.randn((8, S, C, 2), device=dev) * 0.3
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"])
L = habdec_amd.lib(); f = L.hd_debug_dec_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for i in range(12):
    eng.process_device(ring.data_ptr() + (i % 8) * S * C * 8, C, C)
st = np.zeros(4096 * 8, np.uint64); f(st.ctypes.data, 4096 * 8); st = st.reshape(4096, 8).astype(np.float64)
st = st[st[:, 4] > 0]
per = st[:, :4] / st[:, 4:5]
print("workgroups sampled:", len(st), "tiles per WG:", st[0, 4])
print("cycles per tile per wave [wait prefetched tile, stage to LDS, issue next loads, compute+store]:", per.mean(axis=0).round(0).tolist(), "sum", per.sum(axis=1).mean().round(0))
print("kernel ms_front:", eng.timing()["ms_front"])
