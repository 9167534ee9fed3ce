import sys, ctypes, numpy as np, json, subprocess
sys.path.insert(0, '/root/repo')
import torch, bench, habdec_amd
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"])
L = habdec_amd.lib(); f = L.hd_debug_sym_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
acc = np.zeros((5,), np.float64); n = 0; worst = None
for i in range(70):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
    if i >= 10:
        st = np.zeros(S * 8, np.uint64); f(st.ctypes.data, S * 8); st = st.reshape(S, 8).astype(np.int64)
        d = np.diff(st[:, :6], axis=1)     # phase durations in 100 MHz ticks? s_memtime = shader clock cycles
        tot = st[:, 5] - st[:, 0]
        j = int(np.argmax(tot))
        acc += d.mean(axis=0); n += 1
        if worst is None or tot[j] > worst[0]: worst = (int(tot[j]), d[j].tolist(), int(st[j, 6]), int(st[j, 7]))
print("mean cycles per phase [A0, A1(windows), B(search), C(run sums), D]:", (acc / n).round(0).tolist())
print("worst stream: total", worst)
