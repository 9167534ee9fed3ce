import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch, bench, habdec_amd
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"])
eng.set_timing(1)
n = 8
ring = torch.randn((n, S, C, 2), device=dev) * 0.3
torch.cuda.synchronize()
ts = []
for i in range(60):
    eng.process_device(ring.data_ptr() + (i % n) * S * C * 8, C, C)
    if i >= 10: ts.append(eng.timing()["ms_front"])
print(sys.argv[1] if len(sys.argv) > 1 else "", f"stage-1 {np.mean(ts) * 1e3:7.1f} us (min {np.min(ts) * 1e3:.1f}, med {np.median(ts) * 1e3:.1f})")
