# PCIe-inclusive rate of the boundary's host entry point (hd_process_host): the slab is copied H2D inside every call.
import sys, time, ctypes, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, bench, habdec_amd
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"])
for pinned in (True, False):
    t = torch.randn((S, C, 2), dtype=torch.float32) * 0.3
    if pinned: t = t.pin_memory()
    a = t.numpy().view(np.complex64).reshape(S, C)
    for i in range(3): eng.process_host(a, C)
    eng.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 12
    for i in range(K): eng.process_host(a, C)
    eng.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"host-fed ({'pinned' if pinned else 'pageable'} source): {dt * 1e3:.2f} ms per 512 MiB slab = {S * C / dt / 1e9:.2f} GS/s = {S * C * 8 / dt / 1e9:.1f} GB/s over PCIe")
