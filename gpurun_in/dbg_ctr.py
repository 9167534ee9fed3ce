import sys, os, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import habdec_amd, bench
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
torch.cuda.synchronize()
L = habdec_amd.lib(); f = L.hd_debug_step_counters; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
for rep in range(4):
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], pipeline=2)
    buf = np.zeros(32, np.uint32)
    f(eng.h, buf.ctypes.data); print("rep", rep, "initial", buf[:8].tolist(), buf[16:24].tolist())
    for i in range(4):
        t1 = time.time(); eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C); torch.cuda.synchronize(); dt = time.time() - t1
        f(eng.h, buf.ctypes.data); print("  call", i, "t", round(dt, 3), "set0", buf[:8].tolist(), "set1", buf[16:24].tolist(), flush=True)
    eng.flush(); eng.close()
