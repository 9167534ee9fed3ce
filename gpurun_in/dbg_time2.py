import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import habdec_amd, bench
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
torch.cuda.synchronize()
for N in (2, 3, 4, 5, 8):
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], pipeline=int(os.environ.get("PIPE", "2")))
    t0 = time.time(); ts = []
    for i in range(N):
        t1 = time.time(); eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C); torch.cuda.synchronize(); ts.append(round(time.time() - t1, 4))
    t1 = time.time(); eng.flush(); tf = time.time() - t1
    print("N", N, "per-call(+sync)", ts, "flush", round(tf, 4), flush=True)
    eng.close()
