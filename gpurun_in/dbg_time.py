import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import habdec_amd, bench
t0 = time.time()
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
torch.cuda.synchronize(); print("ring", time.time() - t0); t0 = time.time()
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], pipeline=2)
print("engine", time.time() - t0); t0 = time.time()
ts = []
for i in range(40):
    t1 = time.time()
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
    ts.append(time.time() - t1)
print("calls", time.time() - t0, "max call", max(ts), "idx", int(np.argmax(ts)), [round(x, 4) for x in ts[:8]]); t0 = time.time()
eng.flush()
print("flush", time.time() - t0); t0 = time.time()
print(eng.timing())
eng.close()
print("close", time.time() - t0)
