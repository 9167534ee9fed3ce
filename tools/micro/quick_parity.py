"""Quick parity check of the batch path on the GPU box: S streams of noise through three pipelined calls, decimated output of sampled streams bit for bit\nagainst the CPU oracle, with the ranges that differ (which stage-1 tiles are wrong) printed.  S=64 python tools/micro/quick_parity.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import habdec_amd
from oracle import pyoracle
S, C, fs = int(os.environ.get("S", 64)), 65536, 2.048e6
rng = np.random.default_rng(1)
calls = 3
x = (rng.standard_normal((calls, S, C, 2)).astype(np.float32) * 0.3)
dev = torch.device("cuda", 0)
xd = torch.from_numpy(x).to(dev)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, baud=50, rtty_bits=7, rtty_stops=2, pipeline=1)
check = list(range(0, S, max(1, S // 16)))
orcs = {s: pyoracle.Decoder("oracle", factor=64, baud=50, bits=7, stops=2) for s in check}
for k in range(calls):
    eng.process_device(xd[k].data_ptr(), C, C)
    for s, o in orcs.items():
        o(x[k, s].view(np.complex64).reshape(C), fs)
        a, b = eng.decimated(s), o.array("last_decimated")
        bad = np.nonzero(a.view(np.uint64) != b.view(np.uint64))[0]
        if len(bad):
            # runs of bad indices
            brk = np.nonzero(np.diff(bad) > 1)[0]
            starts = np.r_[bad[0], bad[brk + 1]]; ends = np.r_[bad[brk], bad[-1]]
            print("call", k, "stream", s, "bad", len(bad), [(int(a_), int(b_)) for a_, b_ in zip(starts, ends)][:8], "zeros", int(np.sum(a[bad] == 0)))
    print("call", k, "done; variant", eng.timing()["step_variant"], "path", eng.timing()["path"])
