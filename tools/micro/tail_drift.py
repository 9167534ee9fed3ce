"""How a stream tail's duration moves along the benchmark ring (diagnostic build: -DHD_STAMP_TAIL), on-tune vs far-off streams."""
import sys, ctypes, os, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, bench, habdec_amd
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import _variant
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
pipe = bool(int(os.environ.get("PIPE", "1")))
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], pipeline=pipe)
L = habdec_amd.lib(); f = L.hd_debug_step_tail_stamps if pipe else L.hd_debug_tail_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
off = (np.arange(S) % 8) == 7
names = ["initial", "backlog ws", "X", "stage 2", "lp+disc", "F slide", "win sums", "carries", "search loads", "edge search", "run sums", "bits"]
for i in range(140):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
    if i % 10 == 9:
        torch.cuda.synchronize()
        st = np.zeros(S * 24, np.uint64); f(st.ctypes.data, S * 24); st = st.reshape(S, 24).astype(np.int64)
        wall = (st[:, 21] - st[:, 20]) * 10.0 / 1e3
        d = st[:, :20].astype(np.float64)
        grp = lambda m: " ".join(f"{x/1e3:5.1f}" for x in [d[m, 8].mean(), d[m, 9].mean(), d[m, 10].mean(), d[m][:, [15, 16, 17, 19, 6]].sum(axis=1).mean()])
        print(f"call {i:3d}: wall us on-tune p50 {np.percentile(wall[~off], 50):6.1f} max {wall[~off].max():6.1f} | far-off p50 {np.percentile(wall[off], 50):6.1f} max {wall[off].max():6.1f}"
              f" | kcycles [search loads, edge search, run sums, window sums] on-tune {grp(~off)}  far-off {grp(off)}  held(backlog) on/off {eng.symbol_backlog(0)} {eng.symbol_backlog(7)}")
