"""Phase clocks of the stream tail (diagnostic build: HD_EXTRA_FLAGS=-DHD_STAMP_TAIL python -m habdec_amd.build --force)."""
import sys, ctypes, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, bench, habdec_amd
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import _variant
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], arith=int(__import__("os").environ.get("ARITH", "0")), pipeline=bool(int(__import__("os").environ.get("PIPE", "0"))))
L = habdec_amd.lib(); f = getattr(L, ('hd_debug_step_tail_stamps' if int(__import__('os').environ.get('PIPE', '0')) else 'hd_debug_tail_stamps') + ('_fast' if int(__import__('os').environ.get('ARITH', '0')) else '')); f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
acc = np.zeros(20); mx = np.zeros(20); n = 0; spans = []; ends = []
for i in range(40):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
    if i >= 10:
        if int(__import__('os').environ.get('PIPE', '0')): torch.cuda.synchronize()
        st = np.zeros(S * 24, np.uint64); f(st.ctypes.data, S * 24); st = st.reshape(S, 24).astype(np.int64)
        d = st[:, :20].astype(np.float64)
        acc += d.mean(axis=0); mx = np.maximum(mx, d.max(axis=0)); n += 1
        spans.append([np.percentile((st[:, 21] - st[:, 20]) * 10.0, q) for q in (50, 90, 100)])   # 100 MHz realtime ticks -> ns
        t0 = st[:, 20].min(); endus = (st[:, 21] - t0) / 100.0; far = (np.arange(S) % 8) == 7    # (bench.generate_ring: every eighth stream is far off tune)
        ends.append([np.percentile(endus[~far], q) for q in (50, 90, 100)] + [np.percentile(endus[far], q) for q in (50, 90, 100)])
names = ["initial loads", "backlog window sums", "piece: X to LDS", "piece: stage 2", "piece: low-pass+discriminator", "piece: F slide", "piece: window sums",
         "carries+slide", "search loads", "edge search", "run sums", "bits+state",
         "  lp: tap loop", "  lp: exchange+discriminator", "  lp: stores", "  ws: sums", "  ws: flags+mask", "  ws: slide", "  stage 2: tap loop only", "  ws: window_sums8 only"]
for k, a_, m_ in zip(names, acc / n, mx):
    print(f"{k:32s} mean {a_:9.0f} cycles   max {m_:9.0f}")
print("total cycles mean", (acc / n).sum().round(0), " per-stream wall ns p50/p90/max:", np.mean(spans, axis=0).round(0).tolist())
print("a tail's end, us after the launch's first tail started: on-tune streams p50/p90/max %s; far-off-tune streams p50/p90/max %s" % (
    np.mean(ends, axis=0)[:3].round(1).tolist(), np.mean(ends, axis=0)[3:].round(1).tolist()))
