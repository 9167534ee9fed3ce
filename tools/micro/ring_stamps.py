"""Loader and consumer waves of k_step_cu (diagnostic build: HD_EXTRA_FLAGS=-DHD_STAMP_RING): cycles per role and phase in the last launch."""
import sys, ctypes, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, habdec_amd
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import _variant
w = dict(bench.WORKLOADS["cfg4"]); S = 1024; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], arith=int(__import__("os").environ.get("ARITH", "0")), pipeline=bool(int(os.environ.get("PIPE", "1"))))
eng.set_timing(1)
L = habdec_amd.lib(); f = getattr(L, "hd_debug_ring_stamps" + ("_fast" if int(os.environ.get("ARITH", "0")) else "")); f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for i in range(int(os.environ.get("NCALLS", "30"))):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)
eng.flush()
n = 512 * 8 * 8
st = np.zeros(n, np.uint64); f(st.ctypes.data, n); st = st.reshape(512, 8, 8).astype(np.float64)[:256]
t0 = st[:, :, 6][st[:, :, 6] > 0].min()
print("step kernel ms:", eng.timing()["ms_front"], "variant", eng.timing()["step_variant"])
# rows: roles 0-3 = the stage-1 worker waves that exist from the start; roles 4-7 = the tail waves, workers once their tail is done (ring_worker's record)
for wv in range(8):
    c = st[:, wv]
    if not (c[:, 5] > 0).any(): continue
    c = c[c[:, 5] > 0]
    print("worker %d: tiles p0/50/100 %s; per tile: wait for landing %.0f row reads %.0f next run + DMA issue %.0f tap loop %.0f store + loop %.0f; lifetime us p50 %.1f end us p100 %.1f" % (
        wv, np.percentile(c[:, 5], [0, 50, 100]).tolist(), np.median(c[:, 0] / c[:, 5]), np.median(c[:, 3] / c[:, 5]), np.median(c[:, 4] / c[:, 5]), np.median(c[:, 1] / c[:, 5]),
        np.median(c[:, 2] / c[:, 5]), np.median((c[:, 7] - c[:, 6]) / 100), ((c[:, 7] - t0) / 100).max()))
# when does each CU (workgroup) run out of work, and when did its tails hand over?  (realtime counter: 100 ticks per microsecond)
ok = st[:, :, 7] > 0
end_cu = np.where(ok, st[:, :, 7], 0).max(axis=1)
start_cu = np.where(ok, st[:, :, 6], np.inf).min(axis=1)
print("workgroup start us p0/50/100", np.percentile((start_cu - t0) / 100, [0, 50, 100]).round(1).tolist(),
      " end us p0/10/50/90/100", np.percentile((end_cu - t0) / 100, [0, 10, 50, 90, 100]).round(1).tolist())
tl = st[:, 4:8]
tstart = tl[:, :, 6]
print("late waves (tails first): became computing waves at us p0/50/100", np.percentile((tstart[tstart > 0] - t0) / 100, [0, 50, 100]).round(1).tolist(),
      " last of a CU p50/100", np.percentile((tstart.max(axis=1) - t0) / 100, [50, 100]).round(1).tolist())
per_xcd = [np.percentile((end_cu[x::8] - t0) / 100, 100) for x in range(8)]
print("end us of the last workgroup per blockIdx %% 8:", np.round(per_xcd, 1).tolist())
if os.environ.get("PER_CU"):
    # one line per workgroup of two XCDs: when its waves ended (us), tiles of worker 0, when its tail waves turned to stage 1
    for x in (0, 1):
        print(f"-- workgroups with blockIdx % 8 == {x} (sorted by end): end us | worker-0 tiles | worker-0 end | tails done at us (four waves)")
        rows = []
        for b in range(x, 256, 8):
            rows.append(((end_cu[b] - t0) / 100, int(st[b, 0, 5]), (st[b, 0, 7] - t0) / 100, sorted(((st[b, 4:8, 6] - t0) / 100).round(0).tolist())))
        for r in sorted(rows): print("   %6.1f | %3d | %6.1f | %s" % r)
