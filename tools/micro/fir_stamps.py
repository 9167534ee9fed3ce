"""Per-workgroup clocks of k_fir_demod (diagnostic build: HD_EXTRA_FLAGS=-DHD_STAMP_FIR python -m habdec_amd.build --force).  WL=cfg2|cfg3|cfg5."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, habdec_amd
w = dict(bench.WORKLOADS[os.environ.get("WL", "cfg3")]); S = w["S"]; C = w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                        lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"])
L = habdec_amd.lib(); f = L.hd_debug_fir_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for i in range(8):
    eng.process_device(ring.data_ptr() + (i % rc) * S * C * 8, C, C)       # synchronous calls: the kernel has the GPU to itself
st = np.zeros(8192 * 8, np.uint64); f(st.ctypes.data, 8192 * 8); st = st.reshape(8192, 8).astype(np.float64)
st = st[st[:, 6] > 0]
t0 = st[:, 0].min()
print("workgroups seen:", len(st), "(of the first 8192)")
print("100 MHz timeline (us): start p0/50/100 = %s ; end p0/50/100 = %s ; lifetime p50/p90 = %s" % (
    np.percentile((st[:, 0] - t0) / 100.0, [0, 50, 100]).round(1).tolist(), np.percentile((st[:, 1] - t0) / 100.0, [0, 50, 100]).round(1).tolist(),
    np.percentile((st[:, 1] - st[:, 0]) / 100.0, [50, 90]).round(1).tolist()))
print("cycles per workgroup (wave 1) [prologue: parameters + slide + head, staging to LDS + barrier, tap loop, outputs + discriminator + stores]:", st[:, 2:6].mean(axis=0).round(0).tolist())
res = (st[:, 1] - st[:, 0]).sum() / max(st[:, 1].max() - t0, 1)
print("mean resident workgroups (of those seen): %.0f" % res)
