"""A/B of engine builds and environment knobs inside ONE process on ONE box (boxes of the pool differ by several per cent, and so does a box from
minute to minute): the synthetic ring is generated once, every variant gets its own engine from its own library, and the variants take turns
round by round.

    python3 tools/micro/ab_step.py [--workload cfg4] [--steps 100] [--warmup 8] [--rounds 3] [--sync] VARIANT...
    VARIANT = label[:ENV=val[,ENV=val...]]     label `default` = habdec_amd/libhabdec_amd.so, anything else = gpurun_in/variants/libhd_<label>.so
                                              (HD_BUILD_VARIANT=<label> HD_EXTRA_FLAGS="-D..." python3 -m habdec_amd.build); a label may carry a
                                              suffix after `+` to tell two environment settings of one library apart (nt+a:HD_X=1)

Per variant and round: ms per step over the timed region (flush included), the front kernel's average launch time from the engine's HIP events,
and a fingerprint of the decoded output (bits of the first 64 streams, sentences) that must not differ between variants.
"""
import argparse, ctypes as C, os, sys, time, zlib
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch, bench, habdec_amd
from habdec_amd import capi, engine


def load(label):
    base = label.split("+")[0]
    path = capi.LIB_PATH if base == "default" else ROOT / "gpurun_in" / "variants" / f"libhd_{base}.so"
    if "+" in label:                                    # (its own copy: a few knobs are read once per loaded library)
        import shutil, tempfile
        cp = Path(tempfile.mkdtemp()) / f"lib_{label.replace('+', '_')}.so"
        shutil.copy(path, cp); path = cp
    L = C.CDLL(str(path))
    for table in (capi.ENGINE_API, capi.HOST_API):
        for name, (res, args) in table.items():
            if not hasattr(L, name): continue               # (an older build of the library: the entry points the loop uses are all there)
            f = getattr(L, name); f.restype, f.argtypes = res, args
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4"); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--sync", action="store_true"); ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--timing", type=int, default=3)
    ap.add_argument("--no-offsets", action="store_true", help="cfg4 without its far-off-tune streams (every stream within +-200 Hz... of nothing: all on tune)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    w = dict(bench.WORKLOADS[a.workload]); S = a.streams or w["S"]; Cn = w["C"]
    if a.no_offsets: w["offsets"] = False
    dev = torch.device("cuda", 0)
    ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
    base = ring.data_ptr()
    specs = []
    for v in a.variants:
        label, _, envs = v.partition(":")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        specs.append((v, load(label), env))
    res = {v: [] for v, _, _ in specs}
    for r in range(a.rounds):
        for v, L, env in specs:
            for k, x in env.items(): os.environ[k] = x
            capi._lib = L                                   # (Engine() and check() go through capi.lib())
            eng = engine.Engine(n_streams=S, max_chunk=Cn, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                                lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], pipeline=0 if a.sync else 2,
                                **({"arith": int(env["ARITH"])} if "ARITH" in env else {}))      # (ARITH=1: the library's fast mode -- a variant spec like default+f:ARITH=1)
            eng.set_timing(a.timing)
            for i in range(a.warmup): eng.process_device(base + (i % rc) * S * Cn * 8, Cn, Cn)
            eng.flush(); torch.cuda.synchronize()
            for k in env: os.environ.pop(k, None)           # (some knobs are read at the first call, not at engine creation)
            fr = []; seen = eng.timing()["timed_calls"]
            t0 = time.perf_counter()
            for i in range(a.warmup, a.warmup + a.steps):
                eng.process_device(base + (i % rc) * S * Cn * 8, Cn, Cn)
                t = eng.timing()
                if t["timed_calls"] != seen: seen = t["timed_calls"]; fr.append(t["ms_front"])
            t1 = time.perf_counter()
            eng.flush(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tm = eng.timing()
            fp = zlib.crc32(repr(([eng.bits_total(s) for s in range(min(S, 64))], [eng.take_chars(s) for s in range(min(S, 64))], eng.sentences_ok())).encode())
            res[v].append((dt / a.steps * 1e3, float(np.mean(fr)) * 1e3 if fr else float("nan"), (time.perf_counter() - t1) * 1e3, tm["path"], tm["step_variant"], fp))
            eng.close()
    fps = {x[5] for v in res for x in res[v]}
    print(f"workload {a.workload} S={S} steps={a.steps} rounds={a.rounds} {'sync' if a.sync else 'batch'}; output fingerprints {'EQUAL' if len(fps) == 1 else 'DIFFER: ' + str({v: [hex(x[5]) for x in res[v]] for v in res})}")
    for v in res:
        ms = [x[0] for x in res[v]]; us = [x[1] for x in res[v]]; dr = [x[2] for x in res[v]]
        print(f"{v:44s} ms/step {' '.join(f'{m:.4f}' for m in ms)}  (min {min(ms):.4f})  | launch us {' '.join(f'{u:.1f}' for u in us)} | drain ms {' '.join(f'{d:.2f}' for d in dr)} | path {res[v][0][3]} variant {res[v][0][4]}")


if __name__ == "__main__":
    main()
