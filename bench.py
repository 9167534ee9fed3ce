#!/usr/bin/env python3
"""bench.py -- batched RTTY demodulation throughput on MI355X (one process per GPU, no collective on the data path).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg1|cfg2|cfg3|cfg5] [--streams S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N --threads        # ONE process: an engine + a host thread per device; no torchrun, no RCCL

A "step" is one pushSamples()+operator()() round for every stream of the batch (reference
code/websocketServer/main.cpp:240-245): S streams x C IQ samples already resident in HBM go through the whole
chain (decimation -> [DC] -> spectrum/AFC -> low-pass FIR -> FSK discriminator -> symbol extractor on the GPU;
RTTY framing, sentence extraction, CRC on the host).  By default the engine runs its batch (pipelined) mode: up to three
calls are undelivered and each call's text is delivered, in order, while the next calls run; everything is delivered
inside the timed region (hd_flush before the closing barrier).  --sync delivers every step's text before the next step
starts, like Decoder::operator().  Metric: input complex samples consumed per second over all GPUs (IQ Msamples/s).

TWO timed regions per run, in one process: `cold` -- K steps right behind the W warm-up steps, the protocol of rounds 1-3 --
and the sustained one -- the same K steps behind --prewarm untimed steps of the same loop (default: whole passes over the ring, at least 120 steps)
and W more warm-up steps.  The loop's first ~150 launches run about 10 % below the rate it then holds (the power controller settling at the
1400 W cap), whatever kept the GPU busy before, and the driver's timed region is 20 steps long.  `value` is the sustained region; `cold` carries the
other one; --prewarm 0 times the cold region only.  The kernel sample of `roofline` is taken in the sustained state: the timed steps that carried
HIP events plus a sampling pass behind the timed region (at least 30 launches in all).

`--gpus N` started WITHOUT torchrun (and without --threads) starts N ranks itself: the parent -- before it imports torch or touches HIP -- runs
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`, relays rank 0's line and exits with the children's code.
A world size that differs from --gpus is an error, never a silently smaller run.

The default workload is BASELINE.json configs[3] per GPU: 1024 streams @ 2.048 MS/s, /64, 50 baud 7N2, spectrum +
AFC every call ("cfg4" in SURVEY.md's 1-based numbering) -- the batched 2.048 MS/s configuration that exists at
1/2/4/8 GPUs, sharded 1024 streams per GPU (weak scaling).  Each stream carries its own CRC-valid telemetry
sentence in a seamless ring of HBM-resident chunks; the line reports how many sentences were decoded.

Output: ONE JSON line on rank 0 (contract in the round prompt) with `roofline` for the dominant kernel -- in batch mode
the step kernel (this call's first decimation stage, the only code that touches full-rate IQ, with the previous call's
stream tails riding in the same launch); its duration is measured with HIP events on the engine's own stream -- and
`cpu_baseline` (the CPU oracle timed on ALL of this box's host cores on a bounded sample of the same streams, whose
characters, bit counts and sentences are also compared with the GPU's).  `also` carries BASELINE configs[2] (/4, the
VALU-bound single-GPU configuration) measured in the same run.
"""
from __future__ import annotations

import argparse
import collections
import gc
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
HBM_COPY_GBS = 6290.0          # measured float4 copy on gfx950 (same table)
VALU_NONFMA_TFLOPS = 78.6      # FP32 vector peak with separately rounded multiply and add (157.3 TF with FMA)
VALU_FMA_TFLOPS = 157.3        # ... with fused multiply-add: what the fast mode's FIR chain is priced against

WORKLOADS = {
    # name: fs, decimation, baud, bits, stops, streams/GPU, chunk, lowpass_bw, lowpass_trans, ungated, carrier offsets
    "cfg4": dict(fs=2.048e6, D=64, baud=50, bits=7, stops=2, S=1024, C=65536, lp_bw=1500.0, lp_trans=0.025, ungated=False, offsets=True,
                 desc="BASELINE configs[3] per GPU: 1024 streams @ 2.048 MS/s, /64 (/32 212 taps + /2 69 taps), 50 baud 7N2, 161-tap low-pass, spectrum+AFC on every completed 4096-sample buffer (every 4th call)"),
    "cfg1": dict(fs=2.048e6, D=64, baud=300, bits=8, stops=2, S=1024, C=65536, lp_bw=1500.0, lp_trans=0.025, ungated=False, offsets=False,
                 desc="configs[0] batched: 1024 streams @ 2.048 MS/s, /64, 300 baud 8N2"),
    "cfg2": dict(fs=2.5e6, D=16, baud=300, bits=8, stops=2, S=1024, C=65536, lp_bw=3000.0, lp_trans=0.025, ungated=False, offsets=False,
                 desc="configs[1] batched: 1024 streams @ 2.5 MS/s, /16 (/8 54 taps + /2), low-pass 3 kHz, 300 baud 8N2"),
    "cfg3": dict(fs=2.048e6, D=4, baud=300, bits=8, stops=2, S=1024, C=65536, lp_bw=1500.0, lp_trans=0.025, ungated=True, offsets=False,
                 desc="configs[2]: 1024 streams @ 2.048 MS/s, /4 (139 taps), stage-level FIR/demod/symbols at 512 kHz (reference's 160 kHz gate lifted)"),
    "cfg5": dict(fs=10e6, D=256, baud=300, bits=8, stops=2, S=512, C=1 << 20, lp_bw=1500.0, lp_trans=4.0 / 4096, ungated=False, offsets=False,
                 desc="configs[4] per GPU: 512 streams @ 10 MS/s, /256 (/64 348 taps + /4 139 taps), 4097-tap low-pass, 2^20-sample pushes"),
}


def bytes_per_sample(D: int) -> float:
    """Algorithmic HBM bytes per input sample of the whole chain: 8 + 12/D."""
    return 8.0 + 12.0 / D


def flops_per_sample(w) -> float:
    """Multiply-add flops of the FIR chain per input sample (complex x real MAC = 4 flop), exact mode."""
    stages = {64: [(32, 212), (2, 69)], 16: [(8, 54), (2, 69)], 4: [(4, 139)], 256: [(64, 348), (4, 139)]}[w["D"]]
    f, d = 0.0, 1
    for r, t in stages:
        d *= r
        f += 4.0 * t / d
    taps = 4097 if w["D"] == 256 else 161
    return f + 4.0 * taps / w["D"]


def shard(rank: int, world: int, streams_per_gpu: int):
    """Streams are independent: rank r owns global stream ids [r*S, (r+1)*S) -- weak scaling, no data-path collective."""
    return range(rank * streams_per_gpu, (rank + 1) * streams_per_gpu)


def job_time(dist, dt: float, device=None) -> float:
    """Whole-job time of the timed region = the slowest rank's (MAX all-reduce); identity without a process group."""
    if dist is None:
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64)          # a host tensor: the process group is gloo (no RCCL on a collective-free data path)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def stream_text(rank: int, s: int) -> str:
    from habdec_amd import synth
    return synth.make_sentence(f"R{rank}S{s:04d}", f"{s % 10},52,21")      # 23 characters: just above the 20-char scan threshold


def generate_ring(torch, dev, w, S, rank, seed):
    """HBM-resident synthetic IQ as a ring of push slabs [ring_chunks][S][C] cf32 (one slab = what one batched push
    hands over: S streams x C samples, contiguous).  Stream s = its own sentence rendered as continuous-phase 2-FSK,
    padded with mark idle to whole chunks; carrier nudged (< 0.5 Hz) so the ring wraps phase-continuously."""
    from habdec_amd import synth
    fs, baud, C = w["fs"], w["baud"], w["C"]
    texts = [stream_text(rank, s) for s in range(S)]
    bits = np.stack([synth.rtty_bits(t, w["bits"], w["stops"], 2, 0) for t in texts])          # equal lengths by construction
    nb = bits.shape[1]
    ring_chunks = int(math.ceil(nb * fs / baud / C))
    L = ring_chunks * C
    rng = np.random.default_rng(seed)
    if w["offsets"]:   # 7/8 of the streams within the decodable +-200 Hz, 1/8 far off (AFC work only)
        f0 = np.where(np.arange(S) % 8 == 7, rng.uniform(300, 2000, S) * rng.choice([-1, 1], S), rng.uniform(-200, 200, S))
    else:
        f0 = np.zeros(S)
    out = torch.empty((ring_chunks, S, C, 2), dtype=torch.float32, device=dev)
    k = torch.arange(L, device=dev, dtype=torch.int64)
    bit_idx = torch.clamp(torch.div(k * int(round(baud * 1000)), int(round(fs * 1000)), rounding_mode="floor"), max=nb)
    bits_t = torch.from_numpy(np.concatenate([bits, np.ones((S, 1), np.uint8)], axis=1)).to(dev)   # index nb = idle mark
    k1 = (k + 1).to(torch.float64)
    B = max(1, min(S, int(2.0e9 // (L * 8))))
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    for s0 in range(0, S, B):
        s1 = min(S, s0 + B)
        sgn = bits_t[s0:s1].index_select(1, bit_idx).to(torch.float64) * 2.0 - 1.0
        cum = torch.cumsum(sgn, dim=1)
        f0b = torch.from_numpy(f0[s0:s1]).to(dev)[:, None]
        total = (f0b[:, 0] * L + 250.0 * cum[:, -1]) / fs
        f0b = f0b + ((torch.round(total) - total) * fs / L)[:, None]
        cyc = (f0b * k1 + 250.0 * cum) / fs
        ph = (cyc - torch.floor(cyc)) * (2.0 * math.pi)
        del sgn, cum, cyc
        noise = torch.randn((s1 - s0, L, 2), device=dev, dtype=torch.float32, generator=gen) * 0.08
        nb_ = s1 - s0
        out[:, s0:s1, :, 0] = ((0.5 * torch.cos(ph)).to(torch.float32) + noise[..., 0]).view(nb_, ring_chunks, C).transpose(0, 1)
        out[:, s0:s1, :, 1] = ((0.5 * torch.sin(ph)).to(torch.float32) + noise[..., 1]).view(nb_, ring_chunks, C).transpose(0, 1)
        del ph, noise
    return out, ring_chunks, texts


def cpu_baseline(w, host_iq, chunks, C, lookup_mode=1):
    """Oracle (CPU restatement of the reference chain, oracle/liboracle.so) on the host cores: one decoder per thread (std::thread
    inside the library), each fed the chunk sequence the GPU consumed on its stream, repeated with fresh decoders.  The thread count
    is calibrated first -- all logical CPUs, half, a quarter, ... for about a second each (containers often grant fewer cores than
    os.cpu_count() shows, and then more threads only get in each other's way) -- and the best one runs the ~12 s measurement.
    Returns (MS/s, threads used, sample description, calibration table)."""
    from oracle import pyoracle
    kw = dict(fs=w["fs"], factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"],
              lowpass_trans=w["lp_trans"], mathh_context=lookup_mode, ungated=w["ungated"])
    per_pass = len(chunks) * C
    t_one, _ = pyoracle.bench_run(host_iq[:1], chunks, C, 1, **kw)
    nmax = len(host_iq)
    cand = sorted({n for n in (nmax, nmax // 2, nmax // 4, nmax // 8, 64, 32, 16, 8) if 1 <= n <= nmax}, reverse=True)
    table = {}
    for n in cand:
        rep = int(min(max(round(1.0 / max(t_one, 1e-3)), 1), 200))
        dt, _ = pyoracle.bench_run(host_iq[:n], chunks, C, rep, **kw)
        table[n] = n * rep * per_pass / dt / 1e6
    best = max(table, key=table.get)
    repeats = int(min(max(round(12.0 * table[best] * 1e6 / (best * per_pass)), 1), 5000))
    dt, _ = pyoracle.bench_run(host_iq[:best], chunks, C, repeats, **kw)
    total = best * repeats * per_pass
    sample = (f"{best} streams x {len(chunks)} chunks of {C} samples x {repeats} repeats = {total / 1e6:.0f} MS in {dt:.1f} s; one oracle decoder per "
              f"thread, {best} threads; single-thread rate {per_pass / t_one / 1e6:.1f} MS/s")
    return total / dt / 1e6, best, sample, {str(k): round(v, 1) for k, v in table.items()}


OracleLog = collections.namedtuple("OracleLog", "sentences chars bits demod_hash")
ORACLE_LOG_CACHE = {}        # (decoder parameters, chunk sequence, sums over the samples, stream) -> OracleLog; this process only


def oracle_sample_check(w, eng, ring, chunks, C, streams, lookup_mode=1, group=64, exact_floats=True):
    """The self-check every bench line carries: the oracle decodes the chunk sequence the engine consumed on `streams` (one pass, one thread per
    stream, `group` streams at a time: a group's share of the ring is transposed on the GPU and brought over in one copy) and the engine's symbols
    produced, characters and sentences per stream must equal the oracle's.  The headline line checks EVERY stream of the shard."""
    from oracle import pyoracle
    kw = dict(fs=w["fs"], factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"],
              lowpass_trans=w["lp_trans"], mathh_context=lookup_mode, ungated=w["ungated"])
    nuse = max(chunks) + 1
    streams = list(streams)
    t0 = time.perf_counter()
    n_chars = n_bits = n_sent = n_ck = 0
    diff = []
    contiguous = streams == list(range(streams[0], streams[0] + len(streams)))
    # The oracle's answer is a function of its input alone: a second check over the SAME samples in the same order (the fast leg behind the exact leg of one
    # command; the suite's fast twin of an exact all-stream test) reads it from here instead of decoding everything again.  The key carries the decoder's
    # parameters, the chunk sequence and two 64-bit sums over the bit patterns of every sample the check touches.
    import torch
    fp = [0, 0]
    for i in range(nuse):
        v = ring[i].view(torch.int32)
        fp[0] += int(v.sum(dtype=torch.int64)); fp[1] += (i + 1) * int(v[..., 0].sum(dtype=torch.int64)) + int((v[..., 1] >> 3).sum(dtype=torch.int64))
    key0 = (json.dumps(kw, sort_keys=True), tuple(chunks), C, tuple(ring.shape), fp[0], fp[1])
    reused = 0
    for g0 in range(0, len(streams), group):
        grp = streams[g0:g0 + group]
        logs = [ORACLE_LOG_CACHE.get(key0 + (s,)) for s in grp]
        if any(lg is None for lg in logs):
            if contiguous:      # [nuse, G, C, 2] -> [G, nuse * C] complex64 on the device, one D2H copy
                blk = ring[:nuse, grp[0]:grp[0] + len(grp)].transpose(0, 1).contiguous().cpu().numpy().view(np.complex64).reshape(len(grp), -1)
                host_iq = [blk[i] for i in range(len(grp))]
            else:
                host_iq = [ring[:nuse, s].cpu().numpy().view(np.complex64).reshape(-1) for s in grp]
            _, blogs = pyoracle.bench_run(host_iq, chunks, C, 1, **kw)
            del host_iq
            logs = [OracleLog(list(lg), lg.chars, lg.bits, lg.demod_hash) for lg in blogs]
            for s, lg in zip(grp, logs):
                ORACLE_LOG_CACHE[key0 + (s,)] = lg
        else:
            reused += len(grp)
        for s, lg in zip(grp, logs):
            got = (eng.take_sentences(s), eng.take_chars(s), eng.bits_total(s))
            want = (lg.sentences, lg.chars, lg.bits)
            n_chars += len(lg.chars); n_bits += lg.bits; n_sent += len(lg.sentences)
            if exact_floats:                                # exact mode: every call's discriminator output, bit for bit, through the folded checksums
                nck, unk, h = eng.demod_checksum_total(s)
                if unk == 0 and nck == len(chunks):
                    n_ck += nck
                    got, want = got + (h,), want + (lg.demod_hash,)
            for f, g_, o_ in zip(("sentences", "chars", "bits", "discriminator output of every call"), got, want):
                if g_ != o_:
                    if not diff:
                        sys.stderr.write(f"[bench] self-check mismatch on stream {s} ({f}): gpu bits {got[2]} chars {got[1]!r} / oracle bits {want[2]} chars {want[1]!r}\n")
                    diff.append((int(s), f))
    return {"gpu_matches_oracle_on_sample": ((not diff) if n_bits else None), "mismatches": diff[:8],
            "compared": "per stream: symbols produced, characters emitted, sentences" + (", and every call's discriminator output bit for bit (folded checksums)" if exact_floats else "") + " -- GPU engine vs oracle over every step the engine took (both timed regions, pre-warm, warm-up, sampling pass)",
            "streams_in_sample": [int(x) for x in streams] if len(streams) <= 8 else len(streams), "all_streams_of_the_shard": len(streams) == eng.S,
            "bits_in_sample": n_bits, "chars_in_sample": n_chars, "sentences_in_sample": n_sent, "steps_checked": len(chunks),
            "oracle_logs_reused": reused,                   # streams whose oracle log came from an earlier check over the same samples in this process (0 = all decoded here)
            "discriminator_checksums_compared": n_ck,       # (stream, call) pairs whose discriminator output was compared bit for bit (exact mode; 0 in fast mode)
            "check_seconds": round(time.perf_counter() - t0, 1)}


def normwise(a, b):
    """max|a - b| / max|b| (how 'within 1e-5 relative' is evaluated for a buffer of floats); inf when the shapes differ."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return float("inf")
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def fir_normwise(a, b, fir_input):
    """The norm-wise difference of a FIR's OUTPUT, relative to the larger of the output's and the input's peak: a filter that rejects what it is fed (a
    carrier outside the low-pass: every 8th stream of the headline workload) leaves an output far below its input, and the rounding of ANY float
    summation -- the reference's own included -- scales with the terms that cancel (sum |x[t] k[t]|), not with what is left of them.  The low-pass has
    unit gain at DC (FirFilter.h:198-208), so the input's peak is that scale."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return float("inf")
    if a.size == 0:
        return 0.0
    scale = max(float(np.max(np.abs(b))), float(np.max(np.abs(fir_input))) if np.size(fir_input) else 0.0, 1e-30)
    return float(np.max(np.abs(a - b)) / scale)


def demod_excess(gd, od, filt_ref, prev_ref, tol=1e-5, scale=None):
    """The discriminator output of the fast mode against the oracle's: d[i] = arg(y[i] conj(y[i-1])) is the exact mode's code on inputs that differ by
    <= tol * max|y|, so its output may differ by the propagated bound (|dy[i]| / |y[i]| + |dy[i-1]| / |y[i-1]| radians, angles compared modulo 2 pi) --
    1e-5 of pi where the filtered samples are strong, more where y passes near zero (element-wise 1e-5 cannot hold at zero
    crossings).  Returns (largest |difference| / allowed, largest |difference| in radians)."""
    gd, od = np.asarray(gd, np.float64), np.asarray(od, np.float64)
    if gd.shape != od.shape:
        return float("inf"), float("inf")
    if od.size == 0:
        return 0.0, 0.0
    y = np.abs(np.asarray(filt_ref).astype(np.complex128))
    yp = np.concatenate([[abs(complex(prev_ref)) if prev_ref is not None else y[0]], y[:-1]])
    peak = max(float(y.max()), float(scale or 0.0), 1e-30)        # (scale: the peak of the low-pass's input, see fir_normwise)
    d = np.abs(np.angle(np.exp(1j * (gd - od))))
    allowed = np.maximum(tol * math.pi, 1.5 * tol * peak * (1.0 / np.maximum(y, 1e-30) + 1.0 / np.maximum(yp, 1e-30)))
    return float(np.max(d / allowed)), float(np.max(d))


def float_parity_probe(w, ring, ring_chunks, S, C, device_index, steps=6, streams=None):
    """Fast mode only: a short synchronous pass of its own (the first `steps` slabs of the ring, keep_filtered) whose decimated, filtered and discriminator
    floats are compared with the oracle's call by call on a few streams -- norm-wise, the way north_star's 1e-5 is evaluated."""
    import habdec_amd
    from oracle import pyoracle
    streams = sorted({0, min(7, S - 1), S // 2, S - 1}) if streams is None else list(streams)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], device=device_index, pipeline=0, keep_filtered=True, arith=1)
    orcs = {s: pyoracle.Decoder("oracle", factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"], lowpass_trans=w["lp_trans"],
                                ungated=w["ungated"]) for s in streams}
    worst = {"decimated": 0.0, "filtered": 0.0, "demod_over_bound": 0.0, "demod_max_abs_rad": 0.0}
    prev = {s: None for s in streams}
    n = 0
    for k in range(min(steps, ring_chunks)):
        eng.process_device(ring[k].data_ptr(), C, C)
        for s, o in orcs.items():
            o(ring[k, s].cpu().numpy().view(np.complex64).reshape(-1), w["fs"])
            worst["decimated"] = max(worst["decimated"], normwise(eng.decimated(s), o.array("last_decimated")))
            fo, do = o.array("last_filtered"), o.array("last_decimated")
            worst["filtered"] = max(worst["filtered"], fir_normwise(eng.filtered(s), fo, do))
            ex, ab = demod_excess(eng.demodulated(s), o.array("last_demod"), fo, prev[s], scale=float(np.max(np.abs(do))) if do.size else None)
            worst["demod_over_bound"] = max(worst["demod_over_bound"], ex); worst["demod_max_abs_rad"] = max(worst["demod_max_abs_rad"], ab)
            if fo.size:
                prev[s] = fo[-1]
            n += int(fo.size)
    eng.close()
    return {"normwise_max": {k: float("%.3g" % v) for k, v in worst.items()}, "tolerance": 1e-5, "streams": streams, "calls": min(steps, ring_chunks), "filtered_samples_compared": n,
            "within_tolerance": bool(worst["decimated"] <= 1e-5 and worst["filtered"] <= 1e-5 and worst["demod_over_bound"] <= 1.0),
            "how": "per call: max|gpu - oracle| / max|oracle| for the decimated samples; for the filtered samples relative to the larger of their own and the low-pass input's peak (fir_normwise); the discriminator output against the bound those 1e-5 propagate to (demod_excess)"}


def box_identity(torch, dev):
    """What this box's GPU delivers right now, so that a kernel time can be attributed (boxes of the pool differ by +-6 %): a 20 ms calibration
    pass -- a 1 GiB device-to-device copy, the access pattern of the guide's 6.29 TB/s float4 copy figure -- and the clock levels the driver
    reports right after it."""
    out = {"gpu": torch.cuda.get_device_name(dev)}
    try:
        n = 1 << 28
        a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a)
        a.fill_(1.0); b.copy_(a); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.device(dev):
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record(); torch.cuda.synchronize(dev)
        out["copy_GBps"] = round(10 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        out["copy_note"] = "1 GiB torch device-to-device copy x10 (read + write bytes), measured after the timed region"
        del a, b
    except Exception as ex:                      # (never let the calibration take the line down)
        out["copy_GBps"] = None; out["copy_note"] = f"calibration failed: {ex}"
    # clocks as the driver publishes them in sysfs (the active level of pp_dpm_sclk / pp_dpm_mclk).  No child process here: this process has
    # initialised the GPU, and a fork + exec from it (rocm-smi is a script) is refused on this pool.
    try:
        import glob
        pr = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xFF, getattr(pr, "pci_device_id", 0))
        card = next((c for c in glob.glob("/sys/class/drm/card[0-9]*/device") if want in os.path.realpath(c)), None)   # (a node shows all its GPUs in sysfs)
        for key, name in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk")):
            if card and os.path.exists(f"{card}/{name}"):
                act = [ln.split(":", 1)[1].replace("*", "").strip() for ln in open(f"{card}/{name}") if "*" in ln]
                out[key + "_after_run"] = act[0] if act else None
    except Exception:
        pass
    return out


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs in /proc/cpuinfo); None where that cannot be read."""
    try:
        seen, pkg = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((pkg, ln.split(":")[1].strip()))
        return len(seen) or None
    except OSError:
        return None


STAGE1 = {64: "k_decimate<32,212,64>", 16: "k_decimate<8,54,256>", 4: "k_decimate<4,139,256>", 256: "k_decimate<64,348,64>"}
STEP = {64: "k_step<32,212,2,69>", 256: "k_step<64,348,4,139>"}
STEP_CU = {64: "k_step_cu<212,2,69>", 128: "k_step_cu<174,4,139>"}     # one workgroup per CU: a stage-1 worker wave and a stream tail per SIMD
STAGE1_CU = {64: "k_stage1_cu<212,32>", 128: "k_stage1_cu<174,32>", 256: "k_stage1_cu<348,64>", 16: "k_stage1_cu<54,8>", 4: "k_stage1_cu<139,4>"}   # stage 1 alone, one workgroup per CU (worker waves at /32 and /64, loader + computing waves at /8 and /4)
PATHS = {0: "separate kernels", 1: "fused back end", 2: "stream tail kernel", 3: "step kernel (stage 1 + previous call's stream tails)"}


class Shard:
    """One GPU's share of the job: its engine, its HBM-resident ring of push slabs and its step loop (streams are independent: nothing is shared)."""

    def __init__(self, torch, w, S, device_index, rank, sync, arith=0):
        import habdec_amd
        self.torch, self.S, self.C, self.dev = torch, S, w["C"], torch.device("cuda", device_index)
        # (the engine first -- allocations, rocFFT plan, code objects: host-side work during which the GPU idles -- then the synthetic ring, whose
        # generation keeps the GPU busy right up to the warm-up steps)
        self.eng = habdec_amd.Engine(n_streams=S, max_chunk=w["C"], sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"],
                                     rtty_stops=w["stops"], lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"],
                                     device=device_index, pipeline=0 if sync else int(os.environ.get("HD_BENCH_PIPELINE", "2")), arith=arith)
        self.ring, self.ring_chunks, self.texts = generate_ring(torch, self.dev, w, S, rank, seed=1234 + rank)
        # The ring is generated on torch's stream, the engine launches on its own queues: without this wait the warm-up steps run while the last
        # streams' slabs are still being written (rounds 1-3 did -- harmless for the timing, but the self-check then compares the oracle's run over the
        # final data with a GPU run that saw unfinished data in its first calls; it went unnoticed because only the first streams were sampled).
        torch.cuda.synchronize(self.dev)
        self.eng.set_timing(3)     # HIP-event brackets on every 3rd call (each record is a barrier packet worth microseconds of queue time; 3, not 4: every 4th
                                   # launch carries the streams' spectra, and the sample must see light and heavy launches in their true proportion)
        self.base = self.ring.data_ptr()
        self.i = 0
        self.front_ms, self.total_ms, self.host_us, self.sensors = [], [], [], None

    def step(self):
        self.eng.process_device(self.base + (self.i % self.ring_chunks) * self.S * self.C * 8, self.C, self.C)
        self.i += 1                  # (steps taken so far: the engine has consumed chunks [0, i) of the ring, taken modulo its length)

    def warm(self, W):
        for _ in range(W):
            self.step()
        self.eng.flush()

    def timed(self, K, every=1):
        """K steps and the flush that delivers the last step's text; `every`: how often the loop asks the engine for its timing record (a thread
        per device shares the interpreter with its siblings: there the loop body is the bare C call on two steps out of three)."""
        eng = self.eng
        seen = eng.timing()["timed_calls"]
        self.front_ms, self.total_ms, self.host_us = [], [], []
        for j in range(K):
            self.step()
            if every == 1 or j % every == every - 1:
                t = eng.timing()
                if t["timed_calls"] != seen:          # (a call that carried the HIP-event brackets)
                    seen = t["timed_calls"]
                    self.front_ms.append(t["ms_front"])
                    self.total_ms.append(t["ms_total"])
                self.host_us.append((t["host_enqueue_us"], t["host_wait_us"], t["host_text_us"]))
        self.t_loop = time.perf_counter()
        eng.flush()          # the last step's text is delivered inside the timed region
        self.torch.cuda.synchronize(self.dev)
        self.t_end = time.perf_counter()


def gpu_sensor_reader(torch, dev):
    """A function that reads this device's shader clock and average board power from sysfs (hwmon freq1_input / power1_average of the card with the
    device's PCI address) -- file reads only: a child process from a process that has initialised the GPU is refused on this pool."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xFF, getattr(pr, "pci_device_id", 0))
        card = next((c for c in glob.glob("/sys/class/drm/card[0-9]*/device") if want in os.path.realpath(c)), None)
        hw = (glob.glob(f"{card}/hwmon/hwmon*") or [None])[0] if card else None
    except Exception:
        hw = None

    def rd(name, scale):
        try:
            return round(int(open(f"{hw}/{name}").read().strip()) / scale, 1)
        except Exception:
            return None

    def read():
        if not hw:
            return {"sclk_mhz": None, "power_w": None}
        pw = rd("power1_average", 1e6)
        return {"sclk_mhz": rd("freq1_input", 1e6), "power_w": pw if pw is not None else rd("power1_input", 1e6)}
    return read


def run_workload(*args, **kwargs):
    """_run_workload with the garbage collector off from before the engines and rings are built until the last timed step is sampled (timeit does the same).  A
    full collection over the heap an earlier leg's all-stream self-check left behind takes tens of milliseconds: inside a region that is the region; between the
    pre-warm pass and a region it is worse than it looks -- an idle GPU drops its clocks within ~20 ms and the next ~150 launches run 10 % slower
    (tools/micro/r06_warm.py: gap rows).  So the collection runs here, before the ring is generated, and nothing between the warm-up passes and the regions leaves
    the GPU idle; _run_workload switches the collector back on behind its sampling pass, this wrapper on every other way out."""
    gc.collect()
    gc.disable()
    try:
        return _run_workload(*args, **kwargs)
    finally:
        gc.enable()


def _run_workload(torch, dist, dev, rank, local_rank, world, name, K, W, S, sync, cpu_leg, threads=0, prewarm=None, arith=0, min_kernel_samples=32):
    """Time K steps of one workload (after W warm-up steps) and describe the result; rank 0 gets the full dictionary.
    cpu_leg: "full" = the CPU baseline (oracle timed on the host cores) + the self-check on every stream; "check" = the self-check alone on a few
    streams; "check_all" = the self-check alone on every stream of the shard (every line carries one of them); threads = N > 0: one process, a Shard +
    host thread per device; arith: 0 = the engine's exact mode (bit-identical floats), 1 = its fast mode (fused multiply-add, tolerance 1e-5)."""
    import threading
    import habdec_amd
    w = dict(WORKLOADS[name])
    S = S or w["S"]
    C = w["C"]
    if threads:
        world = threads
        shards = [None] * threads

        same = bool(os.environ.get("HD_BENCH_SAME_DEVICE"))        # (tests on a one-GPU box: every shard on device 0 -- the threading is what is exercised)

        def make(d):
            shards[d] = Shard(torch, w, S, 0 if same else d, d, sync, arith)
        th = [threading.Thread(target=make, args=(d,)) for d in range(threads)]
        for t in th: t.start()
        for t in th: t.join()
        if any(x is None for x in shards):
            raise SystemExit("bench.py --threads: a device's engine or ring could not be built")
    else:
        shards = [Shard(torch, w, S, local_rank, rank, sync, arith)]
    sh = shards[0]
    eng, ring, ring_chunks, texts = sh.eng, sh.ring, sh.ring_chunks, sh.texts
    K = K or ring_chunks
    # Untimed steps between the two timed regions -- by default one pass over the ring.  The step loop needs its first ~150 launches (20 ms) to reach the
    # rate it then sustains: 20 timed steps take 0.156-0.162 ms each behind 5 warm-up steps, 0.154-0.161 behind 40, 0.142-0.145 behind 150, 0.136-0.142
    # behind 500 (one box, tools/micro/ab_step.py) -- whichever slabs they read, and a busy GPU beforehand does not replace them.  The
    # driver's command (--warmup 5, 20 steps = 3 ms) behind nothing else times that ramp: it is reported as `cold`.  The self-check covers every step.
    P = min(ring_chunks * -(-120 // ring_chunks), 512) if prewarm is None else max(0, int(prewarm))     # (whole passes over the ring, at least 120 steps)
    base = sh.base
    sensors = [gpu_sensor_reader(torch, x.dev) for x in shards]

    def barrier():
        # (device work first, then the ranks meet on the host -- gloo -- and every device is idle on both sides of the region)
        for x in shards:
            torch.cuda.synchronize(x.dev)
        if dist is not None:
            dist.barrier()

    def region():
        """W untimed warm-up steps, then EXACTLY K timed steps between barriers; the whole job's time (MAX over ranks / shards)."""
        for x in shards:
            x.warm(W)
        barrier()
        t0 = time.perf_counter()
        if threads > 1:
            th = [threading.Thread(target=x.timed, args=(K, 3)) for x in shards]
            for t in th: t.start()
            for t in th: t.join()
        else:
            sh.timed(K, 1)
        # This rank's region ends at ITS device synchronize behind the flush of the K-th step (Shard.timed); the closing barrier is the bracket the ranks meet
        # at, and its own latency -- 0.3 / 0.6 / 0.8 ms for 2 / 4 / 8 ranks over gloo, a tenth to a quarter of a 20-step region -- is not time the K steps took.
        d = max(x.t_end for x in shards) - t0          # (several shards in one process: the slowest device's region, as job_time takes the MAX over ranks)
        barrier()
        return job_time(dist, d, dev)

    dt_cold = region()                                  # behind --warmup only: what rounds 1-3 timed
    cold = {"value": round(world * S * C * K / dt_cold / 1e6, 1), "ms_per_step": round(dt_cold / K * 1e3, 4), "timed_region_ms": round(dt_cold * 1e3, 2),
            "steps": K, "warmup": W, "note": "the same K steps timed right behind the --warmup steps, before the pre-warm pass (the protocol of rounds 1-3)"}
    cold_kernel_ms = list(sh.front_ms)
    if P:
        for x in shards:
            x.warm(P)
        dt = region()                                   # the sustained region: `value`
    else:
        dt = dt_cold
    drain_ms = (sh.t_end - sh.t_loop) * 1e3
    front_ms, total_ms, host_us = list(sh.front_ms), list(sh.total_ms), list(sh.host_us)
    # The kernel sample of `roofline`: the timed steps that carried HIP events, topped up -- outside the timed region, same loop, same state -- to at
    # least 32 launches (the driver's 20 steps alone carry six or seven).  The device's shader clock and board power are read at the end of that pass,
    # with its last launches still queued (two sysfs reads, ~0.3 ms of driver time: NOT inside a timed region).
    n_region = len(front_ms)
    need = max(0, min_kernel_samples - n_region)
    if not threads:
        seen = eng.timing()["timed_calls"]
        for _ in range(max(3 * need + 3, 48 if min_kernel_samples >= 32 else 6)):      # (the suite's all-stream tests ask for a handful: every step taken is a step the oracle decodes)
            sh.step()
            t = eng.timing()
            if t["timed_calls"] != seen:
                seen = t["timed_calls"]
                front_ms.append(t["ms_front"]); total_ms.append(t["ms_total"])
        sh.sensors = sensors[0]()
        eng.flush()
        torch.cuda.synchronize(sh.dev)
    else:
        for x, rd in zip(shards, sensors):
            x.sensors = rd()
    gc.enable()                                         # (the self-check and the CPU baseline below allocate in earnest)
    rank_sensors = [x.sensors or {"sclk_mhz": None, "power_w": None} for x in shards]
    if dist is not None:                                # every rank's clock and power, gathered on rank 0 (a 2-float all-gather, outside the timed regions)
        mine = torch.tensor([rank_sensors[0]["sclk_mhz"] or -1.0, rank_sensors[0]["power_w"] or -1.0], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_sensors = [{"sclk_mhz": (float(a[0]) if float(a[0]) >= 0 else None), "power_w": (float(a[1]) if float(a[1]) >= 0 else None)} for a in allr]
    steps_taken = sh.i
    tm = eng.timing()
    front_bytes, path = tm["front_bytes"], tm.get("path", 0)
    sentences_ok = eng.sentences_ok()
    for x in shards[1:]:
        x.eng.close()
        x.ring = None
    chunks = [i % ring_chunks for i in range(steps_taken)]     # what the engine consumed: both regions, the pre-warm pass, the sampling pass (the self-check follows all of it)
    if rank != 0:
        # Every rank checks ITS shard against the oracle -- behind rank 0's CPU-baseline timing, which has the host cores to itself
        # until the barrier -- and rank 0's line carries the verdicts (per_rank[i].gpu_matches_oracle).
        dist.barrier()
        mine = oracle_sample_check(w, eng, ring, chunks, C, list(range(S)) if cpu_leg in ("full", "check_all") else sorted({0, min(7, S - 1), S // 2, S - 1}),
                                   group=max(8, 128 // world), exact_floats=not arith)
        dist.gather_object({k: mine[k] for k in ("gpu_matches_oracle_on_sample", "all_streams_of_the_shard", "bits_in_sample", "sentences_in_sample", "mismatches", "check_seconds")}, None, dst=0)
        eng.close()
        return None

    value = world * S * C * K / dt / 1e6
    if not front_ms:                            # fewer steps than the timing period
        front_ms.append(tm["ms_front"]); total_ms.append(tm["ms_total"])
    avg_front_ms = float(np.mean(front_ms))
    # The timed kernel and its algorithmic bytes per launch: the step kernel does one call's whole chain --
    # stage 1 of this call and the stream tails of the previous one -- B(D) = 8 + 12/D bytes per input sample;
    # a stage-1 kernel on its own reads the IQ slab and writes its decimated output.
    if path == 3:
        kernel, alg_bytes = (STEP_CU if tm.get("step_variant") == 1 else STEP).get(w["D"], "k_step"), int(S * C * bytes_per_sample(w["D"]))
    else:
        kernel, alg_bytes = (STAGE1_CU if tm.get("step_variant") == 1 else STAGE1).get(w["D"], "k_decimate"), int(front_bytes)
    achieved = alg_bytes / (avg_front_ms * 1e-3) / 1e9
    traffic, prof = None, {}
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists():
        try:
            prof = json.loads(tf.read_text()).get(name + ("_sync" if sync else "") + ("_fast" if arith else ""), {})
            traffic = prof.get("front_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic, prof = None, {}
    traffic_note = None
    if prof.get("front_kernel") != kernel.replace(" ", "") + ("[fast]" if arith else ""):
        traffic, prof = None, {}                               # (the committed profile is of another kernel: say nothing rather than the wrong thing)
    elif prof:
        from habdec_amd.build import source_id
        if prof.get("csrc_id") != source_id():                 # ... or of the same kernel NAME built from other sources
            traffic_note = f"profiles/traffic.json was recorded on sources {prof.get('csrc_id')}, the loaded library's are {source_id()}: traffic and the rocprof figures are not quoted"
            traffic, prof = None, {}
    if prof.get("rocprof_avg_launch_ms"):
        # the committed rocprofv3 --kernel-trace --stats summary of the same command (tools/collect_profiles.py): the live figure must agree with it
        rp = {"rocprof_avg_launch_ms": prof["rocprof_avg_launch_ms"], "rocprof_launches": prof.get("rocprof_launches"), "rocprof_source": prof.get("stats_source"),
              "frac_rocprof": round(alg_bytes / (prof["rocprof_avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    else:
        rp = {}
    valu = w["D"] == 4                                      # configs[2]: the FIR chain's multiply-adds bind, not HBM
    tflops = value / world * 1e6 * flops_per_sample(w) / 1e12
    res = {
        "prewarm_steps": P, "cold": cold,
        "per_rank": [dict(rank=i, **x) for i, x in enumerate(rank_sensors)],
        "value": round(value, 1), "ms_per_step": round(dt / K * 1e3, 4), "steps": K, "S": S, "C": C, "ring_chunks": ring_chunks, "w": w,
        "timed_region_ms": round(dt * 1e3, 2),
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_front_ms, 5),
                     "n_samples": len(front_ms), "min_launch_ms": round(float(np.min(front_ms)), 5), "max_launch_ms": round(float(np.max(front_ms)), 5),
                     "n_samples_in_timed_region": n_region, "cold_avg_launch_ms": round(float(np.mean(cold_kernel_ms)), 5) if cold_kernel_ms else None,
                     "sampling": "HIP events on the engine's queue (riding on the dispatch packet) around every 3rd launch of the sustained timed region, topped up to >= 32 launches by a sampling pass of the same loop right behind it",
                     "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBS, 4)},
        "pipeline": {"bytes_per_sample": round(bytes_per_sample(w["D"]), 3),
                     "hbm_frac_end_to_end": round(value / world * 1e6 * bytes_per_sample(w["D"]) / 1e9 / HBM_PEAK_GBS, 4),
                     "launch_path": PATHS.get(path, str(path)),
                     "call_latency_ms": round(float(np.mean(total_ms)), 4),
                     "sentences_ok_rank0": int(sentences_ok),
                     "drain_ms": round(drain_ms, 3),
                     "host_us_enqueue_p50_p90_max": [round(float(np.percentile([h[0] for h in host_us], q)), 1) for q in (50, 90, 100)],
                     "host_us_per_step": {"enqueue": round(float(np.mean([h[0] for h in host_us])), 1),
                                          "wait_gpu": round(float(np.mean([h[1] for h in host_us])), 1),
                                          "text_stage": round(float(np.mean([h[2] for h in host_us])), 1)},
                     "mode": "sync" if sync else "batch (calls pipelined; up to three undelivered)"},
    }
    res["roofline"].update(rp)
    if valu:
        vpeak = VALU_FMA_TFLOPS if arith else VALU_NONFMA_TFLOPS
        res["roofline"] = {"bound": "valu", "kernel": "FIR chain: " + kernel + " + k_fir_demod (" + ("fast mode: fused multiply-add" if arith else "exact mode: separately rounded multiply and add") + ")",
                           "achieved": round(tflops, 2), "peak": vpeak, "unit": "TFLOP/s", "frac": round(tflops / vpeak, 4),
                           "traffic": traffic, "algorithmic_flop_per_sample": round(flops_per_sample(w), 1),
                           "stage1_hbm": {"achieved": round(achieved, 1), "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_front_ms, 5)}}
    if traffic_note:
        res["roofline"]["traffic_note"] = traffic_note
    if dt * 1e3 < 50.0:
        res["pipeline"]["note"] = f"timed region is only {dt * 1e3:.1f} ms ({K} steps): start-up and the final flush weigh in; --steps 0 times one pass over the ring"
    if not sync and path in (0, 1, 2, 3):
        # In batch mode the timed kernel shares the GPU (or its own launch) with the previous call's back half.  A short synchronous
        # pass (outside the timed region, own engine) gives the stage-1 kernel's isolated duration next to the contract figure above.
        eng1 = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"],
                                 rtty_stops=w["stops"], lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"],
                                 device=local_rank, pipeline=False, arith=arith)
        eng1.set_timing(1)
        iso = []
        for i in range(4 + 24):
            eng1.process_device(base + (i % ring_chunks) * S * C * 8, C, C)
            if i >= 4:
                iso.append(eng1.timing()["ms_front"])
        iso_bytes, iso_cu = eng1.timing()["front_bytes"], eng1.timing().get("step_variant") == 1
        eng1.close()
        iso_ms = float(np.mean(iso))
        iso_bw = iso_bytes / (iso_ms * 1e-3) / 1e9
        tgt = res["roofline"] if not valu else res["roofline"]["stage1_hbm"]
        tgt["isolated"] = {"kernel": (STAGE1_CU if iso_cu else STAGE1).get(w["D"], "k_decimate"), "algorithmic_bytes_per_launch": int(iso_bytes), "avg_launch_ms": round(iso_ms, 5),
                           "achieved": round(iso_bw, 1), "frac": round(iso_bw / HBM_PEAK_GBS, 4),
                           "frac_of_measured_copy_peak": round(iso_bw / HBM_COPY_GBS, 4),
                           "note": "stage 1 alone: synchronous calls (nothing else on the GPU), 24 launches after the timed region"}
    res["box"] = box_identity(torch, dev)
    chunks_cpu = [i % ring_chunks for i in range(W + K)]       # the CPU baseline's bounded sample of the same workload
    if cpu_leg == "full":
        nproc = os.cpu_count() or 1
        nthreads = int(min(nproc, S))
        nuse = min(ring_chunks, W + K)                         # (the chunks the run touched: no need to bring the whole ring over)
        host_iq = [ring[:nuse, s].cpu().numpy().view(np.complex64).reshape(-1) for s in range(nthreads)]
        v, c, sample, calib = cpu_baseline(w, host_iq, chunks_cpu, C)
        del host_iq
        res["cpu_baseline"] = {"value": round(v, 1), "unit": "MS/s", "cores": min(c, physical_cores() or c), "threads": c, "nproc": nproc,
                               "physical_cores": physical_cores(), "kind": "port", "sample": sample,
                               "threads_calibration_MSps": calib}
        if dist is not None:
            dist.barrier()                                     # the other ranks' self-checks start here: the timing above had the host cores to itself
        res["cpu_baseline"].update(oracle_sample_check(w, eng, ring, chunks, C, list(range(S)), group=max(8, 128 // world) if world > 1 else 64, exact_floats=not arith))      # every stream of the shard
    elif cpu_leg in ("check", "check_all"):
        # no CPU timing asked for: the line still says whether what it timed decodes what the oracle decodes (a far-off-tune stream among them)
        res["cpu_baseline"] = {"value": None, "kind": "port", "note": "self-check only (--no-cpu-baseline / secondary workload): the oracle was not timed"}
        if dist is not None:
            dist.barrier()
        res["cpu_baseline"].update(oracle_sample_check(w, eng, ring, chunks, C, list(range(S)) if cpu_leg == "check_all" else sorted({0, min(7, S - 1), S // 2, S - 1}),
                                                       group=max(8, 128 // world) if world > 1 else 64, exact_floats=not arith))
    if "cpu_baseline" in res:
        # every rank's verdict on its own shard beside its clock and power (rank 0's is the line's cpu_baseline block)
        mine = {k: res["cpu_baseline"][k] for k in ("gpu_matches_oracle_on_sample", "all_streams_of_the_shard", "bits_in_sample", "sentences_in_sample", "mismatches", "check_seconds")}
        allc = [mine]
        if dist is not None:
            allc = [None] * world
            dist.gather_object(mine, allc, dst=0)
        for i, c in enumerate(allc):
            if i < len(res["per_rank"]) and c is not None:
                res["per_rank"][i].update({"gpu_matches_oracle": c["gpu_matches_oracle_on_sample"], "self_check": {k: c[k] for k in c if k != "gpu_matches_oracle_on_sample"}})
        res["all_ranks_match_oracle"] = all(c is not None and c["gpu_matches_oracle_on_sample"] is True for c in allc)
    res["arith"] = "fast" if arith else "exact"
    if arith:
        res["float_parity"] = float_parity_probe(w, ring, ring_chunks, S, C, local_rank)
    eng.close()
    del ring
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: one pass over the HBM-resident ring)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm", type=int, default=None, help="untimed steps in front of the warm-up steps (default: whole passes over the ring, at least 120 steps; 0 = none)")
    ap.add_argument("--workload", default="cfg4", choices=list(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary line for BASELINE configs[2] (/4)")
    ap.add_argument("--sync", action="store_true", help="deliver each step's text before the next step starts (no pipelining of calls)")
    ap.add_argument("--threads", action="store_true", help="with --gpus N and no torchrun: ONE process, an engine + a host thread per device (no RCCL anywhere)")
    ap.add_argument("--arith", default="both", choices=["exact", "fast", "both"],
                    help="arithmetic of the FIR sums: exact (bit-identical floats; what `value` always is), fast (fused multiply-add, tolerance 1e-5), both (default: `value` exact and a `fast` block beside it)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: only the launch plumbing (ranks, gloo barrier, MAX over ranks, rank 0's line) -- CPU tests")
    args = ap.parse_args()

    # --gpus N > 1 started as a plain command: this process becomes the launcher -- BEFORE torch is imported or HIP touched -- of N ranks of itself
    # (torch.distributed.run, one per GPU), relays what they print and exits with their code.  (such a command used to run on one GPU.)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.threads:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(Path(__file__).resolve())] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    same_device = bool(os.environ.get("HD_BENCH_SAME_DEVICE")) and not args.threads and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if same_device:
        local_rank = 0          # (tests on a one-GPU box: every rank's shard on device 0 -- the process group, the per-rank self-checks and the gather are what is exercised; the line says so)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not args.threads and "--gpus" not in " ".join(sys.argv[1:]).replace("=", " ").split() and world > 1:
        args.gpus = world                                 # `torchrun --nproc-per-node N bench.py` without --gpus: the job's size is what was asked for
    if not args.threads and world != max(1, args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of {world} rank(s): the line would report another number of GPUs than asked for")
    if args.dry_run:
        # the launch plumbing alone, on CPU (tests/test_sharding_gloo.py): process group, barrier, MAX of the timed region, one line from rank 0
        import torch
        import torch.distributed as dist
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        dt = job_time(dist if world > 1 else None, 1e-3 * (rank + 1))
        verdicts = [{"rank": 0, "gpu_matches_oracle": None}]
        if world > 1:
            # (the real run's last exchange: every rank's self-check verdict gathered on rank 0 -- here a placeholder per rank)
            verdicts = [None] * world
            dist.gather_object({"rank": rank, "gpu_matches_oracle": None}, verdicts if rank == 0 else None, dst=0)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            w = WORKLOADS[args.workload]; S = args.streams or w["S"]; K = args.steps or 1
            print(json.dumps({"metric": "IQ Msamples/s (batched 2.048 MS/s streams)", "value": round(world * S * w["C"] * K / dt / 1e6, 1), "unit": "MS/s", "n_gpus": world,
                              "steps": K, "warmup": args.warmup, "dry_run": True, "slowest_rank_ms": round(dt * 1e3, 3), "per_rank": verdicts, "process_group": "gloo",
                              "shards": [[shard(r, world, S)[0], shard(r, world, S)[-1] + 1] for r in range(world)]}), flush=True)
        return
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to time (the oracle is only the baseline leg)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    threads = 0
    if args.threads:
        if world > 1:
            raise SystemExit("bench.py --threads is the one-process mode: start it without torchrun")
        threads = max(1, args.gpus)
        if torch.cuda.device_count() < threads and not os.environ.get("HD_BENCH_SAME_DEVICE"):
            raise SystemExit(f"bench.py --threads --gpus {threads}: only {torch.cuda.device_count()} device(s) visible")
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo: the ranks share nothing on the data path (streams are sharded) -- what they exchange is a barrier, the MAX of a host
        # timer, two floats of sensor readings and each rank's self-check verdict, all host-side.  No RCCL communicator, no HBM or xGMI traffic beside the kernels.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    world_arg = world
    only_fast = args.arith == "fast"
    r = run_workload(torch, dist, dev, rank, local_rank, world_arg, args.workload, args.steps, args.warmup, args.streams, args.sync,
                     cpu_leg="check" if args.no_cpu_baseline else "full", threads=threads, prewarm=args.prewarm, arith=1 if only_fast else 0)
    # The engine's fast mode (fused multiply-add in every FIR; floats within 1e-5, decisions compared with the oracle's as they are) beside the exact-mode
    # `value`: the same workload, the same regions, its own engine and ring; every stream of the shard through the self-check.
    fast = None
    if args.arith == "both" and not threads:
        fast = run_workload(torch, dist, dev, rank, local_rank, world_arg, args.workload, args.steps, args.warmup, args.streams, args.sync,
                            cpu_leg="check" if args.no_cpu_baseline else "check_all", prewarm=args.prewarm, arith=1)
    if threads:
        world = threads
    also = also_fast = None
    if args.workload == "cfg4" and not args.no_also and world == 1:
        # BASELINE configs[2] ("1024 batched IQ streams @ 2.048 MS/s, dec=2, on 1 MI355X"): the harder, VALU-bound single-GPU
        # configuration, measured beside the headline (its own ring, a short timed region).
        also = run_workload(torch, dist, dev, rank, local_rank, world, "cfg3", 24, 4, 0, args.sync, cpu_leg="check", arith=1 if only_fast else 0)
        if args.arith == "both":
            also_fast = run_workload(torch, dist, dev, rank, local_rank, world, "cfg3", 24, 4, 0, args.sync, cpu_leg="check", arith=1)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    w = r["w"]
    line = {
        "metric": "IQ Msamples/s (batched 2.048 MS/s streams)" if w["fs"] == 2.048e6 else "IQ Msamples/s (batched streams)",
        "value": r["value"], "unit": "MS/s", "n_gpus": world, "steps": r["steps"], "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic: per-stream CRC-valid RTTY sentence as continuous-phase 2-FSK + Gaussian noise, generated on the GPU, HBM-resident ring",
        "config": {"workload": f"{args.workload}: {w['desc']}", "streams_per_gpu": r["S"], "chunk_samples": r["C"], "ring_chunks": r["ring_chunks"],
                   "sharding": f"{r['S']} independent streams per GPU, no data-path collective"},
        "timed_region_ms": r["timed_region_ms"],
        "cold": r["cold"],
        "prewarm_steps": r["prewarm_steps"],
        "timed_region_note": "per rank: from the opening barrier (device synchronize + gloo barrier) to the device synchronize behind the flush that delivers the K-th step's text; MAX over ranks; the closing barrier follows and is not timed",
        "prewarm_note": "`value` / `ms_per_step` are the K steps timed behind the pre-warm pass (untimed steps of the same loop: whole passes over the ring, at least 120) and --warmup more steps: the sustained regime, at the board's 1400 W cap; `cold` is the same K steps timed right behind --warmup alone, before that pass (the protocol of rounds 1-3: the loop's first ~150 launches, DESIGN.md section 6)",
        "per_rank": r["per_rank"],
        "roofline": r["roofline"], "pipeline": r["pipeline"], "box": r.get("box"),
    }
    if threads:
        line["config"]["launcher"] = f"one process, {threads} engine(s) each with its own host thread and device (bench.py --threads)"
    if "cpu_baseline" in r:
        line["cpu_baseline"] = r["cpu_baseline"]
    if also:
        aw = also["w"]
        line["also"] = {"workload": f"cfg3: {aw['desc']}", "value": also["value"], "unit": "MS/s", "ms_per_step": also["ms_per_step"], "steps": also["steps"],
                        "timed_region_ms": also["timed_region_ms"], "roofline": also["roofline"],
                        "hbm_frac_end_to_end": also["pipeline"]["hbm_frac_end_to_end"], "launch_path": also["pipeline"]["launch_path"],
                        "gpu_matches_oracle_on_sample": also["cpu_baseline"]["gpu_matches_oracle_on_sample"],
                        "self_check": {k: also["cpu_baseline"][k] for k in ("streams_in_sample", "bits_in_sample", "chars_in_sample", "sentences_in_sample")}}
    line["arith"] = r["arith"]
    if same_device:
        line["same_device"] = "HD_BENCH_SAME_DEVICE: every rank ran its shard on device 0 -- a plumbing test, not a scaling measurement"
    line["all_ranks_match_oracle"] = r.get("all_ranks_match_oracle")

    def fast_block(f):
        cb = f["cpu_baseline"]
        return {"arith": "fast: v_pk_fma_f32 in every FIR of the chain (hd_engine_config.arith = HD_ARITH_FAST); discriminator and symbol extractor as in the exact mode",
                "value": f["value"], "unit": "MS/s", "ms_per_step": f["ms_per_step"], "steps": f["steps"], "timed_region_ms": f["timed_region_ms"], "cold": f.get("cold"),
                "roofline": f["roofline"], "hbm_frac_end_to_end": f["pipeline"]["hbm_frac_end_to_end"], "launch_path": f["pipeline"]["launch_path"],
                "host_us_per_step": f["pipeline"].get("host_us_per_step"), "per_rank": f.get("per_rank"),
                "parity": {"float": f.get("float_parity"), "streams": cb.get("streams_in_sample"), "all_streams_of_the_shard": cb.get("all_streams_of_the_shard"),
                           "symbols_characters_sentences_equal": cb.get("gpu_matches_oracle_on_sample"), "all_ranks_match_oracle": f.get("all_ranks_match_oracle"),
                           "bits": cb.get("bits_in_sample"), "chars": cb.get("chars_in_sample"), "sentences": cb.get("sentences_in_sample"), "steps_checked": cb.get("steps_checked"),
                           "oracle_logs_reused": cb.get("oracle_logs_reused"),      # streams whose oracle log is the exact leg's (same samples, same chunk sequence): decoded once per line
                           "mismatches": cb.get("mismatches")}}
    if fast:
        line["fast"] = fast_block(fast)
        line["fast"]["speedup_over_exact"] = round(fast["value"] / r["value"], 4) if r["value"] else None
    if also_fast and "also" in line:
        line["also"]["fast"] = fast_block(also_fast)
        line["also"]["fast"]["speedup_over_exact"] = round(also_fast["value"] / also["value"], 4) if also["value"] else None
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
