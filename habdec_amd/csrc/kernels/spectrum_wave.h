// The whole spectrum step of a stream as ONE wave: 4096-point forward transform, half swap, dB power, and the reductions the AFC
// needs (spectrum.hip describes them) -- what rocFFT + k_spectrum_commit do in two launches with a 32 KB round trip through HBM in
// between (fft_raw), done in registers.  Reference: code/Decoder/FFT.cpp:60-87 (FFTW forward, half swap), AFC.h:235-329.
//
// 4096 = 64 x 64.  With n = 64*n1 + n2 and k = k1 + 64*k2:
//     X[k1 + 64 k2] = sum_n2 W64^(n2 k2) * [ W4096^(n2 k1) * sum_n1 x[64 n1 + n2] W64^(n1 k1) ]
// Pass 1: lane n2 loads its 64 samples x[64 n1 + n2] (a wave-wide load per n1: 512 contiguous bytes) and runs a 64-point transform
// over n1 in registers; the result is multiplied by W4096^(n2 k1) (table in HBM/L2, computed in double on the host); an LDS transpose
// (one float plane at a time, 65-float pitch: conflict-free both ways) hands lane k1 the 64 values of all n2; pass 2 is the same
// 64-point transform over n2, after which lane k1 holds X[k1 + 64 k2], k2 = 0..63 -- bins that are 64 apart, so the stores of the
// swapped spectrum and of the power are wave-wide contiguous again.  The 64-point transform is 8 x 8 eight-point transforms on a register
// array with compile-time indices and twiddles.
//
// Parity: like rocFFT's, this transform is compared norm-wise with the exact DFT (FFTW itself is not available to pin against,
// DESIGN.md section 8): same tolerances, same tests.
#pragma once
#include <hip/hip_runtime.h>

#include "launch.h"
#include "sym_common.h"

namespace hd {

namespace specwave {

__device__ static constexpr float kW64[2][32] = {
#include "fft64_tw.inc"
};

// Where bin k of a 64-point transform sits in the register array after fft64 (8 x 8 decomposition, second-level output k2 left in its
// row): X[k1 + 8 k2] = a[8 k1 + k2].
constexpr int xpos(int k) { return 8 * (k & 7) + (k >> 3); }

// 8-point transform of v[0..7] in place, natural order in and out: three radix-2 stages (decimation in time) with W8 = (1 - i)/sqrt(2).
__device__ __forceinline__ void fft8(f32x2 (&v)[8])
{
    constexpr float r = 0.70710678118654752440f;
    // stage 1 on the bit-reversed pairs (0,4) (2,6) (1,5) (3,7)
    const f32x2 a0 = v[0] + v[4], a1 = v[0] - v[4], a2 = v[2] + v[6], a3 = v[2] - v[6];
    const f32x2 a4 = v[1] + v[5], a5 = v[1] - v[5], a6 = v[3] + v[7], a7 = v[3] - v[7];
    // stage 2: twiddles 1, -i
    const f32x2 a3r = (f32x2){a3.y, -a3.x}, a7r = (f32x2){a7.y, -a7.x};
    const f32x2 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3r, b3 = a1 - a3r;
    const f32x2 b4 = a4 + a6, b6 = a4 - a6, b5 = a5 + a7r, b7 = a5 - a7r;
    // stage 3: twiddles 1, W8, -i, W8^3
    const f32x2 t5 = (f32x2){(b5.x + b5.y) * r, (b5.y - b5.x) * r};          // b5 * (1 - i)/sqrt(2)
    const f32x2 t6 = (f32x2){b6.y, -b6.x};                                    // b6 * (-i)
    const f32x2 t7 = (f32x2){(b7.y - b7.x) * r, (-b7.x - b7.y) * r};         // b7 * (-1 - i)/sqrt(2)
    v[0] = b0 + b4; v[4] = b0 - b4;
    v[1] = b1 + t5; v[5] = b1 - t5;
    v[2] = b2 + t6; v[6] = b2 - t6;
    v[3] = b3 + t7; v[7] = b3 - t7;
}

// 64-point transform on the register array: a[n] = x[n] in (natural order), X[k] = a[xpos(k)] out.  64 = 8 x 8 with n = 8 n1 + n2,
// k = k1 + 8 k2: eight 8-point transforms over n1 (one per n2) with the twiddle W64^(n2 k1) folded in, then eight over n2 (one per k1).
// One small transform at a time (the scheduling barriers keep the compiler from interleaving all eight: the array alone takes half of
// the register file, and a spill inside a step launch queues behind stage 1's tile loads).
__device__ __forceinline__ void fft64(f32x2 (&a)[64])
{
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) {
        __builtin_amdgcn_sched_barrier(0);
        f32x2 v[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = a[8 * n1 + n2];
        fft8(v);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            const int m = n2 * k1;                                  // W64^m, m in [0, 49]
            f32x2 w = v[k1];
            if (m != 0) {
                const float c = m < 32 ? kW64[0][m] : -kW64[0][m - 32], sn = m < 32 ? kW64[1][m] : -kW64[1][m - 32];
                w = (f32x2){w.x * c + w.y * sn, w.y * c - w.x * sn};
            }
            a[8 * k1 + n2] = w;
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        __builtin_amdgcn_sched_barrier(0);
        f32x2 v[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) v[n2] = a[8 * k1 + n2];
        fft8(v);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) a[8 * k1 + k2] = v[k2];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// All 64 values become opaque at this point of the program: nothing computed from them can be moved in front of it and nothing they are
// computed from behind it.  (The instruction-scheduler barriers do not bind the optimiser: it started the power computation of the first
// bins in the middle of the second pass, and the extra live values spilled -- a spill inside a step launch queues behind stage 1's loads.)
__device__ __forceinline__ void pin64(f32x2 (&a)[64])
{
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(a[i]));
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}

// first arg-max over the wave: larger value wins, equal values -> smaller index; everybody gets the answer
__device__ __forceinline__ void wave_argmax(float& v, int& idx)
{
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(v, off, 64);
        const int oi = __shfl_down(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    v = __shfl(v, 0, 64); idx = __shfl(idx, 0, 64);
}

}  // namespace specwave

constexpr uint32_t kSpecWaveLds = 64 * 65 * 4;                      // the transpose plane

// x: the stream's 4096 input samples (wave-uniform pointer) -- its row of the spectrum input collection, or, where one call's chunk alone fills the
// buffer, the head of that chunk where the last decimation stage left it (StreamCall::fft_run == 2: no second copy of the samples)
__device__ __forceinline__ void spectrum_wave_body(const float2* __restrict__ x, const float2* __restrict__ tw4096, float2* __restrict__ spec,
                                                   float* __restrict__ power, SpectrumStatsDev* __restrict__ stats, const uint32_t s, const double rate,
                                                   const int bins_sep, float* __restrict__ plane, const uint32_t seq)
{
    const uint32_t l = threadIdx.x & 63u;
    f32x2 a[64];
    // ---- pass 1: transform over n1 for this lane's n2 = l
    // (rows in groups of eight behind a scalar base the compiler cannot fold into the lane offset: one offset register and immediate
    // row offsets instead of 56 more address registers, here and for the stores at the end)
#pragma unroll
    for (int g = 0; g < 64; g += 8) {
        const float2* xg = x + 64 * g;
        asm volatile("" : "+s"(xg));
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float2 v = xg[64 * u + l]; a[g + u] = (f32x2){v.x, v.y}; }
    }
    specwave::fft64(a);
    // ---- twiddle W4096^(n2 k1), then the transpose (lane n2, register k1) -> (lane k1, register n2), one plane at a time
    // (sixteen factors at a time: all 63 gathers hoisted in front of pass 1 would need another 126 registers)
#pragma unroll
    for (int g = 0; g < 64; g += 16) {
        __builtin_amdgcn_sched_barrier(0);
        float2 w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = tw4096[(l * (uint32_t)specwave::xpos(g + u)) & (kFftBins - 1)];   // (cos, -sin); register i holds k1 = xpos(i)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const f32x2 v = a[g + u];
            a[g + u] = (f32x2){v.x * w[u].x - v.y * w[u].y, v.x * w[u].y + v.y * w[u].x};
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k1 = 0; k1 < 64; ++k1) plane[k1 * 65 + l] = a[specwave::xpos(k1)].x;
    __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0): the plane is wave-private
    __builtin_amdgcn_wave_barrier();
    // (the two components move independently: .x of every register is replaced while .y still sits at its pass-1 index)
#pragma unroll
    for (int n2 = 0; n2 < 64; ++n2) a[n2].x = plane[l * 65 + n2];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 0; k1 < 64; ++k1) plane[k1 * 65 + l] = a[specwave::xpos(k1)].y;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 64; ++n2) a[n2].y = plane[l * 65 + n2];
    // ---- pass 2: transform over n2; a[xpos(k2)] = X[l + 64 k2]
    specwave::pin64(a);
    specwave::fft64(a);
    specwave::pin64(a);
    // ---- half swap, dB power, statistics.  Bin k = l + 64 k2 lands at i = (k + 2048) & 4095 = l + 64 j, j = (k2 + 32) & 63.
    float2* so = spec + (size_t)s * kFftBins;
    float* po = power + (size_t)s * kFftBins;
    int mybad = 0;
    double lsum = 0.0;
    // P[l + 64 j] replaces the real part of the bin it was computed from (register xpos((j + 32) & 63)): no second array beside a[]
#define HD_SW_P(j_) a[specwave::xpos(((j_) + 32) & 63)].x
    float2* sg = so;
    float* pg = po;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        if ((j & 7) == 0) {                                         // eight bins at a time: the logarithms' temporaries add up otherwise
            __builtin_amdgcn_sched_barrier(0);
            sg = so + 64 * j; pg = po + 64 * j;
            asm volatile("" : "+s"(sg), "+s"(pg));
        }
        const int k2 = (j + 32) & 63;
        const f32x2 v = a[specwave::xpos(k2)];
        sg[l + 64 * (j & 7)] = make_float2(v.x, v.y);
        float q = (v.x * v.x + v.y * v.y) / (float)kFftBins;
        q = q * q;
        q = (float)((double)q / rate);
        q = 10.0f * log10f(q);
        if (v.x != v.x || v.y != v.y || isinf(v.x) || isinf(v.y) || q != q || isinf(q)) mybad = 1;
        HD_SW_P(j) = q;
        pg[l + 64 * (j & 7)] = q;
        lsum += (double)q;
    }
    const bool bad = __builtin_amdgcn_ballot_w64(mybad != 0) != 0ull;
    const double mean = specwave::wave_sum(lsum) / (double)kFftBins;
    double lvar = 0.0;
    float bv = -__builtin_huge_valf();
    int bi = kFftBins;
#pragma unroll
    for (int j = 0; j < 64; ++j) {                                  // ascending index inside the lane
        const float pj = HD_SW_P(j);
        const double d = (double)pj - mean;
        lvar += d * d;
        if (pj > bv) { bv = pj; bi = (int)l + 64 * j; }
    }
    const double sigma = sqrt(specwave::wave_sum(lvar) / (double)kFftBins);
    specwave::wave_argmax(bv, bi);
    const int p1 = bi;
    const float p1v = bv;
    const int lo = max(p1 - 2 * bins_sep, 0), hi = min(p1 + 2 * bins_sep, (int)kFftBins);
    const float floor0 = __shfl(HD_SW_P(0), 0, 64);                 // P[0]
    float cv = -__builtin_huge_valf();
    int ci = kFftBins;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const int i = (int)l + 64 * j;
        const float pj = HD_SW_P(j);
        if (i >= lo && i < hi && pj > floor0 && abs(i - p1) > bins_sep / 2 && pj > cv) { cv = pj; ci = i; }
    }
    specwave::wave_argmax(cv, ci);
    if (l == 0) {
        int pa = p1, pb = 0;
        float av = p1v, bvv = floor0;
        if (ci < (int)kFftBins) { pb = ci; bvv = cv; }
        if (pb < pa) { const int ti = pa; pa = pb; pb = ti; const float tv = av; av = bvv; bvv = tv; }
        SpectrumStatsDev o;
        o.valid = bad ? 0 : 1;
        o.peak1 = pa; o.peak2 = pb; o.power1 = av; o.power2 = bvv; o.seq = 0u;
        o.mean = mean; o.sigma = sigma;
        stats[s] = o;
        // the call's tag, last (as k_spectrum_commit: the statistics live in mapped host memory and the host waits for the tag, not for an event's fence)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&stats[s].seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#undef HD_SW_P
}

}  // namespace hd
