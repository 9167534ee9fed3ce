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
// swapped spectrum and of the power are wave-wide contiguous again.  The 64-point transform is six radix-2 stages on a register
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

constexpr int brev6(int v) { return ((v & 1) << 5) | ((v & 2) << 3) | ((v & 4) << 1) | ((v & 8) >> 1) | ((v & 16) >> 3) | ((v & 32) >> 5); }

// One radix-2 stage of span M (decimation in time, input in bit-reversed order): a[k+j], a[k+j+M/2] <- u + W t, u - W t, W = W_M^j
template <int M>
__device__ __forceinline__ void fft64_stage(f32x2 (&a)[64])
{
    constexpr int H = M / 2, STEP = 64 / M;
#pragma unroll
    for (int k = 0; k < 64; k += M) {
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int m = j * STEP;                                  // W_64^m = cos - i sin, m in [0, 32)
            const f32x2 u = a[k + j], v = a[k + j + H];
            f32x2 t;
            if (m == 0) t = v;
            else if (m == 16) t = (f32x2){v.y, -v.x};
            else {
                const float c = kW64[0][m], s = kW64[1][m];
                t.x = v.x * c + v.y * s;
                t.y = v.y * c - v.x * s;
            }
            a[k + j] = u + t;
            a[k + j + H] = u - t;
        }
    }
}

__device__ __forceinline__ void fft64(f32x2 (&a)[64])               // a[brev6(n)] = x[n] in, a[k] = X[k] out
{
    // (a barrier for the instruction scheduler between the stages: interleaving them breadth-first keeps two stages' worth of values alive)
    fft64_stage<2>(a); __builtin_amdgcn_sched_barrier(0);
    fft64_stage<4>(a); __builtin_amdgcn_sched_barrier(0);
    fft64_stage<8>(a); __builtin_amdgcn_sched_barrier(0);
    fft64_stage<16>(a); __builtin_amdgcn_sched_barrier(0);
    fft64_stage<32>(a); __builtin_amdgcn_sched_barrier(0);
    fft64_stage<64>(a); __builtin_amdgcn_sched_barrier(0);
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

__device__ __forceinline__ void spectrum_wave_body(const float2* __restrict__ fft_in, const float2* __restrict__ tw4096, float2* __restrict__ spec,
                                                   float* __restrict__ power, SpectrumStatsDev* __restrict__ stats, const uint32_t s, const double rate,
                                                   const int bins_sep, float* __restrict__ plane)
{
    const uint32_t l = threadIdx.x & 63u;
    const float2* x = fft_in + (size_t)s * kFftBins;
    f32x2 a[64];
    // ---- pass 1: transform over n1 for this lane's n2 = l
    // (rows in groups of eight behind a scalar base the compiler cannot fold into the lane offset: one offset register and immediate
    // row offsets instead of 56 more address registers, here and for the stores at the end)
#pragma unroll
    for (int g = 0; g < 64; g += 8) {
        const float2* xg = x + 64 * g;
        asm volatile("" : "+s"(xg));
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float2 v = xg[64 * u + l]; a[specwave::brev6(g + u)] = (f32x2){v.x, v.y}; }
    }
    specwave::fft64(a);
    // ---- twiddle W4096^(n2 k1), then the transpose (lane n2, register k1) -> (lane k1, register n2), one plane at a time
    // (sixteen factors at a time: all 63 gathers hoisted in front of pass 1 would need another 126 registers)
#pragma unroll
    for (int g = 0; g < 64; g += 16) {
        __builtin_amdgcn_sched_barrier(0);
        float2 w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = tw4096[(l * (uint32_t)(g + u)) & (kFftBins - 1)];   // (cos, -sin)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const f32x2 v = a[g + u];
            a[g + u] = (f32x2){v.x * w[u].x - v.y * w[u].y, v.x * w[u].y + v.y * w[u].x};
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k1 = 0; k1 < 64; ++k1) plane[k1 * 65 + l] = a[k1].x;
    __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0): the plane is wave-private
    __builtin_amdgcn_wave_barrier();
    // (the two components move independently: .x of every register is replaced while .y still sits at its pass-1 index)
#pragma unroll
    for (int n2 = 0; n2 < 64; ++n2) a[specwave::brev6(n2)].x = plane[l * 65 + n2];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 0; k1 < 64; ++k1) plane[k1 * 65 + l] = a[k1].y;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 64; ++n2) a[specwave::brev6(n2)].y = plane[l * 65 + n2];
    // ---- pass 2: transform over n2; a[k2] = X[l + 64 k2]
    specwave::fft64(a);
    // ---- half swap, dB power, statistics.  Bin k = l + 64 k2 lands at i = (k + 2048) & 4095 = l + 64 j, j = (k2 + 32) & 63.
    float2* so = spec + (size_t)s * kFftBins;
    float* po = power + (size_t)s * kFftBins;
    int mybad = 0;
    double lsum = 0.0;
    float p[64];                                                    // p[j] = P[l + 64 j]
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
        const f32x2 v = a[k2];
        sg[l + 64 * (j & 7)] = make_float2(v.x, v.y);
        float q = (v.x * v.x + v.y * v.y) / (float)kFftBins;
        q = q * q;
        q = (float)((double)q / rate);
        q = 10.0f * log10f(q);
        if (v.x != v.x || v.y != v.y || isinf(v.x) || isinf(v.y) || q != q || isinf(q)) mybad = 1;
        p[j] = q;
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
        const double d = (double)p[j] - mean;
        lvar += d * d;
        if (p[j] > bv) { bv = p[j]; bi = (int)l + 64 * j; }
    }
    const double sigma = sqrt(specwave::wave_sum(lvar) / (double)kFftBins);
    specwave::wave_argmax(bv, bi);
    const int p1 = bi;
    const float p1v = bv;
    const int lo = max(p1 - 2 * bins_sep, 0), hi = min(p1 + 2 * bins_sep, (int)kFftBins);
    const float floor0 = __shfl(p[0], 0, 64);                       // P[0]
    float cv = -__builtin_huge_valf();
    int ci = kFftBins;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const int i = (int)l + 64 * j;
        if (i >= lo && i < hi && p[j] > floor0 && abs(i - p1) > bins_sep / 2 && p[j] > cv) { cv = p[j]; ci = i; }
    }
    specwave::wave_argmax(cv, ci);
    if (l == 0) {
        int pa = p1, pb = 0;
        float av = p1v, bvv = floor0;
        if (ci < (int)kFftBins) { pb = ci; bvv = cv; }
        if (pb < pa) { const int ti = pa; pa = pb; pb = ti; const float tv = av; av = bvv; bvv = tv; }
        SpectrumStatsDev o;
        o.valid = bad ? 0 : 1;
        o.peak1 = pa; o.peak2 = pb; o.power1 = av; o.power2 = bvv; o._pad = 0.f;
        o.mean = mean; o.sigma = sigma;
        stats[s] = o;
    }
}

}  // namespace hd
