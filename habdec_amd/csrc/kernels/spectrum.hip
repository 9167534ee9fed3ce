// Spectrum post-processing: fftshift + dB power + the reductions the AFC needs, one workgroup per stream.
//
// The 4096-point forward transform itself is rocFFT (batched over streams).  This kernel turns its natural-order
// output into what the reference keeps (code/Decoder/FFT.cpp:77-87 half swap -> freq_out_) and evaluates
// AFC::FftPower / ComputeVariance / FindPeaks (code/Decoder/AFC.h:235-329) with wave/LDS reductions:
//   P[i] = 10*log10f( ((|X[i]|^2 / N)^2) / fsd )          float, the /fsd through double   (AFC.h:267-270, Q22)
//   mean, sigma of P in double                              (AFC.h:103-104, 224-232)
//   p1 = first arg-max of P; p2 = first arg-max of {P[i] > P[0]} within +-2*sep of p1, |i-p1| > sep/2 (AFC.h:303-319)
// The scalar state machine on top of these numbers runs on the host (host/afc_tracker.hpp).
// Parity note: FFTW is not available to pin against, and device log10f differs from glibc's by
// ulps, so P is compared norm-wise, not bit-wise.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace hd {

constexpr int kSpecLanes = 256;
constexpr int kPerLane = kFftBins / kSpecLanes;   // 16

__device__ __forceinline__ double block_sum(double v, double* scratch)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// first arg-max: larger value wins, equal values -> smaller index
__device__ __forceinline__ void block_argmax(float& v, int& idx, float* sv, int* si)
{
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(v, off, 64);
        const int oi = __shfl_down(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sv[wave] = v; si[wave] = idx; }
    __syncthreads();
    v = sv[0]; idx = si[0];
    for (int w = 1; w < 4; ++w)
        if (sv[w] > v || (sv[w] == v && si[w] < idx)) { v = sv[w]; idx = si[w]; }
}

__global__ __launch_bounds__(kSpecLanes) void k_spectrum_commit(const float2* __restrict__ raw, float2* __restrict__ spec,
                                                                  float* __restrict__ power, SpectrumStatsDev* __restrict__ stats,
                                                                  const StreamCall* __restrict__ call, double rate, int bins_sep, const uint32_t seq)
{
    __shared__ float P[kFftBins];
    __shared__ double dscratch[4];
    __shared__ float fscratch[4];
    __shared__ int iscratch[4];
    __shared__ int bad;
    const uint32_t s = blockIdx.x;
    if (!call[s].fft_run) return;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const float2* x = raw + (size_t)s * kFftBins;
    float2* so = spec + (size_t)s * kFftBins;
    float* po = power + (size_t)s * kFftBins;
    int mybad = 0;
    double lsum = 0.0;
    for (int q = 0; q < kPerLane; ++q) {
        const int k = threadIdx.x + q * kSpecLanes;          // natural FFT bin
        const int i = (k + kFftBins / 2) & (kFftBins - 1);    // position after the half swap
        const float2 v = x[k];
        so[i] = v;
        float p = (v.x * v.x + v.y * v.y) / (float)kFftBins;
        p = p * p;
        p = (float)((double)p / rate);
        p = 10.0f * log10f(p);
        if (v.x != v.x || v.y != v.y || isinf(v.x) || isinf(v.y) || p != p || isinf(p)) mybad = 1;
        P[i] = p;
        po[i] = p;
        lsum += (double)p;
    }
    if (mybad) bad = 1;
    const double mean = block_sum(lsum, dscratch) / (double)kFftBins;   // (barriers inside also publish P and bad)
    double lvar = 0.0;
    float bv = -__builtin_huge_valf();
    int bi = kFftBins;
    for (int q = 0; q < kPerLane; ++q) {
        const int i = threadIdx.x * kPerLane + q;             // contiguous per lane: ascending index inside a lane
        const float p = P[i];
        const double d = (double)p - mean;
        lvar += d * d;
        if (p > bv) { bv = p; bi = i; }
    }
    const double sigma = sqrt(block_sum(lvar, dscratch) / (double)kFftBins);
    block_argmax(bv, bi, fscratch, iscratch);
    const int p1 = bi;
    const float p1v = bv;
    // second peak
    const int lo = max(p1 - 2 * bins_sep, 0), hi = min(p1 + 2 * bins_sep, kFftBins);
    const float floor0 = P[0];
    float cv = -__builtin_huge_valf();
    int ci = kFftBins;
    for (int i = lo + (int)threadIdx.x; i < hi; i += kSpecLanes) {
        const float p = P[i];
        if (p > floor0 && abs(i - p1) > bins_sep / 2 && p > cv) { cv = p; ci = i; }
    }
    block_argmax(cv, ci, fscratch, iscratch);
    if (threadIdx.x == 0) {
        int a = p1, b = 0;
        float av = p1v, bvv = floor0;
        if (ci < kFftBins) { b = ci; bvv = cv; }
        if (b < a) { const int ti = a; a = b; b = ti; const float tv = av; av = bvv; bvv = tv; }
        SpectrumStatsDev o;
        o.valid = bad ? 0 : 1;
        o.peak1 = a; o.peak2 = b; o.power1 = av; o.power2 = bvv; o.seq = 0u;
        o.mean = mean; o.sigma = sigma;
        stats[s] = o;
        // the call's tag, last: behind a wait for the stores above (the statistics live in mapped host memory; the engine's events carry no system fence)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&stats[s].seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_spectrum_commit(hipStream_t st, uint32_t n_streams, const float2* raw, float2* spec, float* power,
                            SpectrumStatsDev* stats, const StreamCall* call, double rate, int bins_sep, uint32_t seq)
{
    hipLaunchKernelGGL(k_spectrum_commit, dim3(n_streams), dim3(kSpecLanes), 0, st, raw, spec, power, stats, call, rate, bins_sep, seq);
}

}  // namespace hd
