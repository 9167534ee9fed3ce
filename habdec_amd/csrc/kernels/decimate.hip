// FIR-decimate stage for batched IQ streams on gfx950 (CDNA4).
//
// What it computes (reference code/Decoder/Decimator.h:128-138): out[o] = sum_{t=0}^{T-1} buf[o*D + t] * tap[t]
// with buf = [history(T-1) | this call's input], complex sample times real tap, accumulated in ascending t
// with separately rounded multiply and add (this file is compiled with -ffp-contract=off, so the result is
// bit-identical to the CPU path).  The first stage is the only kernel of the pipeline that touches full-rate
// IQ: 8 bytes read per input sample, 8/D written -- HBM-bound for D >= 8 (kernel table).
//
// Mapping: grid = (output tiles, streams); one lane = one output sample, so the T-term sum stays inside a lane
// and in order.  A tile's (TO-1)*D + T input samples are staged through LDS with coalesced 16-byte global
// loads.  In LDS the samples are laid out linearly with one 8-byte pad after every D samples: lane o then reads
// address o*(D+1) + t + t/D, a stride of 2*(D+1) dwords -- odd multiples of 2 -- which spreads the 32 lanes of
// each ds_read_b64 half-wave over all 64 banks (conflict-free), and the staging writes stay unit-stride.
// Taps are wave-uniform and come through the scalar cache (s_load), not LDS.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace hd {

template <int D, int T, int TO>
__global__ __launch_bounds__(TO) void k_decimate(const float2* __restrict__ in, size_t in_stride,
                                                   const float2* __restrict__ hist, const float* __restrict__ taps,
                                                   float2* __restrict__ out, size_t out_stride,
                                                   const StreamCall* __restrict__ call, int stage, int final_stage,
                                                   uint32_t fir_hist_cap)
{
    constexpr int NJ = (TO - 1) * D + T;       // samples a tile needs
    constexpr int NL = NJ + NJ / D + 2;        // with one pad slot per D samples
    __shared__ float2 tile[NL];

    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t n = stage == 0 ? c.n_in : c.n1;
    const uint32_t nout = n / D;
    const uint32_t o0 = blockIdx.x * TO;
    if (o0 >= nout) return;
    const bool zero_hist = (stage == 0 ? c.zero_hist1 : c.zero_hist2) != 0;
    const float2* in_s = in + (size_t)s * in_stride;
    const float2* hist_s = hist + (size_t)s * (T - 1);
    float2* out_s = out + (size_t)s * out_stride + (final_stage ? (size_t)fir_hist_cap + c.pend_before : 0);

    // tile-local j <-> stream sample x0 + j (negative = history)
    const long x0 = (long)o0 * D - (T - 1);
    constexpr int JS = (T - 1) & 1;            // x0 - JS is even, so pairs are 16-byte aligned in global memory
    auto fetch = [&](long xi) -> float2 {
        if (xi < 0) {
            if (zero_hist || xi < -(long)(T - 1)) return make_float2(0.f, 0.f);
            return hist_s[xi + (T - 1)];
        }
        if (xi < (long)n) return in_s[xi];
        return make_float2(0.f, 0.f);
    };
    for (int k = threadIdx.x; 2 * k - JS < NJ; k += TO) {
        const int j = 2 * k - JS;
        const long xi = x0 + j;
        float2 a, b;
        if (xi >= 0 && xi + 1 < (long)n) {
            const float4 v = *reinterpret_cast<const float4*>(in_s + xi);
            a = make_float2(v.x, v.y);
            b = make_float2(v.z, v.w);
        } else {
            a = fetch(xi);
            b = fetch(xi + 1);
        }
        if (j >= 0) tile[j + j / D] = a;
        if (j + 1 < NJ) tile[(j + 1) + (j + 1) / D] = b;
    }
    __syncthreads();

    const uint32_t o = o0 + threadIdx.x;
    const float2* p = tile + threadIdx.x * (D + 1);
    float ar = 0.f, ai = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float2 x = p[t + t / D];
        const float k = taps[t];
        ar = ar + x.x * k;
        ai = ai + x.y * k;
    }
    if (o < nout) out_s[o] = make_float2(ar, ai);
}

// History carry: hist <- last T-1 input samples.  Q4 (Decimator.h:140-143 with Decoder.h:443-444): the
// reference decimates in place, so history positions that fall inside the first n/D samples hold OUTPUTS.
__global__ void k_decim_history(int D, int T, const float2* __restrict__ in, size_t in_stride,
                                const float2* __restrict__ out, size_t out_stride, float2* __restrict__ hist,
                                const StreamCall* __restrict__ call, int stage, int final_stage, uint32_t fir_hist_cap)
{
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t n = stage == 0 ? c.n_in : c.n1;
    if (!n) return;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (uint32_t)(T - 1)) return;
    const uint32_t nout = n / D;
    const uint32_t idx = n - (T - 1) + j;      // host guarantees n >= T-1
    const float2* out_s = out + (size_t)s * out_stride + (final_stage ? (size_t)fir_hist_cap + c.pend_before : 0);
    hist[(size_t)s * (T - 1) + j] = idx < nout ? out_s[idx] : in[(size_t)s * in_stride + idx];
}

__global__ void k_passthrough(const float2* __restrict__ in, size_t in_stride, float2* __restrict__ out, size_t out_stride,
                              const StreamCall* __restrict__ call, uint32_t fir_hist_cap)
{
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c.n_in) out[(size_t)s * out_stride + fir_hist_cap + c.pend_before + i] = in[(size_t)s * in_stride + i];
}

// Per-chunk DC blocker (reference Decoder.h:450-459): wp = .97*x0; w = x + .97*wp; y = w - wp; wp = w.
// Strictly sequential per stream; one wave per stream walks the chunk 64 samples at a time, broadcasting
// each sample with v_readlane so the recurrence runs uniformly in every lane without LDS or barriers.
__global__ __launch_bounds__(64) void k_dc_remove(float2* __restrict__ fbuf, size_t stride, const StreamCall* __restrict__ call,
                                                    uint32_t fir_hist_cap)
{
    const uint32_t s = blockIdx.x;
    const StreamCall c = call[s];
    if (!c.dc_remove || !c.n2) return;
    float2* x = fbuf + (size_t)s * stride + fir_hist_cap + c.pend_before;
    const int lane = threadIdx.x;
    const float2 x0 = x[0];
    float wr = 0.97f * x0.x, wi = 0.97f * x0.y;
    for (uint32_t base = 0; base < c.n2; base += 64) {
        const uint32_t i = base + lane;
        float2 v = i < c.n2 ? x[i] : make_float2(0.f, 0.f);
        float yr = 0.f, yi = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float xr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), k));
            const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.y), k));
            const float sr = 0.97f * wr, si = 0.97f * wi;
            const float nr = xr + sr, ni = xi + si;
            if (lane == k) { yr = nr - wr; yi = ni - wi; }
            // lanes past the end of the chunk must not disturb the carry (their outputs are not stored)
            if (base + k < c.n2) { wr = nr; wi = ni; }
        }
        if (i < c.n2) x[i] = make_float2(yr, yi);
    }
}

template <int D, int T, int TO>
static void launch_one(hipStream_t st, uint32_t n_streams, uint32_t max_out, const float2* in, size_t in_stride,
                       const float2* hist, const float* taps, float2* out, size_t out_stride, const StreamCall* call,
                       int stage, int final_stage, uint32_t fir_hist_cap)
{
    dim3 grid((max_out + TO - 1) / TO, n_streams);
    hipLaunchKernelGGL((k_decimate<D, T, TO>), grid, dim3(TO), 0, st, in, in_stride, hist, taps, out, out_stride, call, stage,
                       final_stage, fir_hist_cap);
}

bool launch_decimate(hipStream_t st, int ratio, int ntaps, uint32_t n_streams, uint32_t max_out, const float2* in, size_t in_stride,
                     const float2* hist, const float* taps, float2* out, size_t out_stride, const StreamCall* call, int stage,
                     int final_stage, uint32_t fir_hist_cap)
{
    if (!max_out) return true;
#define HD_CASE(D, T, TO) \
    if (ratio == D && ntaps == T) { launch_one<D, T, TO>(st, n_streams, max_out, in, in_stride, hist, taps, out, out_stride, call, stage, final_stage, fir_hist_cap); return true; }
    HD_CASE(2, 69, 256) HD_CASE(4, 139, 256) HD_CASE(8, 280, 256) HD_CASE(8, 54, 256)
    HD_CASE(16, 107, 128) HD_CASE(32, 212, 128) HD_CASE(32, 174, 128) HD_CASE(64, 348, 64)
#undef HD_CASE
    return false;
}

void launch_decim_history(hipStream_t st, int ratio, int ntaps, uint32_t n_streams, const float2* in, size_t in_stride,
                          const float2* out, size_t out_stride, float2* hist, const StreamCall* call, int stage, int final_stage,
                          uint32_t fir_hist_cap)
{
    dim3 grid((ntaps - 1 + 255) / 256, n_streams);
    hipLaunchKernelGGL(k_decim_history, grid, dim3(256), 0, st, ratio, ntaps, in, in_stride, out, out_stride, hist, call, stage,
                       final_stage, fir_hist_cap);
}

void launch_passthrough(hipStream_t st, uint32_t n_streams, uint32_t max_n, const float2* in, size_t in_stride, float2* out,
                        size_t out_stride, const StreamCall* call, uint32_t fir_hist_cap)
{
    if (!max_n) return;
    dim3 grid((max_n + 255) / 256, n_streams);
    hipLaunchKernelGGL(k_passthrough, grid, dim3(256), 0, st, in, in_stride, out, out_stride, call, fir_hist_cap);
}

void launch_dc_remove(hipStream_t st, uint32_t n_streams, float2* fbuf, size_t stride, const StreamCall* call, uint32_t fir_hist_cap)
{
    hipLaunchKernelGGL(k_dc_remove, dim3(n_streams), dim3(64), 0, st, fbuf, stride, call, fir_hist_cap);
}

}  // namespace hd
