// Low-pass FIR fused with the FSK polar discriminator, plus the small data-movement kernels around it.
//
// FIR (reference code/Decoder/FirFilter.h:155-161): y[i] = sum_{t<T} buf[i+t]*tap[t], buf = [history(T-1) | pending
// decimated samples]; same sequential non-FMA accumulation as the decimator (bit-exact).  The discriminator
// (code/Decoder/FSK2_Demod.h:37-40) d[i] = arg(y[i] * conj(y[i-1])) is applied in the epilogue, so the filtered
// samples never go to HBM unless a parity test asks for them: per decimated sample the kernel reads 8 B and writes
// 4 B, the "8/D + 4/D" terms of the pipeline's byte model.
//
// Mapping: grid = (tiles, streams), 256 lanes per tile, one output per lane.  Tiles advance by 255 outputs and
// overlap by one, so every lane finds its predecessor y[i-1] in LDS (lane 0 of a tile only supplies that
// predecessor).  The per-stream input buffer keeps the FIR history right in front of the pending samples, so the
// tile's 255+T inputs are one contiguous, coalesced read into LDS.  Taps are per stream and wave-uniform.
#include <hip/hip_runtime.h>

#include "exact_math.h"
#include "launch.h"

namespace hd {

constexpr int kFirLanes = 256;

__global__ __launch_bounds__(kFirLanes) void k_fir_demod(const float2* __restrict__ fbuf, size_t stride,
                                                          const float* __restrict__ taps, uint32_t taps_stride,
                                                          float* __restrict__ demod, size_t demod_stride,
                                                          float2* __restrict__ filtered,
                                                          const DemodCarry* __restrict__ carry_in,
                                                          DemodCarry* __restrict__ carry_out,
                                                          const StreamCall* __restrict__ call, uint32_t fir_hist_cap,
                                                          float* __restrict__ sym_ring, uint32_t ring_cap,
                                                          const SymState* __restrict__ sym)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];   // [kFirLanes + T - 1] inputs, then reused for outputs
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t m = c.fir_m, T = c.fir_taps;
    if (blockIdx.x == 0 && threadIdx.x == 0 && m == 0) carry_out[s] = carry_in[s];   // idle stream: carry passes through
    if (!m || !T) return;
    const long i0 = (long)blockIdx.x * (kFirLanes - 1) - 1;        // output index of lane 0 (may be -1)
    if (i0 + 1 >= (long)m) return;
    const float2* buf = fbuf + (size_t)s * stride;                 // history occupies [fir_hist_cap-(T-1), fir_hist_cap)
    const long b0 = (long)fir_hist_cap - (long)(T - 1) + i0;       // buffer index of tile-local sample 0
    const uint32_t need = kFirLanes + T - 1;
    const long end = (long)fir_hist_cap + (long)m;                 // one past the last valid input
    for (uint32_t j = threadIdx.x; j < need; j += kFirLanes) {
        const long b = b0 + (long)j;
        float2 v = make_float2(0.f, 0.f);
        if (b >= 0 && b < end && !(c.fir_zero_hist && b < (long)fir_hist_cap)) v = buf[b];
        lds[j] = v;
    }
    __syncthreads();

    const float* tp = taps + (size_t)s * taps_stride;
    const float2* p = lds + threadIdx.x;
    float ar = 0.f, ai = 0.f;
    uint32_t t = 0;
    for (; t + 8 <= T; t += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float2 x = p[t + u];
            const float k = tp[t + u];
            ar = ar + x.x * k;
            ai = ai + x.y * k;
        }
    }
    for (; t < T; ++t) {
        const float2 x = p[t];
        const float k = tp[t];
        ar = ar + x.x * k;
        ai = ai + x.y * k;
    }
    __syncthreads();                       // everyone is done reading inputs: reuse LDS for the outputs
    lds[threadIdx.x] = make_float2(ar, ai);
    __syncthreads();

    const long i = i0 + (long)threadIdx.x;
    if (threadIdx.x == 0 || i >= (long)m) return;
    float pr, pi;
    if (i > 0) { const float2 q = lds[threadIdx.x - 1]; pr = q.x; pi = q.y; }
    else {
        const DemodCarry k = carry_in[s];
        if (k.primed) { pr = k.re; pi = k.im; } else { pr = ar; pi = ai; }   // very first sample: arg(y0*conj(y0))
    }
    const float d = discriminate(ar, ai, pr, pi);
    demod[(size_t)s * demod_stride + i] = d;
    if (sym_ring) {      // append straight into the symbol extractor's ring (SymbolExtractor::pushSamples); a vent
        const SymState st = sym[s];   // (backlog > 30000) restarts the backlog at the same position base + held
        sym_ring[(size_t)s * ring_cap + ((st.base + st.held + (uint32_t)i) & (ring_cap - 1))] = d;
    }
    if (filtered) filtered[(size_t)s * demod_stride + i] = make_float2(ar, ai);
    if (i == (long)m - 1) { DemodCarry k; k.re = ar; k.im = ai; k.primed = 1; k._pad = 0; carry_out[s] = k; }
}

// After the FIR consumed fir_m samples: slide [history | leftover pending] to the front of the OTHER buffer
// (ping-pong instead of an overlapping in-place move).  dst[k] = src[k + fir_m] for k < hist_cap + pend_after.
__global__ void k_fbuf_shift(const float2* __restrict__ src, float2* __restrict__ dst, size_t stride,
                             const StreamCall* __restrict__ call, uint32_t fir_hist_cap)
{
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= fir_hist_cap + c.pend_after) return;
    const uint32_t from = c.clear_pending ? k : k + c.fir_m;
    dst[(size_t)s * stride + k] = src[(size_t)s * stride + from];
}

// Spectrum input collection (reference Decoder.h:467-473): append the HEAD of this call's decimated chunk.
__global__ void k_fft_feed(const float2* __restrict__ fbuf, size_t stride, float2* __restrict__ fft_in,
                           const StreamCall* __restrict__ call, uint32_t fir_hist_cap)
{
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.fft_take) return;
    fft_in[(size_t)s * kFftBins + c.fft_fill + j] = fbuf[(size_t)s * stride + fir_hist_cap + c.pend_before + j];
}

// Per-call parameters come from mapped pinned host memory; one small kernel pulls them into HBM so that the
// thousands of workgroups of the following kernels read them from L2 instead of over PCIe.  (A hipMemcpyAsync
// here stalled the enqueueing thread for milliseconds every few calls on this ROCm release.)
__global__ void k_fetch_params(const uint4* __restrict__ host_src, uint4* __restrict__ dst, uint32_t n16)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = host_src[i];
}

void launch_fetch_params(hipStream_t st, const void* host_mapped, void* dst, size_t bytes)
{
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    hipLaunchKernelGGL(k_fetch_params, dim3((n16 + 255) / 256), dim3(256), 0, st, static_cast<const uint4*>(host_mapped), static_cast<uint4*>(dst), n16);
}

void launch_fir_demod(hipStream_t st, uint32_t n_streams, uint32_t max_m, uint32_t max_taps, const float2* fbuf, size_t stride,
                      const float* taps, uint32_t taps_stride, float* demod, size_t demod_stride, float2* filtered,
                      const DemodCarry* carry_in, DemodCarry* carry_out, const StreamCall* call, uint32_t fir_hist_cap,
                      float* sym_ring, uint32_t ring_cap, const SymState* sym)
{
    const uint32_t tiles = max_m ? (max_m + kFirLanes - 2) / (kFirLanes - 1) : 1;
    const size_t lds = (size_t)(kFirLanes + (max_taps ? max_taps : 1)) * sizeof(float2);
    dim3 grid(tiles, n_streams);
    hipLaunchKernelGGL(k_fir_demod, grid, dim3(kFirLanes), lds, st, fbuf, stride, taps, taps_stride, demod, demod_stride, filtered,
                       carry_in, carry_out, call, fir_hist_cap, sym_ring, ring_cap, sym);
}

void launch_fbuf_shift(hipStream_t st, uint32_t n_streams, const float2* src, float2* dst, size_t stride, const StreamCall* call,
                       uint32_t fir_hist_cap)
{
    dim3 grid((fir_hist_cap + kFirBatch + 255) / 256, n_streams);
    hipLaunchKernelGGL(k_fbuf_shift, grid, dim3(256), 0, st, src, dst, stride, call, fir_hist_cap);
}

void launch_fft_feed(hipStream_t st, uint32_t n_streams, const float2* fbuf, size_t stride, float2* fft_in, const StreamCall* call,
                     uint32_t fir_hist_cap)
{
    dim3 grid(kFftBins / 256, n_streams);
    hipLaunchKernelGGL(k_fft_feed, grid, dim3(256), 0, st, fbuf, stride, fft_in, call, fir_hist_cap);
}

}  // namespace hd
