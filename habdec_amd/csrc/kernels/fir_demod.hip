// Low-pass FIR fused with the FSK polar discriminator, plus the small data-movement kernels around it.
//
// FIR (reference code/Decoder/FirFilter.h:155-161): y[i] = sum_{t<T} buf[i+t]*tap[t], buf = [history(T-1) | pending
// decimated samples]; same sequential non-FMA accumulation as the decimator (bit-exact).  The discriminator
// (code/Decoder/FSK2_Demod.h:37-40) d[i] = arg(y[i] * conj(y[i-1])) is applied in the epilogue, so the filtered
// samples never go to HBM unless a parity test asks for them: per decimated sample the kernel reads 8 B and writes
// 4 B, the "8/D + 4/D" terms of the pipeline's byte model.
//
// Mapping: grid = (tiles, streams), 256 lanes per tile, TWO adjacent outputs per lane (2l, 2l+1 within the tile), so
// that one 16-byte LDS read (samples 2l+2k, 2l+2k+1) feeds four multiply-adds: the kernel is bound by the LDS read
// rate (8 B per complex sample x tap at one output per lane), and two outputs per lane halve that to the level of the
// packed-f32 issue rate.  Tiles advance by 510 outputs and overlap by two: lane 0 only recomputes the predecessor
// y[i-1] that lane 1's first output needs.  The per-stream input buffer keeps the FIR history right in front of the
// pending samples, so a tile's 512+T-1 inputs are one contiguous, coalesced read into LDS.  Taps are per stream and
// wave-uniform (scalar loads).
#include <hip/hip_runtime.h>

#include "exact_math.h"
#include "launch.h"

namespace hd {

constexpr int kFirLanes = 256;
constexpr int kFirTile = 2 * kFirLanes;          // outputs computed per tile
constexpr int kFirAdvance = kFirTile - 2;        // outputs written per tile

#define HD_FIR_PAIR(P, N, k0, k1)                  \
    a0r = a0r + (P).x * (k0); a0i = a0i + (P).y * (k0); \
    a1r = a1r + (P).z * (k0); a1i = a1i + (P).w * (k0); \
    a0r = a0r + (P).z * (k1); a0i = a0i + (P).w * (k1); \
    a1r = a1r + (N).x * (k1); a1i = a1i + (N).y * (k1);

__global__ __launch_bounds__(kFirLanes) void k_fir_demod(const float2* __restrict__ fbuf, size_t stride,
                                                          const float* __restrict__ taps, uint32_t taps_stride,
                                                          float* __restrict__ demod, size_t demod_stride,
                                                          float2* __restrict__ filtered,
                                                          const DemodCarry* __restrict__ carry_in,
                                                          DemodCarry* __restrict__ carry_out,
                                                          const StreamCall* __restrict__ call, uint32_t fir_hist_cap,
                                                          float* __restrict__ sym_ring, uint32_t ring_cap,
                                                          const SymState* __restrict__ sym, float2* __restrict__ fbuf_next,
                                                          const float2* __restrict__ head_in, const uint32_t* __restrict__ head_n_in,
                                                          float2* __restrict__ head_out, uint32_t* __restrict__ head_n_out, uint32_t head_cap)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];   // [kFirTile + T + 1] inputs, then reused for outputs
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t m = c.fir_m, T = c.fir_taps;
    const uint32_t Tp = c.fir_taps_prev ? c.fir_taps_prev : T;     // tap count of the previous run
    const float2* buf = fbuf + (size_t)s * stride;                 // history occupies [fir_hist_cap-(T-1), fir_hist_cap)
    if (fbuf_next) {
        // Slide [history | leftover pending] to the front of the OTHER buffer for the next call (ping-pong instead of
        // an overlapping in-place move): next[k] = buf[k + fir_m], k < hist_cap + pend_after.  The stream's tiles share it.
        float2* nx = fbuf_next + (size_t)s * stride;
        const uint32_t cnt = fir_hist_cap + c.pend_after, off = c.clear_pending ? 0u : m;
        constexpr int SB = 4;
        for (uint32_t k0 = blockIdx.x * kFirLanes + threadIdx.x; k0 < cnt; k0 += SB * gridDim.x * kFirLanes) {
            float2 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) { const uint32_t k = k0 + u * gridDim.x * kFirLanes; v[u] = k < cnt ? buf[k + off] : make_float2(0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < SB; ++u) { const uint32_t k = k0 + u * gridDim.x * kFirLanes; if (k < cnt) nx[k] = v[u]; }
        }
    }
    // head of this run's input for a later run with a different tap count (FirHistory, dev_types.h); an idle stream keeps its old one
    if (blockIdx.x == 0) {
        const float2* hi = head_in + (size_t)s * head_cap;
        float2* ho = head_out + (size_t)s * head_cap;
        const uint32_t hn = (m && T) ? min(m, head_cap) : head_n_in[s];
        for (uint32_t k = threadIdx.x; k < hn; k += kFirLanes) ho[k] = (m && T) ? buf[fir_hist_cap + k] : hi[k];
        if (threadIdx.x == 0) head_n_out[s] = hn;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && m == 0) carry_out[s] = carry_in[s];   // idle stream: carry passes through
    if (!m || !T) return;
    const long i0 = (long)blockIdx.x * kFirAdvance - 2;            // output index of lane 0's first output (may be -2)
    if (i0 + 2 >= (long)m) return;
    const long b0 = (long)fir_hist_cap - (long)(T - 1) + i0;       // buffer index of tile-local sample 0
    const uint32_t live = (uint32_t)min((long)kFirTile, (long)m - i0);   // outputs of this tile that exist
    const uint32_t need = ((live + 1u) & ~1u) + T + 1;             // + the pair read one past the last tap
    const long end = (long)fir_hist_cap + (long)m;                 // one past the last valid input
    constexpr int LB = 4;                                          // loads in flight per lane before the first LDS store
    for (uint32_t j0 = threadIdx.x; j0 < need; j0 += LB * kFirLanes) {
        float2 v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const long b = b0 + (long)(j0 + u * kFirLanes);
            v[u] = make_float2(0.f, 0.f);
            if (j0 + u * kFirLanes < need && b >= 0 && b < end && !(c.fir_zero_hist && b < (long)fir_hist_cap)) v[u] = buf[b];
        }
        if (Tp != T && !c.fir_zero_hist) {                       // first run after a tap-count change (wave-uniform, rare): FirHistory
            const uint32_t head_n = head_n_in[s];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const long b = b0 + (long)(j0 + u * kFirLanes);
                if (j0 + u * kFirLanes < need && b >= (long)fir_hist_cap - (long)(T - 1) && b < (long)fir_hist_cap) {
                    const uint32_t j = (uint32_t)(b - ((long)fir_hist_cap - (long)(T - 1)));      // history slot, 0 = oldest
                    v[u] = make_float2(0.f, 0.f);
                    if (j < Tp - 1) v[u] = buf[fir_hist_cap - (Tp - 1) + j];
                    else if (j - (Tp - 1) < head_n) v[u] = head_in[(size_t)s * head_cap + j - (Tp - 1)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) if (j0 + u * kFirLanes < need) lds[j0 + u * kFirLanes] = v[u];
    }
    __syncthreads();

    const float* tp = taps + (size_t)s * taps_stride;
    const bool active = 2u * threadIdx.x < live;
    float a0r = 0.f, a0i = 0.f, a1r = 0.f, a1i = 0.f;
    if (active) {
        const float4* p = reinterpret_cast<const float4*>(lds) + threadIdx.x;    // pair k: samples 2l+2k, 2l+2k+1
        uint32_t t = 0;
        float4 P = p[0];
        if (T >= 8) {       // software-pipelined: the LDS pairs and the (scalar) taps of block t+8 are requested before block t's math
            float4 N0 = p[1], N1 = p[2], N2 = p[3], N3 = p[4];
            float k0 = tp[0], k1 = tp[1], k2 = tp[2], k3 = tp[3], k4 = tp[4], k5 = tp[5], k6 = tp[6], k7 = tp[7];
            for (; t + 8 <= T; t += 8) {
                float4 M0 = N3, M1 = N3, M2 = N3, M3 = N3;
                float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f, q6 = 0.f, q7 = 0.f;
                if (t + 16 <= T) {
                    const float4* pn = p + (t >> 1) + 5;
                    M0 = pn[0]; M1 = pn[1]; M2 = pn[2]; M3 = pn[3];
                    const float* tn = tp + t + 8;
                    q0 = tn[0]; q1 = tn[1]; q2 = tn[2]; q3 = tn[3]; q4 = tn[4]; q5 = tn[5]; q6 = tn[6]; q7 = tn[7];
                }
                HD_FIR_PAIR(P, N0, k0, k1)
                HD_FIR_PAIR(N0, N1, k2, k3)
                HD_FIR_PAIR(N1, N2, k4, k5)
                HD_FIR_PAIR(N2, N3, k6, k7)
                P = N3;
                N0 = M0; N1 = M1; N2 = M2; N3 = M3;
                k0 = q0; k1 = q1; k2 = q2; k3 = q3; k4 = q4; k5 = q5; k6 = q6; k7 = q7;
            }
        }
        for (; t + 2 <= T; t += 2) {
            const float4 N = p[(t >> 1) + 1];
            const float k0 = tp[t], k1 = tp[t + 1];
            HD_FIR_PAIR(P, N, k0, k1)
            P = N;
        }
        if (t < T) {
            const float k0 = tp[t];
            a0r = a0r + P.x * k0; a0i = a0i + P.y * k0;
            a1r = a1r + P.z * k0; a1i = a1i + P.w * k0;
        }
    }
    __syncthreads();                       // everyone is done reading inputs: reuse LDS for the outputs
    if (active) reinterpret_cast<float4*>(lds)[threadIdx.x] = make_float4(a0r, a0i, a1r, a1i);
    __syncthreads();

    if (threadIdx.x == 0 || !active) return;           // lane 0 only supplied lane 1's predecessor
    const long i = i0 + 2l * (long)threadIdx.x;        // >= 0 here
    float pr, pi;
    if (i > 0) { const float2 q = lds[2 * threadIdx.x - 1]; pr = q.x; pi = q.y; }
    else {
        const DemodCarry k = carry_in[s];
        if (k.primed) { pr = k.re; pi = k.im; } else { pr = a0r; pi = a0i; }   // very first sample: arg(y0*conj(y0))
    }
    const bool two = i + 1 < (long)m;
    const float d0 = discriminate(a0r, a0i, pr, pi);
    const float d1 = two ? discriminate(a1r, a1i, a0r, a0i) : 0.f;
    float* dm = demod + (size_t)s * demod_stride + i;
    if (two) *reinterpret_cast<float2*>(dm) = make_float2(d0, d1); else dm[0] = d0;     // i is even, demod_stride is even
    if (sym_ring) {      // append straight into the symbol extractor's ring (SymbolExtractor::pushSamples); a vent
        const SymState st = sym[s];   // (backlog > 30000) restarts the backlog at the same position base + held
        float* ring = sym_ring + (size_t)s * ring_cap;
        const uint32_t pos = st.base + st.held + (uint32_t)i;
        ring[pos & (ring_cap - 1)] = d0;
        if (two) ring[(pos + 1) & (ring_cap - 1)] = d1;
    }
    if (filtered) {
        filtered[(size_t)s * demod_stride + i] = make_float2(a0r, a0i);
        if (two) filtered[(size_t)s * demod_stride + i + 1] = make_float2(a1r, a1i);
    }
    const long last = (long)m - 1;
    if (i == last || (two && i + 1 == last)) {
        DemodCarry k; k.primed = 1; k._pad = 0;
        if (i == last) { k.re = a0r; k.im = a0i; } else { k.re = a1r; k.im = a1i; }
        carry_out[s] = k;
    }
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
                      float* sym_ring, uint32_t ring_cap, const SymState* sym, float2* fbuf_next,
                      const float2* head_in, const uint32_t* head_n_in, float2* head_out, uint32_t* head_n_out, uint32_t head_cap)
{
    const uint32_t tiles = max_m ? (max_m + kFirAdvance - 1) / kFirAdvance : 1;
    const size_t lds = (size_t)(kFirTile + (max_taps ? max_taps : 1) + 2) * sizeof(float2);
    dim3 grid(tiles, n_streams);
    hipLaunchKernelGGL(k_fir_demod, grid, dim3(kFirLanes), lds, st, fbuf, stride, taps, taps_stride, demod, demod_stride, filtered,
                       carry_in, carry_out, call, fir_hist_cap, sym_ring, ring_cap, sym, fbuf_next, head_in, head_n_in, head_out, head_n_out, head_cap);
}

void launch_fft_feed(hipStream_t st, uint32_t n_streams, const float2* fbuf, size_t stride, float2* fft_in, const StreamCall* call,
                     uint32_t fir_hist_cap)
{
    dim3 grid(kFftBins / 256, n_streams);
    hipLaunchKernelGGL(k_fft_feed, grid, dim3(256), 0, st, fbuf, stride, fft_in, call, fir_hist_cap);
}

}  // namespace hd
