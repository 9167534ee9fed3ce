// Low-pass FIR fused with the FSK polar discriminator, plus the small data-movement kernels around it.
//
// FIR (reference code/Decoder/FirFilter.h:155-161): y[i] = sum_{t<T} buf[i+t]*tap[t], buf = [history(T-1) | pending
// decimated samples]; same sequential non-FMA accumulation as the decimator (bit-exact).  The discriminator
// (code/Decoder/FSK2_Demod.h:37-40) d[i] = arg(y[i] * conj(y[i-1])) is applied in the epilogue, so the filtered
// samples never go to HBM unless a parity test asks for them: per decimated sample the kernel reads 8 B and writes
// 4 B, the "8/D + 4/D" terms of the pipeline's byte model.
//
// Mapping: grid = (tiles, streams), 256 lanes per tile, SIX adjacent outputs per lane (6l .. 6l+5 within the tile) on a rolling
// register window: one 16-byte LDS read (two samples) feeds twelve packed multiply-adds, and the six independent sums let a wave
// issue back to back (one sum per lane waits out the adder on every tap).  Six, not four: consecutive lanes then start 48 bytes
// apart, and three 16-byte chunks is an odd stride -- the sixteen lanes of each ds_read_b128 lane group fall into sixteen different
// bank quads.  With four outputs per lane (32 bytes apart, round 3) two lanes of every group shared a quad: half of the kernel's LDS
// cycles were bank conflicts (profiles/r03_other_shapes_pmc_sq.txt).  Taps come in blocks of sixteen through the scalar cache, requested a block ahead.
// Tiles advance by 1530 outputs and overlap by six: lane 0 only recomputes the predecessor y[i-1] that lane 1's first output needs.  The per-stream input buffer keeps the FIR history right in front of the
// pending samples, so a tile's 512+T-1 inputs are one contiguous, coalesced read into LDS.  Taps are per stream and
// wave-uniform (scalar loads).
#include <hip/hip_runtime.h>

#include "arith.h"
#include "exact_math.h"
#include "launch.h"

namespace hd {
namespace HD_ARITH_NS {

#ifdef HD_STAMP_FIR   // diagnostic build only (tools/micro/fir_stamps.py): per-workgroup clocks of k_fir_demod
__device__ unsigned long long g_fir_stamps[8192 * 8];
extern "C" void HD_DBG_NAME(hd_debug_fir_stamps)(unsigned long long* host, size_t n) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fir_stamps), n * 8); }
#define FSTAMP_DECL unsigned long long fs_t = __builtin_amdgcn_s_memtime(), fs_acc[4] = {0, 0, 0, 0}; const unsigned long long fs_r0 = __builtin_amdgcn_s_memrealtime()
#define FSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); fs_acc[i] += t_ - fs_t; fs_t = t_; } while (0)
#define FSTAMP_WRITE() do { const uint32_t w_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 64 && w_ < 8192) { unsigned long long* g_ = g_fir_stamps + w_ * 8; \
        g_[0] = fs_r0; g_[1] = __builtin_amdgcn_s_memrealtime(); for (int i_ = 0; i_ < 4; ++i_) g_[2 + i_] = fs_acc[i_]; g_[6] = 1; } } while (0)
#else
#define FSTAMP_DECL do { } while (0)
#define FSTAMP(i) do { } while (0)
#define FSTAMP_WRITE() do { } while (0)
#endif

constexpr int kFirLanes = 256;
#ifndef HD_FIR_OUT
#define HD_FIR_OUT 6
#endif
constexpr int kFirOut = HD_FIR_OUT;              // adjacent outputs per lane (even; 6: lanes 48 bytes apart, conflict-free 16-byte LDS reads)
constexpr int kFirTile = kFirOut * kFirLanes;    // outputs computed per tile
constexpr int kFirAdvance = kFirTile - kFirOut;  // outputs written per tile (lane 0 only supplies lane 1's predecessor)
constexpr int kFirWin = (16 + kFirOut) / 2;      // 16-byte pairs a block of sixteen taps touches (samples 0 .. 14 + kFirOut of the lane's window)
constexpr int kFirSlack = 16 + 2 * kFirWin;      // samples a lane may read past its last tap (the re-read last block of a kFirOut-output window)
static_assert(kFirOut % 2 == 0 && kFirOut >= 2 && kFirOut <= 8, "outputs per lane: pairs of samples, lane windows 16-byte aligned");

typedef float fd_f32x2 __attribute__((ext_vector_type(2)));
typedef float fd_f32x4 __attribute__((ext_vector_type(4)));

// Sixteen taps for a lane's kFirOut adjacent outputs.  w[] holds the samples x[kFirOut l + t0 ..] of the block (pairs); output q takes sample
// j + q with tap j.  kFirOut independent sums: per tap kFirOut products, then kFirOut adds -- every sum still receives its products in ascending tap
// order with separately rounded multiply and add, and one wave keeps issuing back to back (a single chain waits out the add on every tap).
__device__ __forceinline__ void fir_block16(fd_f32x2 (&acc)[kFirOut], const fd_f32x4 (&w)[kFirWin], const float (&k)[16])
{
    HD_FIR_ARITH
    auto smp = [&](int i) -> fd_f32x2 { return (i & 1) ? w[i >> 1].zw : w[i >> 1].xy; };
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        fd_f32x2 pr[kFirOut];
#pragma unroll
        for (int q = 0; q < kFirOut; ++q) pr[q] = smp(j + q) * k[j];
#pragma unroll
        for (int q = 0; q < kFirOut; ++q) acc[q] = acc[q] + pr[q];
    }
}

__global__ __launch_bounds__(kFirLanes) void k_fir_demod(const float2* __restrict__ fbuf, size_t stride,
                                                          const float* __restrict__ taps, uint32_t taps_stride,
                                                          float* __restrict__ demod, size_t demod_stride,
                                                          float2* __restrict__ filtered,
                                                          const DemodCarry* __restrict__ carry_in,
                                                          DemodCarry* __restrict__ carry_out,
                                                          const StreamCall* __restrict__ call, uint32_t fir_hist_cap,
                                                          float* __restrict__ sym_ring, uint32_t ring_cap,
                                                          const SymState* __restrict__ sym, float2* __restrict__ fbuf_next,
                                                          float2* __restrict__ head_side /* [S][head_cap]: FirHistory heads moved aside (dev_types.h) */, uint32_t head_cap,
                                                          const float2* __restrict__ fbuf_prev /* the previous call's low-pass buffers */,
                                                          uint32_t* __restrict__ ck_acc /* [S][2]: the call's discriminator checksum, accumulated by the stream's tiles (or null) */,
                                                          const float2* __restrict__ pre /* fast mode, long filters: the low-pass output already computed by FFT (k_lp_gather ... below), [S][pre_stride], to be scaled by pre_scale; null = filter here */,
                                                          const uint32_t pre_stride, const float pre_scale)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];   // [kFirTile + T + kFirSlack] inputs, then reused for outputs
    __shared__ uint32_t s_ck[2];
    const uint32_t s = blockIdx.y;
    FSTAMP_DECL;
    const StreamCall c = call[s];
    const uint32_t m = c.fir_m, T = c.fir_taps;
    const uint32_t Tp = sc_taps_prev(c) ? sc_taps_prev(c) : T;     // tap count of the previous run
    const float2* buf = fbuf + (size_t)s * stride;                 // history occupies [fir_hist_cap-(T-1), fir_hist_cap)
    if (fbuf_next) {
        // Slide [history | leftover pending] to the front of the OTHER buffer for the next call (ping-pong instead of
        // an overlapping in-place move): next[k] = buf[k + fir_m], k < hist_cap + pend_after.  The stream's tiles share it.
        // (Round 6: after a run only the LAST T-1 history slots are moved -- whatever tap count the next run has, it reads its history out of those and out of the
        // FirHistory head (dev_types.h).  The whole region used to move on every call: fir_hist_cap is the filter's CAPACITY -- 133 KB per stream and call at /4,
        // 35 KB at /16, a quarter of k_fir_demod's traffic.  A call without a run moves the buffer as it is: what the last run left is not known here.)
        float2* nx = fbuf_next + (size_t)s * stride;
        const uint32_t lo = (m && T) ? fir_hist_cap - (T - 1u) : 0u;
        const uint32_t cnt = fir_hist_cap + c.pend_after - lo, off = c.clear_pending ? 0u : m;
        constexpr int SB = 4;
        for (uint32_t k0 = blockIdx.x * kFirLanes + threadIdx.x; k0 < cnt; k0 += SB * gridDim.x * kFirLanes) {
            float2 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) { const uint32_t k = k0 + u * gridDim.x * kFirLanes; v[u] = k < cnt ? buf[lo + k + off] : make_float2(0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < SB; ++u) { const uint32_t k = k0 + u * gridDim.x * kFirLanes; if (k < cnt) nx[lo + k] = v[u]; }
        }
    }
    // FirHistory head (dev_types.h): no run writes one; a stream that ran in the previous call and does not run in this one has its head moved aside
    const float2* head_prevbuf = fbuf_prev + (size_t)s * stride + fir_hist_cap;
    if (blockIdx.x == 0 && sc_head_save(c)) {
        float2* ho = head_side + (size_t)s * head_cap;
        for (uint32_t k = threadIdx.x; k < sc_head_n(c); k += kFirLanes) ho[k] = head_prevbuf[k];
    }
    const float2* head_in = sc_head_prev(c) ? head_prevbuf : head_side + (size_t)s * head_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0 && m == 0) carry_out[s] = carry_in[s];   // idle stream: carry passes through
    if (!m || !T) return;
    FSTAMP(0);
    const long i0 = (long)blockIdx.x * kFirAdvance - kFirOut;      // output index of lane 0's first output (may be -4)
    if (i0 + kFirOut >= (long)m) return;
    const long b0 = (long)fir_hist_cap - (long)(T - 1) + i0;       // buffer index of tile-local sample 0
    const uint32_t live = (uint32_t)min((long)kFirTile, (long)m - i0);   // outputs of this tile that exist
    const bool active = (uint32_t)kFirOut * threadIdx.x < live;
    fd_f32x2 acc[kFirOut];
#pragma unroll
    for (int q = 0; q < kFirOut; ++q) acc[q] = fd_f32x2{0.f, 0.f};
    if (pre) {
        // the filtered samples exist already (the transform route): this kernel is the discriminator, the symbol ring's append, the carries and the slide
        if (active) {
            const float2* ps = pre + (size_t)s * pre_stride;
#pragma unroll
            for (int q = 0; q < kFirOut; ++q) {
                const long i = i0 + (long)kFirOut * (long)threadIdx.x + q;
                if (i >= 0 && i < (long)m) { const float2 v = ps[i]; acc[q] = fd_f32x2{v.x * pre_scale, v.y * pre_scale}; }
            }
        }
    } else {
    const uint32_t need = ((live + (uint32_t)kFirOut - 1u) / (uint32_t)kFirOut) * (uint32_t)kFirOut + T + kFirSlack;   // + what whole blocks read past the last tap (zero-filled)
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
            const uint32_t head_n = sc_head_n(c);
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const long b = b0 + (long)(j0 + u * kFirLanes);
                if (j0 + u * kFirLanes < need && b >= (long)fir_hist_cap - (long)(T - 1) && b < (long)fir_hist_cap) {
                    const uint32_t j = (uint32_t)(b - ((long)fir_hist_cap - (long)(T - 1)));      // history slot, 0 = oldest
                    v[u] = make_float2(0.f, 0.f);
                    if (j < Tp - 1) v[u] = buf[fir_hist_cap - (Tp - 1) + j];
                    else if (j - (Tp - 1) < head_n) v[u] = head_in[j - (Tp - 1)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) if (j0 + u * kFirLanes < need) lds[j0 + u * kFirLanes] = v[u];
    }
    __syncthreads();
    FSTAMP(1);

    typedef const float __attribute__((address_space(4)))* ctaps_t;
    const ctaps_t tp = (ctaps_t)(uintptr_t)(taps + (size_t)s * taps_stride);   // per stream, wave-uniform: scalar loads
    if (active) {
        // lane l's outputs 6l .. 6l+5 take samples 6l + t + q: a rolling window of twenty-two samples per sixteen taps, eleven 16-byte reads per
        // block of 192 packed operations, the next block's samples and taps requested before this block's arithmetic.
        const fd_f32x4* p = reinterpret_cast<const fd_f32x4*>(lds) + (uint32_t)(kFirOut / 2) * threadIdx.x;    // pair index: sample kFirOut l + 2k
        const uint32_t nblk = T / 16u;
        fd_f32x4 w[kFirWin], wn[kFirWin];
        float k[16], kn[16];
        auto load_block = [&](fd_f32x4 (&ww)[kFirWin], float (&kk)[16], const uint32_t b) {
            const fd_f32x4* pn = p + 8u * b;
#pragma unroll
            for (int i = 0; i < kFirWin; ++i) ww[i] = pn[i];
            const ctaps_t tn = tp + 16u * b;
#pragma unroll
            for (int j = 0; j < 16; ++j) kk[j] = tn[j];
        };
        if (nblk) load_block(w, k, 0u);
        // two blocks per trip, the two register images taking turns: the block after next is requested into the image just used up, so
        // nothing is copied between trips (a one-block loop moved ten 16-byte pairs and sixteen taps per 128 packed operations)
        uint32_t b = 0;
        for (; b + 2u <= nblk; b += 2u) {
            load_block(wn, kn, b + 1u);
            fir_block16(acc, w, k);
            load_block(w, k, b + 2u < nblk ? b + 2u : nblk - 1u);      // (unconditional: the last trip re-reads the last block, unused)
            fir_block16(acc, wn, kn);
        }
        if (b < nblk) fir_block16(acc, w, k);                       // an odd count: the last block sits in the first image
        for (uint32_t t = nblk * 16u; t < T; ++t) {                 // the taps behind the last whole block, one at a time
            HD_FIR_ARITH
            const float kt = tp[t];
            const float2* x = lds + (uint32_t)kFirOut * threadIdx.x + t;
            fd_f32x2 pr[kFirOut];
#pragma unroll
            for (int q = 0; q < kFirOut; ++q) { const float2 v = x[q]; pr[q] = fd_f32x2{v.x, v.y} * kt; }
#pragma unroll
            for (int q = 0; q < kFirOut; ++q) acc[q] = acc[q] + pr[q];
        }
    }
    }
    FSTAMP(2);
    __syncthreads();                       // everyone is done reading inputs: reuse LDS for the outputs
    if (active) {
#pragma unroll
        for (int q = 0; q < kFirOut; q += 2)
            reinterpret_cast<float4*>(lds)[(uint32_t)(kFirOut / 2) * threadIdx.x + (uint32_t)(q / 2)] = make_float4(acc[q].x, acc[q].y, acc[q + 1].x, acc[q + 1].y);
    }
    if (threadIdx.x < 2) s_ck[threadIdx.x] = 0u;
    __syncthreads();

    if (threadIdx.x != 0 && active) {                   // (lane 0 only supplied lane 1's predecessor)
    const long i = i0 + (long)kFirOut * (long)threadIdx.x;         // >= 0 here
    float pr, pi;
    if (i > 0) { const float2 q = lds[kFirOut * threadIdx.x - 1]; pr = q.x; pi = q.y; }
    else {
        const DemodCarry kc = carry_in[s];
        if (kc.primed) { pr = kc.re; pi = kc.im; } else { pr = acc[0].x; pi = acc[0].y; }   // very first sample: arg(y0*conj(y0))
    }
    const uint32_t nv = (uint32_t)min((long)kFirOut, (long)m - i);  // outputs of this lane that exist
    float d[kFirOut];
    d[0] = discriminate(acc[0].x, acc[0].y, pr, pi);
#pragma unroll
    for (int q = 1; q < kFirOut; ++q) d[q] = (uint32_t)q < nv ? discriminate(acc[q].x, acc[q].y, acc[q - 1].x, acc[q - 1].y) : 0.f;
    float* dm = demod + (size_t)s * demod_stride + i;                // i is even, demod_stride is even
#pragma unroll
    for (int q = 0; q < kFirOut; q += 2) {
        if ((uint32_t)q + 1u < nv) *reinterpret_cast<float2*>(dm + q) = make_float2(d[q], d[q + 1]);
        else if ((uint32_t)q < nv) dm[q] = d[q];
    }
    if (sym_ring) {      // append straight into the symbol extractor's ring (SymbolExtractor::pushSamples); a vent
        const SymState st = sym[s];   // (backlog > 30000) restarts the backlog at the same position base + held
        float* ring = sym_ring + (size_t)s * ring_cap;
        const uint32_t pos = st.base + st.held + (uint32_t)i;
#pragma unroll
        for (int q = 0; q < kFirOut; ++q) if ((uint32_t)q < nv) ring[(pos + (uint32_t)q) & (ring_cap - 1)] = d[q];
    }
    if (filtered) {
#pragma unroll
        for (int q = 0; q < kFirOut; ++q) if ((uint32_t)q < nv) filtered[(size_t)s * demod_stride + i + q] = make_float2(acc[q].x, acc[q].y);
    }
    const long last = (long)m - 1;
    if (i <= last && last < i + (long)kFirOut) {
        DemodCarry kc; kc.primed = 1; kc._pad = 0;
        fd_f32x2 y = acc[0];
#pragma unroll
        for (int q = 1; q < kFirOut; ++q) if (last == i + q) y = acc[q];
        kc.re = y.x; kc.im = y.y;
        carry_out[s] = kc;
    }
    if (ck_acc) {       // BitsHeader::demod_ck of this call: sum of the outputs' bit patterns, and of (index + 1) * bit pattern (mod 2^32)
        uint32_t c0 = 0, c1 = 0;
#pragma unroll
        for (int q = 0; q < kFirOut; ++q) if ((uint32_t)q < nv) { const uint32_t b = __builtin_bit_cast(uint32_t, d[q]); c0 += b; c1 += ((uint32_t)i + (uint32_t)q + 1u) * b; }
        atomicAdd(&s_ck[0], c0); atomicAdd(&s_ck[1], c1);
    }
    }
    if (ck_acc) {
        __syncthreads();
        if (threadIdx.x < 2) atomicAdd(&ck_acc[2 * (size_t)s + threadIdx.x], s_ck[threadIdx.x]);     // (k_symbols of this call hands the sums to the result slot and clears them)
    }
    FSTAMP(3);
    FSTAMP_WRITE();
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
                      float2* head_side, uint32_t head_cap, const float2* fbuf_prev, uint32_t* ck_acc,
                      const float2* pre, uint32_t pre_stride, float pre_scale)
{
    const uint32_t tiles = max_m ? (max_m + kFirAdvance - 1) / kFirAdvance : 1;
    const size_t lds = (size_t)(kFirTile + (max_taps ? max_taps : 1) + kFirSlack + 4) * sizeof(float2);
    dim3 grid(tiles, n_streams);
    hipLaunchKernelGGL(k_fir_demod, grid, dim3(kFirLanes), lds, st, fbuf, stride, taps, taps_stride, demod, demod_stride, filtered,
                       carry_in, carry_out, call, fir_hist_cap, sym_ring, ring_cap, sym, fbuf_next, head_side, head_cap, fbuf_prev, ck_acc, pre, pre_stride, pre_scale);
}

// ---- Fast mode, long low-pass filters (configs[4]: 4097 taps over 4096 samples per call = 67 MFLOP per stream and call done directly): the same correlation
// y[i] = sum_t buf[i + t] tap[t] (FirFilter.h:155-161) through N-point transforms -- x = [history (T-1) | the run's m inputs | zeros], Y = IFFT(FFT(x) conj(FFT(taps))),
// y[i] = Y[i] / N for i < m (no wrap-around: T - 1 + m <= N).  rocFFT does the transforms in place on a [S][N] work buffer; these kernels build the input image --
// with the history exactly as k_fir_demod's tile loader builds it: zeros on a restart (Q5), the FirHistory reconstruction after a tap-count change -- and
// multiply the spectra; k_fir_demod then runs with `pre` set.  The result differs from the direct sum by the transforms' rounding (~1e-6 of the input's peak,
// tests/test_gpu_fast.py); the exact mode never takes this route.
__global__ __launch_bounds__(256) void k_lp_gather(const float2* __restrict__ fbuf, size_t stride, float2* __restrict__ work, uint32_t N, const StreamCall* __restrict__ call,
                                                    uint32_t fir_hist_cap, const float2* __restrict__ head_side, uint32_t head_cap, const float2* __restrict__ fbuf_prev)
{
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t m = c.fir_m, T = c.fir_taps;
    if (!m || !T) return;
    const uint32_t Tp = sc_taps_prev(c) ? sc_taps_prev(c) : T, H = T - 1, L = H + m;
    const float2* buf = fbuf + (size_t)s * stride;
    const bool refold = Tp != T && !c.fir_zero_hist;
    const uint32_t head_n = refold ? sc_head_n(c) : 0u;
    const float2* head_in = sc_head_prev(c) ? fbuf_prev + (size_t)s * stride + fir_hist_cap : head_side + (size_t)s * head_cap;
    float2* w = work + (size_t)s * N;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < N; j += gridDim.x * 256u) {
        float2 v = make_float2(0.f, 0.f);
        if (j < L) {
            if (j >= H) v = buf[fir_hist_cap + (j - H)];                     // the run's inputs: pending + new decimated samples
            else if (c.fir_zero_hist) v = make_float2(0.f, 0.f);              // history restarts from zeros
            else if (!refold) v = buf[fir_hist_cap - H + j];
            else if (j < Tp - 1) v = buf[fir_hist_cap - (Tp - 1) + j];        // first run after a tap-count change (FirHistory, dev_types.h)
            else if (j - (Tp - 1) < head_n) v = head_in[j - (Tp - 1)];
        }
        w[j] = v;
    }
}
// the taps as a complex sequence of N (zero-padded): transformed once per design into `kf`
__global__ __launch_bounds__(256) void k_lp_taps_gather(const float* __restrict__ taps, uint32_t taps_stride, const uint32_t* __restrict__ ntaps, float2* __restrict__ kf, uint32_t N)
{
    const uint32_t s = blockIdx.y, T = ntaps[s];
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < N; j += gridDim.x * 256u)
        kf[(size_t)s * N + j] = make_float2(j < T ? taps[(size_t)s * taps_stride + j] : 0.f, 0.f);
}
// X[f] *= conj(K[f])
__global__ __launch_bounds__(256) void k_lp_mul(float2* __restrict__ work, const float2* __restrict__ kf, uint32_t N, const StreamCall* __restrict__ call)
{
    const uint32_t s = blockIdx.y;
    if (!call[s].fir_m || !call[s].fir_taps) return;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < N; j += gridDim.x * 256u) {
        const float2 x = work[(size_t)s * N + j], k = kf[(size_t)s * N + j];
        work[(size_t)s * N + j] = make_float2(x.x * k.x + x.y * k.y, x.y * k.x - x.x * k.y);
    }
}
void launch_lp_gather(hipStream_t st, uint32_t n_streams, const float2* fbuf, size_t stride, float2* work, uint32_t N, const StreamCall* call, uint32_t fir_hist_cap,
                      const float2* head_side, uint32_t head_cap, const float2* fbuf_prev)
{
    hipLaunchKernelGGL(k_lp_gather, dim3((N + 1023) / 1024, n_streams), dim3(256), 0, st, fbuf, stride, work, N, call, fir_hist_cap, head_side, head_cap, fbuf_prev);
}
void launch_lp_taps_gather(hipStream_t st, uint32_t n_streams, const float* taps, uint32_t taps_stride, const uint32_t* ntaps, float2* kf, uint32_t N)
{
    hipLaunchKernelGGL(k_lp_taps_gather, dim3((N + 1023) / 1024, n_streams), dim3(256), 0, st, taps, taps_stride, ntaps, kf, N);
}
void launch_lp_mul(hipStream_t st, uint32_t n_streams, float2* work, const float2* kf, uint32_t N, const StreamCall* call)
{
    hipLaunchKernelGGL(k_lp_mul, dim3((N + 1023) / 1024, n_streams), dim3(256), 0, st, work, kf, N, call);
}

void launch_fft_feed(hipStream_t st, uint32_t n_streams, const float2* fbuf, size_t stride, float2* fft_in, const StreamCall* call,
                     uint32_t fir_hist_cap)
{
    dim3 grid(kFftBins / 256, n_streams);
    hipLaunchKernelGGL(k_fft_feed, grid, dim3(256), 0, st, fbuf, stride, fft_in, call, fir_hist_cap);
}

}  // namespace HD_ARITH_NS
}  // namespace hd
