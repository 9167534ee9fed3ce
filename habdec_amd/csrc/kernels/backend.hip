// Fused back end for two-stage decimation plans: second FIR-decimate stage + low-pass FIR + FSK discriminator + the
// buffer slide, one workgroup per stream.
//
// At the headline shape (1024 streams, 2048 stage-1 samples per stream and call) the separate kernels
// (k_decimate<2,69,256>, k_fir_demod) are pure latency: a few hundred bytes per workgroup behind four or five dependent
// global round trips each, plus a launch and an end-of-kernel cache write-back apiece.  Here a stream's whole call lives
// in LDS: the stage-1 output chunk with its stage-2 history, the stage-2 output appended behind the low-pass history and
// the pending samples, and every global read the workgroup needs is requested up front (one round trip).
//
// Arithmetic is the same as in decimate.hip / fir_demod.hip: per output a T-term sum in ascending tap order with
// separately rounded multiply and add (reference Decimator.h:128-138, FirFilter.h:155-161; compiled with
// -ffp-contract=off), then arg(y[i] * conj(y[i-1])) (FSK2_Demod.h:37-40).  Buffer conventions (history in front of the
// pending samples, ping-pong slide, spectrum feed from the head of the chunk, history carry with the in-place quirk Q4)
// are those of the kernels it replaces, so the engine can switch between the two paths from call to call.
//
// LDS: 28.5 KB per stream at the headline shape.  A variant that walks stage 2 in two half-size images (20 KB, two
// workgroups beside six stage-1 workgroups per CU) was measured and dropped: alone it is as fast, next to a classic stage-1
// grid it gains 7 %, but under the 6-per-CU stage-1 split that batch mode uses it LOSES 8 % (two back-end workgroups per CU
// take more VALU/LDS time from stage 1 than their head start returns).
#include <hip/hip_runtime.h>

#include "arith.h"
#include "exact_math.h"
#include "launch.h"

namespace hd {
namespace HD_ARITH_NS {

constexpr int kBeLanes = 256;
constexpr int kBePass = 2 * kBeLanes;            // low-pass outputs per pass: two adjacent outputs per lane

#define HD_BE_PAIR(P, N, k0, k1)                        \
    a0r = a0r + (P).x * (k0); a0i = a0i + (P).y * (k0); \
    a1r = a1r + (P).z * (k0); a1i = a1i + (P).w * (k0); \
    a0r = a0r + (P).z * (k1); a0i = a0i + (P).w * (k1); \
    a1r = a1r + (N).x * (k1); a1i = a1i + (N).y * (k1);

#ifdef HD_STAMP_BE   // diagnostic build only (tools/micro/be_stamps.py): s_memtime at the phase boundaries, per stream
__device__ unsigned long long g_be_stamps[8192 * 8];
#define BSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_be_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" void HD_DBG_NAME(hd_debug_be_stamps)(unsigned long long* host, size_t n) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_be_stamps), n * 8); }
#else
#define BSTAMP(i) do { } while (0)
#endif

template <int D2, int T2>
__global__ __launch_bounds__(kBeLanes) void k_backend(const float2* __restrict__ dec1, size_t dec1_stride,
                                                       const float2* __restrict__ hist2_in, float2* __restrict__ hist2_out,
                                                       const float* __restrict__ taps2,
                                                       const float2* __restrict__ fbuf, float2* __restrict__ fbuf_w,
                                                       float2* __restrict__ fbuf_next, size_t fbuf_stride, uint32_t fir_hist_cap,
                                                       const float* __restrict__ lp_taps, uint32_t taps_stride,
                                                       float* __restrict__ demod, size_t demod_stride, float2* __restrict__ filtered,
                                                       const DemodCarry* __restrict__ carry_in, DemodCarry* __restrict__ carry_out,
                                                       const StreamCall* __restrict__ call, float2* __restrict__ fft_in,
                                                       float* __restrict__ sym_ring, uint32_t ring_cap, const SymState* __restrict__ sym,
                                                       uint32_t xin_cap /* float2 slots of the stage-2 input image, even */,
                                                       float2* __restrict__ head_buf /* [S][head_cap]: FirHistory heads moved aside (dev_types.h) */,
                                                       uint32_t head_cap, const float2* __restrict__ fbuf_prev /* the previous call's low-pass buffers */)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    // [xin: (T2-1) history + n1 chunk samples; reused for the low-pass outputs of a pass][fin: (T-1) history | pending | new]
    float2* xin = smem;
    float2* fin = smem + xin_cap;
    __shared__ float2 s_ylast;
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    const StreamCall c = call[s];
    const uint32_t n1 = c.n1, n2 = c.n2, m = c.fir_m, T = c.fir_taps, pb = c.pend_before;
    const float2* in_s = dec1 + (size_t)s * dec1_stride;
    const float2* cur = fbuf + (size_t)s * fbuf_stride;
    float2* cur_w = fbuf_w + (size_t)s * fbuf_stride;
    float2* nxt = fbuf_next + (size_t)s * fbuf_stride;
    const uint32_t H = T ? T - 1 : 0;                       // low-pass history length in use
    const uint32_t Tp = sc_taps_prev(c) ? sc_taps_prev(c) : T;   // tap count of the previous run (FirHistory, dev_types.h)
    float2* head_side = head_buf + (size_t)s * head_cap;
    const float2* head_prevbuf = fbuf_prev + (size_t)s * fbuf_stride + fir_hist_cap;
    const float2* head_in = sc_head_prev(c) ? head_prevbuf : head_side;
    const uint32_t f_old = H + pb;                          // fin slots that come from global memory

    BSTAMP(0);
    // ---- every global read, issued back to back: stage-2 history, the stage-1 chunk (16-byte loads), low-pass history +
    // pending, the deep part of the slide (buffer -> buffer), and one word per 64-byte line of this stream's low-pass taps
    // (each stream has its own tap vector: without the touch the tap loop's first pass waits on a scalar-cache miss per block)
    const float* tp = lp_taps + (size_t)s * taps_stride;
    const uint32_t sl_cnt = fir_hist_cap + c.pend_after, sl_off = c.clear_pending ? 0u : m;
    const uint32_t lds_from = fir_hist_cap - H;             // buffer indices from here on are imaged in `fin`
    const uint32_t sl_deep = lds_from > sl_off ? min(lds_from - sl_off, sl_cnt) : 0u;   // slide elements whose source is below the image
    float tap_touch = 0.f;
    {
        constexpr int XB = 6, FB = 3, SB = 6;               // first batch covers n1 <= 3072, H + pending <= 768, 1536 deep slide elements
        {   // unconditional, independent scalar loads (index clamped into the stream's tap vector): issued back to back
            const uint32_t tl = T ? T - 1 : 0;
            float tt[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) tt[u] = tp[min((uint32_t)u * 16u, tl)];
#pragma unroll
            for (int u = 0; u < 16; ++u) tap_touch += tt[u];
        }
        // element k of the low-pass image [history (H) | pending]: the buffer as it is, or -- first run after a tap-count change
        // -- the reference's view of its one buffer (FirHistory, dev_types.h)
        const bool refold = Tp != T && !c.fir_zero_hist;          // wave-uniform, and false except right after a redesign
        const uint32_t head_n = refold ? sc_head_n(c) : 0u;
        auto fold = [&](uint32_t k) -> float2 {
            if (k < H && c.fir_zero_hist) return make_float2(0.f, 0.f);
            if (!refold || k >= H) return cur[fir_hist_cap - H + k];
            if (k < Tp - 1) return cur[fir_hist_cap - (Tp - 1) + k];
            const uint32_t h = k - (Tp - 1);
            return h < head_n ? head_in[h] : make_float2(0.f, 0.f);
        };
        float2 ts[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) { const uint32_t k = tid + u * kBeLanes; ts[u] = k < sl_deep ? cur[k + sl_off] : make_float2(0.f, 0.f); }
        float2 th = make_float2(0.f, 0.f);
        if (tid < (uint32_t)(T2 - 1) && !c.zero_hist2) th = hist2_in[(size_t)s * (T2 - 1) + tid];
        float4 tx[XB];
        float2 tf[FB];
        const float4* in4 = reinterpret_cast<const float4*>(in_s);
        const uint32_t n1p = n1 >> 1;                       // n1 is a multiple of D2 (even for D2 = 2, 4)
#pragma unroll
        for (int u = 0; u < XB; ++u) { const uint32_t k = tid + u * kBeLanes; tx[u] = k < n1p ? in4[k] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const uint32_t k = tid + u * kBeLanes;
            tf[u] = make_float2(0.f, 0.f);
            if (k < f_old) tf[u] = cur[fir_hist_cap - H + k];             // steady state: one plain load per element
        }
        if (tid < (uint32_t)(T2 - 1)) xin[tid] = th;
        if (c.fir_zero_hist || refold) {                                  // history restarts from zeros / first run after a redesign
#pragma unroll
            for (int u = 0; u < FB; ++u) { const uint32_t k = tid + u * kBeLanes; if (k < H) tf[u] = fold(k); }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) { const uint32_t k = tid + u * kBeLanes; if (k < sl_deep) nxt[k] = ts[u]; }
        for (uint32_t k = tid + SB * kBeLanes; k < sl_deep; k += kBeLanes) nxt[k] = cur[k + sl_off];
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const uint32_t k = tid + u * kBeLanes;
            if (k < n1p) { xin[(T2 - 1) + 2 * k] = make_float2(tx[u].x, tx[u].y); xin[(T2 - 1) + 2 * k + 1] = make_float2(tx[u].z, tx[u].w); }
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) { const uint32_t k = tid + u * kBeLanes; if (k < f_old) fin[k] = tf[u]; }
        for (uint32_t k = tid + XB * kBeLanes; k < n1p; k += kBeLanes) {
            const float4 v = in4[k];
            xin[(T2 - 1) + 2 * k] = make_float2(v.x, v.y); xin[(T2 - 1) + 2 * k + 1] = make_float2(v.z, v.w);
        }
        for (uint32_t k = tid + FB * kBeLanes; k < f_old; k += kBeLanes) fin[k] = fold(k);
    }
    __syncthreads();
    BSTAMP(1);

    // ---- stage-2 decimation: y2[o] = sum_t x[o*D2 + t] * h2[t], one output per lane and step.  Lane o starts at sample
    // o*D2 (even), so its samples come as 16-byte pairs; for D2 = 2 consecutive lanes read consecutive pairs (conflict-free).
    // Taps are consumed eight at a time in a rolled loop: only eight of them are live in scalar registers.
    float2* y2 = fin + f_old;
    for (uint32_t o = tid; o < n2; o += 2 * kBeLanes) {     // two outputs (o, o + 256) share each block of taps
        HD_FIR_ARITH
        const bool has_b = o + kBeLanes < n2;
        const float4* p4 = reinterpret_cast<const float4*>(xin + (size_t)o * D2);
        const float4* q4 = has_b ? reinterpret_cast<const float4*>(xin + (size_t)(o + kBeLanes) * D2) : p4;
        float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
        int t = 0;
#pragma unroll 1
        for (; t + 8 <= T2; t += 8) {
            const float4 x0 = p4[(t >> 1)], x1 = p4[(t >> 1) + 1], x2 = p4[(t >> 1) + 2], x3 = p4[(t >> 1) + 3];
            const float4 z0 = q4[(t >> 1)], z1 = q4[(t >> 1) + 1], z2 = q4[(t >> 1) + 2], z3 = q4[(t >> 1) + 3];
            const float* tb = taps2 + t;
            const float k0 = tb[0], k1 = tb[1], k2 = tb[2], k3 = tb[3], k4 = tb[4], k5 = tb[5], k6 = tb[6], k7 = tb[7];
            ar = ar + x0.x * k0; ai = ai + x0.y * k0; br = br + z0.x * k0; bi = bi + z0.y * k0;
            ar = ar + x0.z * k1; ai = ai + x0.w * k1; br = br + z0.z * k1; bi = bi + z0.w * k1;
            ar = ar + x1.x * k2; ai = ai + x1.y * k2; br = br + z1.x * k2; bi = bi + z1.y * k2;
            ar = ar + x1.z * k3; ai = ai + x1.w * k3; br = br + z1.z * k3; bi = bi + z1.w * k3;
            ar = ar + x2.x * k4; ai = ai + x2.y * k4; br = br + z2.x * k4; bi = bi + z2.y * k4;
            ar = ar + x2.z * k5; ai = ai + x2.w * k5; br = br + z2.z * k5; bi = bi + z2.w * k5;
            ar = ar + x3.x * k6; ai = ai + x3.y * k6; br = br + z3.x * k6; bi = bi + z3.y * k6;
            ar = ar + x3.z * k7; ai = ai + x3.w * k7; br = br + z3.z * k7; bi = bi + z3.w * k7;
        }
#pragma unroll
        for (int u = 0; u < (T2 % 8); u += 2) {            // t is even here
            const float4 x = p4[(t + u) >> 1], z = q4[(t + u) >> 1];
            ar = ar + x.x * taps2[t + u]; ai = ai + x.y * taps2[t + u]; br = br + z.x * taps2[t + u]; bi = bi + z.y * taps2[t + u];
            if (u + 1 < (T2 % 8)) {
                ar = ar + x.z * taps2[t + u + 1]; ai = ai + x.w * taps2[t + u + 1]; br = br + z.z * taps2[t + u + 1]; bi = bi + z.w * taps2[t + u + 1];
            }
        }
        auto emit = [&](uint32_t oo, float2 y) {
            y2[oo] = y;
            cur_w[fir_hist_cap + pb + oo] = y;              // the decimated chunk stays readable (getters, unfused path next call)
            if (fft_in && oo < c.fft_take) fft_in[(size_t)s * kFftBins + c.fft_fill + oo] = y;   // reference Decoder.h:467-473
        };
        emit(o, make_float2(ar, ai));
        if (has_b) emit(o + kBeLanes, make_float2(br, bi));
    }
    __syncthreads();
    BSTAMP(2);

    // ---- stage-2 history carry for the next call (Decimator.h:140-143, with the in-place quirk Q4 of Decoder.h:443-444)
    if (n1) {
        if (tid < (uint32_t)(T2 - 1)) {
            const uint32_t idx = n1 - (T2 - 1) + tid;       // host guarantees n1 >= T2-1
            hist2_out[(size_t)s * (T2 - 1) + tid] = idx < n2 ? y2[idx] : xin[(T2 - 1) + idx];
        }
    } else if (tid < (uint32_t)(T2 - 1)) {
        hist2_out[(size_t)s * (T2 - 1) + tid] = hist2_in[(size_t)s * (T2 - 1) + tid];      // idle stream: passes through
    }

    // ---- slide [history | leftover pending] to the front of the other buffer: next[k] = buf[k + fir_m]; the deep part went
    // buffer to buffer with the first batch, the rest comes from the LDS image (which also holds this call's new samples)
    for (uint32_t k = sl_deep + tid; k < sl_cnt; k += kBeLanes) {
        const uint32_t j = k + sl_off;                      // >= lds_from
        float2 v = make_float2(0.f, 0.f);
        if (j - lds_from < f_old + n2) {
            v = fin[j - lds_from];
            if ((c.fir_zero_hist || Tp != T) && j < fir_hist_cap) v = cur[j];   // the image differs there; the slide moves the buffer as it is
        } else v = cur[j];
        nxt[k] = v;
    }
    if (sc_head_save(c)) {   // FirHistory head, lazily (dev_types.h): the stream ran in the previous call and does not in this one -- its head moves aside
        for (uint32_t k = tid; k < sc_head_n(c); k += kBeLanes) head_side[k] = head_prevbuf[k];
    }
    if (!m || !T) {
        if (tid == 0) carry_out[s] = carry_in[s];           // low-pass did not run: discriminator carry passes through
        return;
    }

    BSTAMP(3);
    // ---- low-pass + discriminator, kBePass outputs per pass, two adjacent outputs per lane (see fir_demod.hip)
    __syncthreads();                                        // xin is free now: reused for a pass's outputs
    float2* yout = xin;
    if (tap_touch == 12345.678f) demod[0] = tap_touch;      // keeps the tap touch alive; never true for a low-pass design
    const DemodCarry kin = carry_in[s];
    const SymState st = sym_ring ? sym[s] : SymState{};
    for (uint32_t i0 = 0; i0 < m; i0 += kBePass) {
        const uint32_t live = min((uint32_t)kBePass, m - i0);
        const bool active = 2u * tid < live;
        float a0r = 0.f, a0i = 0.f, a1r = 0.f, a1i = 0.f;
        if (active) {
            HD_FIR_ARITH
            const float4* p = reinterpret_cast<const float4*>(fin + i0) + tid;    // pair k: samples i0 + 2l + 2k, +1
            uint32_t t = 0;
            float4 P = p[0];
            if (T >= 8) {
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
                    HD_BE_PAIR(P, N0, k0, k1)
                    HD_BE_PAIR(N0, N1, k2, k3)
                    HD_BE_PAIR(N1, N2, k4, k5)
                    HD_BE_PAIR(N2, N3, k6, k7)
                    P = N3;
                    N0 = M0; N1 = M1; N2 = M2; N3 = M3;
                    k0 = q0; k1 = q1; k2 = q2; k3 = q3; k4 = q4; k5 = q5; k6 = q6; k7 = q7;
                }
            }
            for (; t + 2 <= T; t += 2) {
                const float4 N = p[(t >> 1) + 1];
                const float k0 = tp[t], k1 = tp[t + 1];
                HD_BE_PAIR(P, N, k0, k1)
                P = N;
            }
            if (t < T) {
                const float k0 = tp[t];
                a0r = a0r + P.x * k0; a0i = a0i + P.y * k0;
                a1r = a1r + P.z * k0; a1i = a1i + P.w * k0;
            }
            reinterpret_cast<float4*>(yout)[tid] = make_float4(a0r, a0i, a1r, a1i);
        }
        __syncthreads();
        if (i0 == 0) BSTAMP(4);
        if (active) {
            const uint32_t i = i0 + 2 * tid;
            float pr, pi;
            if (tid > 0) { const float2 q = yout[2 * tid - 1]; pr = q.x; pi = q.y; }
            else if (i0 > 0) { pr = s_ylast.x; pi = s_ylast.y; }
            else if (kin.primed) { pr = kin.re; pi = kin.im; }
            else { pr = a0r; pi = a0i; }                     // very first sample: arg(y0*conj(y0))
            const bool two = 2u * tid + 1 < live;
            const float d0 = discriminate(a0r, a0i, pr, pi);
            const float d1 = two ? discriminate(a1r, a1i, a0r, a0i) : 0.f;
            float* dm = demod + (size_t)s * demod_stride + i;
            if (two) *reinterpret_cast<float2*>(dm) = make_float2(d0, d1); else dm[0] = d0;
            if (sym_ring) {      // append straight into the symbol extractor's ring (SymbolExtractor::pushSamples); a vent
                float* ring = sym_ring + (size_t)s * ring_cap;   // (backlog > 30000) restarts the backlog at the same position
                const uint32_t pos = st.base + st.held + i;
                ring[pos & (ring_cap - 1)] = d0;
                if (two) ring[(pos + 1) & (ring_cap - 1)] = d1;
            }
            if (filtered) {
                filtered[(size_t)s * demod_stride + i] = make_float2(a0r, a0i);
                if (two) filtered[(size_t)s * demod_stride + i + 1] = make_float2(a1r, a1i);
            }
            const uint32_t last = m - 1;
            if (i == last || (two && i + 1 == last)) {
                DemodCarry k; k.primed = 1; k._pad = 0;
                if (i == last) { k.re = a0r; k.im = a0i; } else { k.re = a1r; k.im = a1i; }
                carry_out[s] = k;
            }
        }
        __syncthreads();                                    // yout is read above; the next pass rewrites it
        if (i0 == 0) BSTAMP(5);
        if (tid == kBeLanes - 1 && i0 + kBePass < m) s_ylast = make_float2(a1r, a1i);   // (a full pass: lane 255 holds its last output)
        // s_ylast is read by lane 0 only behind the next pass's barrier
    }
    BSTAMP(6);
}

size_t backend_lds_bytes(int ntaps2, uint32_t max_n1, uint32_t max_n2, uint32_t max_taps)
{
    const size_t xin_cap = ((size_t)(ntaps2 - 1) + max_n1 + 1) & ~(size_t)1;
    const size_t xin_need = xin_cap > (size_t)kBePass ? xin_cap : (size_t)kBePass;
    const size_t fin_cap = (size_t)(max_taps ? max_taps - 1 : 0) + (kFirBatch - 1) + max_n2 + 16;   // history | pending (< one batch) | new
    return (xin_need + fin_cap) * sizeof(float2);
}

bool launch_backend(hipStream_t st, int ratio2, int ntaps2, uint32_t n_streams, uint32_t max_n1, uint32_t max_n2, uint32_t max_taps,
                    const float2* dec1, size_t dec1_stride, const float2* hist2_in, float2* hist2_out, const float* taps2,
                    const float2* fbuf, float2* fbuf_w, float2* fbuf_next, size_t fbuf_stride, uint32_t fir_hist_cap, const float* lp_taps,
                    uint32_t taps_stride, float* demod, size_t demod_stride, float2* filtered, const DemodCarry* carry_in,
                    DemodCarry* carry_out, const StreamCall* call, float2* fft_in, float* sym_ring, uint32_t ring_cap, const SymState* sym,
                    float2* head_buf, uint32_t head_cap, const float2* fbuf_prev)
{
    const size_t lds = backend_lds_bytes(ntaps2, max_n1, max_n2, max_taps);
    if (lds > 64 * 1024) return false;
    size_t xin_cap = ((size_t)(ntaps2 - 1) + max_n1 + 1) & ~(size_t)1;
    if (xin_cap < (size_t)kBePass) xin_cap = kBePass;
#define HD_BE_CASE(D, T)                                                                                                              \
    if (ratio2 == D && ntaps2 == T) {                                                                                                 \
        hipLaunchKernelGGL((k_backend<D, T>), dim3(n_streams), dim3(kBeLanes), lds, st, dec1, dec1_stride, hist2_in, hist2_out, taps2, \
                           fbuf, fbuf_w, fbuf_next, fbuf_stride, fir_hist_cap, lp_taps, taps_stride, demod, demod_stride, filtered,   \
                           carry_in, carry_out, call, fft_in, sym_ring, ring_cap, sym, (uint32_t)xin_cap, head_buf, head_cap, fbuf_prev);                            \
        return true;                                                                                                                  \
    }
    HD_BE_CASE(2, 69) HD_BE_CASE(4, 139)
#undef HD_BE_CASE
    return false;
}

}  // namespace HD_ARITH_NS
}  // namespace hd
