// The stream tail: everything behind the first decimation stage of a two-stage plan, for ONE stream and one call, as a
// device function -- second FIR-decimate stage, low-pass FIR, FSK discriminator, symbol extractor (window sums, flag
// mask, edge search, run means, bits), plus the bookkeeping the other kernels expect (history carries, buffer slide,
// spectrum feed, FirHistory head, result slot).
//
// Why it exists.  k_backend + k_symbols (backend.hip, symbols.hip) give a stream a 256-lane workgroup each and image the
// whole call in LDS.  Next to the HBM-bound stage-1 kernel only one such workgroup fits per CU, so in batch mode the back
// half of call k took about as long as stage 1 of call k+1 and the step was the SUM of the two.  This body needs one
// wave, ~17 KB of LDS and no barrier: the call is walked in pieces of P stage-2 outputs whose inputs are prefetched into
// registers one piece ahead, every stage hands its output to the next through small sliding LDS windows, and the symbol
// extractor consumes the discriminator output as it appears.  It runs either as its own kernel (k_tail, tail.hip) or as
// the first blockIdx range of the stage-1 launch of the NEXT call (k_step, decimate.hip), where the hardware dispatcher
// deals tails and stage-1 tile runs to the CUs' eight slots as they free up.
//
// Arithmetic and semantics are those of the kernels it replaces (and therefore of the reference): T-term sums in
// ascending tap order with separately rounded multiply and add (Decimator.h:128-138, FirFilter.h:155-161; compiled with
// -ffp-contract=off), arg(y[i] * conj(y[i-1])) through exact_math.h (FSK2_Demod.h:37-40), window sums and run sums
// added left to right by one lane (SymbolExtractor.h:162-224).  Global state (buffers, rings, SymState) is laid out as
// the other kernels leave and expect it, so the engine can switch between the paths from call to call.
#pragma once
#include <hip/hip_runtime.h>

#include "exact_math.h"
#include "launch.h"
#include "sym_common.h"

namespace hd {

constexpr uint32_t kTailHdrBytes = 64;          // small shared scalars at the base of the workgroup's LDS scratch
constexpr uint32_t kTailStrip = 512;            // samples per run-sum step (per wave)

#define HD_TB_PAIR(P_, N_, k0, k1)                        \
    a0r = a0r + (P_).x * (k0); a0i = a0i + (P_).y * (k0); \
    a1r = a1r + (P_).z * (k0); a1i = a1i + (P_).w * (k0); \
    a0r = a0r + (P_).z * (k1); a0i = a0i + (P_).w * (k1); \
    a1r = a1r + (N_).x * (k1); a1i = a1i + (N_).y * (k1);

template <int NT>
__device__ __forceinline__ void tb_sync()
{
    __syncthreads();      // a 64-lane workgroup is one wave: the compiler drops the s_barrier and keeps the memory ordering
}

// NT lanes, OP stage-2 outputs per lane and piece, stage-2 design (D2, T2).
template <int NT, int OP, int D2, int T2>
__device__ __forceinline__ void tail_body(const TailArgs& a, const uint32_t s, unsigned char* __restrict__ lds)
{
    constexpr int P = NT * OP;                      // stage-2 outputs per piece
    constexpr int XCH = P * D2;                     // stage-1 samples a piece consumes
    constexpr int XN = (T2 - 1) + XCH;              // X image: [stage-2 history | piece]
    constexpr int XB = XCH / 2 / NT;                // 16-byte loads per lane and piece
    constexpr int HB = (T2 - 1 + NT - 1) / NT;
    constexpr int NW = NT / 64;
    constexpr uint32_t B = P + (kFirBatch - 1);     // most discriminator outputs one piece can release
    static_assert(((T2 - 1) & 1) == 0 && XCH % (2 * NT) == 0, "16-byte pairs must stay aligned");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    uint32_t* sh = reinterpret_cast<uint32_t*>(lds);                    // [0] nfl [1] overflow [2] frontier [3] carry sum [4] flagged
    float2* X = reinterpret_cast<float2*>(lds + kTailHdrBytes);
    float2* Y = X + ((XN + 1) & ~1);                                    // [1]: predecessor of this pass's first output; outputs from [2]
    float2* F = reinterpret_cast<float2*>(lds + a.f_off);               // low-pass input window, F[0] = input index fbase
    float* V = reinterpret_cast<float*>(lds + a.v_off);                 // discriminator output from position c0 on
    float* WS = reinterpret_cast<float*>(lds + a.ws_off);               // window sums of [c0 - R, c0), then the new ones
    unsigned long long* words = reinterpret_cast<unsigned long long*>(lds + a.words_off);

    const StreamCall c = a.call[s];
    const uint32_t n1 = c.n1, n2 = c.n2, m = c.fir_m, T = c.fir_taps, pb = c.pend_before;
    const uint32_t H = T ? T - 1 : 0;
    const uint32_t Tp = c.fir_taps_prev ? c.fir_taps_prev : T;
    const bool run = m && T;
    const bool keepf = !c.clear_pending;
    const uint32_t nS = a.n_streams;
    const float2* in_s = a.dec1 + (size_t)s * a.dec1_stride;
    float2* cur = a.fbuf + (size_t)s * a.fbuf_stride;
    float2* nxt = a.fbuf_next + (size_t)s * a.fbuf_stride;
    const float2* head_in = a.head_buf + ((size_t)a.head_par * nS + s) * a.head_cap;
    float2* head_out = a.head_buf + ((size_t)(a.head_par ^ 1u) * nS + s) * a.head_cap;
    const uint32_t* head_n_in = a.head_cnt + (size_t)a.head_par * nS + s;
    uint32_t* head_n_out = a.head_cnt + (size_t)(a.head_par ^ 1u) * nS + s;
    const float* tp = a.lp_taps + (size_t)s * a.taps_stride;
    const uint32_t fhc = a.fir_hist_cap;
    const uint32_t hn = run ? min(m, a.head_cap) : 0u;

    // ---- symbol extractor: state after this call's push (SymbolExtractor.h:116-124), what this call has to do
    const uint32_t ring_cap = a.ring_cap, rmask = ring_cap - 1;
    float* vring = a.ring + (size_t)s * ring_cap;
    float* gw = a.wsum + (size_t)s * ring_cap;
    unsigned long long* gmask = a.flipmask + (size_t)s * (ring_cap / 64);
    uint32_t* slot = a.slots + (size_t)s * a.slot_words;
    BitsHeader* hdr = reinterpret_cast<BitsHeader*>(slot);
    const SymState old = a.sym[s];
    const SymbolParams q = a.sp[s];
    const DemodCarry kin = a.carry_in[s];
    const uint32_t R = q.R;
    SymState st = old;
    bool search = false, do_sums = false;
    if (m) {
        st = state_after_push(old, q, m);
        search = !(st.held < q.min_held || st.held < q.spb);
        do_sums = search || !(q.min_held == 0xFFFFFFFFu || st.held < R);
    }
    const uint32_t end_old = old.base + old.held;                       // ring position of this call's first discriminator output
    uint32_t c0 = st.cached;                                            // next position whose window sum is due; V[0] is sample c0
    uint32_t vcnt = 0;                                                  // samples in V
    unsigned long long carry_word = 0ull;                               // flags of the positions [c0 & ~63, c0), as earlier sweeps left them
    const uint32_t old_cnt = do_sums ? end_old - c0 : 0u;              // backlog samples whose windows are not final yet (R-1 in steady state)

    // ---- every global read the first piece needs, issued back to back (one round trip): a touch of each 64-byte line of the
    // stream's low-pass taps (scalar cache), stage-2 history, the first piece of the stage-1 chunk, low-pass history + pending,
    // the backlog samples and window sums the first window sums build on, the boundary word of the flag mask.
    float tap_touch = 0.f;
    {
        const uint32_t tl = T ? T - 1 : 0;
        float tt[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) tt[u] = tp[min((uint32_t)u * 16u, tl)];
#pragma unroll
        for (int u = 0; u < 16; ++u) tap_touch += tt[u];
    }
    const bool refold = Tp != T && !c.fir_zero_hist && run;           // first run after a tap-count change (FirHistory, dev_types.h)
    const uint32_t head_n = refold ? head_n_in[0] : 0u;
    auto fold = [&](uint32_t k) -> float2 {                             // element k of [history (H) | pending]
        if (k < H && c.fir_zero_hist) return make_float2(0.f, 0.f);
        if (!refold || k >= H) return cur[fhc - H + k];
        if (k < Tp - 1) return cur[fhc - (Tp - 1) + k];
        const uint32_t h = k - (Tp - 1);
        return h < head_n ? head_in[h] : make_float2(0.f, 0.f);
    };
    const uint32_t f_old = H + pb;
    const float4* in4 = reinterpret_cast<const float4*>(in_s);
    const uint32_t n1p = n1 >> 1;
    float4 tx[XB];                                                      // register prefetch of the next piece
    auto prefetch = [&](uint32_t pc) {
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const uint32_t k = pc * (XCH / 2) + tid + u * NT;
            tx[u] = k < n1p ? in4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    constexpr int FBN = 512 / NT, VBN = 512 / NT;
    const uint32_t take0 = min(old_cnt, (uint32_t)(VBN * NT));
    {
        float2 th[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const uint32_t k = tid + u * NT;
            th[u] = (k < (uint32_t)(T2 - 1) && !c.zero_hist2) ? a.hist2_in[(size_t)s * (T2 - 1) + k] : make_float2(0.f, 0.f);
        }
        prefetch(0);
        float2 tf[FBN];
#pragma unroll
        for (int u = 0; u < FBN; ++u) { const uint32_t k = tid + u * NT; tf[u] = k < f_old ? cur[fhc - H + k] : make_float2(0.f, 0.f); }
        float tv[VBN], tw[VBN];
#pragma unroll
        for (int u = 0; u < VBN; ++u) { const uint32_t k = tid + u * NT; tv[u] = k < take0 ? vring[(c0 + k) & rmask] : 0.f; }
#pragma unroll
        for (int u = 0; u < VBN; ++u) { const uint32_t k = tid + u * NT; tw[u] = (do_sums && k < R) ? gw[(c0 - R + k) & rmask] : 0.f; }
        if (do_sums && (c0 & 63u)) carry_word = gmask[(c0 & rmask) >> 6] & ((1ull << (c0 & 63u)) - 1ull);
        if (c.fir_zero_hist || refold) {
#pragma unroll
            for (int u = 0; u < FBN; ++u) { const uint32_t k = tid + u * NT; if (k < H) tf[u] = fold(k); }
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) { const uint32_t k = tid + u * NT; if (k < (uint32_t)(T2 - 1)) X[k] = th[u]; }
#pragma unroll
        for (int u = 0; u < FBN; ++u) { const uint32_t k = tid + u * NT; if (k < f_old) F[k] = tf[u]; }
        for (uint32_t k = tid + FBN * NT; k < f_old; k += NT) F[k] = fold(k);
#pragma unroll
        for (int u = 0; u < VBN; ++u) { const uint32_t k = tid + u * NT; if (k < take0) V[k] = tv[u]; }
#pragma unroll
        for (int u = 0; u < VBN; ++u) { const uint32_t k = tid + u * NT; if (do_sums && k < R) WS[k] = tw[u]; }
        if (do_sums) for (uint32_t k = tid + VBN * NT; k < R; k += NT) WS[k] = gw[(c0 - R + k) & rmask];
        if (tid == 0) Y[1] = make_float2(kin.re, kin.im);
    }
    vcnt = take0;
    tb_sync<NT>();
    for (uint32_t k = tid; k < pb && k < hn; k += NT) head_out[k] = F[H + k];     // pending samples open this run's input (FirHistory)

    // ---- window sums + flags for every position whose right window is complete: positions [c0, c0 + vcnt - R + 1).
    // W(p) = v[p] + ... + v[p+R-1] is both the reference's right window of p and its left window of p + R (symbols.hip).
    auto sym_feed = [&]() {
        int32_t npos = (int32_t)vcnt - (int32_t)R + 1;
        uint32_t off = 0;
        const float tiny = (float)R * 2.8e-45f;
        auto avg_sign = [&](float wsum) { return __builtin_fabsf(wsum) > tiny ? sgnf(wsum) : sgnf(wsum / (float)R); };
        while (npos > 0) {
            const uint32_t cnt = min((uint32_t)npos, (uint32_t)(kAvgPos * NT));
            const uint32_t wb0 = c0 & ~63u, wsh = c0 & 63u;
            const uint32_t nwords = (wsh + cnt + 63u) >> 6;
            if (tid < (uint32_t)(kAvgPos * NT / 64 + 2)) words[tid] = 0ull;
            tb_sync<NT>();
            const bool any = kAvgPos * tid < cnt;
            float wp[kAvgPos];
            if (any) {
                window_sums(V + off + kAvgPos * tid, R, wp);
#pragma unroll
                for (int j = 0; j < kAvgPos; ++j) {
                    WS[off + R + kAvgPos * tid + j] = wp[j];
                    if (kAvgPos * tid + j < cnt) gw[(c0 + kAvgPos * tid + j) & rmask] = wp[j];
                }
            }
            tb_sync<NT>();
            if (any) {
                unsigned int bits = 0;
#pragma unroll
                for (int j = 0; j < kAvgPos; ++j)
                    if (kAvgPos * tid + j < cnt && avg_sign(WS[off + kAvgPos * tid + j]) != avg_sign(wp[j])) bits |= 1u << j;
                if (bits) {
                    const uint32_t bp = wsh + tid * kAvgPos, shb = bp & 63u;
                    atomicOr(&words[bp >> 6], (unsigned long long)bits << shb);
                    if (shb > 60u) atomicOr(&words[(bp >> 6) + 1], (unsigned long long)bits >> (64u - shb));
                }
            }
            tb_sync<NT>();
            if (tid < nwords) {
                unsigned long long w = words[tid];
                if (tid == 0) w |= carry_word;
                gmask[((wb0 + tid * 64u) & rmask) >> 6] = w;
            }
            {   // the word the next sweep starts in, as far as it is filled
                const uint32_t e = wsh + cnt, ei = e >> 6, eb = e & 63u;
                unsigned long long w = words[ei];
                if (ei == 0) w |= carry_word;
                carry_word = eb ? (w & ((1ull << eb) - 1ull)) : 0ull;
            }
            tb_sync<NT>();
            c0 += cnt; off += cnt; npos -= (int32_t)cnt;
        }
        if (off) {                                                      // slide both windows down by the positions done
            const uint32_t nv = vcnt - off;
            for (uint32_t k0 = 0; k0 < nv; k0 += 4 * NT) {
                float t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; t4[u] = k < nv ? V[k + off] : 0.f; }
                tb_sync<NT>();
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < nv) V[k] = t4[u]; }
            }
            for (uint32_t k0 = 0; k0 < R; k0 += 4 * NT) {
                float t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; t4[u] = k < R ? WS[k + off] : 0.f; }
                tb_sync<NT>();
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < R) WS[k] = t4[u]; }
            }
            vcnt = nv;
            tb_sync<NT>();
        }
    };
    if (do_sums) {
        sym_feed();
        for (uint32_t left = old_cnt - take0; left;) {                  // only after a parameter change: the whole backlog is due
            const uint32_t take = min(left, B);
            for (uint32_t k = tid; k < take; k += NT) V[vcnt + k] = vring[(c0 + vcnt + k) & rmask];
            vcnt += take; left -= take;
            tb_sync<NT>();
            sym_feed();
        }
    }

    // ---- the call, piece by piece
    uint32_t fbase = 0, fcount = f_old, i_done = 0;
    const uint32_t npieces = (n2 + P - 1) / P;
    if (tap_touch == 12345.678f) a.demod[0] = tap_touch;                // keeps the tap touch alive; never true for a low-pass design
    for (uint32_t pc = 0; pc < npieces; ++pc) {
        const uint32_t o0 = pc * P, po = min((uint32_t)P, n2 - o0);
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const uint32_t k = tid + u * NT;
            *reinterpret_cast<float4*>(X + (T2 - 1) + 2 * k) = tx[u];
        }
        tb_sync<NT>();
        if (pc + 1 < npieces) prefetch(pc + 1);

        // stage 2: y2[o] = sum_t x[o*D2 + t] * h2[t]; two outputs (o, o + NT) share each block of eight taps
#pragma unroll
        for (int jo = 0; jo < OP; jo += 2) {
            const uint32_t oa = tid + jo * NT, ob = oa + NT;
            if (oa < po) {
                const bool has_b = (jo + 1 < OP) && ob < po;
                const float4* p4 = reinterpret_cast<const float4*>(X + (size_t)oa * D2);
                const float4* q4 = has_b ? reinterpret_cast<const float4*>(X + (size_t)ob * D2) : p4;
                float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
                int t = 0;
#pragma unroll 1
                for (; t + 8 <= T2; t += 8) {
                    const float4 x0 = p4[(t >> 1)], x1 = p4[(t >> 1) + 1], x2 = p4[(t >> 1) + 2], x3 = p4[(t >> 1) + 3];
                    const float4 z0 = q4[(t >> 1)], z1 = q4[(t >> 1) + 1], z2 = q4[(t >> 1) + 2], z3 = q4[(t >> 1) + 3];
                    const float* tb = a.taps2 + t;
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
                for (int u = 0; u < (T2 % 8); u += 2) {
                    const float4 x = p4[(t + u) >> 1], z = q4[(t + u) >> 1];
                    ar = ar + x.x * a.taps2[t + u]; ai = ai + x.y * a.taps2[t + u]; br = br + z.x * a.taps2[t + u]; bi = bi + z.y * a.taps2[t + u];
                    if (u + 1 < (T2 % 8)) {
                        ar = ar + x.z * a.taps2[t + u + 1]; ai = ai + x.w * a.taps2[t + u + 1]; br = br + z.z * a.taps2[t + u + 1]; bi = bi + z.w * a.taps2[t + u + 1];
                    }
                }
                auto emit = [&](uint32_t ol, float2 y) {
                    const uint32_t oo = o0 + ol;
                    if (keepf || npieces == 1) F[f_old + oo - fbase] = y;
                    cur[fhc + pb + oo] = y;                             // the decimated chunk stays readable (getters, the unfused path next call)
                    if (a.fft_in && oo < c.fft_take) a.fft_in[(size_t)s * kFftBins + c.fft_fill + oo] = y;   // Decoder.h:467-473
                    if (pb + oo < hn) head_out[pb + oo] = y;
                };
                emit(oa, make_float2(ar, ai));
                if (has_b) emit(ob, make_float2(br, bi));
            }
        }
        // the last T2-1 samples of this image are the next piece's history
        float2 xt[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) { const uint32_t k = tid + u * NT; xt[u] = k < (uint32_t)(T2 - 1) ? X[XCH + k] : make_float2(0.f, 0.f); }
        if (pc + 1 == npieces && n1) {
            // stage-2 history carry for the next call (Decimator.h:140-143, with the in-place quirk Q4 of Decoder.h:443-444: history
            // positions inside the first n2 samples hold OUTPUTS -- only for inputs so short that this is the only piece)
            tb_sync<NT>();
            for (uint32_t k = tid; k < (uint32_t)(T2 - 1); k += NT) {
                const uint32_t idx = n1 - (T2 - 1) + k;                 // host guarantees n1 >= T2-1
                a.hist2_out[(size_t)s * (T2 - 1) + k] = idx < n2 ? F[f_old + idx - fbase] : in_s[idx];
            }
        }
        tb_sync<NT>();
#pragma unroll
        for (int u = 0; u < HB; ++u) { const uint32_t k = tid + u * NT; if (k < (uint32_t)(T2 - 1)) X[k] = xt[u]; }
        fcount += po;

        // low-pass + discriminator over the outputs that became computable, 2*NT per pass, two adjacent outputs per lane
        if (run) {
            const uint32_t i_hi = (pc + 1 == npieces) ? m : min(m, (pb + o0 + po) & ~1u);
            for (uint32_t i0 = i_done; i0 < i_hi; i0 += 2 * NT) {
                const uint32_t live = min((uint32_t)(2 * NT), i_hi - i0);
                const bool active = 2u * tid < live;
                float a0r = 0.f, a0i = 0.f, a1r = 0.f, a1i = 0.f;
                if (active) {
                    const float4* p = reinterpret_cast<const float4*>(F + (i0 - fbase)) + tid;
                    uint32_t t = 0;
                    float4 Pq = p[0];
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
                            HD_TB_PAIR(Pq, N0, k0, k1)
                            HD_TB_PAIR(N0, N1, k2, k3)
                            HD_TB_PAIR(N1, N2, k4, k5)
                            HD_TB_PAIR(N2, N3, k6, k7)
                            Pq = N3;
                            N0 = M0; N1 = M1; N2 = M2; N3 = M3;
                            k0 = q0; k1 = q1; k2 = q2; k3 = q3; k4 = q4; k5 = q5; k6 = q6; k7 = q7;
                        }
                    }
                    for (; t + 2 <= T; t += 2) {
                        const float4 N = p[(t >> 1) + 1];
                        const float k0 = tp[t], k1 = tp[t + 1];
                        HD_TB_PAIR(Pq, N, k0, k1)
                        Pq = N;
                    }
                    if (t < T) {
                        const float k0 = tp[t];
                        a0r = a0r + Pq.x * k0; a0i = a0i + Pq.y * k0;
                        a1r = a1r + Pq.z * k0; a1i = a1i + Pq.w * k0;
                    }
                    reinterpret_cast<float4*>(Y)[1 + tid] = make_float4(a0r, a0i, a1r, a1i);
                }
                tb_sync<NT>();
                if (active) {
                    const uint32_t i = i0 + 2 * tid;
                    float pr, pi;
                    const float2 qv = Y[2 * tid + 1];
                    if (i == 0 && !kin.primed) { pr = a0r; pi = a0i; }   // very first sample: arg(y0*conj(y0)) (FSK2_Demod.h:35)
                    else { pr = qv.x; pi = qv.y; }
                    const bool two = 2u * tid + 1 < live;
                    const float d0 = discriminate(a0r, a0i, pr, pi);
                    const float d1 = two ? discriminate(a1r, a1i, a0r, a0i) : 0.f;
                    float* dm = a.demod + (size_t)s * a.demod_stride + i;
                    if (two) *reinterpret_cast<float2*>(dm) = make_float2(d0, d1); else dm[0] = d0;
                    const uint32_t pos = end_old + i;                   // SymbolExtractor::pushSamples: append to the backlog ring
                    vring[pos & rmask] = d0;
                    if (two) vring[(pos + 1) & rmask] = d1;
                    if (do_sums) { V[pos - c0] = d0; if (two) V[pos + 1 - c0] = d1; }
                    if (a.filtered) {
                        a.filtered[(size_t)s * a.demod_stride + i] = make_float2(a0r, a0i);
                        if (two) a.filtered[(size_t)s * a.demod_stride + i + 1] = make_float2(a1r, a1i);
                    }
                    const uint32_t last = m - 1;
                    if (i == last || (two && i + 1 == last)) {
                        DemodCarry k; k.primed = 1; k._pad = 0;
                        if (i == last) { k.re = a0r; k.im = a0i; } else { k.re = a1r; k.im = a1i; }
                        a.carry_out[s] = k;
                    }
                }
                tb_sync<NT>();                                          // Y is read above; its carry slot and the next pass rewrite it
                if (active && (2u * tid + 2 >= live)) Y[1] = (2u * tid + 1 < live) ? make_float2(a1r, a1i) : make_float2(a0r, a0i);
            }
            i_done = i_hi;
            {   // slide the low-pass window: what is still needed starts at input index i_done
                const uint32_t shf = i_done - fbase, keep = fcount - shf;
                if (shf) {
                    tb_sync<NT>();
                    for (uint32_t k0 = 0; k0 < keep; k0 += 4 * NT) {
                        float2 t4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; t4[u] = k < keep ? F[k + shf] : make_float2(0.f, 0.f); }
                        tb_sync<NT>();
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < keep) F[k] = t4[u]; }
                    }
                    fbase = i_done; fcount = keep;
                }
            }
            if (do_sums) {
                vcnt = end_old + i_done - c0;
                tb_sync<NT>();
                sym_feed();
            }
        }
    }
    tb_sync<NT>();

    // ---- what the next call finds: stage-2 history of an idle stream, [history | leftover pending] at the front of the other
    // low-pass buffer, the head of this run's input (FirHistory), the discriminator carry
    if (!n1) for (uint32_t k = tid; k < (uint32_t)(T2 - 1); k += NT) a.hist2_out[(size_t)s * (T2 - 1) + k] = a.hist2_in[(size_t)s * (T2 - 1) + k];
    if (run) {
        // F[0 ..) is input index m on: the last H inputs of this run followed by the leftover pending samples
        for (uint32_t k = tid; k < H + c.pend_after; k += NT) nxt[fhc - H + k] = F[k];
    } else {
        // low-pass did not run: the buffer moves as it is (history of whatever length the last run had, pending, new)
        __threadfence_block();
        tb_sync<NT>();
        const uint32_t cnt = fhc + c.pend_after;
        for (uint32_t k = tid; k < cnt; k += NT) nxt[k] = cur[k];
    }
    {
        const uint32_t hk = run ? hn : head_n_in[0];
        if (!run) for (uint32_t k = tid; k < hk; k += NT) head_out[k] = head_in[k];
        if (tid == 0) head_n_out[0] = hk;
    }
    if (!run && tid == 0) a.carry_out[s] = kin;

    // ---- symbol extractor, second half: edge search, run means, bits (SymbolExtractor.h:129-158)
    uint32_t* outw = slot + sizeof(BitsHeader) / 4;
    const uint32_t cap_bits = (a.slot_words - sizeof(BitsHeader) / 4) * 32;
    if (!m) {
        if (tid == 0) { hdr->nbits = 0; hdr->held_after = old.held; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = old.base + old.held - old.cached; }
        return;
    }
    const uint32_t h = st.held;
    const uint32_t end = st.base + h;
    const uint32_t pend = end - R + 1;
    if (!do_sums) {
        if (tid == 0) { a.sym[s] = st; hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = st.base + h - st.cached; }
        return;
    }
    if (!search) {
        if (tid == 0) {
            if ((int32_t)(pend - st.cached) > 0) st.cached = pend;
            a.sym[s] = st;
            hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = end - st.cached;
        }
        return;
    }
    const uint32_t limit = h - q.spb;                                   // backlog indices searched: [R, limit)
    unsigned long long* lmask = reinterpret_cast<unsigned long long*>(lds + a.lmask_off);
    uint32_t* flips = reinterpret_cast<uint32_t*>(lds + a.flips_off);
    uint32_t* runinfo = flips + a.fl_cap;
    float* strips = reinterpret_cast<float*>(lds + a.strips_off);
    float* wc = reinterpret_cast<float*>(lds + a.wc_off);
    const uint32_t fl_cap = a.fl_cap;
    // this call's window sums, flags and samples were written to the global rings above; what follows reads them back
    __threadfence_block();
    tb_sync<NT>();
    const uint32_t wc_n = limit <= a.wc_cap ? limit : 0u;               // window sums of the searchable backlog, when they fit
    {
        const uint32_t wr0 = (st.base + R) & ~63u;
        const uint32_t nw = ((st.base + limit) - wr0 + 63u) >> 6;
        constexpr int MB = 512 / NT, CB = 1024 / NT;
        for (uint32_t i0 = 0; i0 < nw; i0 += MB * NT) {
            unsigned long long tm[MB];
#pragma unroll
            for (int u = 0; u < MB; ++u) { const uint32_t i = i0 + tid + u * NT; tm[u] = i < nw ? gmask[((wr0 + 64u * i) & rmask) >> 6] : 0ull; }
            if (i0 == 0) {
                float t[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) { const uint32_t k = tid + u * NT; t[u] = k < wc_n ? gw[(st.base + k) & rmask] : 0.f; }
#pragma unroll
                for (int u = 0; u < CB; ++u) { const uint32_t k = tid + u * NT; if (k < wc_n) wc[k] = t[u]; }
            }
#pragma unroll
            for (int u = 0; u < MB; ++u) { const uint32_t i = i0 + tid + u * NT; if (i < nw) lmask[((wr0 + 64u * i) & rmask) >> 6] = tm[u]; }
        }
        for (uint32_t k = tid + CB * NT; k < wc_n; k += NT) wc[k] = gw[(st.base + k) & rmask];
    }
    tb_sync<NT>();
    auto wsum = [&](uint32_t i) -> float { return i < wc_n ? wc[i] : gw[(st.base + i) & rmask]; };
    if (wave == 0) {
        uint32_t pos = R, nfl = 0, overflow = 0;
        uint32_t frontier = 0xFFFFFFFFu;                                // backlog index no future flip point can precede
        while (pos < limit) {
            const uint32_t lo = find_flag_lds(lmask, st.base, rmask, pos, limit, true);
            if (lo == 0xFFFFFFFFu) break;
            const uint32_t hi = find_flag_lds(lmask, st.base, rmask, lo + 1, limit, false);
            if (hi == 0xFFFFFFFFu) { frontier = lo; break; }            // an edge zone is open: the next flip lies at or after lo
            unsigned long long key = 0ull;                              // first maximum of the weight over [lo, hi): (weight bits, ~index)
            constexpr int ZB = 4;
            for (uint32_t i0 = lo + lane; i0 < hi; i0 += 64 * ZB) {
                float wr_[ZB], wl_[ZB];
#pragma unroll
                for (int u = 0; u < ZB; ++u) {
                    const uint32_t i = i0 + 64 * u;
                    wr_[u] = i < hi ? wsum(i) : 0.0f;
                    wl_[u] = i < hi ? wsum(i - R) : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < ZB; ++u) {
                    const uint32_t i = i0 + 64 * u;
                    if (i < hi) {
                        const float d = wr_[u] / (float)R - wl_[u] / (float)R;   // avg_r - avg_l
                        const float w = q.float_abs ? __builtin_fabsf(d) : (float)abs((int)d);
                        const unsigned long long k = ((unsigned long long)__builtin_bit_cast(uint32_t, w) << 32) | (uint32_t)~i;
                        key = k > key ? k : key;
                    }
                }
            }
            const uint32_t f = ~(uint32_t)wave_max_u64(key);
            if (nfl < fl_cap) { if (lane == 0) flips[nfl] = f; ++nfl; } else { overflow = 1; break; }
            pos = f + R;
        }
        if (frontier == 0xFFFFFFFFu) frontier = max(pos, limit);        // nothing flagged in [pos, limit)
        if (lane == 0) { sh[0] = nfl; sh[1] = overflow; sh[2] = frontier; }
    }
    tb_sync<NT>();
    const uint32_t nfl = sh[0];

    // per-run sums in element order (std::accumulate); the run in progress is carried across calls (SymState::run_sum covers
    // [base, run_pos)) up to the search frontier -- the same left-to-right chain of adds the reference performs in one go
    float* sb = strips + wave * kTailStrip;
    constexpr int NX = kTailStrip / 64;
    auto chain = [&](float acc, uint32_t a0, uint32_t b) -> float {     // acc + v[a0] + ... + v[b-1] (backlog indices)
        if (a0 >= b) return acc;
        float nx[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) { const uint32_t k = a0 + 64 * j + lane; nx[j] = k < b ? vring[(st.base + k) & rmask] : 0.0f; }
        for (uint32_t k0 = a0; k0 < b; k0 += kTailStrip) {
#pragma unroll
            for (int j = 0; j < NX; ++j) sb[64 * j + lane] = nx[j];
            if (k0 + kTailStrip < b) {
#pragma unroll
                for (int j = 0; j < NX; ++j) { const uint32_t k = k0 + kTailStrip + 64 * j + lane; nx[j] = k < b ? vring[(st.base + k) & rmask] : 0.0f; }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): the strip has landed (wave-private)
            __builtin_amdgcn_wave_barrier();
            const uint32_t cnt = min(kTailStrip, b - k0);
            const float4* s4 = reinterpret_cast<const float4*>(sb);
            uint32_t i = 0;
            if (cnt >= 32) {
                float4 cc[8], nn[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) cc[u] = s4[u];
                for (; i + 64 <= cnt; i += 32) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) nn[u] = s4[((i + 32) >> 2) + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { acc = acc + cc[u].x; acc = acc + cc[u].y; acc = acc + cc[u].z; acc = acc + cc[u].w; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) cc[u] = nn[u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc = acc + cc[u].x; acc = acc + cc[u].y; acc = acc + cc[u].z; acc = acc + cc[u].w; }
                i += 32;
            }
            for (; i < cnt; ++i) acc = acc + sb[i];
            __builtin_amdgcn_wave_barrier();
        }
        return acc;
    };
    const uint32_t carried_to = st.run_pos - st.base;
    for (uint32_t r = wave; r < nfl; r += NW) {
        const uint32_t ra = r ? flips[r - 1] : 0u, rb = flips[r];
        const float acc = r ? chain(0.0f, ra, rb) : chain(st.run_sum, min(carried_to, rb), rb);
        if (lane == 0) {
            const float mean = acc / (float)(rb - ra);
            const uint32_t cnt = (uint32_t)roundf((float)(rb - ra) / (float)q.spb);
            runinfo[r] = (cnt << 1) | (mean > 0.0f ? 1u : 0u);
        }
    }
    const uint32_t frontier = sh[2];
    if (wave == (nfl % NW)) {
        const uint32_t ra = nfl ? flips[nfl - 1] : carried_to;
        const float acc = chain(nfl ? 0.0f : st.run_sum, ra, max(ra, frontier));
        if (lane == 0) sh[3] = __builtin_bit_cast(uint32_t, acc);
    }
    tb_sync<NT>();
    if (tid == 0) {
        uint32_t nbits = 0, curw = 0, overflow = sh[1];
        for (uint32_t r = 0; r < nfl; ++r) {
            const uint32_t bit = runinfo[r] & 1u;
            for (uint32_t k = runinfo[r] >> 1; k; --k) {
                if (nbits >= cap_bits) { overflow = 1; break; }
                curw |= bit << (nbits & 31);
                if ((nbits & 31) == 31) { outw[nbits >> 5] = curw; curw = 0; }
                ++nbits;
            }
        }
        if (nbits & 31) outw[nbits >> 5] = curw;
        if (a.flips_dbg)
            for (uint32_t r = 0; r < nfl && r < a.flips_cap; ++r) a.flips_dbg[(size_t)s * a.flips_cap + r] = flips[r];
        const uint32_t last = nfl ? flips[nfl - 1] : 0u;               // erase the consumed prefix (SymbolExtractor.h:156-157) = advance the base
        st.run_pos = st.base + max(nfl ? last : carried_to, frontier);
        st.run_sum = __builtin_bit_cast(float, sh[3]);
        st.cached = pend;
        st.base += last;
        st.held = h - last;
        a.sym[s] = st;
        hdr->nbits = nbits; hdr->held_after = st.held; hdr->nflips = nfl; hdr->overflow = overflow; hdr->uncached = end - st.cached;
    }
}

}  // namespace hd
