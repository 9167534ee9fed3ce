// Symbol extractor on the GPU: demodulated samples -> bits, per stream.
//
// Reference behaviour (code/Decoder/SymbolExtractor.h:108-255): demodulated samples are appended to a per-stream
// backlog (dropped wholesale if more than 30000 are already held); once at least 3 symbols are held, bit edges
// ("flip points") are searched left to right: slide i until the mean of the R samples left of i and the mean of
// the R samples right of i differ in sign, keep sliding until they agree again, and take the position in that
// span where |mean_r - mean_l| is largest (first maximum); continue R samples after it.  The search stops at
// size - samples_per_bit.  Each run between flips yields round(len/spb) copies of (mean > 0); consumed samples
// are erased.
//
// The reference re-evaluates O(backlog * R) window sums on every call.  Here:
//   * the backlog lives in a per-stream power-of-two RING (the discriminator kernel appends straight into it), so
//     "erase" is just advancing the base position -- no copies;
//   * the two window means of a candidate position are pure functions of the samples around it, so they are
//     computed ONCE, when the position's right window is complete (k_sym_avg: one lane per new position, each lane
//     adding its R samples left-to-right from LDS exactly like std::accumulate, so the floats are bit-identical),
//     and cached as one "signs differ" bit (ballot -> 64-bit mask words) plus one weight per position;
//   * the inherently sequential edge search runs on those mask words with one wave per stream (k_sym_scan: 4096
//     positions per step through ballots and find-first-set); per-run sums are accumulated in element order with
//     the wave loading 256 samples at a time and v_readlane feeding a wave-uniform accumulator; lane 0 packs bits.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace hd {

constexpr int kAvgLanes = 256;
constexpr uint32_t kMaxFlipsPerCall = 1024;   // LDS flip list of the scan kernel (overflow is flagged)

__device__ __forceinline__ int sgnf(float v) { return (0.0f < v) - (v < 0.0f); }

// State after this call's samples were appended (SymbolExtractor.h:116-124: the vent happens before the append).
__device__ __forceinline__ SymState state_after_push(SymState st, const SymbolParams& q, uint32_t m)
{
    if (st.held > kVentLimit) { st.base += st.held; st.held = 0; st.cached = st.base; }
    if (q.reset) st.cached = st.base;
    st.held += m;
    return st;
}

// Window sums for the NEW candidate positions.  One lane owns 4 consecutive positions: it walks its R+3 left-window
// samples (then the R+3 right-window samples) once with 16-byte LDS reads and adds each sample to every one of its
// four accumulators whose window contains it.  Each accumulator still receives exactly its own R samples in index
// order, so the sums are bit-identical to std::accumulate, with 1/16 of the LDS instructions of the scalar form.
constexpr int kAvgPos = 4;                                  // positions per lane
constexpr int kAvgSpan = kAvgLanes * kAvgPos;               // positions per workgroup (1024)

__device__ __forceinline__ void window_sums(const float* __restrict__ w, uint32_t R, float acc[kAvgPos])
{
    // accumulator j sums w[j .. j+R)
#pragma unroll
    for (int j = 0; j < kAvgPos; ++j) acc[j] = 0.0f;
    const uint32_t total = R + kAvgPos - 1;                 // samples touched: w[0 .. R+3)
    uint32_t e = 0;
    {   // head chunk: element e feeds accumulators j <= e
        const float4 x = *reinterpret_cast<const float4*>(w);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < kAvgPos; ++j)
                if (j <= u && (uint32_t)u < (uint32_t)j + R) acc[j] = acc[j] + xs[u];
        e = 4;
    }
    for (; e + 4 <= R; e += 4) {                            // interior: every element feeds all four
        const float4 x = *reinterpret_cast<const float4*>(w + e);
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.x;
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.y;
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.z;
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.w;
    }
    for (; e < total; e += 4) {                             // tail chunks: element e feeds accumulators with e < j + R
        const float4 x = *reinterpret_cast<const float4*>(w + e);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < kAvgPos; ++j)
                if (e + u >= (uint32_t)j && e + u < (uint32_t)j + R) acc[j] = acc[j] + xs[u];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One workgroup (4 waves) per stream does the whole symbol stage of a call:
//   A. window sums for the positions that became computable (4 positions per lane, 1024 per sweep) -> weights (global ring,
//      they are needed again by later calls) and sign-difference mask words (global ring + an LDS image of every mask word
//      the edge search can touch, so the search never waits on HBM/L2);
//   B. wave 0: edge search on the LDS mask image (4096 positions per step);
//   C. runs are dealt round-robin to the 4 waves; each wave sums its run in element order: 256 samples at a time go
//      global -> registers -> a private LDS strip, then wave-uniform (broadcast) 16-byte LDS reads feed one v_add per
//      sample while the next 256 are already in flight;
//   D. lane 0 packs the bits, advances the ring base (the reference's erase) and writes the result slot.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSymLanes = 256;
constexpr uint32_t kRunStrip = 256;                       // samples per run-sum step

__device__ __forceinline__ uint32_t find_flag_lds(const unsigned long long* lmask, uint32_t base, uint32_t rmask,
                                                  uint32_t from, uint32_t to, bool want)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wr0 = (base + from) & ~63u;
    for (uint32_t it = 0;; ++it) {
        const uint32_t wstart = wr0 + it * 4096u;
        if ((int32_t)(wstart - base) >= (int32_t)to) break;
        const uint32_t wr = wstart + lane * 64u;
        const int32_t lw = (int32_t)(wr - base);
        unsigned long long w = 0;
        if (lw < (int32_t)to) {
            w = lmask[(wr & rmask) >> 6];
            if (!want) w = ~w;
            const int32_t lo = (int32_t)from - lw;
            if (lo > 0) w = lo >= 64 ? 0ull : (w & (~0ull << lo));
            const int32_t hi = (int32_t)to - lw;
            if (hi < 64) w &= (1ull << hi) - 1ull;
        }
        const unsigned long long hit = __ballot(w != 0ull);
        if (hit) {
            const int src = __ffsll((long long)hit) - 1;
            const unsigned long long ww = __shfl(w, src, 64);
            return (uint32_t)((int32_t)(wstart - base) + src * 64 + (__ffsll((long long)ww) - 1));
        }
    }
    return 0xFFFFFFFFu;
}

__global__ __launch_bounds__(kSymLanes) void k_symbols(const float* __restrict__ tail, uint32_t ring_cap, SymState* __restrict__ sym,
                                                        unsigned long long* __restrict__ flipmask, float* __restrict__ weight,
                                                        const SymbolParams* __restrict__ sp, const StreamCall* __restrict__ call,
                                                        uint32_t* __restrict__ slots, uint32_t slot_words,
                                                        uint32_t* __restrict__ flips_dbg, uint32_t flips_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve: [lmask: ring_cap/64 u64][win: kAvgSpan + 2*R + 16 floats][strips: 4 waves x 2 x kRunStrip floats]
    unsigned long long* lmask = reinterpret_cast<unsigned long long*>(smem);
    __shared__ unsigned long long words[kAvgSpan / 64];
    __shared__ uint32_t flips[kMaxFlipsPerCall];
    __shared__ uint32_t runinfo[kMaxFlipsPerCall];          // (count << 1) | bit
    __shared__ uint32_t s_nfl, s_overflow;
    const uint32_t s = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* slot = slots + (size_t)s * slot_words;
    BitsHeader* hdr = reinterpret_cast<BitsHeader*>(slot);
    uint32_t* outw = slot + sizeof(BitsHeader) / 4;
    const uint32_t cap_bits = (slot_words - sizeof(BitsHeader) / 4) * 32;
    const uint32_t m = call[s].fir_m;
    const SymState old = sym[s];
    if (!m) {                                               // symbol stage not reached this call
        if (tid == 0) { hdr->nbits = 0; hdr->held_after = old.held; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = old.base + old.held - old.cached; }
        return;
    }
    const SymbolParams q = sp[s];
    SymState st = state_after_push(old, q, m);
    const uint32_t h = st.held;
    if (h < q.min_held || h < q.spb) {
        if (tid == 0) { sym[s] = st; hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = st.base + h - st.cached; }
        return;
    }
    const uint32_t R = q.R, rmask = ring_cap - 1;
    float* win = reinterpret_cast<float*>(lmask + ring_cap / 64);
    float* strips = win + ((kAvgSpan + 2 * R + 16 + 3) & ~3u);
    const float* v = tail + (size_t)s * ring_cap;
    unsigned long long* gmask = flipmask + (size_t)s * (ring_cap / 64);
    float* gw = weight + (size_t)s * ring_cap;
    const uint32_t end = st.base + h;
    const uint32_t pend = end - R + 1;                      // first position whose right window is still incomplete
    const uint32_t limit = h - q.spb;                       // backlog indices searched: [R, limit)

    // ---- A0: LDS image of the cached mask words the search can touch
    {
        const uint32_t wr0 = (st.base + R) & ~63u;                     // ring position of the first word
        const uint32_t nw = ((st.base + limit) - wr0 + 63u) >> 6;      // modular difference: safe across the 2^32 wrap
        for (uint32_t i = tid; i < nw; i += kSymLanes) {
            const uint32_t wi = ((wr0 + 64u * i) & rmask) >> 6;
            lmask[wi] = gmask[wi];
        }
    }
    // ---- A1: window sums for the new positions, one sweep of kAvgSpan positions at a time
    for (uint32_t c0 = st.cached & ~63u; (int32_t)(pend - c0) > 0; c0 += kAvgSpan) {
        __syncthreads();                                    // previous sweep's `win`/`words` consumers are done; A0 stores ordered
        const uint32_t w0 = c0 - R;
        const uint32_t wn = kAvgSpan + 2 * R + 8;
        for (uint32_t k = tid; k < wn; k += kSymLanes) win[k] = v[(w0 + k) & rmask];
        if (tid < kAvgSpan / 64) words[tid] = 0ull;
        __syncthreads();
        const uint32_t p0 = c0 + tid * kAvgPos;
        if ((int32_t)(pend - p0) > 0) {
            float sl[kAvgPos], sr[kAvgPos];
            window_sums(win + tid * kAvgPos, R, sl);
            {   // right windows start R samples later; R is not a multiple of 4 in general: realign through a residual shift
                const uint32_t off = tid * kAvgPos + R;
                const uint32_t al = off & ~3u, sh = off & 3u;
                const float* w = win + al;
#pragma unroll
                for (int j = 0; j < kAvgPos; ++j) sr[j] = 0.0f;
                const uint32_t total = sh + R + kAvgPos - 1;
                for (uint32_t e = 0; e < total; e += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(w + e);
                    const float xs[4] = {x.x, x.y, x.z, x.w};
                    if (e >= sh + kAvgPos - 1 && e + 4 <= sh + R) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int j = 0; j < kAvgPos; ++j) sr[j] = sr[j] + xs[u];
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int j = 0; j < kAvgPos; ++j)
                                if (e + u >= sh + (uint32_t)j && e + u < sh + (uint32_t)j + R) sr[j] = sr[j] + xs[u];
                    }
                }
            }
            unsigned int bits = 0;
#pragma unroll
            for (int j = 0; j < kAvgPos; ++j) {
                const uint32_t p = p0 + j;
                if ((int32_t)(pend - p) > 0) {
                    const float al = sl[j] / (float)R, ar = sr[j] / (float)R;   // no clamping: p >= base+R is all the search reads, p+R <= end
                    if (sgnf(al) != sgnf(ar)) bits |= 1u << j;
                    const float d = ar - al;
                    gw[p & rmask] = q.float_abs ? __builtin_fabsf(d) : (float)abs((int)d);
                }
            }
            if (bits) atomicOr(&words[tid >> 4], (unsigned long long)bits << ((tid & 15) * 4));
        }
        __syncthreads();
        if (tid < kAvgSpan / 64 && (int32_t)(pend - (c0 + tid * 64)) > 0) {
            const uint32_t wi = ((c0 + tid * 64) & rmask) >> 6;
            gmask[wi] = words[tid];
            lmask[wi] = words[tid];
        }
    }
    __threadfence_block();                                  // this call's weights (global) are read back by wave 0 below
    __syncthreads();

    // ---- B: edge search (wave 0)
    if (wave == 0) {
        uint32_t pos = R, nfl = 0, overflow = 0;
        while (pos < limit) {
            const uint32_t lo = find_flag_lds(lmask, st.base, rmask, pos, limit, true);
            if (lo == 0xFFFFFFFFu) break;
            const uint32_t hi = find_flag_lds(lmask, st.base, rmask, lo + 1, limit, false);
            if (hi == 0xFFFFFFFFu) break;
            float bw = -1.0f;                               // first maximum of the weight over [lo, hi)
            uint32_t bi = 0xFFFFFFFFu;
            for (uint32_t i = lo + lane; i < hi; i += 64) {
                const float w = gw[(st.base + i) & rmask];
                if (bi == 0xFFFFFFFFu || w > bw) { bw = w; bi = i; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const float ow = __shfl_down(bw, off, 64);
                const uint32_t oi = __shfl_down(bi, off, 64);
                if (oi != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || ow > bw || (ow == bw && oi < bi))) { bw = ow; bi = oi; }
            }
            const uint32_t f = __shfl(bi, 0, 64);
            if (nfl < kMaxFlipsPerCall) { if (lane == 0) flips[nfl] = f; ++nfl; } else { overflow = 1; break; }
            pos = f + R;
        }
        if (lane == 0) { s_nfl = nfl; s_overflow = overflow; }
    }
    __syncthreads();
    const uint32_t nfl = s_nfl;

    // ---- C: per-run sums in element order (std::accumulate), runs dealt round-robin to the waves
    float* strip = strips + wave * (2 * kRunStrip);
    for (uint32_t r = wave; r < nfl; r += kSymLanes / 64) {
        const uint32_t a = r ? flips[r - 1] : 0u, b = flips[r];
        float acc = 0.0f;
        float nx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k = a + 64 * j + lane;
            nx[j] = k < b ? v[(st.base + k) & rmask] : 0.0f;
        }
        uint32_t par = 0;
        for (uint32_t k0 = a; k0 < b; k0 += kRunStrip, par ^= 1u) {
            float* sb = strip + par * kRunStrip;
#pragma unroll
            for (int j = 0; j < 4; ++j) sb[64 * j + lane] = nx[j];
            if (k0 + kRunStrip < b) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t k = k0 + kRunStrip + 64 * j + lane;
                    nx[j] = k < b ? v[(st.base + k) & rmask] : 0.0f;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): this wave's strip writes have landed (wave-private strip)
            __builtin_amdgcn_wave_barrier();
            const uint32_t cnt = min(kRunStrip, b - k0);
            const float4* s4 = reinterpret_cast<const float4*>(sb);
            uint32_t i = 0;
            for (; i + 16 <= cnt; i += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 x = s4[(i >> 2) + u];      // wave-uniform address: one broadcast LDS read feeds four adds
                    acc = acc + x.x; acc = acc + x.y; acc = acc + x.z; acc = acc + x.w;
                }
            }
            for (; i < cnt; ++i) acc = acc + sb[i];
        }
        if (lane == 0) {
            const float mean = acc / (float)(b - a);
            const uint32_t cnt = (uint32_t)roundf((float)(b - a) / (float)q.spb);
            runinfo[r] = (cnt << 1) | (mean > 0.0f ? 1u : 0u);
        }
    }
    __syncthreads();

    // ---- D: bits, ring advance, result slot
    if (tid == 0) {
        uint32_t nbits = 0, cur = 0, overflow = s_overflow;
        for (uint32_t r = 0; r < nfl; ++r) {
            const uint32_t bit = runinfo[r] & 1u;
            for (uint32_t k = runinfo[r] >> 1; k; --k) {
                if (nbits >= cap_bits) { overflow = 1; break; }
                cur |= bit << (nbits & 31);
                if ((nbits & 31) == 31) { outw[nbits >> 5] = cur; cur = 0; }
                ++nbits;
            }
        }
        if (nbits & 31) outw[nbits >> 5] = cur;
        if (flips_dbg)
            for (uint32_t r = 0; r < nfl && r < flips_cap; ++r) flips_dbg[(size_t)s * flips_cap + r] = flips[r];
        const uint32_t last = nfl ? flips[nfl - 1] : 0u;   // erase the consumed prefix (SymbolExtractor.h:156-157) = advance the base
        st.cached = pend;
        st.base += last;
        st.held = h - last;
        sym[s] = st;
        hdr->nbits = nbits; hdr->held_after = st.held; hdr->nflips = nfl; hdr->overflow = overflow; hdr->uncached = end - st.cached;
    }
}

void launch_symbols(hipStream_t st, uint32_t n_streams, uint32_t max_m, uint32_t max_new, uint32_t max_R, const float* tail,
                    uint32_t ring_cap, SymState* sym, unsigned long long* flipmask, float* weight, const SymbolParams* sp,
                    const StreamCall* call, uint32_t* slots, uint32_t slot_words, uint32_t* flips_dbg, uint32_t flips_cap)
{
    (void)max_m; (void)max_new;
    const size_t lds = (size_t)(ring_cap / 64) * 8 + (size_t)((kAvgSpan + 2 * max_R + 16 + 3) & ~3u) * 4 + (size_t)(kSymLanes / 64) * 2 * kRunStrip * 4;
    hipLaunchKernelGGL(k_symbols, dim3(n_streams), dim3(kSymLanes), lds, st, tail, ring_cap, sym, flipmask, weight, sp, call, slots,
                       slot_words, flips_dbg, flips_cap);
}

}  // namespace hd
