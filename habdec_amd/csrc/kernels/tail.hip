// The stream tail as a kernel of its own: one workgroup per stream, 64 lanes (batches) or 256 lanes (a handful of streams, where the
// latency of the one wave would show).  The same body rides in the stage-1 launch in batch mode (k_step, decimate.hip).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "tail_body.h"

namespace hd {
namespace HD_ARITH_NS {

template <int NT, int OP, int D2, int T2>
__global__ __launch_bounds__(NT) void k_tail(const TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_lds[];
    tail_body<NT, OP, D2, T2>(a, blockIdx.x, tail_lds);
}

#ifdef HD_RING_FAULT   // the tails of THIS translation unit (k_tail: synchronous delivery, the drain at hd_flush)
extern "C" void HD_DBG_NAME(hd_debug_tail_fault_arm_no_tag)()
{
    const unsigned int one = 1, zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tail_fault_armed), &one, sizeof one);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tail_fault_fired), &zero, sizeof zero);
}
#endif
#ifdef HD_STAMP_TAIL
extern "C" void HD_DBG_NAME(hd_debug_tail_stamps)(unsigned long long* host, size_t n) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tail_stamps), n * 8); }
#endif

// Stage-2 outputs per lane and piece: four for the one-wave tail; the 256-lane tail (a handful of streams) takes one -- pieces of 256 outputs, every wave busy
// on short calls -- or four: a call of 1024 decimated samples is then ONE piece instead of four (a single stream's 65536-sample push at /64: 55.7 against
// 69.8 us per push, round 5).  The engine picks by the call's size.
static constexpr int tail_op_default(int lanes) { return lanes == 64 ? 4 : 1; }

bool tail_layout(TailArgs& a, int lanes, int ratio2, int ntaps2, uint32_t max_taps, uint32_t max_R, uint32_t min_R, uint32_t ring_cap, uint32_t pend_max, uint32_t lds_limit, int op)
{
    if (!op) op = tail_op_default(lanes);
    if (!(lanes == 64 ? op == 4 : (op == 1 || op == 4))) return false;
    a.op = (uint32_t)op;
    if (!((ratio2 == 2 && ntaps2 == 69) || (ratio2 == 4 && ntaps2 == 139))) return false;
    if (lanes != 64 && lanes != 256) return false;
    if (pend_max > kFirBatch - 1) pend_max = kFirBatch - 1;
    a.pend_max = pend_max;
    const uint32_t NT = (uint32_t)lanes, P = NT * (uint32_t)op, B = 2 * P + pend_max;
    const uint32_t XN = (uint32_t)(ntaps2 - 1) + P * (uint32_t)ratio2;
    const uint32_t H = max_taps ? max_taps - 1 : 0;
    // 64 lanes: the compact carve (tail_body.h: kCompact) -- the discriminator's exchange array lives in the part of the stage-1 image that is dead
    // while the low-pass runs, and the read-ahead slack behind X and F is the next region (values read there are never used).  It is what lets
    // four tails sit beside FIVE stage-1 tile slots in a CU's LDS (decimate.hip: k_step_cu).
    const bool compact = lanes == 64;
#ifdef HD_X_NOPAD
    const bool xpad = false;
#else
    const bool xpad = compact && op == 4 && ratio2 == 2;      // the padded stage-1 image (tail_body.h: kXPad): 16 bytes behind every 8 samples
#endif
    const uint32_t XNP = xpad ? XN + 2u * (XN >> 3) + 2u : XN;
    uint32_t off = kTailHdrBytes + (compact ? ((XNP + 1) & ~1u) : ((XN + 4 + 1) & ~1u) + 2 * NT + 2) * 8;
    a.f_off = off; off += ((H + B + (compact ? 0u : 16u) + 1) & ~1u) * 8;
    a.v_off = off; off += ((max_R + B + 16 + 3) & ~3u) * 4;
    a.ws_off = off; off += ((max_R + B + 8 + 3) & ~3u) * 4;
    a.words_off = off; off += (kWidePos * NT / 64 + 2) * 8;
    a.tp_off = off; off += ((max_taps + 8 + 3) & ~3u) * 4;
    a.h2_off = off; off += (((uint32_t)ntaps2 + 7) & ~3u) * 4;
    const uint32_t stream_phase = off;
    // search phase (overlays the windows above): flip list, run info and run-sum strips at fixed offsets, then the region each stream carves for
    // itself (tail_body.h): flag-mask image of the backlog it searches | cached samples | cached window sums
    off = kTailHdrBytes;
    uint32_t fl_cap = ring_cap / (min_R ? min_R : 4u) + 2u;              // a flip point moves the search on by R
    fl_cap = (fl_cap + 63u) & ~63u;
    if (fl_cap > kMaxFlipsPerCall) fl_cap = kMaxFlipsPerCall;           // (the same bound on every path: sym_common.h)
    a.fl_cap = fl_cap;
    a.flips_off = off; off += fl_cap * 8;
    a.strips_off = off; off += (NT / 64) * kTailStrip * 4;
    a.dyn_off = off;
    // the mask image of the longest backlog there can be (the whole ring) must fit; in the common case (a few symbols of backlog) the mask takes
    // a few hundred bytes and the region should hold at least 1024 cached window sums + 512 cached samples
    const uint32_t dyn_min = (ring_cap / 64) * 8 > 6144u + 512u ? (ring_cap / 64) * 8 : 6144u + 512u;
    uint32_t need = stream_phase > off + dyn_min ? stream_phase : off + dyn_min;
    if (need < kTailHdrBytes + kSpecWaveLds) need = kTailHdrBytes + kSpecWaveLds;   // the spectrum's transpose plane (spectrum_wave.h) reuses the scratch at the end
    if (need > lds_limit) return false;
    uint32_t want = off + 512u + 4 * (2560 + 1536);                      // mask + window sums of ~four symbols of backlog at 50 baud + samples of one call and a half
    if (want < need) want = need;
    a.lds_bytes = (want < lds_limit ? want : lds_limit) & ~15u;
    return true;
}

bool launch_tail(hipStream_t st, int lanes, int ratio2, int ntaps2, uint32_t n_streams, const TailArgs& a, hipEvent_t ev_stop)
{
#define HD_TAIL_CASE(NT, OP, D, T)                                                                                    \
    if (lanes == NT && a.op == OP && ratio2 == D && ntaps2 == T) {                                                    \
        if (ev_stop) hipExtLaunchKernelGGL((k_tail<NT, OP, D, T>), dim3(n_streams), dim3(NT), a.lds_bytes, st, nullptr, ev_stop, 0u, a); \
        else hipLaunchKernelGGL((k_tail<NT, OP, D, T>), dim3(n_streams), dim3(NT), a.lds_bytes, st, a);               \
        return true;                                                                                                  \
    }
    HD_TAIL_CASE(64, 4, 2, 69) HD_TAIL_CASE(64, 4, 4, 139) HD_TAIL_CASE(256, 1, 2, 69) HD_TAIL_CASE(256, 1, 4, 139) HD_TAIL_CASE(256, 4, 2, 69) HD_TAIL_CASE(256, 4, 4, 139)
#undef HD_TAIL_CASE
    return false;
}

}  // namespace HD_ARITH_NS
}  // namespace hd
