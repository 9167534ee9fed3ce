// The stream tail as a kernel of its own: one workgroup per stream, 64 lanes (batches) or 256 lanes (a handful of streams, where the
// latency of the one wave would show).  The same body rides in the stage-1 launch in batch mode (k_step, decimate.hip).
#include <hip/hip_runtime.h>

#include "tail_body.h"

namespace hd {

template <int NT, int OP, int D2, int T2>
__global__ __launch_bounds__(NT) void k_tail(const TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_lds[];
    tail_body<NT, OP, D2, T2>(a, blockIdx.x, tail_lds);
}

static constexpr int tail_op(int lanes) { return lanes == 64 ? 4 : 1; }

bool tail_layout(TailArgs& a, int lanes, int ratio2, int ntaps2, uint32_t max_taps, uint32_t max_R, uint32_t min_R, uint32_t ring_cap, uint32_t lds_limit)
{
    if (!((ratio2 == 2 && ntaps2 == 69) || (ratio2 == 4 && ntaps2 == 139))) return false;
    if (lanes != 64 && lanes != 256) return false;
    const uint32_t NT = (uint32_t)lanes, P = NT * (uint32_t)tail_op(lanes), B = P + (kFirBatch - 1);
    const uint32_t XN = (uint32_t)(ntaps2 - 1) + P * (uint32_t)ratio2;
    const uint32_t H = max_taps ? max_taps - 1 : 0;
    uint32_t off = kTailHdrBytes + (((XN + 1) & ~1u) + 2 * NT + 2) * 8;
    a.f_off = off; off += ((H + (kFirBatch - 1) + P + 4 + 1) & ~1u) * 8;
    a.v_off = off; off += ((max_R + B + 8 + 3) & ~3u) * 4;
    a.ws_off = off; off += ((max_R + B + 4 + 3) & ~3u) * 4;
    a.words_off = off; off += (kAvgPos * NT / 64 + 2) * 8;
    const uint32_t stream_phase = off;
    // search phase (overlays the windows above): flag-mask image, flip list, run info, run-sum strips, window-sum cache
    off = kTailHdrBytes;
    a.lmask_off = off; off += (ring_cap / 64) * 8;
    uint32_t fl_cap = ring_cap / (min_R ? min_R : 4u) + 2u;              // a flip point moves the search on by R
    fl_cap = (fl_cap + 63u) & ~63u;
    if (fl_cap > kMaxFlipsPerCall) fl_cap = kMaxFlipsPerCall;
    a.fl_cap = fl_cap;
    a.flips_off = off; off += fl_cap * 8;
    a.strips_off = off; off += (NT / 64) * kTailStrip * 4;
    a.wc_off = off;
    const uint32_t need = stream_phase > off + 4096 ? stream_phase : off + 4096;   // at least 1024 cached window sums
    if (need > lds_limit) return false;
    a.lds_bytes = lds_limit < 65536 ? lds_limit : 65536;
    if (a.lds_bytes < need) a.lds_bytes = need;
    a.wc_cap = (a.lds_bytes - a.wc_off) / 4;
    return true;
}

bool launch_tail(hipStream_t st, int lanes, int ratio2, int ntaps2, uint32_t n_streams, const TailArgs& a)
{
#define HD_TAIL_CASE(NT, D, T)                                                                                        \
    if (lanes == NT && ratio2 == D && ntaps2 == T) {                                                                  \
        hipLaunchKernelGGL((k_tail<NT, tail_op(NT), D, T>), dim3(n_streams), dim3(NT), a.lds_bytes, st, a);           \
        return true;                                                                                                  \
    }
    HD_TAIL_CASE(64, 2, 69) HD_TAIL_CASE(64, 4, 139) HD_TAIL_CASE(256, 2, 69) HD_TAIL_CASE(256, 4, 139)
#undef HD_TAIL_CASE
    return false;
}

}  // namespace hd
