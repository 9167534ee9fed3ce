// First decimation stage (/32) as loader waves + consumer waves around LDS tile slots -- the stage-1 half of the per-CU step kernel
// (k_step_cu, decimate.hip).
//
// What it computes is decimate.hip's sum (reference code/Decoder/Decimator.h:128-138): out[o] = sum_t buf[32 o + t] * tap[t], every output's T
// products added by ONE lane in ascending tap order with separately rounded multiply and add.  What differs is who moves the bytes.
// In the single-wave stage-1 workgroup (decimate_body) a wave is EITHER issuing the next tile's loads OR computing: its tile time is the
// sum of an issue burst, the data's flight, an LDS store pass and the tap loop, the tile in flight sits in 72 VGPRs, and beside the
// stream tails of a step launch -- which hold half of a CU's wave slots for the first half of the launch -- the four stage-1 waves left
// on a CU pull a third of the CU's share of HBM.  Here ONE wave per CU (two: RingGeom::nl) does nothing but keep loads in flight: LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 bytes straight into LDS, no VGPR destination, no ds_write) into its tile slots, a tile
// issued as soon as a slot is free and published through an LDS word once the wave's vector-memory counter says it has landed
// (tools/micro/loader_bw.hip: one such wave per CU streams 5.5 TB/s over the chip, two 6.2); the other waves take published tiles in
// order and spend their time in the tap loop.  Loads in flight no longer depend on what the computing waves are doing, and a tile in
// flight costs no registers.
//
// Slot layout = the padded rows the tap loop reads without bank conflicts: rows of 32 samples (256 B, one output's stride) at a pitch
// of 272 B, so that lane o's window starts 17 sixteen-byte chunks after lane o-1's and the lanes' ds_read_b128 fall into different
// bank groups.  An LDS-DMA instruction writes 1 KiB CONTIGUOUSLY (wave-uniform base + lane * 16), so the pad cannot be skipped on the
// LDS side; it is made on the SOURCE side instead: lane l of instruction i owns chunk P = 64 i + l of the slot, which is column
// P % 17 of row P / 17 -- a data chunk (column < 16: it loads that chunk of the stream) or the pad (column 16: it loads the row's
// last chunk again, never read).  64 rows are exactly 17 instructions; the HR halo rows in front (the T-1 samples before the tile's
// first output's stride) are two more -- or, for a stream's first tile, HR dword-wide LDS-DMA rows out of the stage history, whose
// samples sit at odd 8-byte offsets.
//
// Protocol (LDS words, RingCtl; round 5's version).  Every tile slot has ONE 64-bit word: (state, descriptor).  A loader owns a range of slots.  It puts
// its next tile into any of them whose word says FREE (a slot inside a tail's LDS slice only once that tail is done), issues the DMA, and -- when
// IB_STS.VM_CNT shows the tile has arrived -- stores (READY, stream, tile) into the slot's word: a plain store, nothing to wait for.  A computing wave reads
// all slot words at once (lane j reads word j), picks a READY one and takes it with ONE compare-and-swap of the whole word (READY, descriptor) -> TAKEN:
// whoever wins has the descriptor already; whoever loses looks again.  When its lanes' rows are in registers it stores FREE.  A word is written by the
// slot's loader (FREE -> READY), by the one consumer whose swap succeeded (READY -> TAKEN -> FREE) and by nobody else, so there is no publication queue to
// overflow and nothing to acknowledge (rounds 3-4 published through a sixteen-entry queue of sequence numbers: a loader paid two returning LDS atomics per
// tile for it -- in-kernel clocks: a lone loader wave spent its whole launch in that chain and the computing waves waited 40 % of theirs -- and a
// consumer that did not look for sixteen publications could find its word overwritten).  How many loader waves there are, and when they
// join, is nobody else's business -- stage 1 alone runs two from the start, a step launch starts with one and the first stream tail that finishes becomes
// the second (one loader's 6-bit vmcnt holds three tiles' DMA instructions).  `live` counts the loaders that may still publish: a consumer that finds
// no READY word and sees live == 0 is done.  Runs of tiles come from the per-XCD counters of the step launches (StepClaim, launch.h): the feeding consumer
// draws them -- a returning atomic, which a wave without DMA in flight can simply wait for -- and hands them to the loaders through run_q, "no more"
// sentinels included, for as long as it lives.  No s_barrier after the start: the stream tails in the workgroup's other waves never take part; every
// wait is bounded and reported (RingArgs::gave_up).
#pragma once
#include <hip/hip_runtime.h>

#include "arith.h"
#include "launch.h"

namespace hd {
namespace HD_ARITH_NS {

typedef float r_f32x2 __attribute__((ext_vector_type(2)));
typedef float r_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRingRowBytes = 272;                  // 32 samples + one 16-byte pad
constexpr int kRingNSLAlone = 4;                      // tile slots per loader wave of the loader / consumer kernels (k_stage1_cu at /8 and /4: eight slots)
constexpr int kWorkSlotBytes = 64 * kRingRowBytes;    // a worker wave's tile slot (ring_worker: the /32 stages): 64 padded rows
constexpr int kRingCtlBytes = 384;                 // (256 bytes of RingCtl; every 16 bytes count when a fifth tile slot has to fit beside four tails)
constexpr uint32_t kRingSpinLimit = 1u << 22;   // polls before a waiting wave gives up (seconds; a correct run waits microseconds)
template <int T> constexpr int ring_halo_rows() { return (T - 1 + 31) / 32; }
template <int T> constexpr int ring_slot_bytes() { return (64 + ring_halo_rows<T>()) * kRingRowBytes; }
template <int T, int NSL = kRingNSLAlone> constexpr int ring_bytes() { return 2 * NSL * ring_slot_bytes<T>() + kRingCtlBytes; }   // NSL ring slots per loader
// The /32 stages run the SYSTOLIC tap loop (ring_consumer): a lane keeps only ITS row of 32 samples in registers and the accumulators travel from
// lane to lane, so the 64 rows of a tile yield 64 - HR outputs (the sums that would wrap past lane 63 belong to the next tile) and consecutive
// tiles of a stream advance by that many rows: tile k holds rows [origin, origin + 64) of the stream's history-extended row sequence (rows 0 .. HR-1
// are the stage history), origin = min(k * ADV, R - ADV) with R = n / 32 rows in the call -- the last tile is pulled back so that it ends with the
// call's last row and never reads past the stream's input; the outputs it shares with its neighbour are computed twice, identically.
template <int T> constexpr int ring_adv() { return 64 - ring_halo_rows<T>(); }

__host__ __device__ constexpr uint32_t ring_sys_tiles(uint32_t n, uint32_t adv) { return (n / 32u + adv - 1u) / adv; }

#ifdef HD_STAMP_RING   // diagnostic build only (tools/micro/ring_stamps.py): where the loader and the consumers of k_step_cu spend their cycles
__device__ unsigned long long g_ring_stamps[512 * 8 * 8];
extern "C" void HD_DBG_NAME(hd_debug_ring_stamps)(unsigned long long* host, size_t n) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ring_stamps), n * 8); }
#define RSTAMP_DECL unsigned long long rs_t = __builtin_amdgcn_s_memtime(), rs_acc[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long rs_r0 = __builtin_amdgcn_s_memrealtime()
#define RSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); rs_acc[i] += t_ - rs_t; rs_t = t_; } while (0)
#define RSTAMP_WRITE(wave_, n_) do { if ((threadIdx.x & 63u) == 0 && blockIdx.x < 512 && (wave_) < 8u) { unsigned long long* g_ = g_ring_stamps + ((size_t)blockIdx.x * 8 + (wave_)) * 8; \
        for (int i_ = 0; i_ < 5; ++i_) g_[i_] = rs_acc[i_]; g_[5] = (n_); g_[6] = rs_r0; g_[7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define RSTAMP_DECL do { } while (0)
#define RSTAMP(i) do { } while (0)
#define RSTAMP_WRITE(wave_, n_) do { } while (0)
#endif

#ifdef HD_RING_FAULT
__device__ unsigned int g_ring_fault_fired, g_ring_fault_mode;
// (one fault per arming.  Mode 0: a loader never publishes its second tile and stops -- the computing waves' waits run out.  Mode 1: the feeding wave
// stops handing out runs after its second -- the loaders' waits run out.)
static void ring_fault_arm(unsigned int mode) { const unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_fault_mode), &mode, sizeof mode); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_fault_fired), &z, sizeof z); }
extern "C" void HD_DBG_NAME(hd_debug_ring_fault_arm)() { ring_fault_arm(0u); }
extern "C" void HD_DBG_NAME(hd_debug_ring_fault_arm_starve)() { ring_fault_arm(1u); }
#endif

constexpr uint32_t kSlotFree = 0u, kSlotReady = 1u, kSlotTaken = 2u;     // state (high word) of a tile slot's word
struct RingCtl {
    unsigned long long slot[16];   // per tile slot (global slot number, RingGeom): (state << 32) | (stream << 12) | tile.  Limits (checked by the launcher):
                            // 2^20 streams, 4096 tiles per stream and call.  Zero = FREE.
    uint32_t live;          // loader waves that may still publish
    uint32_t run_tail;      // runs the feeding consumer has put into run_q so far
    uint32_t run_head;      // runs the loaders have claimed (fetch-add)
    uint32_t _pad0;
    uint32_t run_q[4];      // drawn run numbers (0xFFFFFFFF: no more)
    uint32_t simd_rank[4];  // waves of the workgroup that have arrived on each SIMD (role assignment, k_step_cu)
    uint32_t roles_taken;   // bit w: role w has a wave
    uint32_t _pad[3];
};
static_assert(sizeof(RingCtl) <= kRingCtlBytes, "ring control block");
// first thing in the kernel, before the workgroup's first barrier: everything zero (every slot FREE), `live` = the loaders that exist from the start
__device__ __forceinline__ void ring_ctl_init(RingCtl* ctl, const uint32_t n_live)
{
    const uint32_t i = threadIdx.x;
    if (i < (uint32_t)kRingCtlBytes / 4u) reinterpret_cast<uint32_t*>(ctl)[i] = i == (uint32_t)offsetof(RingCtl, live) / 4u ? n_live : 0u;
}

// Where the tile slots of a loader / consumer kernel are: slot j < nb at ring + j * SLOT.  A loader owns a contiguous range of them.
struct RingGeom {
    unsigned char* ring; uint32_t nb;
};

struct RingArgs {
    const float2* in; size_t in_stride;             // this call's IQ slab
    const float2* hist_in; float2* hist_out;        // stage history [S][T-1], ping-pong
    const float* taps;
    float2* out; size_t out_stride;                 // stage-1 output [S][n/32]
    uint32_t n;                                     // samples per stream this call (uniform, multiple of 2048)
    uint32_t ntiles;                                // n / 2048
    StepClaim claim;
    unsigned int* gave_up;                          // mapped host word: set by a wave whose bounded wait ran out (never in a correct run; the engine then stays failed)
    // a single-stage plan (/4): the stage is the FINAL one -- its outputs go behind the pending samples of the low-pass buffer, and the head of the
    // chunk feeds the spectrum collection (decimate_body's final_stage bookkeeping; Decoder.h:443-444, :467-473)
    const StreamCall* call; uint32_t fir_hist_cap; float2* fft_in;       // call == nullptr: not the final stage
};

__device__ __forceinline__ uint32_t lds_addr_of(const void* p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(uint32_t n)
{
#define HD_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        HD_VMC(0) HD_VMC(1) HD_VMC(2) HD_VMC(3) HD_VMC(4) HD_VMC(5) HD_VMC(6) HD_VMC(7) HD_VMC(8) HD_VMC(9)
        HD_VMC(10) HD_VMC(11) HD_VMC(12) HD_VMC(13) HD_VMC(14) HD_VMC(15) HD_VMC(16) HD_VMC(17) HD_VMC(18) HD_VMC(19)
        HD_VMC(20) HD_VMC(21) HD_VMC(22) HD_VMC(23) HD_VMC(24) HD_VMC(25) HD_VMC(26) HD_VMC(27) HD_VMC(28) HD_VMC(29)
        HD_VMC(30) HD_VMC(31) HD_VMC(32) HD_VMC(33) HD_VMC(34) HD_VMC(35) HD_VMC(36) HD_VMC(37) HD_VMC(38) HD_VMC(39)
        HD_VMC(40) HD_VMC(41) HD_VMC(42) HD_VMC(43) HD_VMC(44) HD_VMC(45) HD_VMC(46) HD_VMC(47) HD_VMC(48) HD_VMC(49)
        HD_VMC(50) HD_VMC(51) HD_VMC(52) HD_VMC(53) HD_VMC(54) HD_VMC(55) HD_VMC(56) HD_VMC(57) HD_VMC(58) HD_VMC(59)
        HD_VMC(60) HD_VMC(61) HD_VMC(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef HD_VMC
}

// One LDS-DMA instruction: lane l's 16 (4) bytes at base + off go to LDS byte lds_dst + 16 l (4 l).  M0 carries the LDS address and is
// written in the statement that uses it (the compiler does not preserve it around inline asm).  `base` must be wave-uniform.
__device__ __forceinline__ void glds16(const void* base, uint32_t off, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* base, uint32_t off, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_dst) : "memory");
}

// Seventeen of them back to back -- a tile's 64 rows -- with M0 stepped by 1 KiB in between: one scalar instruction per DMA instruction
// instead of five (the loader's issue time per tile is what bounds a CU's stream when nothing else does).
template <bool HEAD_KEPT = false /* the tile's first rows are another tile's last ones and that tile has not come yet (a run's first tile): default policy there too */>
__device__ __forceinline__ void glds16_x17(const void* base, const uint32_t (&off)[17], uint32_t lds_dst)
{
    unsigned keep, scc_keep;                      // (s_add_u32 writes SCC, which compiler code around the statement may hold live: saved and restored)
// Cache policy of the body rows' LDS-DMA: nt -- these bytes are read exactly once (measured on one box, alternating builds: step launch 158.8 ->
// 156.2 us, stage 1 alone 115.6 -> 111.2; MI355X_MICROARCH.md rows ldsdma-fill / nt-weights) -- EXCEPT the last two of the seventeen instructions:
// they carry the tile's rows 56.5 .. 63, whose last seven are the NEXT tile's halo; left on the default policy they stay in this XCD's L2 for that
// re-read (with nt on all seventeen the PMC traffic of a step launch rose from 630.6 to 647.6 MB: the halo reads went to HBM).  The halo rows
// themselves are default-policy loads.
#define HD_GLDS_BODY_POLICY " nt"
#define HD_G1(n) "global_load_lds_dwordx4 %" #n ", %19" HD_GLDS_BODY_POLICY "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define HD_G1D(n) "global_load_lds_dwordx4 %" #n ", %19\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define HD_G17_OPS : "=&s"(keep), "=&s"(scc_keep)                                                                                                                   \
                   : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "v"(off[8]), "v"(off[9]),              \
                     "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]), "v"(off[14]), "v"(off[15]), "v"(off[16]), "s"(base), "s"(lds_dst)                      \
                   : "memory"
    // (Round 5: the worker waves of an XCD walk neighbouring runs at the same time, so the rows a run's FIRST tile shares with the previous run's last tile are
    // loaded long before that last tile comes -- with nt they were gone by then: 16 MB per step launch fetched twice at runs of four.  Such a tile keeps its
    // first two instructions -- rows 0 .. 7.5 -- on the default policy as well.)
    if constexpr (HEAD_KEPT)
        asm volatile("s_cselect_b32 %1, 1, 0\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %20\n\ts_nop 0\n\t"
                     HD_G1D(2) HD_G1D(3) HD_G1(4) HD_G1(5) HD_G1(6) HD_G1(7) HD_G1(8) HD_G1(9) HD_G1(10) HD_G1(11) HD_G1(12) HD_G1(13) HD_G1(14) HD_G1(15) HD_G1(16) HD_G1D(17)
                     "global_load_lds_dwordx4 %18, %19\n\ts_mov_b32 m0, %0\n\ts_cmp_lg_u32 %1, 0" HD_G17_OPS);
    else
        asm volatile("s_cselect_b32 %1, 1, 0\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %20\n\ts_nop 0\n\t"
                     HD_G1(2) HD_G1(3) HD_G1(4) HD_G1(5) HD_G1(6) HD_G1(7) HD_G1(8) HD_G1(9) HD_G1(10) HD_G1(11) HD_G1(12) HD_G1(13) HD_G1(14) HD_G1(15) HD_G1(16) HD_G1D(17)
                     "global_load_lds_dwordx4 %18, %19\n\ts_mov_b32 m0, %0\n\ts_cmp_lg_u32 %1, 0" HD_G17_OPS);
#undef HD_G17_OPS
#undef HD_G1D
#undef HD_G1
}
// HR history rows, one dword-wide DMA instruction per row (pitch 272 bytes)
template <int HR>
__device__ __forceinline__ void glds4_rows(const void* base, const uint32_t (&off)[HR], uint32_t lds_dst)
{
#pragma unroll
    for (int r = 0; r < HR; ++r) glds4(base, off[r], lds_dst + (uint32_t)r * kRingRowBytes);
}

__device__ __forceinline__ const void* uniform_ptr(const void* p)
{
    const uint64_t b = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                                         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b));
}

// ---------------------------------------------------------------------------------------------------------------- a loader wave
// The calling wave has been counted in RingCtl::live (ring_ctl_init for the loaders that exist from the start; a late one counts itself in,
// k_step_cu).  It owns the global slots [slot0, slot0 + nslots), nslots <= 8.
template <int T>
__device__ __forceinline__ void ring_loader(const RingArgs& a, const RingGeom& geo, RingCtl* __restrict__ ctl, const uint32_t slot0, const uint32_t nslots,
                                            const uint32_t stamp_row /* diagnostic builds */)
{
    constexpr int HR = ring_halo_rows<T>();
    constexpr int SLOT = ring_slot_bytes<T>();
    constexpr int NBODY = 17;                       // 64 rows x 17 chunks = 17 x 64 chunks
    constexpr int NHALO = (HR * 17 + 63) / 64;      // halo rows out of the stream itself (every tile but a stream's first)
    const uint32_t lane = threadIdx.x & 63u;
    // lane j < nslots watches my slot j (global slot slot0 + j)
    const uint32_t gs = slot0 + lane;
    const uint32_t my_dst = lds_addr_of(geo.ring) + gs * (uint32_t)SLOT;

    // per-lane source offsets (bytes from the tile's first body row / first halo row)
    uint32_t boff[NBODY], hoff[NHALO], hist_off[HR];
#pragma unroll
    for (int i = 0; i < NBODY; ++i) { const uint32_t P = 64u * i + lane, row = P / 17u, col = P - row * 17u; boff[i] = row * 256u + (col < 16u ? col : 15u) * 16u; }
#pragma unroll
    for (int i = 0; i < NHALO; ++i) {
        uint32_t P = 64u * i + lane; if (P >= (uint32_t)(HR * 17)) P = HR * 17 - 1;      // (the last instruction runs past the halo: those lanes are masked off below)
        const uint32_t row = P / 17u, col = P - row * 17u; hoff[i] = row * 256u + (col < 16u ? col : 15u) * 16u;
    }
#pragma unroll
    for (int r = 0; r < HR; ++r) {                  // history row r: dword `lane` of the row is component lane & 1 of stream sample (r - HR) * 32 + lane / 2
        const int h = (r - HR) * 32 + (int)(lane >> 1) + (T - 1);                          // index into the T-1 history samples (< 0: in front of them, never read)
        hist_off[r] = (uint32_t)(h < 0 ? 0 : h) * 8u + (lane & 1u) * 4u;
    }

    // ---- runs of tiles: drawn from the XCDs' counters by the feeding consumer (ring_consumer), handed over through ctl->run_q.  (A loader
    // issues nothing but LDS-DMA: a returning GLOBAL atomic among them would have to be counted by hand too, and its destination register is
    // the compiler's to move before the value has arrived.  The LDS atomics below return through the other counter.)
    const uint32_t run_len = a.claim.run_len;
    bool have = false, ended = false, claimed = false;
    uint32_t s = 0, tile = 0, left = 0, my_run = 0;
    uint32_t issued = 0, landed = 0;                // my tiles
    uint32_t inflight_instr = 0;                    // vector-memory instructions of my tiles in flight (at most three tiles: the counter holds 63)
    unsigned long long fifo = 0, sfifo = 0;         // ... per tile, oldest in the low byte; and the slots they are going to
    const uint32_t max_fly = nslots >= 3u ? 3u : 2u;
    uint32_t my_state = 0, my_desc = 0;             // lane j < nslots, my slot j: 0 = mine to fill, 1 = a tile is on its way into it, 2 = published (until its word says FREE again);
                                                    // (stream << 12) | tile of the tile there now
    uint32_t idle_spins = 0;
    RSTAMP_DECL;

    for (;;) {
        RSTAMP(0);
        // ---- one look at my slots' words: the only LDS round trip of an iteration
        unsigned long long w = 0;
        if (lane < nslots) w = __hip_atomic_load(&ctl->slot[gs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // A published slot is mine again once its consumer has stored FREE (its lanes' rows are in registers by then).  (Looked at BEFORE this iteration
        // publishes anything: the word of a slot whose tile is still on its way says FREE too -- the loader writes nothing when it issues.)
        if (my_state == 2u && (uint32_t)(w >> 32) == kSlotFree) my_state = 0u;
        // ---- publish what has landed, without blocking: the wave's count of outstanding vector-memory instructions is readable
        // (IB_STS.VM_CNT, low four bits in [3:0], high two in [23:22]); the oldest tile in flight has landed once no more than the
        // instructions issued after it are outstanding.
        while (issued != landed) {
            const uint32_t ib = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 7);
            const uint32_t oldest = (uint32_t)(fifo & 0xFFu), younger = inflight_instr - oldest;
            if (((ib & 15u) | (((ib >> 22) & 3u) << 4)) > younger) break;
            wait_vmcnt(younger);                                       // (returns at once -- the counter has said so -- but it is the instruction whose completion semantics the hand-off relies on)
            const uint32_t lslot = (uint32_t)(sfifo & 0xFFu) - slot0;   // my slot number
            inflight_instr -= oldest; fifo >>= 8; sfifo >>= 8;
            ++landed;
#ifdef HD_RING_FAULT   // fault-injection build (libhabdec_amd_fault.so, tests/test_gpu_fault.py): ONE loader of the process never publishes its second tile and stops
            if (landed == 2u && g_ring_fault_mode == 0u) {
                unsigned int won = 1u;
                if (lane == 0) won = atomicCAS(&g_ring_fault_fired, 0u, 1u);
                if ((unsigned int)__builtin_amdgcn_readfirstlane((int)won) == 0u) { RSTAMP_WRITE(stamp_row, issued); return; }   // (still counted in `live`: the consumers' waits run out)
            }
#endif
            // the lane that watches the slot stores its word: (READY, descriptor) -- a plain store, in order behind everything this wave has done to LDS
            if (lane == lslot) {
                __hip_atomic_store(&ctl->slot[gs], ((unsigned long long)kSlotReady << 32) | my_desc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                my_state = 2u;
            }
        }
        RSTAMP(2);
        if (!have && !ended) {                                         // the current run is used up: take the next one the feeder has drawn
            if (!claimed) {
                uint32_t idx = 0;
                if (lane == 0) idx = __hip_atomic_fetch_add(&ctl->run_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                my_run = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
                claimed = true;
            }
            const uint32_t tail = __hip_atomic_load(&ctl->run_tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((int32_t)(tail - my_run) > 0) {
                const uint32_t rr = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->run_q[my_run & 3u]);
                claimed = false;
                if (rr != 0xFFFFFFFFu) {
                    const uint32_t g0 = rr * run_len;                  // (rr: the run's number in the launch, whichever XCD's share it was drawn from)
                    s = g0 / a.ntiles; tile = g0 - s * a.ntiles; left = run_len; have = true;
                } else {
                    ended = true;
                }
            }
        }
        // which of my slots can take a tile?
        unsigned long long free_mask = 0;
        if (have && issued - landed < max_fly && inflight_instr + (uint32_t)(NBODY + (HR > NHALO ? HR : NHALO)) <= 63u)
            free_mask = __ballot(lane < nslots && my_state == 0u);
        if (free_mask) {
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(free_mask));      // my slot number; global slot slot0 + slot
            const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)my_dst, (int)slot);
            if (lane == slot) { my_state = 1u; my_desc = (s << 12) | tile; }
            uint32_t cnt = NBODY;
            const unsigned char* body = reinterpret_cast<const unsigned char*>(a.in + (size_t)s * a.in_stride) + (size_t)tile * (64u * 256u);
            if (tile == 0) {
                const unsigned char* hb = reinterpret_cast<const unsigned char*>(a.hist_in + (size_t)s * (T - 1));
                glds4_rows<HR>(hb, hist_off, dst);
                cnt += HR;
            } else {
                const unsigned char* hb = body - HR * 256;
#pragma unroll
                for (int i = 0; i < NHALO; ++i) {
                    if (64 * (i + 1) <= HR * 17 || lane < (uint32_t)(HR * 17 - 64 * i)) glds16(hb, hoff[i], dst + 1024u * i);
                }
                cnt += NHALO;
            }
            glds16_x17(body, boff, dst + (uint32_t)(HR * kRingRowBytes));
            fifo |= (unsigned long long)cnt << (8u * (issued - landed));
            sfifo |= (unsigned long long)(slot0 + slot) << (8u * (issued - landed));
            inflight_instr += cnt;
            ++issued;
            ++tile; --left;
            if (!left) have = false;
            idle_spins = 0;
            RSTAMP(1);
            continue;
        }
        if (!have && ended && issued == landed) break;                 // nothing left to issue, nothing in flight
        if (issued == landed && ++idle_spins > kRingSpinLimit) {       // (bounded, see the consumers' wait)
            if (lane == 0) __hip_atomic_store(a.gave_up, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (a plain store: atomics on mapped host memory need PCIe atomics)
            break;
        }
        __builtin_amdgcn_s_sleep(1);                                   // my slots are busy, the next run has not been drawn yet, or loads are on their way
        RSTAMP(3);
    }
    // every tile of mine is published: count myself out (the consumers' end condition)
    if (lane == 0) (void)__hip_atomic_fetch_sub(&ctl->live, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    RSTAMP_WRITE(stamp_row, issued);
}

// -------------------------------------------------------------------------------------------------------------- a consumer wave
// Sixteen taps of ONE sum, scheduled by hand: the sum is a serial chain of adds by definition (ascending tap order, one accumulator), the
// products are not -- but written as "acc = acc + x * k" the compiler multiplies into a temporary right in front of each add, and a lone wave
// then waits out the multiplier's latency sixteen times per chunk (tools/micro/valu_rate.hip: 6.4 cycles per instruction against 5.2 for
// independent ones).  Here the products run three taps ahead of the adds through four rotating temporaries, so an independent multiply sits
// between consecutive adds.  Same instructions, same order of the adds, separately rounded multiply and add: bit-identical results.
// Taps arrive as eight aligned scalar pairs; op_sel picks the pair's low or high word for both halves of the packed multiply.
// (Fast mode: sixteen v_pk_fma_f32 into ONE accumulator -- the single-wave kernels' form; the worker waves split the chain in two, ring_fma16x2.)
__device__ __forceinline__ void ring_mac16_asm(r_f32x2& acc, const r_f32x4 (&x)[8], const r_f32x2 (&kp)[8])
{
#ifdef HD_FAST_ARITH
#define HD_FMA_E(xi, ki) "v_pk_fma_f32 %0, %" #xi ", %" #ki ", %0 op_sel_hi:[1,0,1]\n\t"
#define HD_FMA_O(xi, ki) "v_pk_fma_f32 %0, %" #xi ", %" #ki ", %0 op_sel:[0,1,0]\n\t"
    asm volatile(
        HD_FMA_E(1, 17) HD_FMA_O(2, 17) HD_FMA_E(3, 18) HD_FMA_O(4, 18) HD_FMA_E(5, 19) HD_FMA_O(6, 19) HD_FMA_E(7, 20) HD_FMA_O(8, 20)
        HD_FMA_E(9, 21) HD_FMA_O(10, 21) HD_FMA_E(11, 22) HD_FMA_O(12, 22) HD_FMA_E(13, 23) HD_FMA_O(14, 23) HD_FMA_E(15, 24) "v_pk_fma_f32 %0, %16, %24, %0 op_sel:[0,1,0]"
        : "+v"(acc)
        : "v"(x[0].xy), "v"(x[0].zw), "v"(x[1].xy), "v"(x[1].zw), "v"(x[2].xy), "v"(x[2].zw), "v"(x[3].xy), "v"(x[3].zw),
          "v"(x[4].xy), "v"(x[4].zw), "v"(x[5].xy), "v"(x[5].zw), "v"(x[6].xy), "v"(x[6].zw), "v"(x[7].xy), "v"(x[7].zw),
          "s"(kp[0]), "s"(kp[1]), "s"(kp[2]), "s"(kp[3]), "s"(kp[4]), "s"(kp[5]), "s"(kp[6]), "s"(kp[7]));
#undef HD_FMA_E
#undef HD_FMA_O
#else
    r_f32x2 t0, t1, t2, t3;
#define HD_MUL_E(t, xi, ki) "v_pk_mul_f32 %" #t ", %" #xi ", %" #ki " op_sel_hi:[1,0]\n\t"
#define HD_MUL_O(t, xi, ki) "v_pk_mul_f32 %" #t ", %" #xi ", %" #ki " op_sel:[0,1]\n\t"
#define HD_ADD(t) "v_pk_add_f32 %0, %0, %" #t "\n\t"
    // operands: 0 acc; 1-4 temporaries; 5-20 the sixteen sample pairs (x[0].xy, x[0].zw, x[1].xy, ...); 21-28 the eight tap pairs
    asm volatile(
        HD_MUL_E(1, 5, 21) HD_MUL_O(2, 6, 21) HD_MUL_E(3, 7, 22)
        HD_ADD(1) HD_MUL_O(4, 8, 22)
        HD_ADD(2) HD_MUL_E(1, 9, 23)
        HD_ADD(3) HD_MUL_O(2, 10, 23)
        HD_ADD(4) HD_MUL_E(3, 11, 24)
        HD_ADD(1) HD_MUL_O(4, 12, 24)
        HD_ADD(2) HD_MUL_E(1, 13, 25)
        HD_ADD(3) HD_MUL_O(2, 14, 25)
        HD_ADD(4) HD_MUL_E(3, 15, 26)
        HD_ADD(1) HD_MUL_O(4, 16, 26)
        HD_ADD(2) HD_MUL_E(1, 17, 27)
        HD_ADD(3) HD_MUL_O(2, 18, 27)
        HD_ADD(4) HD_MUL_E(3, 19, 28)
        HD_ADD(1) HD_MUL_O(4, 20, 28)
        HD_ADD(2) HD_ADD(3) "v_pk_add_f32 %0, %0, %4"
        : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(x[0].xy), "v"(x[0].zw), "v"(x[1].xy), "v"(x[1].zw), "v"(x[2].xy), "v"(x[2].zw), "v"(x[3].xy), "v"(x[3].zw),
          "v"(x[4].xy), "v"(x[4].zw), "v"(x[5].xy), "v"(x[5].zw), "v"(x[6].xy), "v"(x[6].zw), "v"(x[7].xy), "v"(x[7].zw),
          "s"(kp[0]), "s"(kp[1]), "s"(kp[2]), "s"(kp[3]), "s"(kp[4]), "s"(kp[5]), "s"(kp[6]), "s"(kp[7]));
#undef HD_MUL_E
#undef HD_MUL_O
#undef HD_ADD
#endif
}
#ifdef HD_FAST_ARITH
// Fast mode, worker waves: the sixteen taps of a chunk as two chains of eight -- even slots into `a`, odd slots into `b` (added once, when the sum leaves
// the wave) -- so that consecutive v_pk_fma_f32 never depend on each other: a lone wave on its SIMD issues them back to back.
__device__ __forceinline__ void ring_fma16x2(r_f32x2& a, r_f32x2& b, const r_f32x4 (&x)[8], const r_f32x2 (&kp)[8])
{
#define HD_FMA_A(xi, ki) "v_pk_fma_f32 %0, %" #xi ", %" #ki ", %0 op_sel_hi:[1,0,1]\n\t"
#define HD_FMA_B(xi, ki) "v_pk_fma_f32 %1, %" #xi ", %" #ki ", %1 op_sel:[0,1,0]\n\t"
    asm volatile(
        HD_FMA_A(2, 18) HD_FMA_B(3, 18) HD_FMA_A(4, 19) HD_FMA_B(5, 19) HD_FMA_A(6, 20) HD_FMA_B(7, 20) HD_FMA_A(8, 21) HD_FMA_B(9, 21)
        HD_FMA_A(10, 22) HD_FMA_B(11, 22) HD_FMA_A(12, 23) HD_FMA_B(13, 23) HD_FMA_A(14, 24) HD_FMA_B(15, 24) HD_FMA_A(16, 25) "v_pk_fma_f32 %1, %17, %25, %1 op_sel:[0,1,0]"
        : "+v"(a), "+v"(b)
        : "v"(x[0].xy), "v"(x[0].zw), "v"(x[1].xy), "v"(x[1].zw), "v"(x[2].xy), "v"(x[2].zw), "v"(x[3].xy), "v"(x[3].zw),
          "v"(x[4].xy), "v"(x[4].zw), "v"(x[5].xy), "v"(x[5].zw), "v"(x[6].xy), "v"(x[6].zw), "v"(x[7].xy), "v"(x[7].zw),
          "s"(kp[0]), "s"(kp[1]), "s"(kp[2]), "s"(kp[3]), "s"(kp[4]), "s"(kp[5]), "s"(kp[6]), "s"(kp[7]));
#undef HD_FMA_A
#undef HD_FMA_B
}
#endif

// One input sample for EIGHT adjacent outputs (the /4 stage: output q takes this sample with tap t0 - 4 q): eight products, then eight adds, as ONE
// statement -- every sum still receives its products in ascending tap order, separately rounded.  (As separate statements the compiler's block
// scheduler put all of a tile's 1112 products first and spilled them.)  ODD: the taps are the high words of their scalar pairs.
template <bool ODD>
__device__ __forceinline__ void ring_slot8(r_f32x2 (&acc)[8], const r_f32x2 smp, const r_f32x2 k0, const r_f32x2 k1, const r_f32x2 k2, const r_f32x2 k3,
                                           const r_f32x2 k4, const r_f32x2 k5, const r_f32x2 k6, const r_f32x2 k7)
{
#ifdef HD_FAST_ARITH
#define HD_S8_FMA(a, k) "v_pk_fma_f32 %" #a ", %8, %" #k ", %" #a " op_sel_hi:[1,0,1]\n\t"
#define HD_S8_FMAO(a, k) "v_pk_fma_f32 %" #a ", %8, %" #k ", %" #a " op_sel:[0,1,0]\n\t"
#define HD_S8F_OPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])                      \
                   : "v"(smp), "s"(k0), "s"(k1), "s"(k2), "s"(k3), "s"(k4), "s"(k5), "s"(k6), "s"(k7)
    if constexpr (ODD)
        asm volatile(HD_S8_FMAO(0, 9) HD_S8_FMAO(1, 10) HD_S8_FMAO(2, 11) HD_S8_FMAO(3, 12) HD_S8_FMAO(4, 13) HD_S8_FMAO(5, 14) HD_S8_FMAO(6, 15) "v_pk_fma_f32 %7, %8, %16, %7 op_sel:[0,1,0]" HD_S8F_OPS);
    else
        asm volatile(HD_S8_FMA(0, 9) HD_S8_FMA(1, 10) HD_S8_FMA(2, 11) HD_S8_FMA(3, 12) HD_S8_FMA(4, 13) HD_S8_FMA(5, 14) HD_S8_FMA(6, 15) "v_pk_fma_f32 %7, %8, %16, %7 op_sel_hi:[1,0,1]" HD_S8F_OPS);
#undef HD_S8_FMA
#undef HD_S8_FMAO
#undef HD_S8F_OPS
#else
    r_f32x2 t0, t1, t2, t3, t4, t5, t6, t7;
#define HD_S8_MUL(t, k) "v_pk_mul_f32 %" #t ", %16, %" #k " op_sel_hi:[1,0]\n\t"
#define HD_S8_MULO(t, k) "v_pk_mul_f32 %" #t ", %16, %" #k " op_sel:[0,1]\n\t"
#define HD_S8_ADDS "v_pk_add_f32 %0, %0, %8\n\tv_pk_add_f32 %1, %1, %9\n\tv_pk_add_f32 %2, %2, %10\n\tv_pk_add_f32 %3, %3, %11\n\t" \
                   "v_pk_add_f32 %4, %4, %12\n\tv_pk_add_f32 %5, %5, %13\n\tv_pk_add_f32 %6, %6, %14\n\tv_pk_add_f32 %7, %7, %15"
#define HD_S8_OPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),                       \
                    "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)                                            \
                  : "v"(smp), "s"(k0), "s"(k1), "s"(k2), "s"(k3), "s"(k4), "s"(k5), "s"(k6), "s"(k7)
    if constexpr (ODD)
        asm volatile(HD_S8_MULO(8, 17) HD_S8_MULO(9, 18) HD_S8_MULO(10, 19) HD_S8_MULO(11, 20) HD_S8_MULO(12, 21) HD_S8_MULO(13, 22) HD_S8_MULO(14, 23) HD_S8_MULO(15, 24) HD_S8_ADDS HD_S8_OPS);
    else
        asm volatile(HD_S8_MUL(8, 17) HD_S8_MUL(9, 18) HD_S8_MUL(10, 19) HD_S8_MUL(11, 20) HD_S8_MUL(12, 21) HD_S8_MUL(13, 22) HD_S8_MUL(14, 23) HD_S8_MUL(15, 24) HD_S8_ADDS HD_S8_OPS);
#undef HD_S8_MUL
#undef HD_S8_MULO
#undef HD_S8_ADDS
#undef HD_S8_OPS
#endif
}
// ... and for ONE output (the slots at either end of a lane's window, which not every output covers)
template <bool ODD>
__device__ __forceinline__ void ring_slot1(r_f32x2& acc, const r_f32x2 smp, const r_f32x2 k)
{
#ifdef HD_FAST_ARITH
    if constexpr (ODD) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(smp), "s"(k));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(smp), "s"(k));
#else
    r_f32x2 t;
    if constexpr (ODD) asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel:[0,1]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(t) : "v"(smp), "s"(k));
    else asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(t) : "v"(smp), "s"(k));
#endif
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant inside the body
template <class F, int... I>
__device__ __forceinline__ void ring_for_each_index(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

template <int T, int D>
__device__ __forceinline__ void ring_consumer(const RingArgs& a, const RingGeom& geo, RingCtl* __restrict__ ctl, const bool feeder,
                                              const uint32_t role /* the wave's role number in the workgroup (diagnostic builds: its row in the stamp table) */)
{
    constexpr int HR = ring_halo_rows<T>();
    constexpr int JS = HR * 32 - (T - 1);           // slot of tap 0, counted from column 0 of the lane's first row
    constexpr int C0 = JS / 16;                     // first 16-slot chunk that carries taps
    static_assert(D == 16 || D == 8 || D == 4 || D == 2, "a lane's row of 32 samples is 32 / D outputs (the /32 stages run ring_worker)");
    static_assert(T <= 64 || (32 / D) % 8 == 0, "long filters of the small ratios: outputs in groups of eight");
    const uint32_t lane = threadIdx.x & 63u;
    typedef const float __attribute__((address_space(4)))* ctaps_t;
    const ctaps_t taps = (ctaps_t)(uintptr_t)a.taps - JS;                                   // taps[slot]
    auto coff = [](int c) { return (c >> 1) * kRingRowBytes + (c & 1) * 128; };
    RSTAMP_DECL;
#ifdef HD_STAMP_RING
    uint32_t n_done = 0;
#endif
    const uint32_t my_wave = role; (void)my_wave;
    // The feeding consumer also draws the runs for the loader(s) from this XCD's counter (StepClaim, launch.h) -- a returning atomic the
    // compiler counts and waits for, which a wave without DMA in flight can afford -- and keeps two of them ready in ctl->run_q.
    const bool feeding = feeder;
    uint32_t fed = 0;
#ifdef HD_RING_FAULT
    uint32_t starving = 0;                          // 0: not decided, 1: this is the wave that starves its loaders, 2: another one is
#endif
    bool exhausted = false;                         // the XCD's counter has run out: from here on the loaders are fed "no more" sentinels, on demand, for
                                                    // as long as this wave lives -- and it lives until every loader has counted itself out
    const uint32_t xcd0 = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;     // XCC_ID
    const uint32_t xcd = xcd0 < a.claim.n_xcd ? xcd0 : a.claim.n_xcd - 1;
    // The runs of a launch are dealt to the XCDs in equal shares, one counter each, and an XCD's CUs draw from their own share only (the queue entry is the
    // run's number in the whole launch).  Round 4 let a CU whose XCD had run dry go on with the neighbours' counters: bit-identical once the feeder tried one
    // counter per visit, and 1 % SLOWER (NOTES.md) -- the XCDs that end late are the ones whose tails end late, not the ones with tiles left.
    unsigned int* my_ctr = a.claim.ctr + (size_t)xcd * 32;
    unsigned int* next_ctr = a.claim.ctr_next + (size_t)xcd * 32;
    auto feed = [&]() {
        while (feeding) {
            const uint32_t head = __hip_atomic_load(&ctl->run_head, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((int32_t)(fed - head) >= 2) break;  // two unclaimed entries are ready (entries are reused four later: a claimed one has been read long before)
            uint32_t v = 0xFFFFFFFFu;
            if (!exhausted) {
                // (ONE returning agent-scope atomic per visit, a microsecond or two)
#ifdef HD_RING_FAULT   // fault mode 1: ONE feeding wave of the process stops handing out runs after its second -- its loaders starve, their bounded waits run out
                if (fed >= 2u && g_ring_fault_mode == 1u) {
                    if (!starving) { unsigned int won = 1u; if (lane == 0) won = atomicCAS(&g_ring_fault_fired, 0u, 1u); starving = (unsigned int)__builtin_amdgcn_readfirstlane((int)won) == 0u ? 1u : 2u; }
                    if (starving == 1u) break;
                }
#endif
                unsigned int t = 0;
                if (lane == 0) t = __hip_atomic_fetch_add(my_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
                // the first ticket past the end -- exactly one per XCD and launch -- resets the XCD's counter of the other set for the next step launch
                if (t == a.claim.runs_per_xcd && lane == 0) (void)__hip_atomic_exchange(next_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t < a.claim.runs_per_xcd) v = xcd * a.claim.runs_per_xcd + t; else exhausted = true;
            }
            if (lane == 0) {
                ctl->run_q[fed & 3u] = v;
                __hip_atomic_store(&ctl->run_tail, fed + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            ++fed;
        }
    };

    const uint32_t n_slots_all = geo.nb;            // (at most sixteen: RingCtl::slot)
    const uint32_t look_from = (role * 5u) & 15u;   // the waves of a CU start looking at different slots: fewer of them go for the same READY word
    for (;;) {
        RSTAMP(2);
        unsigned long long pw = 0;
        uint32_t slot = 0;
        bool last_look = false;
        for (uint32_t spin = 0;; ++spin) {
            feed();
            // every slot's word at once (lane j reads word j); a READY one is taken by swapping the WHOLE word -- state and descriptor -- for TAKEN: the
            // wave whose swap succeeds holds the descriptor already, a wave that was beaten to it looks again
            unsigned long long w = 0;
            if (lane < n_slots_all) w = __hip_atomic_load(&ctl->slot[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t ready = (uint32_t)__ballot((uint32_t)(w >> 32) == kSlotReady) & 0xFFFFu;
            if (ready) {
                const uint32_t rot = ((ready >> look_from) | (ready << (16u - look_from))) & 0xFFFFu;
                slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)__builtin_ctz(rot) + look_from)) & 15u;
                pw = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), (int)slot) << 32) |
                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, (int)slot);
                uint32_t won = 0;
                if (lane == 0) {
                    unsigned long long expect = pw;
                    won = __hip_atomic_compare_exchange_strong(&ctl->slot[slot], &expect, (unsigned long long)kSlotTaken << 32, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
                }
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)won)) break;
                continue;
            }
            if (last_look) {                        // nothing READY after every loader had counted itself out: done
                RSTAMP(0); RSTAMP_WRITE(my_wave, n_done);
                return;
            }
            if (__hip_atomic_load(&ctl->live, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                last_look = true;                   // no loader will publish any more -- but one may have between my look and its count-down (it publishes, THEN counts itself out)
                continue;
            }
            if (spin > kRingSpinLimit) {            // (never in a correct run: a bounded wait cannot hang the device, and the engine reports it)
                if (lane == 0) __hip_atomic_store(a.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        RSTAMP(0);
        const uint32_t plo = (uint32_t)pw, s = plo >> 12, tile = plo & 0xFFFu;
        const unsigned char* p = geo.ring + slot * (uint32_t)ring_slot_bytes<T>() + lane * (uint32_t)kRingRowBytes;

        {
        // Smaller ratios: the lane's row of 32 samples is OPL = 32 / D adjacent outputs, output q taking the window that starts D * q
        // slots further on -- OPL independent sums (each its own T products in ascending tap order), one pass over the lane's
        // (OPL - 1) * D + T slots, the chunk loop unrolled with compile-time ranges (which taps of which output a 16-slot chunk carries).
        // A short filter (54 taps at /8) sits in scalar registers for the whole launch; a long one (139 taps at /4) is read chunk by chunk
        // through the scalar cache (a chunk carries at most 16 + D * (OPL - 1) different taps).
        constexpr int OPL = 32 / D;
        constexpr int NSW = (OPL - 1) * D + T;          // slots a lane reads: [JS, JS + NSW)
        constexpr int CW1 = (JS + NSW - 1) / 16;        // last chunk
        constexpr bool kResident = T <= 64;
        float k[kResident ? T : 1];
        if constexpr (kResident) {
#pragma unroll
            for (int t = 0; t < T; ++t) k[t] = taps[JS + t];
        }
        r_f32x2 acc[OPL];
#pragma unroll
        for (int q = 0; q < OPL; ++q) acc[q] = (r_f32x2){0.f, 0.f};
        r_f32x4 xw[2][8];
        auto rdw = [&](r_f32x4 (&x)[8], const int c) {
            const unsigned char* pc = p + coff(c);
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) x[q8] = *reinterpret_cast<const r_f32x4*>(pc + 16 * q8);
        };
        rdw(xw[0], C0);
        if constexpr (CW1 > C0) rdw(xw[1], C0 + 1);
        ring_for_each_index([&](auto ci) {
            constexpr int c = C0 + decltype(ci)::value;
            const r_f32x4 (&x)[8] = xw[(c - C0) & 1];
            if constexpr (kResident) {
                HD_FIR_ARITH
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int slot_j = 16 * c + j;
                    const r_f32x2 smp = (j & 1) ? x[j >> 1].zw : x[j >> 1].xy;
#pragma unroll
                    for (int q = 0; q < OPL; ++q) {
                        const int t = slot_j - JS - D * q;
                        if (t >= 0 && t < T) acc[q] = acc[q] + smp * k[t];
                    }
                }
            } else {
                // The chunk's taps as aligned scalar PAIRS (tap 2i, 2i + 1), fetched here -- a chunk touches at most (16 + D (OPL - 1)) / 2 + 1 of them -- and
                // multiplied straight out of the scalar registers: op_sel picks the pair's low or high word for both halves of the packed multiply.
                // (Written as `smp * tap` the compiler broadcasts every tap into a VECTOR register pair first -- 2 T registers, spilled at T = 139.)
                constexpr int TLO = 16 * c - JS - D * (OPL - 1) < 0 ? 0 : 16 * c - JS - D * (OPL - 1);
                constexpr int THI = 16 * c + 15 - JS >= T ? T - 1 : 16 * c + 15 - JS;          // taps [TLO, THI] are used by this chunk
                constexpr int P0 = TLO / 2, NP = THI >= TLO ? THI / 2 - P0 + 1 : 0;
                typedef const r_f32x2 __attribute__((address_space(4)))* cpair_t;
                const cpair_t tp2 = (cpair_t)(uintptr_t)a.taps;                                  // (the tap table is 8-byte aligned and padded to an even count)
                r_f32x2 kp[NP > 0 ? NP : 1];
#pragma unroll
                for (int i = 0; i < NP; ++i) kp[i] = tp2[P0 + i];
                ring_for_each_index([&](auto ji) {
                    constexpr int j = decltype(ji)::value, slot_j = 16 * c + j, t0 = slot_j - JS;      // output q takes this sample with tap t0 - D q
                    const r_f32x2 smp = (j & 1) ? x[j >> 1].zw : x[j >> 1].xy;
                    constexpr bool odd = (t0 & 1) != 0;                                               // (D q is even: one parity for all of the slot's taps)
                    // the lane's outputs in groups of eight (one at /4, two at /2): a group whose eight windows all cover this sample takes the
                    // one-statement form, a group at either end of the lane's span goes output by output
                    ring_for_each_index([&](auto gi) {
                        constexpr int g = decltype(gi)::value, tg = t0 - D * 8 * g;
                        if constexpr (tg - D * 7 >= 0 && tg < T) {
                            ring_slot8<odd>(reinterpret_cast<r_f32x2 (&)[8]>(acc[8 * g]), smp, kp[(tg - 0 * D) / 2 - P0], kp[(tg - 1 * D) / 2 - P0],
                                            kp[(tg - 2 * D) / 2 - P0], kp[(tg - 3 * D) / 2 - P0], kp[(tg - 4 * D) / 2 - P0], kp[(tg - 5 * D) / 2 - P0],
                                            kp[(tg - 6 * D) / 2 - P0], kp[(tg - 7 * D) / 2 - P0]);
                        } else {
                            ring_for_each_index([&](auto qi) {
                                constexpr int q = 8 * g + decltype(qi)::value, t = t0 - D * q;
                                if constexpr (t >= 0 && t < T) ring_slot1<odd>(acc[q], smp, kp[t / 2 - P0]);
                            }, std::make_integer_sequence<int, 8>{});
                        }
                    }, std::make_integer_sequence<int, OPL / 8>{});
                }, std::make_integer_sequence<int, 16>{});
                __builtin_amdgcn_sched_barrier(0);      // (the next chunk's tap pairs are fetched behind this chunk's arithmetic, not in front of the whole loop)
            }
            if constexpr (c + 2 <= CW1) rdw(xw[(c - C0) & 1], c + 2);
        }, std::make_integer_sequence<int, CW1 - C0 + 1>{});
        RSTAMP(1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&ctl->slot[slot], (unsigned long long)kSlotFree << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t o0 = (tile * 64u + lane) * (uint32_t)OPL;                                 // the lane's first output of the stream's chunk
        if (!a.call) {
            float2* dst = a.out + (size_t)s * a.out_stride + o0;                                     // (16-byte aligned: the stride is even)
#pragma unroll
            for (int q = 0; q < OPL; q += 2) *reinterpret_cast<float4*>(dst + q) = make_float4(acc[q].x, acc[q].y, acc[q + 1].x, acc[q + 1].y);
        } else {
            // the final stage of a single-stage plan: behind the pending low-pass input, and the head of the chunk into the spectrum collection
            const uint32_t pend = a.call[s].pend_before, ftake = a.call[s].fft_take, ffill = a.call[s].fft_fill;
            float2* dst = a.out + (size_t)s * a.out_stride + a.fir_hist_cap + pend + o0;
            if (((a.fir_hist_cap + pend) & 1u) == 0u) {
#pragma unroll
                for (int q = 0; q < OPL; q += 2) *reinterpret_cast<float4*>(dst + q) = make_float4(acc[q].x, acc[q].y, acc[q + 1].x, acc[q + 1].y);
            } else {
#pragma unroll
                for (int q = 0; q < OPL; ++q) dst[q] = make_float2(acc[q].x, acc[q].y);
            }
            if (a.fft_in && o0 < ftake) {
                float2* fd = a.fft_in + (size_t)s * kFftBins + ffill + o0;
                if (o0 + (uint32_t)OPL <= ftake && (ffill & 1u) == 0u) {
#pragma unroll
                    for (int q = 0; q < OPL; q += 2) *reinterpret_cast<float4*>(fd + q) = make_float4(acc[q].x, acc[q].y, acc[q + 1].x, acc[q + 1].y);
                } else {
#pragma unroll
                    for (int q = 0; q < OPL; ++q) if (o0 + (uint32_t)q < ftake) fd[q] = make_float2(acc[q].x, acc[q].y);
                }
            }
        }
        if (tile + 1 == a.ntiles) {                 // the stream's last tile: carry the last T-1 inputs (Decimator.h:140-143)
            const float2* in_s = a.in + (size_t)s * a.in_stride;
            float2* hout = a.hist_out + (size_t)s * (T - 1);
            for (uint32_t j = lane; j < (uint32_t)(T - 1); j += 64u) hout[j] = in_s[a.n - (T - 1) + j];
        }
        }
    }
}


// ------------------------------------------------------------------------------------------------- a worker wave (the /32 stages)
// Round 5.  Rounds 3-4 split stage 1 into loader waves and computing waves around shared tile slots.  In-kernel clocks of a step launch said where that
// ends: while the four stream tails of a CU run, its ONE loader wave is busy every cycle of the launch (two LDS round trips and seventeen DMA issues per
// tile in one instruction stream, three tiles in flight at most: its 6-bit vmcnt) and the three computing waves wait for tiles 40 % of their time -- the
// chip moves 2.8 TB/s of stage 1 until the tails are done, whatever the layout (NOTES.md, round 4).  With the systolic tap loop below a wave needs its
// tile's LDS only until its lanes' rows are in registers, so every stage-1 wave can be its own loader: a WORKER owns one 64-row slot and walks runs of
// tiles by itself --
//     wait for my tile (s_waitcnt vmcnt(0)) -> sixteen ds_read_b128: my row -> issue the DMA of my NEXT tile into the same slot -> tap loop -> store
// -- so the next tile is in flight while this one is summed, loads in flight scale with the waves that exist (four workers beside four tails, eight
// once the tails are done: a finished tail wave becomes a worker whose slot is its own LDS slice), and there is NOTHING shared between the waves of a
// CU: no publication words, no slot hand-over, no loader to feed, no wait that is not a hardware counter (the bounded-wait reports and the
// fault-injection build concern the loader / consumer kernels of the smaller ratios only).  Runs of tiles come straight from the XCD's counter
// (StepClaim): the draw for the next run is issued behind the DMA of the current run's last tile and read a tile later, behind the same wait.
//
// The SYSTOLIC tap loop.  Output o of the stream is the sum over the rows o .. o + HR of the history-extended row sequence (its window starts at slot JS
// of row o and ends with slot 0 of row o + HR).  Lane l holds row origin + l of the tile -- ITS 32 samples, all that is read of the tile (a lane that
// walks its output's whole window reads 6.6 x as much) -- and the ACCUMULATORS move: the sum of output origin + o starts in lane o, takes the taps that
// fall on that lane's row, and is handed to lane o + 1 (one DPP rotate per component) for the next row.  Every sum still receives its T products in
// ascending tap order, separately rounded: bit-identical to the lane-owns-the-window loop by construction.  After HR hand-overs lane l >= HR holds
// output origin + l - HR; the sums that wrapped past lane 63 are the next tile's (ring_adv).
// (Row length D: the stage's ratio -- 32 samples at /32, 64 at /64 -- so that output o's window starts in row o.  A row is D / 2 sixteen-byte chunks plus one pad
// chunk; 64 rows are D / 2 + 1 DMA instructions.)
template <int D> constexpr int work_row_bytes() { return D * 8 + 16; }
template <int D> constexpr int work_slot_bytes() { return 64 * work_row_bytes<D>(); }
template <int T, int D> constexpr int work_halo_rows() { return (T - 1 + D - 1) / D; }
static_assert(work_slot_bytes<32>() == kWorkSlotBytes && work_row_bytes<32>() == kRingRowBytes, "the /32 worker slot");

// Eleven LDS-DMA instructions back to back with M0 stepped by 1 KiB in between (a /64 tile's 64 rows are three such blocks); NT = how many of them, from the
// first on, carry the nt policy (the rows the NEXT tile shares stay on the default policy: glds16_x17).
template <int NT>
__device__ __forceinline__ void glds16_x11(const void* base, const uint32_t* off, uint32_t lds_dst)
{
    unsigned keep, scc_keep;
#define HD_GN(n) "global_load_lds_dwordx4 %" #n ", %13 nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define HD_GD(n) "global_load_lds_dwordx4 %" #n ", %13\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define HD_G11_OPS : "=&s"(keep), "=&s"(scc_keep) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "v"(off[8]), "v"(off[9]), "v"(off[10]), "s"(base), "s"(lds_dst) : "memory"
#define HD_G11_HEAD "s_cselect_b32 %1, 1, 0\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %14\n\ts_nop 0\n\t"
#define HD_G11_TAIL "s_mov_b32 m0, %0\n\ts_cmp_lg_u32 %1, 0"
    if constexpr (NT >= 11)
        asm volatile(HD_G11_HEAD HD_GN(2) HD_GN(3) HD_GN(4) HD_GN(5) HD_GN(6) HD_GN(7) HD_GN(8) HD_GN(9) HD_GN(10) HD_GN(11) HD_GN(12) HD_G11_TAIL HD_G11_OPS);
    else
        asm volatile(HD_G11_HEAD HD_GN(2) HD_GN(3) HD_GN(4) HD_GN(5) HD_GN(6) HD_GN(7) HD_GN(8) HD_GD(9) HD_GD(10) HD_GD(11) HD_GD(12) HD_G11_TAIL HD_G11_OPS);
#undef HD_GN
#undef HD_GD
#undef HD_G11_OPS
#undef HD_G11_HEAD
#undef HD_G11_TAIL
}

template <int T, int D = 32>
__device__ __forceinline__ void ring_worker(const RingArgs& a, unsigned char* __restrict__ slot, const uint32_t role /* diagnostic builds: the wave's row in the stamp table */)
{
    static_assert(D == 32 || D == 64, "a row is one output's stride");
    constexpr int HR = work_halo_rows<T, D>();
    constexpr uint32_t ADV = 64u - (uint32_t)HR;    // outputs per tile
    constexpr int JS = HR * D - (T - 1);            // slot of tap 0, counted from column 0 of the lane's first row
    constexpr int NS = JS + T;                      // taps sit on slots [JS, NS)
    constexpr int CPR = D / 2 + 1;                  // sixteen-byte chunks per padded row = DMA instructions per 64 rows
    constexpr int ROWB = work_row_bytes<D>();
    constexpr int NX = D / 2;                       // a lane's row: NX 16-byte reads
    constexpr int UPS = D / 32;                     // units of 32 slots (two 16-tap chunks) per step
    constexpr int NU = (HR + 1) * UPS;              // units in all; the sum moves on to the next lane behind every UPS-th
    constexpr int NB0 = ((int)ADV * CPR + 63) / 64; // DMA instructions for the ADV input rows of a stream's first tile
    constexpr int HQ = D / 32;                      // dword-wide DMA instructions per history row
    static_assert(NS == HR * D + 1, "the window ends with slot 0 of row HR");
    const uint32_t lane = threadIdx.x & 63u;
    typedef const float __attribute__((address_space(4)))* ctaps_t;
    const ctaps_t taps = (ctaps_t)(uintptr_t)a.taps - JS;                                   // taps[slot]
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr_of(slot));      // (wave-uniform: M0 takes it)
    const unsigned char* p = slot + lane * (uint32_t)ROWB;
    const uint32_t rows = a.n / (uint32_t)D;
    RSTAMP_DECL;
    uint32_t n_done = 0; (void)n_done;

    // per-lane source offsets of the CPR DMA instructions of 64 rows (chunk P = 64 i + lane of the slot = column P % CPR of row P / CPR; the last column,
    // the pad, loads the row's last chunk again) and of the HR history rows of a stream's first tile (dword-wide: the history sits at odd 8-byte offsets)
    uint32_t boff[CPR], hist_off[HR * HQ];
#pragma unroll
    for (int i = 0; i < CPR; ++i) { const uint32_t P = 64u * i + lane, row = P / (uint32_t)CPR, col = P - row * (uint32_t)CPR; boff[i] = row * (uint32_t)(D * 8) + (col < (uint32_t)(CPR - 1) ? col : (uint32_t)(CPR - 2)) * 16u; }
#pragma unroll
    for (int r = 0; r < HR * HQ; ++r) {             // instruction r: dwords [64 (r % HQ), +64) of history row r / HQ
        const int h = (r / HQ - HR) * D + (r % HQ) * 32 + (int)(lane >> 1) + (T - 1);      // index into the T-1 history samples (< 0: in front of them, never read)
        hist_off[r] = (uint32_t)(h < 0 ? 0 : h) * 8u + (lane & 1u) * 4u;
    }
    auto issue = [&](const uint32_t s, const uint32_t tile, const bool run_start) {
        const unsigned char* in_s = reinterpret_cast<const unsigned char*>(a.in + (size_t)s * a.in_stride);
        if (tile == 0) {                            // rows 0 .. HR-1: the stage history; rows HR .. 63: the call's first ADV rows
            const unsigned char* hb = reinterpret_cast<const unsigned char*>(a.hist_in + (size_t)s * (T - 1));
#pragma unroll
            for (int r = 0; r < HR * HQ; ++r) glds4(hb, hist_off[r], dst + (uint32_t)((r / HQ) * ROWB + (r % HQ) * 256));
#pragma unroll
            for (int i = 0; i < NB0; ++i)
                if (64 * (i + 1) <= (int)ADV * CPR || lane < (uint32_t)((int)ADV * CPR - 64 * i)) glds16(in_s, boff[i], dst + (uint32_t)(HR * ROWB) + 1024u * i);
        } else {
            const uint32_t origin = tile * ADV < rows - ADV ? tile * ADV : rows - ADV;
            const unsigned char* src = in_s + (size_t)(origin - (uint32_t)HR) * (uint32_t)(D * 8);
            if constexpr (D == 32) { if (run_start) glds16_x17<true>(src, boff, dst); else glds16_x17<false>(src, boff, dst); }
            else { glds16_x11<11>(src, boff, dst); glds16_x11<11>(src, boff + 11, dst + 11u * 1024u); glds16_x11<7>(src, boff + 22, dst + 22u * 1024u); }
        }
    };

    // runs of tiles from this XCD's counter (launch.h: StepClaim; the first ticket past the end resets the counter of the other set for the next launch).
    // A run is run_len consecutive tiles of the slab -- of one stream where run_len divides its tile count, on into the next stream where it does not.
    const uint32_t xcd0 = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;     // XCC_ID
    const uint32_t xcd = xcd0 < a.claim.n_xcd ? xcd0 : a.claim.n_xcd - 1;
    unsigned int* my_ctr = a.claim.ctr + (size_t)xcd * 32;
    unsigned int* next_ctr = a.claim.ctr_next + (size_t)xcd * 32;
    // Guided hand-out (round 6): the XCD's first tickets are runs of run_len tiles (neighbouring tiles share their halo rows through the XCD's L2), its last
    // ones single tiles -- the workers of an XCD then run dry within one tile's time of each other instead of one run's (a launch used to end over 9 us,
    // p10 to p90 of its workgroups, with runs of four to the end).
    const uint32_t runs = a.claim.runs_per_xcd, run_len = a.claim.run_len, short_from = a.claim.short_from < runs ? a.claim.short_from : runs;
    const uint32_t tpx = a.claim.tiles_per_xcd ? a.claim.tiles_per_xcd : runs * run_len;
    auto draw = [&]() -> unsigned int { unsigned int t = 0; if (lane == 0) t = __hip_atomic_fetch_add(my_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
    uint32_t s = 0, tile = 0, left = 0, this_run = 0;   // the run's cursor: (s, tile) is the next tile to issue, `left` of the run's this_run tiles are still to be issued
    auto take_run = [&](const unsigned int ticket) -> bool {          // the ticket's run, or false: the XCD's share is used up
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        if (t == runs && lane == 0) (void)__hip_atomic_exchange(next_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t >= runs) return false;
        const uint32_t g0 = xcd * tpx + (t < short_from ? t * run_len : short_from * run_len + (t - short_from));
        s = g0 / a.ntiles; tile = g0 - s * a.ntiles; left = this_run = t < short_from ? run_len : 1u;
        return true;
    };
    auto advance = [&]() { ++tile; --left; if (tile == a.ntiles) { tile = 0; ++s; } };

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (a former tail wave: every LDS access of the tail has completed before DMA lands in its slice)
    if (!take_run(draw())) { RSTAMP_WRITE(role, 0); return; }
    uint32_t cs = s, ct = tile;                     // the tile in (or on its way into) my slot
    issue(cs, ct, true); advance();
    unsigned int ticket = 0;                        // the draw in flight (lane 0's register) once the run's last tile has been issued
    if (!left) ticket = draw();

    // taps of the sixteen slots [16 c, 16 c + 16) of the lane-relative slot sequence as eight scalar pairs; slots outside [JS, NS) carry no tap
    auto ldk = [&](r_f32x2 (&k)[8], auto cc) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            k[j >> 1].x = (16 * c + j >= JS && 16 * c + j < NS) ? taps[16 * c + j] : 0.f;
            k[j >> 1].y = (16 * c + j + 1 >= JS && 16 * c + j + 1 < NS) ? taps[16 * c + j + 1] : 0.f;
        }
    };
    for (;;) {
        RSTAMP(2);
        // my tile has landed (and my last store is acknowledged, and the ticket -- if one was drawn -- has arrived): the wave's only wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(ticket));            // (the compiler's scoreboard learns here, where it costs nothing, that the draw has returned -- or it waits for it
                                                    // where the register is written next: behind the DMA of a run's last tile)
        RSTAMP(0);
        r_f32x2 kk[2][2][8];                        // [unit parity][chunk of the unit][pair]
        r_f32x4 x[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) x[q] = *reinterpret_cast<const r_f32x4*>(p + 16 * q);
        ldk(kk[0][0], std::integral_constant<int, 0>{}); ldk(kk[0][1], std::integral_constant<int, 1>{});   // (unit 0's taps travel with the row)
        __builtin_amdgcn_sched_barrier(0);
        // the rows are in registers: the slot can take the next tile at once -- its DMA flies while this tile is summed
#pragma unroll
        for (int q = 0; q < NX; q += 8)
            asm volatile("" : "+v"(x[q]), "+v"(x[q + 1]), "+v"(x[q + 2]), "+v"(x[q + 3]), "+v"(x[q + 4]), "+v"(x[q + 5]), "+v"(x[q + 6]), "+v"(x[q + 7]) :: "memory");
        RSTAMP(3);
        bool more = true;
        if (!left) more = take_run(ticket);         // (drawn behind the DMA of the run's last tile: it arrived with that tile)
        uint32_t ns = 0, nt = 0;
        if (more) {
            ns = s; nt = tile;
            issue(ns, nt, left == this_run); advance();
            if (!left) ticket = draw();
        }
        RSTAMP(4);

        const uint32_t origin = ct * ADV < rows - ADV ? ct * ADV : rows - ADV;
        r_f32x2 acc = {0.f, 0.f};
#ifdef HD_FAST_ARITH
        r_f32x2 acc_b = {0.f, 0.f};                     // fast mode: the odd slots' chain (ring_fma16x2); it travels with `acc` and joins it at the end
#endif
        // chunk c of the slot sequence = chunk c % (D / 16) of the row of step c / (D / 16): all sixteen taps (the hand-scheduled form), or the few that exist
        auto mac_chunk = [&](const r_f32x2 (&k)[8], auto cc) {
            constexpr int c = decltype(cc)::value, j0 = JS > 16 * c ? JS - 16 * c : 0, j1 = NS < 16 * c + 16 ? NS - 16 * c : 16;
            const r_f32x4 (&xh)[8] = reinterpret_cast<const r_f32x4 (&)[8]>(x[8 * (c % (D / 16))]);
#ifdef HD_FAST_ARITH
            if constexpr (j0 == 0 && j1 == 16) ring_fma16x2(acc, acc_b, xh, k);
            else {
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    const r_f32x2 smp = (j & 1) ? xh[j >> 1].zw : xh[j >> 1].xy;
                    const float kj = (j & 1) ? k[j >> 1].y : k[j >> 1].x;
                    if (j & 1) acc_b = __builtin_elementwise_fma(smp, (r_f32x2){kj, kj}, acc_b);
                    else acc = __builtin_elementwise_fma(smp, (r_f32x2){kj, kj}, acc);
                }
            }
#else
            if constexpr (j0 == 0 && j1 == 16) ring_mac16_asm(acc, xh, k);
            else {
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    const r_f32x2 smp = (j & 1) ? xh[j >> 1].zw : xh[j >> 1].xy;
                    acc = acc + smp * ((j & 1) ? k[j >> 1].y : k[j >> 1].x);
                }
            }
#endif
        };
        auto rot1 = [&](r_f32x2& v) {                   // lane l takes lane l - 1's sum (wave_ror:1)
            // (through scalars: __builtin_bit_cast of an ext-vector ELEMENT reads the vector's first element whichever was named -- clang 20)
            const float re = v.x, im = v.y;
            v.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, re), 0x13C, 0xF, 0xF, false));
            v.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, im), 0x13C, 0xF, 0xF, false));
        };
        auto rot = [&]() {
            rot1(acc);
#ifdef HD_FAST_ARITH
            rot1(acc_b);
#endif
        };
        // A unit's taps (32 slots) are requested a unit ahead through the scalar cache.  Scalar loads return out of order, so waiting for any of them waits for
        // all that are outstanding: the wait for unit u's taps (forced by `arrived`) therefore sits IN FRONT of the request for unit u + 1's, which then has
        // the whole of unit u's arithmetic to come back; the scheduling barrier keeps the compiler from sinking the request to its use.
        auto arrived = [&](const r_f32x2 (&k)[2][8]) {
            asm volatile("" :: "s"(k[0][0]), "s"(k[0][1]), "s"(k[0][2]), "s"(k[0][3]), "s"(k[0][4]), "s"(k[0][5]), "s"(k[0][6]), "s"(k[0][7]),
                               "s"(k[1][0]), "s"(k[1][1]), "s"(k[1][2]), "s"(k[1][3]), "s"(k[1][4]), "s"(k[1][5]), "s"(k[1][6]), "s"(k[1][7]));
        };
        ring_for_each_index([&](auto ui) {
            constexpr int u = decltype(ui)::value;      // unit u: slots [32 u, 32 u + 32) of the slot sequence = part u % UPS of the row of step u / UPS
            arrived(kk[u & 1]);
            if constexpr (u + 1 < NU) {
                ldk(kk[(u + 1) & 1][0], std::integral_constant<int, 2 * u + 2>{});
                ldk(kk[(u + 1) & 1][1], std::integral_constant<int, 2 * u + 3>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (JS < 32 * u + 16 && 32 * u < NS) mac_chunk(kk[u & 1][0], std::integral_constant<int, 2 * u>{});
            if constexpr (32 * u + 16 < NS && JS < 32 * u + 32) mac_chunk(kk[u & 1][1], std::integral_constant<int, 2 * u + 1>{});
            if constexpr ((u + 1) % UPS == 0 && u + 1 < NU) rot();
        }, std::make_integer_sequence<int, NU>{});
#ifdef HD_FAST_ARITH
        acc = acc + acc_b;
#endif
#ifdef HD_STAMP_RING
        asm volatile("" : "+v"(acc));
        ++n_done;
#endif
        RSTAMP(1);
        if (lane >= (uint32_t)HR) a.out[(size_t)cs * a.out_stride + origin + lane - (uint32_t)HR] = make_float2(acc.x, acc.y);
        if (ct + 1 == a.ntiles) {                   // the stream's last tile: carry the last T-1 inputs (Decimator.h:140-143)
            // (all loads, then all stores: ONE wait -- which the DMA in flight has to see out anyway)
            const float2* in_s = a.in + (size_t)cs * a.in_stride + (a.n - (uint32_t)(T - 1));
            float2* hout = a.hist_out + (size_t)cs * (T - 1);
            constexpr int NH = (T - 1 + 63) / 64;
            float2 h[NH];
#pragma unroll
            for (int k = 0; k < NH; ++k) h[k] = in_s[lane + 64u * k < (uint32_t)(T - 1) ? lane + 64u * k : 0u];
#pragma unroll
            for (int k = 0; k < NH; ++k) if (lane + 64u * k < (uint32_t)(T - 1)) hout[lane + 64u * k] = h[k];
        }
        if (!more) break;
        cs = ns; ct = nt;
    }
    RSTAMP(2);
    RSTAMP_WRITE(role, n_done);
}

}  // namespace HD_ARITH_NS
}  // namespace hd
