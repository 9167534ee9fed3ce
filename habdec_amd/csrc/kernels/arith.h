// The arithmetic mode of the FIR sums (hd_engine_config.arith), chosen per translation unit: every kernel file that carries a FIR of the chain --
// decimate.hip (+ stage1_ring.h, tail_body.h), fir_demod.hip, backend.hip, tail.hip -- is compiled twice (habdec_amd/build.py), and what it defines lives
// in the mode's namespace: hd::exact or hd::fast.
//
//   exact (default)  out = (...((x[0] k[0]) + x[1] k[1]) + ...): every product and every sum rounded separately, ascending tap order, one accumulator --
//                    the reference's CPU arithmetic (code/Decoder/Decimator.h:128-138, FirFilter.h:155-161 built without contraction), bit for bit.
//                    The whole library is compiled with -ffp-contract=off; the hand-scheduled blocks are v_pk_mul_f32 + v_pk_add_f32.
//   fast             acc = fma(x[t], k[t], acc) (v_pk_fma_f32): half the vector instructions of every FIR, and where a lone chain would wait out the
//                    adder, two accumulators over alternating taps, added at the end.  north_star's tolerance for intermediate floats is 1e-5 relative;
//                    measured against the exact mode: <= 3e-7 norm-wise (tests/test_gpu_fast.py).  Everything behind the FIRs is the exact mode's code:
//                    the discriminator (exact_math.h) and the symbol extractor produce the decisions, and those are tested for identity with the reference's, as they are.
//
// Why not fold the symmetric taps -- (x[t] + x[T-1-t]) k[t], all eight decimator tables (filtercoef.h:27-1449) and the Blackman-Harris design
// (FirFilter.h:198-208) are symmetric --: with a packed FMA a folded pair costs one v_pk_add_f32 and one v_pk_fma_f32, the same two instructions as the two
// v_pk_fma_f32 of the unfolded pair, and it needs the two halves of a window at once -- in the systolic stage 1 (stage1_ring.h) they sit in different
// lanes, in the lane-owns-a-window loops it doubles the LDS read streams.  Folding halves the MULTIPLIES, not the instructions; the instruction count is
// what the fused form already halves.
#pragma once

#ifdef HD_FAST_ARITH
#define HD_ARITH_NS fast
#define HD_FIR_ARITH _Pragma("clang fp contract(fast)")     /* at the head of a block: a * b + c inside it may fuse (the rest of the file stays contract-off) */
#define HD_DBG_NAME(name) name##_fast                       /* extern "C" diagnostics of the fast-mode copy of a translation unit */
#else
#define HD_ARITH_NS exact
#define HD_FIR_ARITH
#define HD_DBG_NAME(name) name
#endif

namespace hd {
namespace HD_ARITH_NS {
#ifdef HD_FAST_ARITH
constexpr bool kFastArith = true;
#else
constexpr bool kFastArith = false;
#endif
// Fast mode also takes the symbol extractor's window sums (SymbolExtractor.h:162-224: std::accumulate over R samples per candidate position) by sliding from an
// exactly summed anchor (sym_common.h: window_sums_slide); -DHD_FAST_EXACT_WINDOWS keeps them in the exact mode's order (A/B builds).
#ifdef HD_FAST_EXACT_WINDOWS
constexpr bool kFastWindows = false;
#else
constexpr bool kFastWindows = true;
#endif
}  // namespace HD_ARITH_NS
}  // namespace hd
