// Decimation stage plan and coefficient tables for the MI355X engine (host side).
//
// habdec picks one or two FIR-decimate stages for a power-of-two total factor
// (reference code/Decoder/Decoder.h:286-320); the coefficient tables are the GQRX designs tabulated in
// code/Decoder/filtercoef.h:27-1449, kept here as float32 bit patterns (decim_taps.inc, generated).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace hd {

#include "decim_taps.inc"

struct DecimStage {
    int ratio = 0;                 // samples in per sample out
    std::vector<float> taps;       // T coefficients, applied oldest-sample-first
};

inline std::vector<float> decim_table(int total_name, int ratio)
{
    const uint32_t* bits = nullptr;
    unsigned n = 0;
#define HD_PICK(N, R) if (total_name == N && ratio == R) { bits = kDecimTapBits_##N##_##R; n = kDecimTapBits_##N##_##R##_n; }
    HD_PICK(2, 2) HD_PICK(4, 4) HD_PICK(8, 8) HD_PICK(16, 8) HD_PICK(32, 16) HD_PICK(64, 32) HD_PICK(128, 32) HD_PICK(256, 64)
#undef HD_PICK
    std::vector<float> t(n);
    if (n) std::memcpy(t.data(), bits, n * sizeof(float));
    return t;
}

// Returns false for factors habdec rejects (anything but 1,2,4,...,256).  Factor 1 = no stages.
inline bool decim_plan(unsigned total, std::vector<DecimStage>& out)
{
    struct Row { unsigned total; int r0, n0, r1, n1; };
    static const Row rows[] = {
        {256, 64, 256, 4, 4}, {128, 32, 128, 4, 4}, {64, 32, 64, 2, 2}, {32, 16, 32, 2, 2},
        {16, 8, 16, 2, 2},    {8, 8, 8, 0, 0},      {4, 4, 4, 0, 0},    {2, 2, 2, 0, 0},
    };
    out.clear();
    if (total == 1) return true;
    for (const Row& r : rows) {
        if (r.total != total) continue;
        out.push_back({r.r0, decim_table(r.n0, r.r0)});
        if (r.r1) out.push_back({r.r1, decim_table(r.n1, r.r1)});
        return true;
    }
    return false;
}

}  // namespace hd
