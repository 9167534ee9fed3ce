// Windowed-sinc low-pass design used in front of the FSK discriminator (host side).
//
// Same numbers as the reference's FirFilter::LP_BlackmanHarris (code/Decoder/FirFilter.h:173-209) with
// its sinc / 4-term Blackman-Harris helpers (code/Decoder/habdec_windows.h:27-53): tap count
// size_t(4/transition) clamped to the batch length and forced odd; taps normalised by a double sum.
// The design runs on the host with the host libm, exactly where the reference runs it.
//
// `float_trig`: the reference calls sin/cos unqualified from a global template, so the overload depends
// on its translation unit (DESIGN.md "lookup context"): true = float sinf/cosf everywhere, false = the
// double functions with the window summed in double and rounded once.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace hd {

struct LowpassDesigner {
    bool float_trig = true;
    std::vector<float> taps;        // current design (empty until the first successful design)
    size_t batch = 0;               // input length of the most recent filter call (0 = never fed)

    // Returns true when `taps` changed.
    bool design(float cutoff_rel, float transition)
    {
        if (!batch) return false;                                   // "No Input set."
        const float tw = transition ? transition : cutoff_rel * cutoff_rel;
        size_t n = static_cast<size_t>(4.0f / tw);
        if (n > batch) n = batch;
        n |= 1;
        if (n <= 4 || n == taps.size()) return false;               // keeps the old design, even for a new cutoff
        taps.assign(n, 0.0f);
        const int centre = static_cast<int>(n / 2);
        double norm = 0.0;
        for (int i = 0; i < static_cast<int>(n); ++i) {
            const float arg = 2.0f * cutoff_rel * (i - centre);
            taps[i] = sinc(arg) * window(static_cast<size_t>(i), n);
            norm += taps[i];
        }
        for (float& t : taps) t /= norm;
        return true;
    }

private:
    float sinc(float x) const
    {
        if (!x) return 1.0f;
        return float_trig ? sinf(x) / x : static_cast<float>(std::sin(static_cast<double>(x)) / static_cast<double>(x));
    }
    float window(size_t i, size_t n) const
    {
        static const float c0 = 0.35874, c1 = 0.48829, c2 = 0.14128, c3 = 0.01168;
        static const float w2 = 2.0 * M_PI, w4 = 4.0 * M_PI, w6 = 6.0 * M_PI;
        const float span = n - 1;
        const float p2 = w2 * i / span, p4 = w4 * i / span, p6 = w6 * i / span;
        if (float_trig) return c0 - c1 * cosf(p2) + c2 * cosf(p4) - c3 * cosf(p6);
        const double w = c0 - c1 * std::cos(static_cast<double>(p2)) + c2 * std::cos(static_cast<double>(p4)) - c3 * std::cos(static_cast<double>(p6));
        return static_cast<float>(w);
    }
};

}  // namespace hd
