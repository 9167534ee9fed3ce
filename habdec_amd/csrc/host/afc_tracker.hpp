// Automatic-frequency-control state machine (host side, a few scalar operations per stream per call).
//
// The GPU reduces each 4096-bin spectrum to `SpectrumStats` (mean / sigma of the dB power array, the
// strongest bin and the strongest second bin about one FSK shift away); this class is the sequential
// part of the reference's AFC::process (code/Decoder/AFC.h:108-184): running averages
// (code/Decoder/Average.h:39-70) of the noise statistics and of the two peak positions, the detection
// threshold, the "stable peak" test and the latched frequency correction.  It advances once per
// process() call, also on calls that did not produce a new spectrum (Decoder.h:505-507).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace hd {

struct SpectrumStats {          // produced by the spectrum kernel, one per stream
    int32_t valid;              // 0 when the spectrum or its power held NaN/Inf (AFC.h:250-283)
    int32_t peak1, peak2;       // ordered so peak1 < peak2 (AFC.h:321-325)
    float power1, power2;       // dB power at those bins
    uint32_t seq;               // (device side: the call's tag, see dev_types.h SpectrumStatsDev)
    double mean, sigma;         // AFC.h:103-104
};

template <typename T>
class WindowMean {              // Average<T>: plain mean until `cap` samples, then an exponential blend
public:
    explicit WindowMean(size_t cap) : cap_(std::max<size_t>(1, cap)) { push(T(0)); }   // the initial 0 counts
    double value() const { return n_ ? double(acc_) / n_ : double(acc_); }
    double push(T v)
    {
        const double before = value() - v;
        if (n_ == cap_) acc_ = T(value() * (cap_ - 1) + v);
        else { ++n_; acc_ += v; }
        return before;
    }
    void restart(T v) { acc_ = v; n_ = 1; }
private:
    T acc_ = 0;
    size_t n_ = 0, cap_;
};

class AfcTracker {
public:
    double correction = 0, shift_hz = 0, noise_floor = 0, noise_sigma = 0;
    int gui_left = 0, gui_right = 0;

    // `bins`/`rate` describe the spectrum the stats came from; `have` = a spectrum exists at all.
    void step(bool have, const SpectrumStats& st, size_t bins, double rate)
    {
        if (!have || !st.valid) { correction = 0; return; }
        noise_floor = st.mean;
        noise_sigma = st.sigma;
        floor_avg_.push(st.mean);
        sigma_avg_.push(st.sigma);
        int p1 = st.peak1, p2 = st.peak2;
        const float threshold = float(floor_avg_.value() + 3 * std::fabs(sigma_avg_.value()));
        const bool seen1 = st.power1 > threshold, seen2 = st.power2 > threshold;
        bool steady_l = false, steady_r = false;
        if (seen1 && seen2) {
            if (p2 < p1) std::swap(p1, p2);
            steady_l = left_.push(p1) <= 2;        // signed difference, as in the reference
            steady_r = right_.push(p2) <= 2;
        }
        gui_left = seen1 ? (steady_l ? int(left_.value()) : int(-left_.value())) : 0;
        gui_right = seen2 ? (steady_r ? int(right_.value()) : int(-right_.value())) : 0;
        if (steady_l && steady_r) {
            const int l = int(std::round(left_.value())), r = int(std::round(right_.value()));
            const int gap = r - l;
            const double hz_per_bin = rate / double(bins);
            shift_hz = hz_per_bin * gap;
            const double centre = l + gap / 2;
            const double off_bins = centre - double(bins) / 2;
            if (std::abs(off_bins) > 4) correction = hz_per_bin * off_bins;
        }
    }

    void reset(double applied_hz, size_t bins, double rate)     // AFC.h:187-194
    {
        const double bins_per_hz = double(bins) / rate;
        left_.restart(int(std::max(0.0, left_.value() - applied_hz * bins_per_hz)));
        right_.restart(int(std::max(0.0, right_.value() - applied_hz * bins_per_hz)));
        correction = 0;
    }

private:
    WindowMean<double> floor_avg_{100}, sigma_avg_{100};
    WindowMean<int> left_{4}, right_{4};
};

}  // namespace hd
