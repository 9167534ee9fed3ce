// SpectrumInfo<T> -- power spectrum values plus the peak / noise metadata the GUIs draw.  Same public members as the
// reference's code/Decoder/SpectrumInfo.h:35-85 (the websocket layer reads them by name).
#pragma once
#include <vector>

namespace habdec {

template <typename T>
class SpectrumInfo : public std::vector<T> {
public:
    typedef T TValue;

    mutable T min_ = 0, max_ = 0;
    mutable double noise_floor_ = 0, noise_variance_ = 0, sampling_rate_ = 0, shift_ = 0;
    mutable int peak_left_ = 0, peak_right_ = 0;
    mutable bool peak_left_valid_ = false, peak_right_valid_ = false;

    SpectrumInfo() = default;

    template <typename U>
    SpectrumInfo(const SpectrumInfo<U>& o)
        : std::vector<T>(o.begin(), o.end()), noise_floor_(o.noise_floor_), noise_variance_(o.noise_variance_),
          peak_left_(o.peak_left_), peak_right_(o.peak_right_), peak_left_valid_(o.peak_left_valid_), peak_right_valid_(o.peak_right_valid_)
    {
    }
    template <typename U>
    const SpectrumInfo& operator=(const SpectrumInfo<U>& o)
    {
        this->assign(o.begin(), o.end());
        noise_floor_ = o.noise_floor_; noise_variance_ = o.noise_variance_;
        peak_left_ = o.peak_left_; peak_right_ = o.peak_right_;
        peak_left_valid_ = o.peak_left_valid_; peak_right_valid_ = o.peak_right_valid_;
        return *this;
    }
    const SpectrumInfo& operator=(const std::vector<T>& values)
    {
        std::vector<T>::operator=(values);
        return *this;
    }
};

}  // namespace habdec
