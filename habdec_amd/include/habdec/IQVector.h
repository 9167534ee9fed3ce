// IQVector<T> -- the argument type of Decoder<T>::pushSamples: a std::vector of complex samples tagged with its
// sampling rate.  Same public surface as the reference's code/Decoder/IQVector.h:33-48 (data(), size(), resize(),
// samplingRate() getter/setter, copy keeps the rate) so code written against habdec compiles unchanged.
#pragma once
#include <atomic>
#include <complex>
#include <ostream>
#include <vector>

namespace habdec {

template <typename T>
class IQVector : public std::vector<std::complex<T>> {
    using Base = std::vector<std::complex<T>>;

public:
    typedef T TValue;
    typedef std::complex<T> TComplex;

    IQVector() = default;
    IQVector(const IQVector& o) : Base(o) { rate_.store(o.rate_.load()); }
    IQVector& operator=(const IQVector& o)
    {
        Base::operator=(o);
        rate_.store(o.rate_.load());
        return *this;
    }

    double samplingRate() const { return rate_.load(); }
    void samplingRate(double r) { rate_.store(r); }

private:
    std::atomic<double> rate_{0.0};
};

// the samples as "(re,im) (re,im) ..." (IQVector.h:67-73)
template <typename T>
std::ostream& operator<<(std::ostream& os, const IQVector<T>& v)
{
    for (const auto& z : v) os << z << " ";
    return os;
}

}  // namespace habdec
