// TEST INFRASTRUCTURE ONLY: the reference's own payload writers -- SerializeSpectrum / SerializeDemodulation
// (websocketServer/NetTransport.h:61-102) over CompressedVector (websocketServer/CompressedVector.{h,cpp}) and SpectrumInfo
// (Decoder/SpectrumInfo.h) -- compiled from where they lie, behind a C ABI.  The zoom / thinning steps in front of them live
// in websocketServer/habdec_ws_protocol.cpp:338-405, a file that needs boost to compile, so THAT part is restated here
// (ref_shrink, the zoom arithmetic) and is pinned only by reading: "parity unpinned" for zoom/ShrinkVector, pinned for the
// quantisation and the headers.
#include <cstring>
#include <sstream>
#include <vector>

#include "Decoder/SpectrumInfo.h"
#include "websocketServer/NetTransport.h"

namespace {
template <typename T>
void ref_shrink(T& vec, size_t new_size)            // habdec_ws_protocol.cpp:338-351
{
    if (new_size >= vec.size()) return;
    for (size_t i = 0; i < new_size; ++i) {
        float i_0_1 = float(i) / new_size;
        size_t I = i_0_1 * vec.size();
        vec[i] = vec[I];
    }
    vec.resize(new_size);
}
size_t emit(const std::stringstream& ss, unsigned char* out, size_t cap)
{
    const std::string s = ss.str();
    if (out && cap >= s.size()) std::memcpy(out, s.data(), s.size());
    return s.size();
}
}  // namespace

extern "C" {

size_t ref_spectrum_payload(const float* bins, size_t n, double noise_floor, double noise_variance, double sampling_rate, double shift,
                            int peak_left, int peak_right, float zoom, int resolution, int type_size, unsigned char* out, size_t cap,
                            size_t* bins_sent)
{
    using namespace std;
    habdec::SpectrumInfo<float> si;
    si = std::vector<float>(bins, bins + n);
    *bins_sent = 0;
    if (!si.size()) return 0;
    si.noise_floor_ = noise_floor; si.noise_variance_ = noise_variance; si.sampling_rate_ = sampling_rate; si.shift_ = shift;
    si.peak_left_ = std::abs(peak_left); si.peak_left_valid_ = peak_left > 0;          // Decoder.h:823-828
    si.peak_right_ = std::abs(peak_right); si.peak_right_valid_ = peak_right > 0;
    // habdec_ws_protocol.cpp:365-393
    zoom = min(max(zoom, 0.01f), 0.99f);
    const size_t zb = zoom / 2 * si.size();
    const size_t ze = (1.0f - zoom / 2) * si.size();
    si.erase(si.begin() + ze, si.end());
    si.erase(si.begin(), si.begin() + zb);
    si.peak_left_ -= zb;
    if (si.peak_left_ < 0 || si.peak_left_ > si.size()) { si.peak_left_ = 0; si.peak_left_valid_ = false; }
    si.peak_right_ -= zb;
    if (si.peak_right_ < 0 || si.peak_right_ > si.size()) { si.peak_right_ = 0; si.peak_right_valid_ = false; }
    if (resolution < si.size()) {
        si.peak_left_ = double(si.peak_left_) * resolution / si.size();
        si.peak_right_ = double(si.peak_right_) * resolution / si.size();
        ref_shrink(si, resolution);
    }
    if (!si.size()) return 0;
    std::stringstream ss;
    if (type_size == 1) SerializeSpectrum(si, ss, (unsigned char*)0);
    if (type_size == 2) SerializeSpectrum(si, ss, (unsigned short int*)0);
    if (type_size == 4) SerializeSpectrum(si, ss, (float*)0);
    *bins_sent = si.size();
    return emit(ss, out, cap);
}

size_t ref_demod_payload(const float* trace, size_t n, int resolution, int type_size, unsigned char* out, size_t cap, size_t* sent)
{
    std::vector<float> v(trace, trace + n);
    *sent = 0;
    if (!v.size()) return 0;
    ref_shrink(v, resolution);
    if (!v.size()) return 0;
    std::stringstream ss;
    if (type_size == 1) SerializeDemodulation(v, ss, (unsigned char*)0);
    if (type_size == 2) SerializeDemodulation(v, ss, (unsigned short int*)0);
    if (type_size == 4) SerializeDemodulation(v, ss, (float*)0);
    *sent = v.size();
    return emit(ss, out, cap);
}

}
