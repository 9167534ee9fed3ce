// GUI data path: the websocket server's binary payloads for the spectrum and the demodulated trace,
// built from what the engine's getters return -- zoom to the centre bins and thin to the requested resolution
// (websocketServer/habdec_ws_protocol.cpp:338-351, 355-405), quantise to 8 / 16 / 32 bits against the vector's own min/max
// (CompressedVector.h:50-62, CompressedVector.cpp:60-118), prepend the 52-byte / 20-byte header (NetTransport.h:29-102).
// A monitoring front end for thousands of streams needs exactly these bytes per stream; the webClient reads them unchanged.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <limits>
#include <vector>

namespace hd {
namespace gui {

struct SpectrumMeta {           // what SpectrumInfo<float> carries besides the bins (SpectrumInfo.h:35-56)
    double noise_floor = 0, noise_variance = 0, sampling_rate = 0, shift = 0;
    int peak_left = 0, peak_right = 0;
    bool peak_left_valid = false, peak_right_valid = false;
};

// ShrinkVector (habdec_ws_protocol.cpp:338-351): keep new_size samples, sample i taken from index float(i)/new_size * size
inline void shrink(std::vector<float>& v, size_t new_size)
{
    if (new_size >= v.size()) return;
    for (size_t i = 0; i < new_size; ++i) {
        const float f = float(i) / new_size;
        const size_t I = f * v.size();
        v[i] = v[I];
    }
    v.resize(new_size);
}

// CompressedVector<T>(std::vector<float>): min/max of the vector, then per value  float(v - min) / (max - min) * T_max,
// truncated (CompressedVector.cpp:75-118); T = float copies the values.
inline void quantise(const std::vector<float>& v, int type_size, double& vmin, double& vmax, std::vector<uint8_t>& out)
{
    vmin = *std::min_element(v.begin(), v.end());
    vmax = *std::max_element(v.begin(), v.end());
    out.clear();
    if (type_size == 4) {
        out.resize(v.size() * 4);
        std::memcpy(out.data(), v.data(), out.size());
        return;
    }
    out.reserve(v.size() * (size_t)type_size);
    for (float x : v) {
        x = float(x - vmin) / (vmax - vmin);
        if (type_size == 1) {
            const unsigned char q = x * std::numeric_limits<unsigned char>::max();
            out.push_back(q);
        } else {
            const uint16_t q = x * std::numeric_limits<uint16_t>::max();
            out.push_back((uint8_t)(q & 0xFF));
            out.push_back((uint8_t)(q >> 8));
        }
    }
}

inline void put32(std::vector<uint8_t>& o, const void* p) { const uint8_t* b = static_cast<const uint8_t*>(p); o.insert(o.end(), b, b + 4); }
inline void put_i(std::vector<uint8_t>& o, int32_t v) { put32(o, &v); }
inline void put_f(std::vector<uint8_t>& o, float v) { put32(o, &v); }

// SpectrumToStream (habdec_ws_protocol.cpp:355-405) + SerializeSpectrum (NetTransport.h:61-85).  type_size: 1, 2 or 4.
// Returns the number of bins sent (0: nothing to send).
inline size_t spectrum_payload(std::vector<float> bins, SpectrumMeta m, float zoom, int resolution, int type_size, std::vector<uint8_t>& out)
{
    out.clear();
    if (bins.empty() || (type_size != 1 && type_size != 2 && type_size != 4)) return 0;
    zoom = std::min(std::max(zoom, 0.01f), 0.99f);
    const size_t begin = zoom / 2 * bins.size();
    const size_t end = (1.0f - zoom / 2) * bins.size();
    bins.erase(bins.begin() + end, bins.end());
    bins.erase(bins.begin(), bins.begin() + begin);
    m.peak_left -= (int)begin;
    if (m.peak_left < 0 || (size_t)m.peak_left > bins.size()) { m.peak_left = 0; m.peak_left_valid = false; }
    m.peak_right -= (int)begin;
    if (m.peak_right < 0 || (size_t)m.peak_right > bins.size()) { m.peak_right = 0; m.peak_right_valid = false; }
    if (resolution >= 0 && (size_t)resolution < bins.size()) {
        m.peak_left = double(m.peak_left) * resolution / bins.size();
        m.peak_right = double(m.peak_right) * resolution / bins.size();
        shrink(bins, (size_t)resolution);
    }
    if (bins.empty()) return 0;
    double vmin, vmax;
    std::vector<uint8_t> vals;
    quantise(bins, type_size, vmin, vmax, vals);
    put_i(out, 52);                                       // SpectrumInfoHeader (NetTransport.h:29-47)
    put_f(out, (float)m.noise_floor); put_f(out, (float)m.noise_variance); put_f(out, (float)m.sampling_rate); put_f(out, (float)m.shift);
    put_i(out, m.peak_left); put_i(out, m.peak_right); put_i(out, m.peak_left_valid ? 1 : 0); put_i(out, m.peak_right_valid ? 1 : 0);
    put_f(out, (float)vmin); put_f(out, (float)vmax);
    put_i(out, type_size); put_i(out, (int32_t)bins.size());
    out.insert(out.end(), vals.begin(), vals.end());
    return bins.size();
}

// DemodToStream (habdec_ws_protocol.cpp:408-429) + SerializeDemodulation (NetTransport.h:88-102)
inline size_t demod_payload(std::vector<float> trace, int resolution, int type_size, std::vector<uint8_t>& out)
{
    out.clear();
    if (trace.empty() || (type_size != 1 && type_size != 2 && type_size != 4)) return 0;
    if (resolution >= 0) shrink(trace, (size_t)resolution);
    if (trace.empty()) return 0;
    double vmin, vmax;
    std::vector<uint8_t> vals;
    quantise(trace, type_size, vmin, vmax, vals);
    put_i(out, 20);                                       // DemodHeader (NetTransport.h:50-57)
    put_f(out, (float)vmin); put_f(out, (float)vmax);
    put_i(out, type_size); put_i(out, (int32_t)trace.size());
    out.insert(out.end(), vals.begin(), vals.end());
    return trace.size();
}

}  // namespace gui
}  // namespace hd
