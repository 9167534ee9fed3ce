// Bit-exact device replacements for the two libm calls on the sample path.
//
// The reference discriminator is std::arg(x[i] * conj(x[i-1])) = atan2f(im, re) from the host libm
// (code/Decoder/FSK2_Demod.h:37-40).  glibc 2.35's atan2f/atanf are the classic fdlibm single-precision
// algorithms evaluated in plain float arithmetic (argument reduction to one of four intervals, an odd/even
// split degree-11 polynomial, hi/lo constants).  Restated below with the same operation order and compiled
// with -ffp-contract=off, every result is bit-identical to the host libm (checked on 4e8 inputs incl.
// NaN/Inf/denormal bit patterns in tests/test_host_logic.py::test_atan2_restatement_matches_libm and on the
// GPU in tests/test_gpu_parity.py).  That makes the demodulated floats, and therefore every flip point,
// bit and character downstream, identical to the CPU path by construction instead of "almost always".
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define HD_HD __host__ __device__ __forceinline__
#else
#define HD_HD inline
#endif

namespace hd {

HD_HD uint32_t f32_bits(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
HD_HD float bits_f32(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }

// Written branch-light on purpose: on the GPU a divergent branch executes both sides, so the four argument reductions select their
// numerator and denominator first and share ONE division (the same IEEE division of the same operands as in the branchy original),
// and atan2f reaches its single atanf body from both of the places the original calls it from.
HD_HD float exact_atanf(float x)
{
    const float hi0 = bits_f32(0x3eed6338u), hi1 = bits_f32(0x3f490fdau), hi2 = bits_f32(0x3f7b985eu), hi3 = bits_f32(0x3fc90fdau);
    const float lo0 = bits_f32(0x31ac3769u), lo1 = bits_f32(0x33222168u), lo2 = bits_f32(0x33140fb4u), lo3 = bits_f32(0x33a22168u);
    const float c0 = bits_f32(0x3eaaaaabu), c1 = bits_f32(0xbe4ccccdu), c2 = bits_f32(0x3e124925u), c3 = bits_f32(0xbde38e38u),
                c4 = bits_f32(0x3dba2e6eu), c5 = bits_f32(0xbd9d8795u), c6 = bits_f32(0x3d886b35u), c7 = bits_f32(0xbd6ef16bu),
                c8 = bits_f32(0x3d4bda59u), c9 = bits_f32(0xbd15a221u), c10 = bits_f32(0x3c8569d7u);
    const int32_t hx = (int32_t)f32_bits(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {                       // |x| >= 2^25 (or NaN)
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x31000000) return x;                // |x| < 2^-29
    const bool small = ix < 0x3ee00000;           // |x| < 7/16: no reduction
    const float ax = __builtin_fabsf(x);
    // reduction intervals [7/16, 11/16), [11/16, 19/16), [19/16, 39/16), [39/16, 2^25)
    const bool s0 = ix < 0x3f300000, s1 = ix < 0x3f980000, s2 = ix < 0x401c0000;
    const float num = s0 ? 2.0f * ax - 1.0f : s1 ? ax - 1.0f : s2 ? ax - 1.5f : -1.0f;
    const float den = s0 ? 2.0f + ax : s1 ? ax + 1.0f : s2 ? 1.0f + 1.5f * ax : ax;
    const float hi = s0 ? hi0 : s1 ? hi1 : s2 ? hi2 : hi3;
    const float lo = s0 ? lo0 : s1 ? lo1 : s2 ? lo2 : lo3;
    const float xr = small ? x : num / den;
    const float z = xr * xr;
    const float w = z * z;
    const float p1 = z * (c0 + w * (c2 + w * (c4 + w * (c6 + w * (c8 + w * c10)))));
    const float p2 = w * (c1 + w * (c3 + w * (c5 + w * (c7 + w * c9))));
    if (small) return xr - xr * (p1 + p2);
    const float r = hi - ((xr * (p1 + p2) - lo) - xr);
    return hx < 0 ? -r : r;
}

HD_HD float exact_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f;
    const float pi_o_4 = bits_f32(0x3f490fdbu), pi_o_2 = bits_f32(0x3fc90fdbu), pi = bits_f32(0x40490fdbu), pi_lo = bits_f32(0xb3bbbd2eu);
    const int32_t hx = (int32_t)f32_bits(x), hy = (int32_t)f32_bits(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;            // NaN
    const bool x_one = hx == 0x3f800000;                              // x == 1: atanf(y), whatever y is
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                // 2*sign(x) + sign(y)
    if (!x_one) {
        if (iy == 0) {
            if (m < 2) return y;
            return m == 2 ? pi + tiny : -pi - tiny;
        }
        if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
        if (ix == 0x7f800000) {
            if (iy == 0x7f800000) {
                switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
                }
            }
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
        if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    }
    const int32_t k = (iy - ix) >> 23;
    float z = exact_atanf(x_one ? y : __builtin_fabsf(y / x));
    if (x_one) return z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    switch (m) {
    case 0: return z;
    case 1: return bits_f32(f32_bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// The ordinary case of exact_atan2f as straight-line code: both arguments finite and non-zero, x != 1, exponents within 2^60 of each
// other and t = |y / x| inside atanf's polynomial range [2^-29, 2^25).  Same operations in the same order as the general function
// above takes for such arguments -- one third of its instructions, because the special cases and their divergent branches are gone.
// A wave takes this path when ALL of its active lanes qualify (discriminate below); the host build decides per element.
HD_HD bool atan2_plain_args(float y, float x)
{
    const int32_t hx = (int32_t)f32_bits(x), hy = (int32_t)f32_bits(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int32_t k = (iy - ix) >> 23;
    return ix != 0 && iy != 0 && ix < 0x7f800000 && iy < 0x7f800000 && hx != 0x3f800000 && k <= 60 && k >= -60;
}
HD_HD bool atan2_plain_quotient(float t) { const int32_t it = (int32_t)f32_bits(t); return it >= 0x31000000 && it < 0x4c000000; }
HD_HD float exact_atan2f_plain(float t /* |y / x| */, float y, float x)
{
    const float hi0 = bits_f32(0x3eed6338u), hi1 = bits_f32(0x3f490fdau), hi2 = bits_f32(0x3f7b985eu), hi3 = bits_f32(0x3fc90fdau);
    const float lo0 = bits_f32(0x31ac3769u), lo1 = bits_f32(0x33222168u), lo2 = bits_f32(0x33140fb4u), lo3 = bits_f32(0x33a22168u);
    const float c0 = bits_f32(0x3eaaaaabu), c1 = bits_f32(0xbe4ccccdu), c2 = bits_f32(0x3e124925u), c3 = bits_f32(0xbde38e38u),
                c4 = bits_f32(0x3dba2e6eu), c5 = bits_f32(0xbd9d8795u), c6 = bits_f32(0x3d886b35u), c7 = bits_f32(0xbd6ef16bu),
                c8 = bits_f32(0x3d4bda59u), c9 = bits_f32(0xbd15a221u), c10 = bits_f32(0x3c8569d7u);
    const float pi = bits_f32(0x40490fdbu), pi_lo = bits_f32(0xb3bbbd2eu);
    const int32_t it = (int32_t)f32_bits(t);
    const bool small = it < 0x3ee00000;
    const bool s0 = it < 0x3f300000, s1 = it < 0x3f980000, s2 = it < 0x401c0000;
    const float num = s0 ? 2.0f * t - 1.0f : s1 ? t - 1.0f : s2 ? t - 1.5f : -1.0f;
    const float den = s0 ? 2.0f + t : s1 ? t + 1.0f : s2 ? 1.0f + 1.5f * t : t;
    const float hi = s0 ? hi0 : s1 ? hi1 : s2 ? hi2 : hi3;
    const float lo = s0 ? lo0 : s1 ? lo1 : s2 ? lo2 : lo3;
    const float xr = small ? t : num / den;
    const float z = xr * xr;
    const float w = z * z;
    const float p1 = z * (c0 + w * (c2 + w * (c4 + w * (c6 + w * (c8 + w * c10)))));
    const float p2 = w * (c1 + w * (c3 + w * (c5 + w * (c7 + w * c9))));
    const float q = xr * (p1 + p2);
    const float a = small ? xr - q : hi - ((q - lo) - xr);            // atanf(t), t > 0
    const uint32_t hxu = f32_bits(x), hyu = f32_bits(y);
    const float b = (hxu >> 31) ? pi - (a - pi_lo) : a;               // x < 0: second / third quadrant ((a - pi_lo) - pi is exactly -(pi - (a - pi_lo)))
    return bits_f32(f32_bits(b) ^ (hyu & 0x80000000u));               // y < 0: mirrored
}

// arg(cur * conj(prev)) with the naive complex product the reference's std::complex operator* uses
// (re = a*c - b*d, im = a*d + b*c with d = -prev.im): SURVEY.md Q13.
HD_HD float discriminate(float a, float b, float pr, float pi_)
{
    const float c = pr, d = -pi_;
    const float re = a * c - b * d;
    const float im = a * d + b * c;
    const float t = __builtin_fabsf(im / re);
    const bool plain = atan2_plain_args(im, re) && atan2_plain_quotient(t);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(!plain) == 0ull) return exact_atan2f_plain(t, im, re);   // every active lane is an ordinary case
    return exact_atan2f(im, re);
#else
    return plain ? exact_atan2f_plain(t, im, re) : exact_atan2f(im, re);
#endif
}

}  // namespace hd
