/* ORACLE -- test infrastructure, never part of the product path.
 *
 * The reference discriminator is std::arg(x[i] * conj(x[i-1])) = atan2f(im, re) of whatever libm the reference was built against
 * (code/Decoder/FSK2_Demod.h:37-40).  On glibc 2.35 (this image) that is the classic fdlibm single-precision algorithm; a box with
 * another libm (e.g. a correctly rounded atan2f) would make every bit-exact assert on the discriminator output go red although
 * nothing is wrong with the kernels.  The oracle therefore carries its own restatement of that algorithm, in the branchy form of
 * the fdlibm sources (s_atanf.c / e_atan2f.c: argument reduction to one of four intervals, odd/even split degree-11 polynomial,
 * hi/lo constants), evaluated in plain float arithmetic; orc_atan2f_libm_mismatches() compares it with the box's libm and only
 * REPORTS a difference (tests/test_host_logic.py warns).  The product has its own, differently structured restatement
 * (habdec_amd/csrc/kernels/exact_math.h): the two are written independently on purpose and are compared bit for bit by the tests.
 */
#pragma once
#include <stdint.h>
#include <string.h>

static inline uint32_t orc_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float orc_bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float orc_atanf_fdlibm(float x)
{
    const float hi0 = orc_bits_f32(0x3eed6338u), hi1 = orc_bits_f32(0x3f490fdau), hi2 = orc_bits_f32(0x3f7b985eu), hi3 = orc_bits_f32(0x3fc90fdau);
    const float lo0 = orc_bits_f32(0x31ac3769u), lo1 = orc_bits_f32(0x33222168u), lo2 = orc_bits_f32(0x33140fb4u), lo3 = orc_bits_f32(0x33a22168u);
    const float c0 = orc_bits_f32(0x3eaaaaabu), c1 = orc_bits_f32(0xbe4ccccdu), c2 = orc_bits_f32(0x3e124925u), c3 = orc_bits_f32(0xbde38e38u),
                c4 = orc_bits_f32(0x3dba2e6eu), c5 = orc_bits_f32(0xbd9d8795u), c6 = orc_bits_f32(0x3d886b35u), c7 = orc_bits_f32(0xbd6ef16bu),
                c8 = orc_bits_f32(0x3d4bda59u), c9 = orc_bits_f32(0xbd15a221u), c10 = orc_bits_f32(0x3c8569d7u);
    const int32_t hx = (int32_t)orc_f32_bits(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {                       // |x| >= 2^25 (or NaN)
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    int seg;
    float hi = 0.f, lo = 0.f;
    if (ix < 0x3ee00000) {                        // |x| < 7/16
        if (ix < 0x31000000) return x;            // |x| < 2^-29
        seg = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {                    // |x| < 19/16
            if (ix < 0x3f300000) { seg = 0; hi = hi0; lo = lo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { seg = 1; hi = hi1; lo = lo1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { seg = 2; hi = hi2; lo = lo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { seg = 3; hi = hi3; lo = lo3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (c0 + w * (c2 + w * (c4 + w * (c6 + w * (c8 + w * c10)))));
    const float s2 = w * (c1 + w * (c3 + w * (c5 + w * (c7 + w * c9))));
    if (seg < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

static inline float orc_atan2f_fdlibm(float y, float x)
{
    const float tiny = 1.0e-30f;
    const float pi_o_4 = orc_bits_f32(0x3f490fdbu), pi_o_2 = orc_bits_f32(0x3fc90fdbu), pi = orc_bits_f32(0x40490fdbu), pi_lo = orc_bits_f32(0xb3bbbd2eu);
    const int32_t hx = (int32_t)orc_f32_bits(x), hy = (int32_t)orc_f32_bits(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;            // NaN
    if (hx == 0x3f800000) return orc_atanf_fdlibm(y);                      // x == 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                // 2*sign(x) + sign(y)
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
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = orc_atanf_fdlibm(__builtin_fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return orc_bits_f32(orc_f32_bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

