/*
 * habdec_oracle.cpp -- CPU ORACLE: restatement of the reference's RTTY demodulation chain.
 *
 * TEST INFRASTRUCTURE ONLY (see habdec_oracle.h).  Nothing under habdec_amd/ may include, link or
 * load this file.  Build: `make -C oracle` -> oracle/liboracle.so  (g++ -O3 -std=c++17, no
 * -march=native, no fast-math: the reference's own flags, code/Decoder/CMakeLists.txt:43-47, so
 * every sum below is a sequence of separately rounded float mul and add, never an FMA).
 *
 * Each function cites the reference lines it follows.  Quirk numbers (Qn) refer to SURVEY.md section 9.
 */
#include "habdec_oracle.h"
#include "orc_atan2f.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <map>
#include <regex>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "decim_taps_ref.inc"

typedef std::complex<float> cf32;

/* ------------------------------------------------------------------------------------------------
 * Decimation tables and stage plan
 * ---------------------------------------------------------------------------------------------- */
static const float* bits_as_float(const uint32_t* p) { return reinterpret_cast<const float*>(p); }

size_t orc_decim_taps(int total_name, int ratio, const float** taps)
{
#define ORC_T(N, R) if (total_name == N && ratio == R) { *taps = bits_as_float(kOracleTaps_##N##_##R); return kOracleTaps_##N##_##R##_n; }
    ORC_T(2, 2) ORC_T(4, 4) ORC_T(8, 8) ORC_T(16, 8) ORC_T(32, 16) ORC_T(64, 32) ORC_T(128, 32) ORC_T(256, 64)
#undef ORC_T
    *taps = nullptr;
    return 0;
}

/* Decoder.h:286-320: which (ratio, table) pairs make up a total factor. */
int orc_decim_plan(int f, int ratio[2], int name[2])
{
    switch (f) {
    case 256: ratio[0] = 64; name[0] = 256; ratio[1] = 4; name[1] = 4; return 2;
    case 128: ratio[0] = 32; name[0] = 128; ratio[1] = 4; name[1] = 4; return 2;
    case 64:  ratio[0] = 32; name[0] = 64;  ratio[1] = 2; name[1] = 2; return 2;
    case 32:  ratio[0] = 16; name[0] = 32;  ratio[1] = 2; name[1] = 2; return 2;
    case 16:  ratio[0] = 8;  name[0] = 16;  ratio[1] = 2; name[1] = 2; return 2;
    case 8:   ratio[0] = 8;  name[0] = 8;   return 1;
    case 4:   ratio[0] = 4;  name[0] = 4;   return 1;
    case 2:   ratio[0] = 2;  name[0] = 2;   return 1;
    case 1:   return 0;
    default:  return -1;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Decimator stage -- Decimator.h:69-80 (setInput), :99-146 (operator())
 * ---------------------------------------------------------------------------------------------- */
struct orc_decimator {
    int factor = 0;
    std::vector<float> taps;
    std::vector<cf32> buff;   /* p_buff_: [history(T-1) | input] */
};

orc_decimator* orc_decimator_new(int factor, const float* taps, size_t ntaps)
{
    auto* d = new orc_decimator;
    d->factor = factor;
    d->taps.assign(taps, taps + ntaps);
    return d;
}
void orc_decimator_free(orc_decimator* d) { delete d; }

size_t orc_decimator_run(orc_decimator* d, float* iq_f, size_t n)
{
    cf32* iq = reinterpret_cast<cf32*>(iq_f);
    const size_t T = d->taps.size();
    const size_t D = (size_t)d->factor;
    /* Q4: with n < T-1 the reference's wrap memcpy reads before the caller's buffer. */
    if (n + 1 < T) return (size_t)-1;
    /* setInput, Decimator.h:74-79 -- Q5: growth re-zeroes the first T entries (the history) */
    const size_t want = n + T + D;
    if (d->buff.size() < want) {
        d->buff.resize(want);
        std::fill(d->buff.begin(), d->buff.begin() + T, cf32(0, 0));
    }
    /* Decimator.h:122-124 */
    std::memcpy(d->buff.data() + T - 1, iq, n * sizeof(cf32));
    /* Decimator.h:128-138 -- output aliases input (Decoder.h:443-444, Q3) */
    size_t out = 0;
    for (size_t in = 0; in < n && out < n / D; in += D, ++out) {
        float ar = 0.0f, ai = 0.0f;
        const cf32* b = d->buff.data() + in;
        for (size_t t = 0; t < T; ++t) {
            const float k = d->taps[t];
            ar = ar + b[t].real() * k;
            ai = ai + b[t].imag() * k;
        }
        iq[out] = cf32(ar, ai);
    }
    /* Decimator.h:140-143 -- Q4: the history is taken from the caller's buffer AFTER its head was
     * overwritten by the outputs. */
    std::memcpy(d->buff.data(), iq + n - (T - 1), (T - 1) * sizeof(cf32));
    return out;
}

/* ------------------------------------------------------------------------------------------------
 * DC removal -- Decoder.h:450-459 (Q6: re-seeded from x[0] on every call)
 * ---------------------------------------------------------------------------------------------- */
void orc_dc_remove(float* iq_f, size_t n)
{
    if (!n) return;
    cf32* x = reinterpret_cast<cf32*>(iq_f);
    cf32 wp(0.97f * x[0].real(), 0.97f * x[0].imag());
    for (size_t i = 0; i < n; ++i) {
        const cf32 s(0.97f * wp.real(), 0.97f * wp.imag());
        const cf32 w(x[i].real() + s.real(), x[i].imag() + s.imag());
        x[i] = cf32(w.real() - wp.real(), w.imag() - wp.imag());
        wp = w;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Spectrum -- FFT.cpp:90-99 (forward c2c, unnormalised) + :77-87 (half swap).  FFTW is absent:
 * PARITY UNPINNED at this boundary; this is the exact DFT evaluated in double and rounded once.
 * ---------------------------------------------------------------------------------------------- */
void orc_fft_shifted(const float* in_iq, float* out_iq, size_t n)
{
    std::vector<std::complex<double>> a(n);
    unsigned lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    for (size_t i = 0; i < n; ++i) {
        size_t r = 0;
        for (unsigned b = 0; b < lg; ++b) if (i & ((size_t)1 << b)) r |= (size_t)1 << (lg - 1 - b);
        a[r] = std::complex<double>(in_iq[2 * i], in_iq[2 * i + 1]);
    }
    const double pi = 3.14159265358979323846;
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < len / 2; ++k) {
                const double ang = -2.0 * pi * (double)k / (double)len;
                const std::complex<double> w(std::cos(ang), std::sin(ang));
                const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
        }
    }
    const size_t half = n / 2;
    for (size_t i = 0; i < n; ++i) {
        const size_t src = (i + half) % n;   /* out[i] <-> out[i+half] */
        out_iq[2 * i] = (float)a[src].real();
        out_iq[2 * i + 1] = (float)a[src].imag();
    }
}

/* ------------------------------------------------------------------------------------------------
 * Running average -- Average.h:31-70
 * ---------------------------------------------------------------------------------------------- */
template <typename T>
struct OrcAverage {
    T sum = 0;
    size_t count = 0;
    size_t max_count;
    explicit OrcAverage(size_t m) : max_count(std::max<size_t>(1, m)) { add(T(0)); }   /* :34-37, Q21 */
    double get() const { return count ? double(sum) / count : double(sum); }           /* :57-62 */
    double add(const T& v)                                                             /* :39-55 */
    {
        const double diff = get() - v;
        if (count == max_count) sum = T(get() * (max_count - 1) + v);
        else { ++count; sum += v; }
        return diff;
    }
    void reset(const T& v) { sum = v; count = 1; }                                     /* :66-70 */
};

/* ------------------------------------------------------------------------------------------------
 * AFC -- AFC.h:92-184 (process), :187-194 (reset), :235-286 (FftPower), :289-329 (FindPeaks)
 * ---------------------------------------------------------------------------------------------- */
struct orc_afc {
    std::vector<cf32> spec;
    double spec_rate = 0;
    std::vector<float> power;
    double correction = 0, noise_floor = 0, noise_var = 0, shift_hz = 0;
    OrcAverage<double> nf_avg{100}, nv_avg{100};
    OrcAverage<int> pl_avg{4}, pr_avg{4};
    int gui_l = 0, gui_r = 0;
};
orc_afc* orc_afc_new(void) { return new orc_afc; }
void orc_afc_free(orc_afc* a) { delete a; }
void orc_afc_set_spectrum(orc_afc* a, const float* iq, size_t nbins, double rate)
{
    a->spec.assign(reinterpret_cast<const cf32*>(iq), reinterpret_cast<const cf32*>(iq) + nbins);
    a->spec_rate = rate;
}

static bool orc_fft_power(const std::vector<cf32>& s, double rate, std::vector<float>& p)
{
    if (s.empty() || !rate) return false;                       /* AFC.h:242-246 */
    for (const cf32& c : s)                                     /* :250-260 */
        if (c.real() != c.real() || c.imag() != c.imag() || std::isinf(c.real()) || std::isinf(c.imag()))
            return false;
    p.resize(s.size());
    for (size_t i = 0; i < s.size(); ++i) {                     /* :265-271, Q22 */
        float v = (s[i].real() * s[i].real() + s[i].imag() * s[i].imag()) / (float)s.size();
        v = v * v;
        v = (float)((double)v / rate);
        p[i] = 10.0f * log10f(v);
    }
    for (float v : p) if (v != v || std::isinf(v)) return false; /* :273-283 */
    return true;
}

static void orc_find_peaks(const std::vector<float>& v, double rel_sep, int* o1, int* o2)
{
    int sep = (int)std::round(rel_sep * (double)v.size());      /* AFC.h:299-300 */
    sep = std::max(8, sep);
    int p1 = (int)(std::max_element(v.begin(), v.end()) - v.begin());   /* first maximum */
    const int begin = std::max(p1 - 2 * sep, 0), end = std::min(p1 + 2 * sep, (int)v.size());
    int p2 = 0;
    float p2v = v[0];
    for (int i = begin; i < end; ++i)                           /* :311-319 */
        if (v[i] > p2v && std::abs(i - p1) > sep / 2) { p2 = i; p2v = v[i]; }
    if (p2 < p1) std::swap(p1, p2);
    *o1 = p1; *o2 = p2;
}

double orc_afc_process(orc_afc* a)
{
    if (!orc_fft_power(a->spec, a->spec_rate, a->power)) { a->correction = 0; return 0; }   /* :96-100 */
    double acc = 0.0;                                           /* :103 */
    for (float v : a->power) acc += v;
    a->noise_floor = acc / a->power.size();
    double var = 0;                                             /* :224-232 */
    for (float v : a->power) var += (v - a->noise_floor) * (v - a->noise_floor);
    var /= a->power.size();
    a->noise_var = std::sqrt(var);
    a->nf_avg.add(a->noise_floor);
    a->nv_avg.add(a->noise_var);

    const float fsk_shift = 500;                                /* :110-111 */
    const float rel_sep = (float)(fsk_shift / a->spec_rate);
    int p1, p2;
    orc_find_peaks(a->power, rel_sep, &p1, &p2);

    const float thr = (float)(a->nf_avg.get() + 3 * std::fabs(a->nv_avg.get()));   /* :119 */
    const bool d1 = a->power[p1] > thr, d2 = a->power[p2] > thr;
    bool sl = false, sr = false;
    if (d1 && d2) {                                             /* :133-142 */
        if (p2 < p1) std::swap(p1, p2);
        if (a->pl_avg.add(p1) <= 2) sl = true;
        if (a->pr_avg.add(p2) <= 2) sr = true;
    }
    a->gui_l = 0;                                               /* :146-162 */
    if (d1) a->gui_l = sl ? (int)a->pl_avg.get() : (int)(-a->pl_avg.get());
    a->gui_r = 0;
    if (d2) a->gui_r = sr ? (int)a->pr_avg.get() : (int)(-a->pr_avg.get());

    if (sl && sr) {                                             /* :165-181 */
        const int L = (int)std::round(a->pl_avg.get());
        const int R = (int)std::round(a->pr_avg.get());
        const int dist = R - L;
        const double hz_per_bin = a->spec_rate / a->spec.size();
        a->shift_hz = hz_per_bin * dist;
        const double mid = L + dist / 2;
        const double err_bins = mid - double(a->spec.size()) / 2;
        const double err_hz = hz_per_bin * err_bins;
        if (4 < std::abs(err_bins)) a->correction = err_hz;
    }
    return a->correction;
}

void orc_afc_reset_correction(orc_afc* a, double c)            /* AFC.h:187-194 */
{
    const double bins_per_hz = (double)a->spec.size() / a->spec_rate;
    a->pl_avg.reset((int)std::max(0.0, a->pl_avg.get() - c * bins_per_hz));
    a->pr_avg.reset((int)std::max(0.0, a->pr_avg.get() - c * bins_per_hz));
    a->correction = 0;
}
size_t orc_afc_power(orc_afc* a, const float** p) { *p = a->power.data(); return a->power.size(); }
void orc_afc_get(orc_afc* a, double* c, double* sh, double* nf, double* nv, int* gl, int* gr)
{
    if (c) *c = a->correction; if (sh) *sh = a->shift_hz; if (nf) *nf = a->noise_floor;
    if (nv) *nv = a->noise_var; if (gl) *gl = a->gui_l; if (gr) *gr = a->gui_r;
}

/* ------------------------------------------------------------------------------------------------
 * Low-pass FIR -- FirFilter.h:117-169 (dotProduct), :173-209 (design), habdec_windows.h:27-53
 * ---------------------------------------------------------------------------------------------- */
struct orc_fir {
    size_t in_size = 0;
    std::vector<float> taps;
    std::vector<cf32> buff;
    int float_trig = 1;
};
orc_fir* orc_fir_new(void) { return new orc_fir; }
void orc_fir_free(orc_fir* f) { delete f; }
void orc_fir_set_input_size(orc_fir* f, size_t n) { if (n) f->in_size = n; }   /* FirFilter.h:78-87 */
size_t orc_fir_taps(orc_fir* f, const float** t) { *t = f->taps.data(); return f->taps.size(); }

/* habdec_windows.h:27-53.  The reference calls `sin`/`cos` UNQUALIFIED from a global-namespace
 * template; which overload that finds depends on what the including translation unit declared first
 * (Q9b): with only C++ standard headers in front (libstdc++) just ::sin(double)/::cos(double)
 * are visible, so the trigonometry AND the window sum run in double ("double trig", mode 0, pinned by
 * _ref/libhabdec_ref.so); if anything included <math.h> first, the float overloads win and everything
 * stays float ("float trig", mode 1, the default -- it is what SURVEY.md Q9 observed -- pinned by
 * _ref/libhabdec_ref_mathh.so, the same sources compiled with `-include math.h`). */
static float orc_sinc(float x, int float_trig)
{
    if (!x) return 1.0f;
    if (float_trig) return sinf(x) / x;
    return (float)(sin((double)x) / (double)x);
}
static float orc_bh(size_t x, size_t N, int float_trig)
{
    static const float a0 = 0.35874, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
    static const float PI2 = 2.0 * M_PI, PI4 = 4.0 * M_PI, PI6 = 6.0 * M_PI;
    const float N_1 = N - 1;
    if (float_trig)
        return a0 - a1 * cosf(PI2 * x / N_1) + a2 * cosf(PI4 * x / N_1) - a3 * cosf(PI6 * x / N_1);
    const double w = a0 - a1 * cos((double)(PI2 * x / N_1)) + a2 * cos((double)(PI4 * x / N_1)) - a3 * cos((double)(PI6 * x / N_1));
    return (float)w;
}

void orc_fir_design_mode(orc_fir* f, int float_trig) { f->float_trig = float_trig; }

void orc_fir_design(orc_fir* f, float rel_width, float trans)                  /* FirFilter.h:173-209 */
{
    if (!f->in_size) return;                                   /* Q10 */
    const float tbw = trans ? trans : rel_width * rel_width;
    size_t n = (size_t)(4.0f / tbw);                           /* Q8 */
    if (n > f->in_size) n = f->in_size;
    n |= 1;
    if (n <= 4) return;
    if (n == f->taps.size()) return;
    f->taps.resize(n);
    double sum = 0;
    const int mid = int(n / 2);
    for (int i = 0; i < int(n); i++) {                         /* Q9 */
        f->taps[i] = orc_sinc(2.0f * rel_width * (i - mid), f->float_trig) * orc_bh(size_t(i), n, f->float_trig);
        sum += f->taps[i];
    }
    for (size_t i = 0; i < n; ++i) f->taps[i] /= sum;
}

int orc_fir_run(orc_fir* f, const float* in_f, size_t n, float* out_f)         /* FirFilter.h:117-169 */
{
    const cf32* in = reinterpret_cast<const cf32*>(in_f);
    cf32* out = reinterpret_cast<cf32*>(out_f);
    const size_t T = f->taps.size();
    if (!T) return 1;
    if (T > n + 1) return 2;
    const size_t want = n + T;
    if (f->buff.size() < want) {                               /* :141-147, Q5 */
        f->buff.resize(want);
        std::fill(f->buff.begin(), f->buff.begin() + T, cf32(0, 0));
    }
    std::memcpy(f->buff.data() + T - 1, in, n * sizeof(cf32));
    for (size_t i = 0; i < n; ++i) {                           /* :155-161 */
        float ar = 0.0f, ai = 0.0f;
        const cf32* b = f->buff.data() + i;
        for (size_t t = 0; t < T; ++t) {
            const float k = f->taps[t];
            ar = ar + b[t].real() * k;
            ai = ai + b[t].imag() * k;
        }
        out[i] = cf32(ar, ai);
    }
    std::memcpy(f->buff.data(), in + n - (T - 1), (T - 1) * sizeof(cf32));   /* :163-167 */
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * FSK discriminator -- FSK2_Demod.h:29-42.  Q12: the reference keeps last_value in a
 * thread_local static; here it is per handle (identical for one decoder per thread).  Q13: naive
 * complex product, arg = atan2f.
 * ---------------------------------------------------------------------------------------------- */
struct orc_demod { bool primed = false; cf32 last; };
orc_demod* orc_demod_new(void) { return new orc_demod; }
void orc_demod_free(orc_demod* d) { delete d; }

static inline float orc_disc(cf32 x, cf32 prev)
{
    const float a = x.real(), b = x.imag(), c = prev.real(), d = -prev.imag();
    const float re = a * c - b * d;
    const float im = a * d + b * c;
    return orc_atan2f_fdlibm(im, re);      /* = glibc 2.35's atan2f bit for bit; see orc_atan2f.h and orc_atan2f_libm_mismatches() */
}

/* How many of n pseudo-random argument pairs (plus the special classes) the box's libm atan2f answers differently from the
 * restatement the oracle uses.  0 on glibc 2.35; anything else means "this box has another libm", not "the oracle is wrong". */
size_t orc_atan2f_libm_mismatches(uint64_t seed, size_t n)
{
    size_t bad = 0;
    uint64_t st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto next = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 32); };
    const float specials[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 1e-38f, 3.4e38f, 33554432.0f, 9.3e-10f, 0.4375f, 0.6875f, 1.1875f, 2.4375f};
    const size_t ns = sizeof(specials) / sizeof(specials[0]);
    for (size_t i = 0; i < n + ns * ns; ++i) {
        float y, x;
        if (i < ns * ns) { y = specials[i / ns]; x = specials[i % ns]; }
        else if (i & 1) { y = orc_bits_f32(next()); x = orc_bits_f32(next()); }              /* any bit pattern */
        else { y = ((int32_t)next()) * 4.6566e-10f * 2.0f; x = ((int32_t)next()) * 4.6566e-10f * 2.0f; }   /* ordinary products */
        const float want = atan2f(y, x), got = orc_atan2f_fdlibm(y, x);
        if (want != want && got != got) continue;
        if (orc_f32_bits(want) != orc_f32_bits(got)) ++bad;
    }
    return bad;
}
void orc_demod_run(orc_demod* s, const float* iq_f, size_t n, float* out)
{
    if (!n) return;
    const cf32* x = reinterpret_cast<const cf32*>(iq_f);
    if (!s->primed) { s->last = x[0]; s->primed = true; }      /* static initialiser, first call only */
    for (size_t i = 1; i < n; ++i) out[i] = orc_disc(x[i], x[i - 1]);
    out[0] = orc_disc(x[0], s->last);
    s->last = x[n - 1];
}

/* ------------------------------------------------------------------------------------------------
 * Symbol extractor -- SymbolExtractor.h:108-255
 * ---------------------------------------------------------------------------------------------- */
struct orc_symex {
    double fs = 0, baud = 1;                                   /* :96-97 */
    std::vector<float> v;
    std::vector<uint8_t> bits;
    std::vector<size_t> flips;
    int float_abs = 1;   /* Q15b: which `abs` the reference's unqualified call binds to, see orc_symex_abs_mode */
    size_t spb() const { return size_t(std::round(fs / baud)); }   /* :90 */
};
orc_symex* orc_symex_new(void) { return new orc_symex; }
void orc_symex_free(orc_symex* s) { delete s; }
void orc_symex_rates(orc_symex* s, double fs, double baud) { s->fs = fs; s->baud = baud; }
/* SymbolExtractor.h:217 calls `abs(avg_right-avg_left)` unqualified on floats.  Like sin/cos in the tap
 * design (Q9b) the overload it binds to depends on the including translation unit: after <math.h> or
 * <stdlib.h> (libstdc++'s C-header wrappers) the float overload is visible (mode 1, default, what
 * SURVEY.md Q15 observed; pinned by _ref/libhabdec_ref_mathh.so); with only <c...> headers in front the
 * sole candidate is ::abs(int) and the difference is truncated to an integer before the arg-max
 * (mode 0; pinned by _ref/libhabdec_ref.so). */
void orc_symex_abs_mode(orc_symex* s, int float_abs) { s->float_abs = float_abs; }
size_t orc_symex_held(orc_symex* s) { return s->v.size(); }
size_t orc_symex_last_flips(orc_symex* s, const size_t** f) { *f = s->flips.data(); return s->flips.size(); }

void orc_symex_push(orc_symex* s, const float* x, size_t n)    /* :108-125 */
{
    if (!n) return;
    if (s->v.size() > 3e4) s->v.clear();                       /* Q14 */
    s->v.insert(s->v.end(), x, x + n);
}

static inline int orc_sgn(float v) { return (0.0f < v) - (v < 0.0f); }         /* :33-36 */

static void orc_flip_avg(const std::vector<float>& v, size_t i, size_t R, float* al, float* ar)   /* :51-63, Q15 */
{
    const int L = std::max(int(i - R), 0);
    const int Rr = (int)std::min(i + R, v.size());
    float sl = 0.0f, sr = 0.0f;
    for (size_t k = (size_t)L; k < i; ++k) sl = sl + v[k];
    for (size_t k = i; k < (size_t)Rr; ++k) sr = sr + v[k];
    *al = sl / (float)(i - L);
    *ar = sr / (float)(Rr - i);
}

static size_t orc_first_flip(const orc_symex* s, size_t off)   /* :162-224 */
{
    const std::vector<float>& v = s->v;
    const size_t spb = s->spb();
    if ((v.size() - off) < spb) return 0;
    const size_t R = (size_t)std::max(4, int(spb / 4));
    size_t i = off + R;
    float al, ar;
    orc_flip_avg(v, i, R, &al, &ar);
    while (orc_sgn(al) == orc_sgn(ar)) {
        i += 1;
        if (i >= (v.size() - spb)) return 0;
        orc_flip_avg(v, i, R, &al, &ar);
    }
    const size_t lo = i;
    while (orc_sgn(al) != orc_sgn(ar)) {
        i += 1;
        if (i >= (v.size() - spb)) return 0;
        orc_flip_avg(v, i, R, &al, &ar);
    }
    const size_t hi = i;
    size_t best = lo;
    float bestw = -1.0f;
    for (size_t j = lo; j < hi; ++j) {                         /* first maximum of |avg_r - avg_l| */
        orc_flip_avg(v, j, R, &al, &ar);
        const float w = s->float_abs ? std::fabs(ar - al) : (float)std::abs((int)(ar - al));
        if (j == lo || w > bestw) { best = j; bestw = w; }
    }
    return best;
}

void orc_symex_run(orc_symex* s)                               /* :129-158 */
{
    s->flips.clear();
    if (!s->fs || !s->baud) return;
    if (s->v.size() < s->fs / s->baud * 3) return;
    size_t off = 0, f = orc_first_flip(s, off);                /* :228-241 */
    while (f) { s->flips.push_back(f); off = f; f = orc_first_flip(s, off); }
    if (s->flips.empty()) return;
    size_t last = 0;
    const size_t spb = s->spb();
    for (size_t fp : s->flips) {
        float acc = 0.0f;
        for (size_t k = last; k < fp; ++k) acc = acc + s->v[k];
        const float avg = acc / (float)(fp - last);
        const bool bit = avg > 0;
        size_t cnt = size_t(std::round(float(fp - last) / float(spb)));   /* Q16 */
        last = fp;
        while (cnt--) s->bits.push_back(bit);
    }
    const size_t erase_to = std::min(last, s->v.size());
    s->v.erase(s->v.begin(), s->v.begin() + erase_to);
}

size_t orc_symex_get(orc_symex* s, uint8_t* out, size_t cap)   /* :245-255 */
{
    const size_t n = std::min(cap, s->bits.size());
    std::memcpy(out, s->bits.data(), n);
    s->bits.erase(s->bits.begin(), s->bits.begin() + n);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * RTTY framing -- RTTY.h:59-137
 * ---------------------------------------------------------------------------------------------- */
struct orc_rtty {
    size_t nbits = 0;
    float nstops = 0;
    std::vector<uint8_t> bits;
    std::vector<char> chars;
};
orc_rtty* orc_rtty_new(size_t nbits, float nstops) { auto* r = new orc_rtty; r->nbits = nbits; r->nstops = nstops; return r; }
void orc_rtty_free(orc_rtty* r) { delete r; }
void orc_rtty_push(orc_rtty* r, const uint8_t* b, size_t n) { r->bits.insert(r->bits.end(), b, b + n); }
size_t orc_rtty_pending_bits(orc_rtty* r) { return r->bits.size(); }
size_t orc_rtty_get(orc_rtty* r, char* out, size_t cap)
{
    const size_t n = std::min(cap, r->chars.size());
    std::memcpy(out, r->chars.data(), n);
    r->chars.clear();                                           /* get() clears everything, :68-72 */
    return n;
}

size_t orc_rtty_run(orc_rtty* r)                               /* :77-137 */
{
    if (!r->nbits && !r->nstops) return 0;
    if (r->bits.size() < (1 + r->nbits + r->nstops)) return 0;
    size_t decoded = 0, last = 0;
    const size_t nb = r->nbits;
    for (size_t i = 0; i < r->bits.size();) {
        bool is_char = r->bits[i] == 0;
        const bool fits = (i + 1 + nb + r->nstops) <= r->bits.size();
        is_char &= fits;
        /* Q17: the reference probes the stop bits even when they are out of range; the probe's
         * result is irrelevant then, so it is skipped here. */
        if (fits) for (size_t s = 0; s < r->nstops; ++s) is_char &= r->bits[i + 1 + nb + s] == 1;
        if (!is_char) { ++i; continue; }
        ++i;
        char c = 0;
        for (size_t k = 0; k < nb; ++k) { c += r->bits[i] << k; ++i; }   /* LSB first, Q18 */
        r->chars.push_back(c);
        ++decoded;
        i += r->nstops;
        last = i - 1;
    }
    if (last) r->bits.erase(r->bits.begin(), r->bits.begin() + last + 1);
    return decoded;
}

/* ------------------------------------------------------------------------------------------------
 * Text stage -- CRC.cpp:21-47 (Q26), sentence_extract.cpp:30,58-98 (Q19)
 * ---------------------------------------------------------------------------------------------- */
static std::string orc_crc_str(const std::string& s)
{
    unsigned int crc = 0xffff;
    for (size_t i = 0; i < s.length(); i++) {
        crc ^= (((unsigned int)s[i]) << 8);
        for (int j = 0; j < 8; j++) crc = (crc & 0x8000) ? (crc << 1) ^ 0x1021 : crc << 1;
    }
    static const char* hex = "0123456789ABCDEF";
    std::string r;
    r += hex[(crc >> 12) & 15]; r += hex[(crc >> 8) & 15]; r += hex[(crc >> 4) & 15]; r += hex[crc & 15];
    return r;
}
void orc_crc16(const char* s, size_t n, char out4[5])
{
    const std::string r = orc_crc_str(std::string(s, n));
    std::memcpy(out4, r.c_str(), 5);
}

static const std::regex& orc_regex()
{
    static const std::regex re(R"_(.*?(\$+)([\w,\-,\s]+?),(.+?)(\*|\$)(\w\w\w\w).*)_");
    return re;
}

static bool orc_extract(std::string stream, std::string& call, std::string& data, std::string& crc, std::string& rest)
{
    std::replace(stream.begin(), stream.end(), '\n', ' ');
    const int CRC_LEN = 4;
    if (stream.find("*") < std::string::npos - CRC_LEN) {
        std::smatch m;
        std::regex_match(stream, m, orc_regex());
        if (m.size() >= 5) {
            call = m[2]; data = m[3]; crc = m[5];
            const size_t cut = std::min(stream.size(), size_t(m.position(4) + CRC_LEN));
            rest = stream.substr(cut);
            return true;
        }
    }
    return false;
}
int orc_extract_sentence(const char* stream, size_t n, char* callsign, char* data, char* crc, char* rest, size_t cap)
{
    std::string c, d, k, r;
    if (!orc_extract(std::string(stream, n), c, d, k, r)) return 0;
    auto put = [cap](char* dst, const std::string& s) { const size_t m = std::min(cap - 1, s.size()); std::memcpy(dst, s.data(), m); dst[m] = 0; };
    put(callsign, c); put(data, d); put(crc, k); put(rest, r);
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Decoder -- Decoder.h:206-219 (pushSamples), :268-332 (stage setup), :416-638 (process)
 * ---------------------------------------------------------------------------------------------- */
struct orc_decoder {
    std::vector<cf32> in_queue;           /* iq_in_buffer_ */
    double in_rate = 0;
    std::vector<cf32> temp;               /* iq_samples_temp_ */
    std::vector<cf32> decimated;          /* iq_samples_decimated_ */
    std::vector<cf32> filtered;           /* iq_samples_filtered_ */
    std::vector<orc_decimator*> stages;
    int factor = 1;
    bool dc = false;
    bool with_fft = true;
    bool ungated = false;       /* stage-level studies only: skip the 160 kHz decode gate */
    const size_t nbins = 1 << 12;
    std::vector<cf32> freq_in, freq_out;
    uint64_t fft_count = 0;
    orc_afc afc;
    float lp_bw = 1500, lp_trans = 0.025;
    orc_fir fir;
    std::vector<float> demod;
    orc_demod demod_state;
    orc_symex symex;
    orc_rtty rtty;
    std::string rtty_stream, last_sentence, sentence_log, match_log, chars_log;
    uint64_t total_bits = 0;     /* symbols produced so far (bench self-check) */
    /* every process() call's discriminator output folded into one word, the way the engine folds the checksums its kernels leave in the result slots
     * (include/habdec_amd.h: hd_stream_demod_checksum_total): per call (n, sum of the samples' bit patterns, sum of (i + 1) * bit pattern), FNV-1a over 32-bit words */
    uint64_t demod_hash = 0xCBF29CE484222325ull;
    /* introspection of the last call */
    std::vector<cf32> last_decimated, last_filtered;
    std::vector<float> last_demod;
    std::vector<uint8_t> last_bits;
    double dec_rate() const { return in_rate / factor; }
    ~orc_decoder() { for (auto* s : stages) orc_decimator_free(s); }
};

orc_decoder* orc_decoder_new(void) { return new orc_decoder; }
void orc_decoder_free(orc_decoder* d) { delete d; }

int orc_decoder_setup_factor(orc_decoder* d, size_t f)         /* Decoder.h:268-332 */
{
    if (f < 1 || f > 256) return d->factor;
    for (auto* s : d->stages) orc_decimator_free(s);
    d->stages.clear();
    d->factor = 1;
    int ratio[2], name[2];
    const int ns = orc_decim_plan((int)f, ratio, name);
    if (ns <= 0) return 0;                                     /* factor 1 and non-powers fall to `default` */
    for (int i = 0; i < ns; ++i) {
        const float* taps; const size_t n = orc_decim_taps(name[i], ratio[i], &taps);
        d->stages.push_back(orc_decimator_new(ratio[i], taps, n));
    }
    d->factor = (int)f;
    return d->factor;
}
void orc_decoder_baud(orc_decoder* d, double b) { d->symex.baud = b; }
void orc_decoder_rtty(orc_decoder* d, size_t bits, float stops) { d->rtty.nbits = bits; d->rtty.nstops = stops; }
void orc_decoder_dc_remove(orc_decoder* d, int on) { d->dc = on != 0; }
/* lookup context of the reference's unqualified libm calls (Q9b/Q15b): 1 = <math.h> context (float trig,
 * float abs; default), 0 = <cmath>-only context (double trig, integer abs). */
void orc_decoder_lookup_mode(orc_decoder* d, int mathh_context) { d->fir.float_trig = mathh_context; d->symex.float_abs = mathh_context; }
void orc_decoder_with_fft(orc_decoder* d, int on) { d->with_fft = on != 0; }
void orc_decoder_ungated(orc_decoder* d, int on) { d->ungated = on != 0; }
void orc_decoder_lowpass_bw(orc_decoder* d, float hz)          /* Decoder.h:238-243 */
{
    d->lp_bw = hz;
    orc_fir_design(&d->fir, (float)(d->lp_bw / d->dec_rate()), d->lp_trans);
}
void orc_decoder_lowpass_trans(orc_decoder* d, float t)        /* Decoder.h:252-257 */
{
    d->lp_trans = t;
    orc_fir_design(&d->fir, (float)(d->lp_bw / d->dec_rate()), d->lp_trans);
}
void orc_decoder_push(orc_decoder* d, const float* iq, size_t n, double rate)   /* :206-219, Q1 */
{
    const cf32* x = reinterpret_cast<const cf32*>(iq);
    d->in_queue.insert(d->in_queue.end(), x, x + n);
    if (!d->in_rate) d->in_rate = (float)rate;                 /* init(const float) */
}
void orc_decoder_reset_correction(orc_decoder* d, double c) { orc_afc_reset_correction(&d->afc, c); }

static void orc_decoder_process_body(orc_decoder* d);
void orc_decoder_process(orc_decoder* d)
{
    orc_decoder_process_body(d);
    /* test infrastructure, not reference behaviour: this call's discriminator output (empty when the low-pass did not run) folded into demod_hash */
    uint32_t c0 = 0, c1 = 0;
    for (size_t i = 0; i < d->last_demod.size(); ++i) {
        uint32_t b; std::memcpy(&b, &d->last_demod[i], 4);
        c0 += b; c1 += (uint32_t)(i + 1) * b;
    }
    for (const uint32_t x : {(uint32_t)d->last_demod.size(), c0, c1}) d->demod_hash = (d->demod_hash ^ x) * 0x100000001B3ull;
}
static void orc_decoder_process_body(orc_decoder* d)           /* Decoder.h:416-638 */
{
    d->last_decimated.clear(); d->last_filtered.clear(); d->last_demod.clear(); d->last_bits.clear();
    if (!d->in_rate) return;
    if (int(d->in_queue.size()) < d->factor) return;           /* :429-430 */
    const size_t take = d->in_queue.size() - (d->in_queue.size() % d->factor);   /* Q2 */
    d->temp.assign(d->in_queue.begin(), d->in_queue.begin() + take);
    d->in_queue.erase(d->in_queue.begin(), d->in_queue.begin() + take);

    size_t n = d->temp.size();                                 /* :440-447 */
    for (auto* st : d->stages) {
        n = orc_decimator_run(st, reinterpret_cast<float*>(d->temp.data()), n);
        if (n == (size_t)-1) { n = 0; break; }
    }
    d->temp.resize(n);
    if (d->dc) orc_dc_remove(reinterpret_cast<float*>(d->temp.data()), d->temp.size());   /* :450-459 */
    d->last_decimated = d->temp;
    d->decimated.insert(d->decimated.end(), d->temp.begin(), d->temp.end());   /* :461 */

    if (d->with_fft) {
        if (d->freq_in.size() < d->nbins && d->temp.size()) {  /* :467-473, Q7 */
            const size_t k = std::min(d->nbins - d->freq_in.size(), d->temp.size());
            d->freq_in.insert(d->freq_in.end(), d->temp.begin(), d->temp.begin() + k);
        }
        if (d->freq_in.size() >= d->nbins) {                   /* :479-489 */
            d->freq_out.resize(d->nbins);
            orc_fft_shifted(reinterpret_cast<const float*>(d->freq_in.data()),
                            reinterpret_cast<float*>(d->freq_out.data()), d->nbins);
            d->freq_in.clear();
            ++d->fft_count;
        }
    }
    const size_t batch = 256;                                  /* :492-495 */
    if (d->decimated.size() < batch) return;

    if (d->with_fft) {                                         /* :501-509, Q21 */
        if (d->freq_out.size() == d->nbins)
            orc_afc_set_spectrum(&d->afc, reinterpret_cast<const float*>(d->freq_out.data()), d->nbins, d->dec_rate());
        orc_afc_process(&d->afc);
    }
    if (!d->ungated && d->dec_rate() > 4 * 40e3) {             /* :522-527, Q11 */
        d->temp.clear();
        d->decimated.clear();
        return;
    }
    const size_t m = d->decimated.size() - d->decimated.size() % batch;   /* :532-542 */
    d->filtered.resize(m);
    orc_fir_set_input_size(&d->fir, m);
    orc_fir_design(&d->fir, (float)(d->lp_bw / d->dec_rate()), d->lp_trans);
    orc_fir_run(&d->fir, reinterpret_cast<const float*>(d->decimated.data()), m, reinterpret_cast<float*>(d->filtered.data()));
    d->decimated.erase(d->decimated.begin(), d->decimated.begin() + m);
    d->last_filtered = d->filtered;

    d->demod.resize(d->filtered.size());                       /* :546-555 */
    orc_demod_run(&d->demod_state, reinterpret_cast<const float*>(d->filtered.data()), d->filtered.size(), d->demod.data());
    d->last_demod = d->demod;
    d->symex.fs = d->dec_rate();
    orc_symex_push(&d->symex, d->demod.data(), d->demod.size());

    orc_symex_run(&d->symex);                                  /* :559-566 */
    d->last_bits = d->symex.bits;
    d->total_bits += d->last_bits.size();
    d->symex.bits.clear();
    if (!d->last_bits.empty()) {
        orc_rtty_push(&d->rtty, d->last_bits.data(), d->last_bits.size());
        orc_rtty_run(&d->rtty);
    }
    if (d->rtty.chars.empty()) return;                         /* :568-569 */

    std::vector<char> raw;
    raw.swap(d->rtty.chars);
    std::string printable;                                     /* :575-580 */
    for (char c : raw) if (isprint(c) || c == '\n') printable.push_back(c);
    d->rtty_stream += printable;
    d->chars_log += printable;

    if (d->rtty_stream.size() > 20) {                          /* :591-614 */
        for (;;) {
            std::string call, data, crc, rest;
            if (!orc_extract(d->rtty_stream, call, data, crc, rest)) break;
            d->rtty_stream = rest;
            d->last_sentence = call + "," + data + "*" + crc;
            d->match_log += d->last_sentence + "\n";
            if (crc == orc_crc_str(call + "," + data)) d->sentence_log += d->last_sentence + "\n";
        }
    }
    if (d->rtty_stream.size() > 1000)                          /* :635-636 */
        d->rtty_stream.erase(0, d->rtty_stream.rfind('$'));
}

#define ORC_STR(name, field) size_t name(orc_decoder* d, const char** s) { *s = d->field.c_str(); return d->field.size(); }
ORC_STR(orc_decoder_rtty_stream, rtty_stream)
ORC_STR(orc_decoder_last_sentence, last_sentence)
ORC_STR(orc_decoder_sentence_log, sentence_log)
ORC_STR(orc_decoder_match_log, match_log)
ORC_STR(orc_decoder_chars_log, chars_log)
size_t orc_decoder_last_decimated(orc_decoder* d, const float** p) { *p = reinterpret_cast<const float*>(d->last_decimated.data()); return d->last_decimated.size(); }
size_t orc_decoder_last_filtered(orc_decoder* d, const float** p) { *p = reinterpret_cast<const float*>(d->last_filtered.data()); return d->last_filtered.size(); }
size_t orc_decoder_last_demod(orc_decoder* d, const float** p) { *p = d->last_demod.data(); return d->last_demod.size(); }
size_t orc_decoder_last_bits(orc_decoder* d, const uint8_t** p) { *p = d->last_bits.data(); return d->last_bits.size(); }
size_t orc_decoder_spectrum(orc_decoder* d, const float** p) { *p = reinterpret_cast<const float*>(d->freq_out.data()); return d->freq_out.size(); }
size_t orc_decoder_power(orc_decoder* d, const float** p) { return orc_afc_power(&d->afc, p); }
void orc_decoder_afc(orc_decoder* d, double* c, double* sh, double* nf, double* nv, int* gl, int* gr) { orc_afc_get(&d->afc, c, sh, nf, nv, gl, gr); }
size_t orc_decoder_fir_taps(orc_decoder* d, const float** t) { return orc_fir_taps(&d->fir, t); }
size_t orc_decoder_symex_held(orc_decoder* d) { return d->symex.v.size(); }
uint64_t orc_decoder_fft_count(orc_decoder* d) { return d->fft_count; }

/* ------------------------------------------------------------------------------------------------
 * CPU baseline driver for bench.py: `nthreads` decoders, one per thread, each fed the same chunk
 * sequence of its own stream exactly like DECODER_THREAD (main.cpp:240-245), `repeats` times with a
 * fresh decoder.  Returns wall seconds; the sentence log of each thread's first pass is written to
 * `sentences` separated by '\x1e'.
 * ---------------------------------------------------------------------------------------------- */
double orc_bench_run(const orc_bench_cfg* cfg, const float* const* iq, const uint32_t* chunk_idx, size_t nchunks,
                     size_t chunk, int repeats, int nthreads, char* sentences, size_t cap)
{
    std::vector<std::string> logs(nthreads);
    auto work = [&](int i) {
        for (int r = 0; r < repeats; ++r) {
            orc_decoder* d = orc_decoder_new();
            orc_decoder_setup_factor(d, cfg->factor);
            orc_decoder_baud(d, cfg->baud);
            orc_decoder_rtty(d, cfg->bits, cfg->stops);
            orc_decoder_lookup_mode(d, cfg->mathh_context);
            d->ungated = cfg->ungated != 0;
            d->with_fft = cfg->with_fft != 0;
            orc_decoder_lowpass_bw(d, cfg->lowpass_bw);       /* before any input: stores the value (Q10) */
            orc_decoder_lowpass_trans(d, cfg->lowpass_trans);
            for (size_t c = 0; c < nchunks; ++c) {
                orc_decoder_push(d, iq[i] + 2 * (size_t)chunk_idx[c] * chunk, chunk, cfg->sampling_rate);
                orc_decoder_process(d);
            }
            if (r == 0) logs[i] = d->sentence_log + '\x1f' + d->chars_log + '\x1f' + std::to_string(d->total_bits) + '\x1f' + std::to_string(d->demod_hash);
            orc_decoder_free(d);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < nthreads; ++i) th.emplace_back(work, i);
    for (auto& t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::string all;
    for (int i = 0; i < nthreads; ++i) { all += logs[i]; all.push_back('\x1e'); }
    if (sentences && cap) { const size_t n = std::min(cap - 1, all.size()); std::memcpy(sentences, all.data(), n); sentences[n] = 0; }
    return dt;
}
