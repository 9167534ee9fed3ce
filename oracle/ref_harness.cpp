/*
 * ref_harness.cpp -- C-ABI driver around the REFERENCE's own stage classes.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libhabdec_ref.so, and only in
 * a container where /root/reference exists; the reference sources are compiled where they lie and
 * are never copied into this repository.  Used to (1) pin oracle/habdec_oracle.cpp stage by stage and
 * (2) generate tests/golden/ fixtures (tools/gen_golden.py).
 *
 * What is compiled from the reference, unmodified: Decimator.h, FirFilter.h, habdec_windows.h,
 * FSK2_Demod.h, SymbolExtractor.h, RTTY.h, AFC.h, Average.h, IQVector.h, filtercoef.h, CRC.cpp,
 * sentence_extract.cpp (+ common/console_colors.cpp which RTTY.h pulls in).
 * What is NOT: Decoder.h (needs fftw3.h and ssdv/ssdv.h, both absent from the image; writing stand-ins
 * for them is not allowed) and FFT.cpp (FFTW).  ref_chain_* therefore sequences the reference stage
 * objects with driver code written here after Decoder.h:416-638; its FFT is supplied by the caller.
 */
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "Decoder/IQVector.h"
#include "Decoder/Decimator.h"
#include "Decoder/FirFilter.h"
#include "Decoder/filtercoef.h"
#include "Decoder/FSK2_Demod.h"
#include "Decoder/SymbolExtractor.h"
#include "Decoder/RTTY.h"
#include "Decoder/AFC.h"
#include "Decoder/CRC.h"
#include "Decoder/sentence_extract.h"

using cf32 = std::complex<float>;
using RefDecim = habdec::Decimator<cf32, float>;
using RefFir = habdec::FirFilter<cf32, float>;

/* A worker thread per demodulator: FSK2_Demod keeps its previous sample in a `thread_local static`
 * (FSK2_Demod.h:35), so every independent stream needs a thread of its own to start fresh. */
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has = false, done = false, quit = false;
    Worker() : th([this] { loop(); }) {}
    ~Worker() { { std::lock_guard<std::mutex> l(m); quit = true; } cv.notify_all(); th.join(); }
    void loop()
    {
        std::unique_lock<std::mutex> l(m);
        for (;;) {
            cv.wait(l, [this] { return has || quit; });
            if (quit) return;
            job(); has = false; done = true; cv.notify_all();
        }
    }
    void run(std::function<void()> f)
    {
        std::unique_lock<std::mutex> l(m);
        job = std::move(f); has = true; done = false; cv.notify_all();
        cv.wait(l, [this] { return done; });
    }
};

extern "C" {

/* ---- tables ---- */
size_t ref_decim_taps(int total, int ratio, const float** taps)
{
#define T(N, R) if (total == N && ratio == R) { *taps = d_##N##_r_##R##_kernel; return d_##N##_r_##R##_len; }
    T(2, 2) T(4, 4) T(8, 8) T(16, 8) T(32, 16) T(64, 32) T(128, 32) T(256, 64)
#undef T
    return 0;
}

/* ---- Decimator ---- */
RefDecim* ref_decimator_new(int factor, const float* taps, size_t n) { return new RefDecim(factor, taps, n); }
void ref_decimator_free(RefDecim* d) { delete d; }
size_t ref_decimator_run(RefDecim* d, float* iq, size_t n)
{
    d->setInput(reinterpret_cast<cf32*>(iq), n);
    d->setOutput(reinterpret_cast<cf32*>(iq));
    return (*d)();
}

/* ---- FIR ---- */
RefFir* ref_fir_new() { return new RefFir; }
void ref_fir_free(RefFir* f) { delete f; }
struct FirIO { std::vector<cf32> dummy; };
void ref_fir_set_input(RefFir* f, const float* in, size_t n) { f->setInput(reinterpret_cast<const cf32*>(in), n); }
void ref_fir_design(RefFir* f, float w, float t) { f->LP_BlackmanHarris(w, t); }
size_t ref_fir_ntaps(RefFir* f) { return f->taps_size(); }
void ref_fir_run(RefFir* f, const float* in, size_t n, float* out)
{
    f->setInput(reinterpret_cast<const cf32*>(in), n);
    f->setOutput(reinterpret_cast<cf32*>(out));
    (*f)();
}

/* ---- demod (own thread per handle) ---- */
struct RefDemod { Worker w; };
RefDemod* ref_demod_new() { return new RefDemod; }
void ref_demod_free(RefDemod* d) { delete d; }
void ref_demod_run(RefDemod* d, const float* iq, size_t n, float* out)
{
    d->w.run([=] { habdec::FSK2_Demod<float>(reinterpret_cast<const cf32*>(iq), n, out); });
}

/* ---- symbol extractor ---- */
struct RefSymex { habdec::SymbolExtractor<float> s; };
RefSymex* ref_symex_new() { return new RefSymex; }
void ref_symex_free(RefSymex* s) { delete s; }
void ref_symex_rates(RefSymex* s, double fs, double baud) { s->s.samplingRate(fs); s->s.symbolRate(baud); }
void ref_symex_push(RefSymex* s, const float* v, size_t n) { s->s.pushSamples(std::vector<float>(v, v + n)); }
void ref_symex_run(RefSymex* s) { s->s(); }
size_t ref_symex_get(RefSymex* s, uint8_t* bits, size_t cap)
{
    std::vector<bool> b = s->s.get();
    size_t n = std::min(cap, b.size());
    for (size_t i = 0; i < n; ++i) bits[i] = b[i];
    return b.size();
}

/* ---- RTTY ---- */
struct RefRtty { habdec::RTTY<bool> r; };
RefRtty* ref_rtty_new(size_t nbits, float nstops) { auto* r = new RefRtty; r->r.ascii_bits(nbits); r->r.ascii_stops(nstops); return r; }
void ref_rtty_free(RefRtty* r) { delete r; }
void ref_rtty_push(RefRtty* r, const uint8_t* bits, size_t n)
{
    std::vector<bool> b(n);
    for (size_t i = 0; i < n; ++i) b[i] = bits[i] != 0;
    r->r.push(b);
}
size_t ref_rtty_run(RefRtty* r) { return r->r(); }
size_t ref_rtty_get(RefRtty* r, char* out, size_t cap)
{
    std::vector<char> c = r->r.get();
    size_t n = std::min(cap, c.size());
    std::memcpy(out, c.data(), n);
    return c.size();
}

/* ---- AFC ---- */
struct RefAfc { habdec::AFC<float> a; };
RefAfc* ref_afc_new() { return new RefAfc; }
void ref_afc_free(RefAfc* a) { delete a; }
void ref_afc_set_spectrum(RefAfc* a, const float* iq, size_t nbins, double rate)
{
    habdec::IQVector<float> v;
    v.resize(nbins);
    std::memcpy(v.data(), iq, nbins * sizeof(cf32));
    v.samplingRate(rate);
    a->a.setFftSamples(v);
}
double ref_afc_process(RefAfc* a) { return a->a(); }
void ref_afc_reset_correction(RefAfc* a, double c) { a->a.resetFrequencyCorrection(c); }
size_t ref_afc_power(RefAfc* a, const float** p) { *p = a->a.getPowerSpectrum().data(); return a->a.getPowerSpectrum().size(); }
void ref_afc_get(RefAfc* a, double* c, double* sh, double* nf, double* nv, int* gl, int* gr)
{
    *c = a->a.getFrequencyCorrection(); *sh = a->a.getShift();
    a->a.getNoiseFloor(*nf, *nv); a->a.getPeaks(*gl, *gr);
}

/* ---- text ---- */
void ref_crc16(const char* s, size_t n, char out4[5])
{
    std::string r = habdec::CRC(std::string(s, n));
    std::memcpy(out4, r.c_str(), 5);
}
int ref_extract_sentence(const char* stream, size_t n, char* callsign, char* data, char* crc, char* rest, size_t cap)
{
    auto res = habdec::extractSentence(std::string(stream, n));
    if (res["success"] != "OK") return 0;
    auto put = [cap](char* dst, const std::string& s) { size_t m = std::min(cap - 1, s.size()); std::memcpy(dst, s.data(), m); dst[m] = 0; };
    put(callsign, res["callsign"]); put(data, res["data"]); put(crc, res["crc"]); put(rest, res["stream"]);
    return 1;
}

/* ---- chain: the reference stage objects sequenced as Decoder::process() does ---- */
typedef void (*fft_fn)(const float*, float*, size_t);

struct RefChain {
    Worker w;                      /* all processing on one private thread (demod static) */
    std::vector<cf32> queue, temp, filtered;
    habdec::IQVector<float> decimated, freq_in, freq_out;
    double in_rate = 0;
    std::vector<RefDecim> stages;
    int factor = 1;
    bool dc = false;
    fft_fn fft = nullptr;
    habdec::AFC<float> afc;
    float lp_bw = 1500, lp_trans = 0.025;
    RefFir fir;
    std::vector<float> demod;
    habdec::SymbolExtractor<float> symex;
    habdec::RTTY<bool> rtty;
    std::string stream, last_sentence, sentence_log, match_log, chars_log;
    std::vector<cf32> last_decimated, last_filtered;
    std::vector<float> last_demod;
    std::vector<uint8_t> last_bits;
    double dec_rate() const { return in_rate / factor; }
};

RefChain* ref_chain_new(fft_fn f) { auto* c = new RefChain; c->fft = f; return c; }
void ref_chain_free(RefChain* c) { delete c; }
int ref_chain_setup_factor(RefChain* c, int f)
{
    c->stages.clear(); c->factor = 1;
    switch (f) {
    case 256: c->stages.emplace_back(64, d_256_r_64_kernel, d_256_r_64_len); c->stages.emplace_back(4, d_4_r_4_kernel, d_4_r_4_len); break;
    case 128: c->stages.emplace_back(32, d_128_r_32_kernel, d_128_r_32_len); c->stages.emplace_back(4, d_4_r_4_kernel, d_4_r_4_len); break;
    case 64:  c->stages.emplace_back(32, d_64_r_32_kernel, d_64_r_32_len); c->stages.emplace_back(2, d_2_r_2_kernel, d_2_r_2_len); break;
    case 32:  c->stages.emplace_back(16, d_32_r_16_kernel, d_32_r_16_len); c->stages.emplace_back(2, d_2_r_2_kernel, d_2_r_2_len); break;
    case 16:  c->stages.emplace_back(8, d_16_r_8_kernel, d_16_r_8_len); c->stages.emplace_back(2, d_2_r_2_kernel, d_2_r_2_len); break;
    case 8:   c->stages.emplace_back(8, d_8_r_8_kernel, d_8_r_8_len); break;
    case 4:   c->stages.emplace_back(4, d_4_r_4_kernel, d_4_r_4_len); break;
    case 2:   c->stages.emplace_back(2, d_2_r_2_kernel, d_2_r_2_len); break;
    default: return 0;
    }
    c->factor = f;
    return f;
}
void ref_chain_baud(RefChain* c, double b) { c->symex.symbolRate(b); }
void ref_chain_rtty(RefChain* c, size_t bits, float stops) { c->rtty.ascii_bits(bits); c->rtty.ascii_stops(stops); }
void ref_chain_dc_remove(RefChain* c, int on) { c->dc = on; }
void ref_chain_lowpass_bw(RefChain* c, float hz) { c->lp_bw = hz; c->fir.LP_BlackmanHarris(c->lp_bw / c->dec_rate(), c->lp_trans); }
void ref_chain_lowpass_trans(RefChain* c, float t) { c->lp_trans = t; c->fir.LP_BlackmanHarris(c->lp_bw / c->dec_rate(), c->lp_trans); }
void ref_chain_push(RefChain* c, const float* iq, size_t n, double rate)
{
    const cf32* x = reinterpret_cast<const cf32*>(iq);
    c->queue.insert(c->queue.end(), x, x + n);
    if (!c->in_rate) c->in_rate = (float)rate;
}

static void chain_process(RefChain* c)
{
    using namespace std;
    c->last_decimated.clear(); c->last_filtered.clear(); c->last_demod.clear(); c->last_bits.clear();
    if (!c->in_rate) return;
    if (int(c->queue.size()) < c->factor) return;
    const size_t take = c->queue.size() - (c->queue.size() % c->factor);
    c->temp.assign(c->queue.begin(), c->queue.begin() + take);
    c->queue.erase(c->queue.begin(), c->queue.begin() + take);
    size_t n = c->temp.size();
    for (auto& d : c->stages) { d.setInput(c->temp.data(), n); d.setOutput(c->temp.data()); n = d(); }
    c->temp.resize(n);
    if (c->dc) {
        cf32 w_prev = .97f * cf32(c->temp[0]);
        for (size_t i = 0; i < c->temp.size(); ++i) { cf32 w = c->temp[i] + .97f * w_prev; c->temp[i] = w - w_prev; w_prev = w; }
    }
    c->last_decimated = c->temp;
    c->decimated.insert(c->decimated.end(), c->temp.begin(), c->temp.end());
    c->decimated.samplingRate(c->dec_rate());
    const size_t bins = 4096;
    if (c->fft) {
        if (c->freq_in.size() < bins && c->temp.size()) {
            const size_t k = min(bins - c->freq_in.size(), c->temp.size());
            c->freq_in.insert(c->freq_in.end(), c->temp.begin(), c->temp.begin() + k);
        }
        c->freq_in.samplingRate(c->dec_rate());
        c->freq_out.samplingRate(c->dec_rate());
        if (c->freq_in.size() >= bins) {
            c->freq_out.resize(bins);
            c->fft(reinterpret_cast<const float*>(c->freq_in.data()), reinterpret_cast<float*>(c->freq_out.data()), bins);
            c->freq_in.clear();
        }
    }
    if (c->decimated.size() < 256) return;
    if (c->fft) {
        if (c->freq_out.size() == bins) c->afc.setFftSamples(c->freq_out);
        c->afc();
    }
    if (c->dec_rate() > 4 * 40e3) { c->temp.clear(); c->decimated.clear(); return; }
    const size_t m = c->decimated.size() - c->decimated.size() % 256;
    c->filtered.resize(m);
    c->fir.setInput(c->decimated.data(), m);
    c->fir.setOutput(c->filtered.data());
    c->fir.LP_BlackmanHarris(c->lp_bw / c->dec_rate(), c->lp_trans);
    c->fir();
    c->decimated.erase(c->decimated.begin(), c->decimated.begin() + m);
    c->last_filtered = c->filtered;
    c->demod.resize(c->filtered.size());
    habdec::FSK2_Demod<float>(c->filtered.data(), c->filtered.size(), c->demod.data());
    c->last_demod = c->demod;
    c->symex.samplingRate(c->dec_rate());
    c->symex.pushSamples(c->demod);
    c->symex();
    vector<bool> symbols = c->symex.get();
    for (bool b : symbols) c->last_bits.push_back(b);
    if (symbols.size()) { c->rtty.push(symbols); c->rtty(); }
    if (!c->rtty.size()) return;
    vector<char> raw = c->rtty.get();
    string printable;
    for (char ch : raw) if (isprint(ch) || ch == '\n') printable.push_back(ch);
    c->stream += printable;
    c->chars_log += printable;
    if (c->stream.size() > 20) {
        for (;;) {
            auto res = habdec::extractSentence(c->stream);
            if (res["success"] != "OK") break;
            c->stream = res["stream"];
            c->last_sentence = res["callsign"] + "," + res["data"] + "*" + res["crc"];
            c->match_log += c->last_sentence + "\n";
            if (res["crc"] == habdec::CRC(res["callsign"] + "," + res["data"])) c->sentence_log += c->last_sentence + "\n";
        }
    }
    if (c->stream.size() > 1000) c->stream.erase(0, c->stream.rfind('$'));
}
void ref_chain_process(RefChain* c) { c->w.run([c] { chain_process(c); }); }

#define STR(name, field) size_t name(RefChain* c, const char** s) { *s = c->field.c_str(); return c->field.size(); }
STR(ref_chain_rtty_stream, stream)
STR(ref_chain_last_sentence, last_sentence)
STR(ref_chain_sentence_log, sentence_log)
STR(ref_chain_match_log, match_log)
STR(ref_chain_chars_log, chars_log)
size_t ref_chain_last_decimated(RefChain* c, const float** p) { *p = reinterpret_cast<const float*>(c->last_decimated.data()); return c->last_decimated.size(); }
size_t ref_chain_last_filtered(RefChain* c, const float** p) { *p = reinterpret_cast<const float*>(c->last_filtered.data()); return c->last_filtered.size(); }
size_t ref_chain_last_demod(RefChain* c, const float** p) { *p = c->last_demod.data(); return c->last_demod.size(); }
size_t ref_chain_last_bits(RefChain* c, const uint8_t** p) { *p = c->last_bits.data(); return c->last_bits.size(); }
size_t ref_chain_power(RefChain* c, const float** p) { *p = c->afc.getPowerSpectrum().data(); return c->afc.getPowerSpectrum().size(); }
void ref_chain_afc(RefChain* c, double* corr, double* sh, double* nf, double* nv, int* gl, int* gr)
{
    *corr = c->afc.getFrequencyCorrection(); *sh = c->afc.getShift();
    c->afc.getNoiseFloor(*nf, *nv); c->afc.getPeaks(*gl, *gr);
}
size_t ref_chain_fir_ntaps(RefChain* c) { return c->fir.taps_size(); }

} /* extern "C" */
