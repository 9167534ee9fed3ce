// habdec::Decoder<T> -- source-compatible facade over the MI355X engine (libhabdec_amd.so, include/habdec_amd.h).
//
// Drop-in for the reference's code/Decoder/Decoder.h:51-202: same public methods, typedefs and the three public
// std::function callback members, so code/websocketServer and code/fltkGUI compile against it unchanged
// (INTEGRATION.md shows the two-line CMake change).  One Decoder = one engine with a single stream in synchronous
// mode; batch users drive the C ABI directly.  What runs where:
//   pushSamples()  host: append to the input queue, latch the sampling rate of the first vector (Decoder.h:206-219)
//   process()      host: take floor(n/D)*D samples (Decoder.h:426-436) -> hd_process_host(): decimation, DC blocker,
//                  spectrum, AFC reductions, low-pass FIR, discriminator, symbol extractor on the GPU; AFC state
//                  machine, RTTY framing, sentence scan, CRC on the host; callbacks fire before process() returns.
// Differences from the reference, all outside the decoded data: no SSDV image side-channel (ssdv_callback_ never fires;
// the fsphil/ssdv sources are not part of the reference checkout), and
// changing the decimation factor or the sampling rate after data has flowed restarts the stream state.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <string>
#include <vector>

#include "IQVector.h"
#include "SpectrumInfo.h"
#include "habdec_amd.h"

namespace habdec {

template <typename TReal>
class Decoder {
public:
    using TValue = TReal;
    using TComplex = std::complex<TReal>;
    using TRVector = std::vector<TReal>;
    using TIQVector = habdec::IQVector<TReal>;
    // the reference exposes its stage classes here; they have no host-side equivalent any more
    struct TDecimator {};
    struct TFIR {};

    static_assert(std::is_same<TReal, float>::value, "the MI355X engine computes in float32 (the reference only instantiates Decoder<float>)");

    Decoder() = default;
    Decoder(const Decoder&) = delete;
    Decoder& operator=(const Decoder&) = delete;
    ~Decoder() { if (engine_) hd_engine_destroy(engine_); queue_.release(); }

    // ---- feed
    bool pushSamples(const TIQVector& chunk)
    {
        {
            std::lock_guard<std::mutex> l(queue_mtx_);
            queue_.append(chunk.data(), chunk.size());
        }
        if (!input_rate_) input_rate_ = static_cast<float>(chunk.samplingRate());   // init(const float), first vector only
        return true;
    }

    // ---- options (each forwards to the live engine; before the first process() the values are only stored)
    void lowpass_bw(float hz) { std::lock_guard<std::mutex> l(mtx_); lowpass_bw_ = hz; if (engine_) hd_stream_set_lowpass_bw(engine_, 0, hz); else note_no_input(); }
    float lowpass_bw() const { return lowpass_bw_; }
    void lowpass_trans(float t) { std::lock_guard<std::mutex> l(mtx_); lowpass_trans_ = t; if (engine_) hd_stream_set_lowpass_trans(engine_, 0, t); else note_no_input(); }
    float lowpass_trans() const { return lowpass_trans_; }
    void baud(double b) { std::lock_guard<std::mutex> l(mtx_); baud_ = b; if (engine_) hd_stream_set_baud(engine_, 0, b); }
    double baud() const { return baud_; }
    void rtty_bits(size_t n) { std::lock_guard<std::mutex> l(mtx_); bits_ = n; if (engine_) hd_stream_set_rtty(engine_, 0, (uint32_t)bits_, stops_); }
    size_t rtty_bits() const { return bits_; }
    void rtty_stops(float n) { std::lock_guard<std::mutex> l(mtx_); stops_ = n; if (engine_) hd_stream_set_rtty(engine_, 0, (uint32_t)bits_, stops_); }
    float rtty_stops() const { return stops_; }
    void dc_remove(bool on) { dc_remove_ = on; if (engine_) hd_stream_set_dc_remove(engine_, 0, on); }
    bool dc_remove() const { return dc_remove_; }

    size_t setupDecimationStagesFactor(const size_t factor)
    {
        // reference Decoder.h:268-332, including its edges: out of [1, 256] leaves the plan alone and returns the current factor (:272-276);
        // inside that range the plan is cleared FIRST (:281-284), so a factor without a table -- 1, or anything that is not a power of
        // two -- leaves "no stages, factor 1" behind and returns 0 (:317-319); a supported factor prints its stages (:324-329).
        if (factor < 1 || factor > 256) { std::cout << "Unsupported decimation factor: " << factor << std::endl; return getDecimationFactor(); }
        std::lock_guard<std::mutex> l(mtx_);
        const char* stages = stage_names(factor);
        factor_ = stages ? (int)factor : 1;
        drop_engine();                                       // new stage plan: histories restart (the reference clears its stages too)
        if (!stages) { std::cout << "Unsupported decimation factor: " << factor << std::endl; return 0; }
        std::cout << "Decoder::setupDecimationStagesFactor Decimation Stages: " << stages << std::endl;
        std::cout << "Decoder::setupDecimationStagesFactor Post Decimation Sampling Rate = " << getDecimatedSamplingRate()
                  << ", decimation factor = " << factor_ << std::endl;
        return factor_;
    }
    size_t setupDecimationStagesBW(const double max_rate)
    {
        // reference Decoder.h:336-412: the smallest power-of-two division (2 .. 256) that brings the rate to max_rate or below -- none at
        // all when the input rate already is (no stages, factor 1, returns 1).  The reference would stack a SECOND such division when
        // even /256 is not enough -- a plan of more than two stages, which the engine does not have: that request is refused (message,
        // current plan kept).
        if (!input_rate_) return 0;
        const double r = input_rate_;
        int div = 1;
        if (r > max_rate) {
            for (div = 2; div < 256; div *= 2)
                if (r / div <= max_rate) break;
            if (r / div > max_rate) {
                std::cout << "Decoder::setupDecimationStagesBW more than /256 needed for " << max_rate << " Hz: unsupported, keeping /" << factor_ << std::endl;
                return 0;
            }
        }
        std::lock_guard<std::mutex> l(mtx_);
        factor_ = div;
        drop_engine();
        std::cout << "Decoder::setupDecimationStagesBW Decimation Stages: " << (div > 1 ? stage_names((size_t)div) : "") << std::endl;
        std::cout << "Decoder::setupDecimationStagesBW Post Decimation Sampling Rate = " << r / div << ", decimation factor = " << factor_ << std::endl;
        return factor_;
    }

    // ---- results
    std::string getRTTY() { return text(&hd_stream_rtty); }
    std::string getLastSentence() { return text(&hd_stream_last_sentence); }

    // ---- info
    int getDecimationFactor() const { return factor_; }
    double getInputSamplingRate() const { return input_rate_; }
    double getDecimatedSamplingRate() const { return getInputSamplingRate() / getDecimationFactor(); }
    double getSymbolRate() const { return baud_; }

    // ---- GUI data
    size_t getBinsCount() const { return HD_FFT_BINS; }
    const TIQVector getFFT() const
    {
        TIQVector out;
        std::lock_guard<std::mutex> l(mtx_);
        if (!engine_) return out;
        out.resize(HD_FFT_BINS);
        const size_t n = hd_stream_spectrum(engine_, 0, reinterpret_cast<float*>(out.data()), HD_FFT_BINS);
        out.resize(n);
        out.samplingRate(getDecimatedSamplingRate());
        return out;
    }
    const TRVector getDemodulated() const { return floats(&hd_stream_demodulated, 1 << 16); }
    const TRVector getPowerSpectrum() const { return floats(&hd_stream_power, HD_FFT_BINS); }
    void getPeaks(int& pl, int& pr) { hd_afc_info a = afc(); pl = a.peak_left; pr = a.peak_right; }
    void getNoiseFloor(double& nf, double& nv) { hd_afc_info a = afc(); nf = a.noise_floor; nv = a.noise_variance; }
    double getShift() const { return afc().shift_hz; }
    double getFrequencyCorrection() const { return afc().frequency_correction; }
    void resetFrequencyCorrection(double correction)
    {
        std::lock_guard<std::mutex> l(mtx_);
        if (engine_) hd_stream_reset_frequency_correction(engine_, 0, correction);
    }
    SpectrumInfo<TReal> getSpectrumInfo()
    {
        SpectrumInfo<TReal> info;
        info = getPowerSpectrum();
        if (!info.size()) return info;
        info.min_ = *std::min_element(info.cbegin(), info.cend());
        info.max_ = *std::max_element(info.cbegin(), info.cend());
        int pl, pr;
        getPeaks(pl, pr);
        info.peak_left_ = std::abs(pl); info.peak_left_valid_ = pl > 0;
        info.peak_right_ = std::abs(pr); info.peak_right_valid_ = pr > 0;
        getNoiseFloor(info.noise_floor_, info.noise_variance_);
        info.sampling_rate_ = getDecimatedSamplingRate();
        info.shift_ = getShift();
        return info;
    }

    // ---- run
    void process()
    {
        if (!input_rate_) return;                            // nothing pushed yet
        std::vector<Event> fire;
        std::string chars_now;
        {
            std::lock_guard<std::mutex> l(mtx_);
            // The queue's lock is held across the engine calls, as the reference holds its one mutex across process() (Decoder.h:209, :423: pushSamples
            // waits while a round runs): the samples are read IN PLACE -- the queue lives in page-locked memory the GPU addresses (hd_pinned_alloc), the
            // engine is synchronous, and hd_process_host then hands the kernels the buffer itself instead of copying it.
            std::lock_guard<std::mutex> q(queue_mtx_);
            if ((int)queue_.size() < factor_) return;
            const size_t take = queue_.size() - queue_.size() % (size_t)factor_;
            // Inputs shorter than a decimation stage's history are undefined behaviour in the reference (Decimator.h:140-143) and
            // refused by the engine: leave them queued until the next push has made them long enough.
            if (take < (size_t)hd_min_chunk((uint32_t)factor_)) return;
            if (!ensure_engine(take)) { queue_.consume(take); return; }     // (no device: the samples are dropped, as before)
            events_ = &fire;
            size_t done = 0;
            while (done < take) {                            // one engine call per max_chunk (a single call in normal use)
                size_t n = std::min(take - done, (size_t)max_chunk_);
                const size_t left = take - done - n;
                if (left && left < (size_t)hd_min_chunk((uint32_t)factor_)) n -= (size_t)hd_min_chunk((uint32_t)factor_);   // never leave a too-short last call
                if (hd_process_host(engine_, reinterpret_cast<const float*>(queue_.front() + done), n, nullptr, (uint32_t)n) != HD_OK) {
                    std::cout << "habdec_amd: " << hd_last_error() << std::endl;
                    break;
                }
                done += n;
            }
            queue_.consume(take);
            events_ = nullptr;
            // character_callback_: at most every 250 ms, like the reference (Decoder.h:617-629)
            const auto now = std::chrono::steady_clock::now();
            if (!pending_chars_.empty() && now - last_char_cb_ > std::chrono::milliseconds(250)) {
                chars_now.swap(pending_chars_);
                last_char_cb_ = now;
            }
        }
        // The callbacks run here, on the decoder thread and before process() returns (Decoder.h:604-606, 625-626), but with no lock
        // held: they may call any getter of this decoder, as they can in the reference.
        for (const Event& ev : fire)
            if (sentence_callback_) sentence_callback_(ev.callsign, ev.data, ev.crc);
        if (!chars_now.empty() && character_callback_) character_callback_(chars_now);
    }
    void operator()() { process(); }

    bool livePrint() const { return live_print_; }
    void livePrint(bool on) { live_print_ = on; }
    std::string ssdvBaseFile() const { return ssdv_base_; }
    void ssdvBaseFile(const std::string& f) { ssdv_base_ = f; }

    // callback on each successful sentence decode: callsign, sentence data, CRC
    std::function<void(std::string, std::string, std::string)> sentence_callback_;
    // callback on decoded characters
    std::function<void(std::string)> character_callback_;
    // callback on each decoded SSDV packet: never fired by this implementation (see the header comment)
    std::function<void(std::string, int, std::vector<uint8_t>)> ssdv_callback_;

private:
    struct Event { std::string callsign, data, crc; };
    static void on_sentence(void* self, uint32_t, const char* call, const char* data, const char* crc)
    {
        auto* d = static_cast<Decoder*>(self);
        if (d->events_) d->events_->push_back(Event{call, data, crc});   // delivered by process() once its lock is released
    }
    // What the reference writes to stdout for EVERY sentence its scan finds, CRC-valid or not (Decoder.h:601 -> printHabhubSentence,
    // print_habhub_sentence.cpp:33-62, then `cout<<endl`): the line in magenta + " OK" or red + " ERR", and a running tally whose counters
    // are function-local statics there -- shared by all decoders of the process -- and are so here.
    static void on_match(void*, uint32_t, const char* call, const char* data, const char* crc, int crc_ok)
    {
        static int success = 0, failure = 0;
#ifdef __linux__
        const char *magenta = "\033[1;35m", *red = "\033[1;31m", *off = "\033[0m", *clear = "\033[2K";
#else
        const char *magenta = "", *red = "", *off = "", *clear = "";
#endif
        std::cout << clear << '\r' << "" << (crc_ok ? magenta : red) << call << "," << data << "*" << crc << (crc_ok ? " OK" : " ERR") << off;
        ++(crc_ok ? success : failure);
        std::cout << "\t\tOK:" << success << "  ERR:" << failure << "  Ratio:" << std::setprecision(2) << (float(success) / (success + failure));
        std::cout << std::endl;
    }
    static void on_chars(void* self, uint32_t, const char* chars, size_t n)
    {
        auto* d = static_cast<Decoder*>(self);
        d->pending_chars_.append(chars, n);
        if (d->live_print_) { std::cout.write(chars, (std::streamsize)n); std::cout.flush(); }
    }
    // "/32/2": the stages of a supported total factor as the reference prints them (Decoder.h:286-316, :324-327); nullptr where it has no table
    static const char* stage_names(size_t factor)
    {
        switch (factor) {
            case 256: return "/64/4"; case 128: return "/32/4"; case 64: return "/32/2"; case 32: return "/16/2";
            case 16: return "/8/2"; case 8: return "/8"; case 4: return "/4"; case 2: return "/2";
            default: return nullptr;
        }
    }
    void note_no_input() const { std::cout << "FirFilter::LP_BlackmanHarris No Input set." << std::endl; }
    void drop_engine() { if (engine_) { hd_engine_destroy(engine_); engine_ = nullptr; } }
    bool ensure_engine(size_t take)
    {
        if (engine_ && take <= 0xFFFFFFFFu) return true;
        if (engine_) return true;
        hd_engine_config c;
        hd_engine_config_default(&c);
        const char* dev = std::getenv("HABDEC_AMD_DEVICE");
        c.device = dev ? std::atoi(dev) : 0;
        c.n_streams = 1;
        // room for the usual 65536-sample reads, or for this (larger) first take: the engine's rings and LDS images grow with max_chunk / factor
        max_chunk_ = (uint32_t)std::max<size_t>((size_t)1 << 17, ((take + factor_ - 1) / factor_) * factor_);
        c.max_chunk = max_chunk_;
        c.sampling_rate = input_rate_;
        c.decimation = (uint32_t)factor_;
        c.baud = baud_; c.rtty_bits = (uint32_t)bits_; c.rtty_stops = stops_;
        c.lowpass_bw_hz = lowpass_bw_; c.lowpass_trans = lowpass_trans_; c.dc_remove = dc_remove_;
        if (hd_engine_create(&c, &engine_) != HD_OK) { std::cout << "habdec_amd: " << hd_last_error() << std::endl; engine_ = nullptr; return false; }
        hd_set_sentence_callback(engine_, &Decoder::on_sentence, this);
        hd_set_match_callback(engine_, &Decoder::on_match, this);
        hd_set_chars_callback(engine_, &Decoder::on_chars, this);
        return true;
    }
    std::string text(size_t (*fn)(hd_engine*, uint32_t, char*, size_t))
    {
        std::lock_guard<std::mutex> l(mtx_);
        if (!engine_) return {};
        std::string s(fn(engine_, 0, nullptr, 0), '\0');
        if (!s.empty()) { s.resize(s.size() + 1); fn(engine_, 0, &s[0], s.size()); s.resize(s.size() - 1); }
        return s;
    }
    TRVector floats(size_t (*fn)(hd_engine*, uint32_t, float*, size_t), size_t cap) const
    {
        std::lock_guard<std::mutex> l(mtx_);
        TRVector v;
        if (!engine_) return v;
        v.resize(cap);
        size_t n = fn(engine_, 0, v.data(), cap);
        if (n > cap) { v.resize(n); n = fn(engine_, 0, v.data(), n); }
        v.resize(n);
        return v;
    }
    hd_afc_info afc() const
    {
        hd_afc_info a;
        std::memset(&a, 0, sizeof(a));
        std::lock_guard<std::mutex> l(mtx_);
        if (engine_) hd_stream_afc(engine_, 0, &a);
        return a;
    }

    hd_engine* engine_ = nullptr;
    std::vector<Event>* events_ = nullptr;   // where the engine's sentence callback parks its events during process()
    mutable std::mutex mtx_;                 // process() vs. getters/setters from the server thread
    // iq_in_buffer_ (Decoder.h:206-219) as one linear buffer in page-locked, GPU-mapped memory (ordinary memory where that cannot be had): samples are
    // appended behind `tail`, a round consumes a prefix; what is left (less than one decimation factor, normally) is moved to the front when the space
    // behind `tail` runs out.  Always accessed under queue_mtx_.
    struct InputQueue {
        TComplex* p = nullptr;
        size_t cap = 0, head = 0, tail = 0;
        bool pinned = false;
        size_t size() const { return tail - head; }
        const TComplex* front() const { return p + head; }
        void consume(size_t n) { head += n; if (head == tail) head = tail = 0; }
        void release() { if (p) { if (pinned) hd_pinned_free(p); else std::free(p); } p = nullptr; cap = head = tail = 0; }
        void append(const TComplex* x, size_t n)
        {
            if (tail + n > cap) {
                const size_t keep = tail - head;
                // (the kept samples start at an EVEN index again: the engine's 16-byte loads want the base of a round aligned)
                if (keep + n <= cap) { std::memmove(p, p + head, keep * sizeof(TComplex)); }
                else {
                    size_t ncap = std::max<size_t>((size_t)1 << 18, cap);
                    while (ncap < keep + n) ncap *= 2;
                    bool npinned = true;
                    TComplex* np_ = static_cast<TComplex*>(hd_pinned_alloc(ncap * sizeof(TComplex)));
                    if (!np_) { npinned = false; np_ = static_cast<TComplex*>(std::malloc(ncap * sizeof(TComplex))); }
                    if (!np_) return;                        // out of memory: the push is lost
                    if (keep) std::memcpy(np_, p + head, keep * sizeof(TComplex));
                    if (p) { if (pinned) hd_pinned_free(p); else std::free(p); }
                    p = np_; cap = ncap; pinned = npinned;
                }
                head = 0; tail = keep;
            }
            std::memcpy(p + tail, x, n * sizeof(TComplex));
            tail += n;
        }
    };
    std::mutex queue_mtx_;
    InputQueue queue_;
    double input_rate_ = 0;
    int factor_ = 1;
    uint32_t max_chunk_ = 1u << 20;
    float lowpass_bw_ = 1500, lowpass_trans_ = 0.025f;
    double baud_ = 1;                         // SymbolExtractor default symbol rate
    size_t bits_ = 0;
    float stops_ = 0;
    bool dc_remove_ = false, live_print_ = true;
    std::string ssdv_base_, pending_chars_;
    std::chrono::steady_clock::time_point last_char_cb_ = std::chrono::steady_clock::now();
};

}  // namespace habdec
