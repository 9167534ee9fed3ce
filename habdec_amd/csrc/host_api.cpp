// C ABI over the host-side routines of the engine (include/habdec_amd_host.h).  No HIP calls in here.
#include <algorithm>
#include <cstring>
#include <string>

#include "../../include/habdec_amd_host.h"
#include "host/afc_tracker.hpp"
#include "host/decim_plan.hpp"
#include "host/fir_design.hpp"
#include "host/gui_payload.hpp"
#include "host/telemetry.hpp"
#include "host/sondehub.hpp"
#include "host/text_stage.hpp"
#include "host/iq_file_batch.hpp"
#include "kernels/exact_math.h"

struct hd_host_rtty { hd::RttyFramer f; };
struct hd_host_text { hd::TextStage t; };
struct hd_host_afc { hd::AfcTracker a; };

static void put(char* dst, size_t cap, const std::string& s)
{
    if (!dst || !cap) return;
    const size_t n = std::min(cap - 1, s.size());
    std::memcpy(dst, s.data(), n);
    dst[n] = 0;
}

extern "C" {

int hd_host_decim_plan(unsigned total, int ratio[2], unsigned ntaps[2])
{
    std::vector<hd::DecimStage> st;
    if (!hd::decim_plan(total, st)) return -1;
    for (size_t i = 0; i < st.size(); ++i) { ratio[i] = st[i].ratio; ntaps[i] = (unsigned)st[i].taps.size(); }
    return (int)st.size();
}
size_t hd_host_decim_taps(unsigned total, int stage, float* taps, size_t cap)
{
    std::vector<hd::DecimStage> st;
    if (!hd::decim_plan(total, st) || stage < 0 || (size_t)stage >= st.size()) return 0;
    const auto& t = st[stage].taps;
    std::memcpy(taps, t.data(), std::min(cap, t.size()) * sizeof(float));
    return t.size();
}
size_t hd_host_lowpass_design(float cutoff_rel, float transition, size_t batch, size_t prev_ntaps, int float_trig, float* taps, size_t cap)
{
    hd::LowpassDesigner d;
    d.float_trig = float_trig != 0;
    d.batch = batch;
    d.taps.assign(prev_ntaps, 0.0f);
    if (!d.design(cutoff_rel, transition)) return 0;
    std::memcpy(taps, d.taps.data(), std::min(cap, d.taps.size()) * sizeof(float));
    return d.taps.size();
}

hd_host_rtty* hd_host_rtty_new(size_t nbits, float nstops) { auto* r = new hd_host_rtty; r->f.nbits = nbits; r->f.nstops = nstops; return r; }
void hd_host_rtty_free(hd_host_rtty* r) { delete r; }
size_t hd_host_rtty_pending(const hd_host_rtty* r) { return r ? r->f.pending() : 0; }
size_t hd_host_rtty_push_run(hd_host_rtty* r, const uint8_t* bits, size_t n, char* out, size_t cap)
{
    r->f.push_bits(bits, n);
    std::string s;
    r->f.frame(s);
    std::memcpy(out, s.data(), std::min(cap, s.size()));
    return s.size();
}

void hd_host_crc16(const char* s, size_t n, char out4[5])
{
    const std::string r = hd::crc16_ccitt_hex(std::string(s, n));
    std::memcpy(out4, r.c_str(), 5);
}
int hd_host_extract_sentence(const char* stream, size_t n, char* callsign, char* data, char* crc, char* rest, size_t cap)
{
    hd::SentenceMatch m;
    if (!hd::extract_sentence(std::string(stream, n), m)) return 0;
    put(callsign, cap, m.callsign); put(data, cap, m.data); put(crc, cap, m.crc); put(rest, cap, m.rest);
    return 1;
}

hd_host_text* hd_host_text_new(size_t nbits, float nstops) { auto* t = new hd_host_text; t->t.framer.nbits = nbits; t->t.framer.nstops = nstops; return t; }
void hd_host_text_free(hd_host_text* t) { delete t; }
void hd_host_text_push_bits(hd_host_text* t, const uint8_t* bits, size_t n)
{
    if (n) t->t.framer.push_bits(bits, n);
    t->t.run(n != 0, [](const hd::SentenceMatch&) {});
}
size_t hd_host_text_get(hd_host_text* t, int which, char* buf, size_t cap)
{
    const std::string* s = &t->t.stream;
    if (which == 1) s = &t->t.last_sentence; else if (which == 2) s = &t->t.ok_log; else if (which == 3) s = &t->t.match_log; else if (which == 4) s = &t->t.char_log;
    put(buf, cap, *s);
    return s->size();
}

hd_host_afc* hd_host_afc_new(void) { return new hd_host_afc; }
void hd_host_afc_free(hd_host_afc* a) { delete a; }
void hd_host_afc_step(hd_host_afc* a, int have, int valid, int p1, int p2, float pw1, float pw2, double mean, double sigma, size_t bins, double rate)
{
    hd::SpectrumStats st{};
    st.valid = valid; st.peak1 = p1; st.peak2 = p2; st.power1 = pw1; st.power2 = pw2; st.mean = mean; st.sigma = sigma;
    a->a.step(have != 0, st, bins, rate);
}
void hd_host_afc_reset(hd_host_afc* a, double c, size_t bins, double rate) { a->a.reset(c, bins, rate); }
void hd_host_afc_get(hd_host_afc* a, double* c, double* sh, double* nf, double* ns, int* pl, int* pr)
{
    *c = a->a.correction; *sh = a->a.shift_hz; *nf = a->a.noise_floor; *ns = a->a.noise_sigma; *pl = a->a.gui_left; *pr = a->a.gui_right;
}

void hd_host_atan2f(const float* y, const float* x, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = hd::exact_atan2f(y[i], x[i]); }
void hd_host_discriminate(const float* iq, size_t n, float pr, float pi, float* out)
{
    for (size_t i = 0; i < n; ++i) { out[i] = hd::discriminate(iq[2 * i], iq[2 * i + 1], pr, pi); pr = iq[2 * i]; pi = iq[2 * i + 1]; }
}

/* ---- GUI payloads (habdec_ws_protocol.cpp:338-429, NetTransport.h, CompressedVector.cpp; see host/gui_payload.hpp) ---- */
size_t hd_host_spectrum_payload(const float* bins, size_t n, double noise_floor, double noise_variance, double sampling_rate, double shift,
                                int peak_left, int peak_right, float zoom, int resolution, int type_size, uint8_t* out, size_t cap,
                                size_t* bins_sent)
{
    if (bins_sent) *bins_sent = 0;
    if (!bins && n) return 0;
    hd::gui::SpectrumMeta m;
    m.noise_floor = noise_floor; m.noise_variance = noise_variance; m.sampling_rate = sampling_rate; m.shift = shift;
    // Decoder::getSpectrumInfo (Decoder.h:823-828): the sign of a peak carries its validity
    m.peak_left = peak_left < 0 ? -peak_left : peak_left; m.peak_left_valid = peak_left > 0;
    m.peak_right = peak_right < 0 ? -peak_right : peak_right; m.peak_right_valid = peak_right > 0;
    std::vector<uint8_t> o;
    const size_t sent = hd::gui::spectrum_payload(std::vector<float>(bins, bins + n), m, zoom, resolution, type_size, o);
    if (bins_sent) *bins_sent = sent;
    if (out && cap >= o.size()) std::memcpy(out, o.data(), o.size());
    return o.size();
}
size_t hd_host_demod_payload(const float* trace, size_t n, int resolution, int type_size, uint8_t* out, size_t cap, size_t* sent_out)
{
    if (sent_out) *sent_out = 0;
    if (!trace && n) return 0;
    std::vector<uint8_t> o;
    const size_t sent = hd::gui::demod_payload(std::vector<float>(trace, trace + n), resolution, type_size, o);
    if (sent_out) *sent_out = sent;
    if (out && cap >= o.size()) std::memcpy(out, o.data(), o.size());
    return o.size();
}

/* ---- post-decode telemetry (sentence_parse.cpp, GpsDistance.cpp; see host/telemetry.hpp) ---- */
int hd_host_parse_time(const char* text, int* hour, int* minute, float* second)
{
    if (!text || !hour || !minute || !second) return -1;
    return (int)hd::telemetry::parse_time(text, *hour, *minute, *second);
}
int hd_host_parse_gps_pos(const char* text, float* out)
{
    if (!text || !out) return -1;
    return (int)hd::telemetry::parse_gps_pos(text, *out);
}
int hd_host_parse_sentence(const char* sentence_without_crc, hd_host_telemetry* out)
{
    if (!sentence_without_crc || !out) return -1;
    hd::telemetry::Fields f;
    const auto st = hd::telemetry::parse_sentence(sentence_without_crc, f);
    if (st != hd::telemetry::Status::Ok) return (int)st;
    std::memset(out, 0, sizeof *out);
    std::strncpy(out->callsign, f.callsign.c_str(), sizeof out->callsign - 1);
    out->frame = f.frame; out->hour = f.hour; out->minute = f.minute; out->second = f.second;
    out->lat = f.lat; out->lon = f.lon; out->alt = f.alt;
    return 1;
}
size_t hd_host_timestamp_from_hms(int64_t now_unix, int hour, int minute, float second, char* buf, size_t cap)
{
    const std::string s = hd::telemetry::timestamp_from_hms(now_unix, hour, minute, second);
    if (buf && cap) { const size_t n = std::min(cap - 1, s.size()); std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size();
}
void hd_host_gps_distance(double lat1, double lon1, double alt1, double lat2, double lon2, double alt2, double out[5])
{
    const auto d = hd::telemetry::gps_distance(lat1, lon1, alt1, lat2, lon2, alt2);
    out[0] = d.line; out[1] = d.circle; out[2] = d.radians; out[3] = d.elevation_deg; out[4] = d.bearing_deg;
}

/* ---- sondehub upload batch (sondehub_uploader.cpp:14-69, main.cpp:286-306; see host/sondehub.hpp) ---- */
struct hd_host_sondehub {
    hd::SondehubBatch b;
    std::string pending;      // a body that did not fit the caller's buffer: handed out again by the next take
    size_t pending_n = 0;
    hd_host_sondehub(const char* u, const char* v) : b(u ? u : "", v ? v : "") {}
};
hd_host_sondehub* hd_host_sondehub_new(const char* uploader_callsign, const char* software_version) { return new hd_host_sondehub(uploader_callsign, software_version); }
void hd_host_sondehub_free(hd_host_sondehub* h) { delete h; }
int hd_host_sondehub_push_sentence(hd_host_sondehub* h, uint32_t stream, const char* callsign, const char* data, int64_t now_unix_ns)
{
    if (!h || !callsign || !data) return -1;
    return h->b.push_sentence(stream, callsign, data, now_unix_ns);
}
int hd_host_sondehub_push(hd_host_sondehub* h, const char* payload_callsign, const char* time_received, const char* datetime, int frame, float lat, float lon, float alt)
{
    if (!h || !payload_callsign || !time_received || !datetime) return -1;
    hd::SondeRecord r;
    r.payload_callsign = payload_callsign; r.time_received = time_received; r.datetime = datetime; r.frame = frame; r.lat = lat; r.lon = lon; r.alt = alt;
    h->b.push(r);
    return 1;
}
size_t hd_host_sondehub_size(const hd_host_sondehub* h) { return h ? h->b.size() : 0; }
size_t hd_host_sondehub_take(hd_host_sondehub* h, int64_t now_unix_ns, char* out, size_t cap, size_t* n_records)
{
    if (!h) return 0;
    if (h->pending.empty()) h->pending = h->b.take(now_unix_ns, &h->pending_n);
    const size_t n = h->pending.size();
    if (n_records) *n_records = h->pending_n;
    if (out && cap > n) { std::memcpy(out, h->pending.data(), n); out[n] = 0; h->pending.clear(); h->pending_n = 0; }
    return n;
}
size_t hd_host_utc_iso(int64_t unix_ns, char* buf, size_t cap)
{
    const std::string s = hd::utc_iso_ns(unix_ns);
    if (buf && cap) { const size_t n = std::min(cap - 1, s.size()); std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size();
}
size_t hd_host_json_number(double v, char* buf, size_t cap)
{
    const std::string s = hd::json_number(v);
    if (buf && cap) { const size_t n = std::min(cap - 1, s.size()); std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size();
}

/* ---- batched cf32 file ingest (IQSource_File.h:124-172 per file; see host/iq_file_batch.hpp) ---- */
hd_host_iqfiles* hd_host_iqfiles_open(const char* const* paths, uint32_t n_files, int loop, uint32_t chunk, uint32_t granule, double realtime_rate)
{
    if (!paths || !n_files) return nullptr;
    std::vector<std::string> v;
    for (uint32_t i = 0; i < n_files; ++i) { if (!paths[i]) return nullptr; v.emplace_back(paths[i]); }
    auto* h = new hd_host_iqfiles;
    if (!h->batch.open(v, loop != 0, chunk, granule, realtime_rate)) { delete h; return nullptr; }
    return h;
}
void hd_host_iqfiles_close(hd_host_iqfiles* h) { delete h; }
uint32_t hd_host_iqfiles_streams(const hd_host_iqfiles* h) { return h ? h->batch.streams() : 0; }
uint64_t hd_host_iqfiles_count(const hd_host_iqfiles* h, uint32_t stream) { return (h && stream < h->batch.streams()) ? h->batch.file(stream).count() : 0; }
uint64_t hd_host_iqfiles_rewinds(const hd_host_iqfiles* h, uint32_t stream) { return (h && stream < h->batch.streams()) ? h->batch.file(stream).rewinds() : 0; }
uint32_t hd_host_iqfiles_next(hd_host_iqfiles* h, float* slab, size_t stride, uint32_t* n_per_stream)
{
    if (!h || !slab || !n_per_stream || stride < h->batch.chunk()) return 0;
    return h->batch.next(slab, stride, n_per_stream);
}

}  // extern "C"
