// TEST INFRASTRUCTURE ONLY: a C ABI over the reference's post-decode helpers, compiled together with the reference
// sources where they lie (/root/reference/code/common/{sentence_parse,GpsDistance}.cpp) into oracle/_ref/ -- see Makefile.
// Used to validate habdec_amd/csrc/host/telemetry.hpp and to generate tests/golden/telemetry.json.
#include <cstring>
#include <exception>
#include <string>

#include "common/GpsDistance.h"
#include "common/sentence_parse.h"

extern "C" {

int ref_parse_time(const char* text, int* h, int* m, float* s)
{
    try {
        auto r = habdec::parse_sentence_time(text);
        if (!r) return 0;
        *h = std::get<0>(*r); *m = std::get<1>(*r); *s = std::get<2>(*r);
        return 1;
    } catch (const std::exception&) { return -1; }
}

int ref_parse_gps_pos(const char* text, float* out)
{
    try { *out = habdec::parse_gps_pos(text); return 1; } catch (const std::exception&) { return -1; }
}

// callsign/frame/lat/lon/alt + the full timestamp string the reference builds from the system clock
int ref_parse_sentence(const char* text, char* callsign, size_t cap, int* frame, float* lat, float* lon, float* alt, char* datetime, size_t dcap)
{
    try {
        auto r = habdec::parse_sentence(text);
        if (!r) return 0;
        std::strncpy(callsign, r->payload_callsign.c_str(), cap - 1); callsign[cap - 1] = 0;
        std::strncpy(datetime, r->datetime.c_str(), dcap - 1); datetime[dcap - 1] = 0;
        *frame = r->frame; *lat = r->lat; *lon = r->lon; *alt = r->alt;
        return 1;
    } catch (const std::exception&) { return -1; }
}

size_t ref_timestamp_now(int h, int m, float s, char* buf, size_t cap)
{
    const std::string t = habdec::timestamp_from_HMS(h, m, s);
    std::strncpy(buf, t.c_str(), cap - 1); buf[cap - 1] = 0;
    return t.size();
}

void ref_gps_distance(double lat1, double lon1, double alt1, double lat2, double lon2, double alt2, double out[5])
{
    const habdec::GpsDistance d = habdec::CalcGpsDistance(lat1, lon1, alt1, lat2, lon2, alt2);
    out[0] = d.dist_line_; out[1] = d.dist_circle_; out[2] = d.dist_radians_; out[3] = d.elevation_; out[4] = d.bearing_;
}

}
