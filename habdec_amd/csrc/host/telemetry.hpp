// Post-decode telemetry: what the reference does with a CRC-valid sentence before it is uploaded or
// shown -- field split, time-of-day, decimal / NMEA coordinates (code/common/sentence_parse.cpp:36-196) -- and the
// receiver-to-payload geometry (code/common/GpsDistance.cpp:21-84).  Host code: a fan-in of thousands of decoders still
// produces only a few sentences per second each.
//
// Where the reference lets an exception escape (std::stoi/std::stof on a non-number, std::string::at past the end) these
// functions report Status::Throws instead; everything else follows the reference value for value.
#pragma once
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace hd {
namespace telemetry {

enum class Status { Ok = 1, None = 0, Throws = -1 };

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// std::stof / std::stoi accept leading white space, a sign, then the longest numeric prefix; they throw when no
// conversion can be made (or the value is out of range).  strtof/strtol have the same prefix rules.
inline bool to_float(const std::string& s, float& out)
{
    const char* b = s.c_str();
    char* e = nullptr;
    errno = 0;
    const float v = std::strtof(b, &e);
    if (e == b || errno == ERANGE) return false;
    out = v;
    return true;
}
inline bool to_int(const std::string& s, int& out)
{
    const char* b = s.c_str();
    char* e = nullptr;
    errno = 0;
    const long v = std::strtol(b, &e, 10);
    if (e == b || errno == ERANGE || v < INT32_MIN || v > INT32_MAX) return false;
    out = (int)v;
    return true;
}

// parse_sentence_time (sentence_parse.cpp:47-68): the WHOLE string must be  dd [x] dd [x] [ dd [ . d+ ] ]  with x any one
// non-digit; seconds default to 0.
inline Status parse_time(const std::string& t, int& hours, int& minutes, float& seconds)
{
    size_t i = 0;
    const size_t n = t.size();
    auto two = [&](size_t at) { return at + 1 < n && is_digit(t[at]) && is_digit(t[at + 1]); };
    if (!two(i)) return Status::None;
    const std::string hh = t.substr(i, 2); i += 2;
    if (i < n && !is_digit(t[i])) ++i;                       // \D?
    if (!two(i)) return Status::None;
    const std::string mm = t.substr(i, 2); i += 2;
    // From here the pattern is  \D? (\d\d(\.\d+)?)?  up to the end of the string.
    std::string ss;
    size_t j = i;
    if (j < n && !is_digit(t[j])) ++j;                       // try with the separator taken ...
    auto seconds_at = [&](size_t at, std::string& out) -> bool {   // (\d\d(\.\d+)?)? followed by the end
        if (at == n) { out.clear(); return true; }
        if (!two(at)) return false;
        size_t k = at + 2;
        if (k < n && t[k] == '.') {
            size_t d = k + 1;
            while (d < n && is_digit(t[d])) ++d;
            if (d > k + 1 && d == n) { out = t.substr(at, d - at); return true; }
            return false;
        }
        if (k == n) { out = t.substr(at, 2); return true; }
        return false;
    };
    if (!seconds_at(j, ss)) {
        if (j == i || !seconds_at(i, ss)) return Status::None;       // ... then without it
    }
    hours = std::atoi(hh.c_str());
    minutes = std::atoi(mm.c_str());
    seconds = 0.0f;
    if (!ss.empty() && !to_float(ss, seconds)) return Status::Throws;
    return Status::Ok;
}

// parse_gps_pos (sentence_parse.cpp:106-143): decimal dd.dddd / ddd.dddd as is; NMEA ddmm.mmmm (latitude) or
// dddmm.mmmm (longitude) converted in float arithmetic; anything else is 0.
inline Status parse_gps_pos(const std::string& text, float& out)
{
    if (text.empty()) return Status::Throws;                 // .at(0)
    float sign = 1.0f;
    std::string coord = text;
    if (text[0] == '-') { sign = -1.0f; coord = text.substr(1); }
    const size_t dot = coord.find('.');
    if (dot == 2 || dot == 3) return to_float(text, out) ? Status::Ok : Status::Throws;
    if (dot == 4 || dot == 5) {
        float v;
        if (!to_float(coord, v)) return Status::Throws;
        const float degs = std::trunc(v / 100);
        const float mins = v - 100.0f * degs;
        out = sign * (degs + mins / 60.0f);
        return Status::Ok;
    }
    out = 0.0f;
    return Status::Ok;
}

struct Fields {
    std::string callsign;
    int frame = 0, hour = 0, minute = 0;
    float second = 0, lat = 0, lon = 0, alt = 0;
};

// parse_sentence (sentence_parse.cpp:146-196) without the wall-clock part: "callsign,id,time,lat,lon,alt[,...]".
inline Status parse_sentence(const std::string& sentence_without_crc, Fields& f)
{
    std::vector<std::string> tok;
    size_t start = 0, end;
    while ((end = sentence_without_crc.find(',', start)) != std::string::npos) { tok.push_back(sentence_without_crc.substr(start, end - start)); start = end + 1; }
    tok.push_back(sentence_without_crc.substr(start));
    if (tok.size() < 6) return Status::None;
    std::string cs = tok[0];
    size_t d = cs.find('$');
    if (d != std::string::npos) {                            // drop everything up to and including the run of '$'
        while (true) {
            if (d >= cs.size()) return Status::Throws;       // .at(size) when the callsign ends in '$'
            if (cs[d] != '$') break;
            ++d;
        }
        cs = cs.substr(d);
    }
    if (!to_int(tok[1], f.frame)) return Status::Throws;
    if (!to_float(tok[5], f.alt)) return Status::Throws;
    Status s = parse_gps_pos(tok[3], f.lat);
    if (s != Status::Ok) return s;
    s = parse_gps_pos(tok[4], f.lon);
    if (s != Status::Ok) return s;
    if (!f.lat && !f.lon) return Status::None;               // no GPS fix
    s = parse_time(tok[2], f.hour, f.minute, f.second);
    if (s != Status::Ok) return s;
    f.callsign = cs;
    return Status::Ok;
}

// Days since 1970-01-01 -> civil date (proleptic Gregorian).
inline void civil_from_days(int64_t z, int& y, unsigned& m, unsigned& d)
{
    z += 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const unsigned doe = (unsigned)(z - era * 146097);
    const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const unsigned mp = (5 * doy + 2) / 153;
    d = doy - (153 * mp + 2) / 5 + 1;
    m = mp < 10 ? mp + 3 : mp - 9;
    y = (int)(yoe + era * 400) + (m <= 2);
}

// timestamp_from_HMS (sentence_parse.cpp:73-100) with the clock passed in: today's UTC date in front of the telemetry's
// time of day; around midnight the date is moved a day when the two clocks sit on opposite sides of it.
inline std::string timestamp_from_hms(int64_t now_unix, int hour, int minute, float second)
{
    int64_t days = now_unix >= 0 ? now_unix / 86400 : -((-now_unix + 86399) / 86400);
    const int sys_hour = (int)((now_unix - days * 86400) / 3600);
    if (hour == 23 && sys_hour == 0) days -= 1;
    else if (hour == 0 && sys_hour == 23) days += 1;
    int y; unsigned m, d;
    civil_from_days(days, y, m, d);
    char sec[48];
    std::snprintf(sec, sizeof sec, "%g", (double)second);     // default ostream float formatting (precision 6) ...
    std::string ssec(sec);
    if (ssec.size() < 2) ssec.insert(0, 2 - ssec.size(), '0');   // ... under setfill('0') setw(2)
    char buf[96];
    std::snprintf(buf, sizeof buf, "%04d-%02u-%02uT%02d:%02d:%sZ", y, m, d, hour, minute, ssec.c_str());
    return buf;
}

struct Distance { double line, circle, radians, elevation_deg, bearing_deg; };

// CalcGpsDistance (GpsDistance.cpp:21-84): spherical Earth (r = 6371 km), degrees and metres in, degrees and metres out.
inline Distance gps_distance(double lat1, double lon1, double alt1, double lat2, double lon2, double alt2)
{
    const double r = 6371000.0, rad = M_PI / 180.0;
    lat1 *= rad; lat2 *= rad; lon1 *= rad; lon2 *= rad;
    const double dlon = lon2 - lon1;
    // sin and cos of one argument come from sincos(): that is what g++ -O3 (the reference's build) makes of the pairs, and
    // glibc's sincos is not bit-for-bit sin + cos in every case
    double s1, c1, s2, c2, sd, cd;
    ::sincos(lat1, &s1, &c1);
    ::sincos(lat2, &s2, &c2);
    ::sincos(dlon, &sd, &cd);
    const double east = c2 * sd;                                   // bearing: atan2(east, north) of the great-circle direction at point 1
    const double north = (c1 * s2) - (s1 * c2 * cd);
    double bearing = std::atan2(east, north);
    const double sin_arc = std::sqrt((east * east) + (north * north)); // central angle between the two points
    const double cos_arc = (s1 * s2) + (c1 * c2 * cd);
    const double angle = std::atan2(sin_arc, cos_arc);
    const double rad1 = r + alt1, rad2 = r + alt2;                  // distances from the Earth's centre
    double sn, cs;
    ::sincos(angle, &sn, &cs);
    const double rise = (cs * rad2) - rad1;                         // point 2 seen from point 1: up and along the local horizon
    const double run = sn * rad2;
    const double elevation = std::atan2(rise, run);
    const double line = std::sqrt((rad1 * rad1) + (rad2 * rad2) - 2 * rad2 * rad1 * cs);
    if (bearing < 0) bearing += 2 * M_PI;
    return {line, angle * r, angle, elevation / rad, bearing / rad};
}

}  // namespace telemetry
}  // namespace hd
