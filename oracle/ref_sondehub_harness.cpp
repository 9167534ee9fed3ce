/* ORACLE -- test infrastructure.  Pins the sondehub batch body against the reference's OWN serializer.
 *
 * sondehub_uploader.cpp itself cannot be compiled here (it includes <cpr/cpr.h> and <boost/...>, both absent; no stand-ins).  What
 * makes the body non-trivial -- key order, compact layout, string escaping, the float formatting of nlohmann::json's Grisu2 -- lives
 * in common/json.hpp, which IS in the checkout and is used here unmodified; likewise date.h for the timestamp format of
 * common/utc_now_iso.cpp.  The eleven field assignments of sondehub_uploader.cpp:55-65 and the three statements of
 * utc_now_iso.cpp:13-19 are restated below (that much of the pinning is by inspection). */
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <chrono>

#include "common/json.hpp"
#include "common/date.h"

extern "C" {

/* the loop body of SondeHubUploader::upload (sondehub_uploader.cpp:50-69) for n records */
size_t ref_sondehub_body(const char* uploader_callsign, const char* software_version, const char* upload_time, size_t n,
                         const char* const* payload_callsign, const char* const* time_received, const char* const* datetime,
                         const int* frame, const float* lat, const float* lon, const float* alt, char* out, size_t cap)
{
    std::stringstream s;
    s << "[";
    for (size_t i = 0; i < n; ++i) {
        using json = nlohmann::json;
        json tele_json;
        tele_json["uploader_callsign"] = std::string(uploader_callsign);
        tele_json["software_name"] = "habdec";
        tele_json["software_version"] = std::string(software_version).substr(0, 7);
        tele_json["time_received"] = std::string(time_received[i]);
        tele_json["upload_time"] = std::string(upload_time);
        tele_json["payload_callsign"] = std::string(payload_callsign[i]);
        tele_json["datetime"] = std::string(datetime[i]);
        tele_json["frame"] = frame[i];
        tele_json["lat"] = lat[i];
        tele_json["lon"] = lon[i];
        tele_json["alt"] = static_cast<int>(alt[i]);
        s << tele_json << ",";
    }
    std::string payload{s.str()};
    payload.at(payload.size() - 1) = ']';
    if (out && cap > payload.size()) { std::memcpy(out, payload.data(), payload.size()); out[payload.size()] = 0; }
    return payload.size();
}

/* utc_now_iso() (common/utc_now_iso.cpp:7-22) with the clock reading passed in */
size_t ref_utc_iso(int64_t unix_ns, char* out, size_t cap)
{
    using namespace std;
    using namespace chrono;
    using namespace date;
    const system_clock::time_point sys_now{duration_cast<system_clock::duration>(nanoseconds(unix_ns))};
    auto sys_YMD = year_month_day(floor<days>(sys_now));
    auto sys_HMS = make_time(sys_now - floor<days>(sys_now));
    stringstream ss;
    ss << sys_YMD << "T" << sys_HMS << "Z";
    const string r = ss.str();
    if (out && cap > r.size()) { memcpy(out, r.data(), r.size()); out[r.size()] = 0; }
    return r.size();
}

size_t ref_json_number(double v, char* out, size_t cap)
{
    nlohmann::json j = v;
    const std::string r = j.dump();
    if (out && cap > r.size()) { std::memcpy(out, r.data(), r.size()); out[r.size()] = 0; }
    return r.size();
}

}
