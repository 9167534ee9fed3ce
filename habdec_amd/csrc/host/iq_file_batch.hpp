// Batched cf32 file ingest: S independent IQ files read in lock step into one push slab [S][stride].
//
// One file behaves like the reference's IQSource_File<float>::get (code/IQSource/IQSource_File.h:124-172): raw
// interleaved float32 I,Q, little-endian, no header (:156-157); a read returns what is left; the end of file is only
// NOTICED by the read that runs into it (std::ifstream's eofbit), and the call AFTER that one rewinds when looping
// (:141-155) -- so a file whose length is a whole number of requests delivers one empty read before it starts over,
// exactly like the reference.  The realtime throttle (:165-169) is optional.
//
// The batch adds what a batched decoder needs: every stream hands over a whole multiple of `granule` samples per round
// (the total decimation factor: Decoder::process consumes floor(n/D)*D samples and keeps the rest queued,
// Decoder.h:429-435); the remainder is carried in front of the stream's next round.
#pragma once
#include <stdint.h>

#include <chrono>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace hd {

class IqFile {
  public:
    bool open(const std::string& path, bool loop)
    {
        path_ = path; loop_ = loop;
        {   // IQSource_File::init: count_ = filesize / sizeof(complex<float>)
            std::ifstream in(path, std::ifstream::ate | std::ifstream::binary);
            if (!in.is_open()) return false;
            count_ = (uint64_t)in.tellg() / 8u;
        }
        f_.open(path, std::ios::binary);
        return f_.is_open();
    }
    uint64_t count() const { return count_; }
    bool loops() const { return loop_; }
    uint64_t rewinds() const { return rewinds_; }
    // IQSource_File::get: up to `want` complex samples into dst (2 floats each); returns the number read
    size_t get(float* dst, size_t want)
    {
        if (!f_.is_open()) return 0;
        if (f_.eof()) {
            if (!loop_) return 0;
            f_.clear();
            f_.seekg(0);
            ++rewinds_;
        }
        const uint64_t n = want < count_ ? want : count_;
        f_.read(reinterpret_cast<char*>(dst), (std::streamsize)(n * 8u));
        return (size_t)f_.gcount() / 8u;
    }

  private:
    std::ifstream f_;
    std::string path_;
    bool loop_ = false;
    uint64_t count_ = 0, rewinds_ = 0;
};

class IqFileBatch {
  public:
    // chunk: samples requested per stream and round (the reference asks for 65536, websocketServer/main.cpp:235)
    bool open(const std::vector<std::string>& paths, bool loop, uint32_t chunk, uint32_t granule, double realtime_rate = 0.0)
    {
        if (paths.empty() || !chunk || !granule || chunk < granule) return false;
        chunk_ = chunk; granule_ = granule; rate_ = realtime_rate;
        files_.clear(); carry_.clear(); carry_n_.assign(paths.size(), 0);
        for (const auto& p : paths) {
            files_.emplace_back(new IqFile);
            if (!files_.back()->open(p, loop)) { files_.clear(); return false; }
            carry_.emplace_back(2 * (size_t)granule, 0.0f);
        }
        return true;
    }
    // Rounds smaller than `n` samples (but not empty) are held back and joined with the stream's next round: the engine rejects
    // chunks shorter than its stage histories (undefined behaviour in the reference, Q4), and a file's tail may be that short.
    void set_min_take(uint32_t n)
    {
        min_take_ = n;
        for (auto& c : carry_) if (c.size() < 2 * ((size_t)n + granule_)) c.resize(2 * ((size_t)n + granule_), 0.0f);
    }
    bool looping() const { for (const auto& f : files_) if (f->loops()) return true; return false; }
    uint32_t streams() const { return (uint32_t)files_.size(); }
    uint32_t chunk() const { return chunk_; }
    const IqFile& file(uint32_t s) const { return *files_[s]; }
    // Fill slab[s*stride*2 ...] (stride in complex samples, >= chunk) and n_out[s] (a multiple of granule, <= chunk).
    // Returns the number of streams that read anything this round (0 = every file is exhausted and not looping).
    uint32_t next(float* slab, size_t stride, uint32_t* n_out)
    {
        uint32_t alive = 0;
        uint64_t most = 0;
        for (size_t s = 0; s < files_.size(); ++s) {
            float* dst = slab + s * stride * 2;
            const uint32_t c = carry_n_[s];
            if (c) std::memcpy(dst, carry_[s].data(), (size_t)c * 8);
            const size_t got = files_[s]->get(dst + 2 * (size_t)c, chunk_ - c);
            if (got) ++alive;
            most = got > most ? got : most;
            const uint32_t total = c + (uint32_t)got;
            uint32_t rem = total % granule_;
            if (total - rem && total - rem < min_take_ && total < chunk_) rem = total;     // too short for the engine: wait for more
            n_out[s] = total - rem;
            if (rem) std::memcpy(carry_[s].data(), dst + 2 * (size_t)(total - rem), (size_t)rem * 8);
            carry_n_[s] = rem;
        }
        if (rate_ > 0.0 && most)   // IQSource_File.h:165-169, once per round for the batch
            std::this_thread::sleep_for(std::chrono::duration<double, std::milli>((double)(size_t)((double)most / rate_ * 1000)));
        return alive;
    }

  private:
    std::vector<std::unique_ptr<IqFile>> files_;
    std::vector<std::vector<float>> carry_;
    std::vector<uint32_t> carry_n_;
    uint32_t chunk_ = 0, granule_ = 1, min_take_ = 0;
    double rate_ = 0.0;
};

}  // namespace hd

// handle of the C ABI (habdec_amd_host.h): shared by host_api.cpp (reader) and engine.cpp (the pump that feeds an engine)
struct hd_host_iqfiles { hd::IqFileBatch batch; };
