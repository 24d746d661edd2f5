// hzsdr.hpp -- C++ host-side mirror of the reference's Go interfaces for the hot
// path, over the C ABI of include/hzsdr.h (the reference is compiled code and the
// image has no Go toolchain, so the compiled-language host layer is C++; the cgo
// package in INTEGRATION.md is this file transliterated).
//
// Names follow the reference: sdr::Reader (reader.go:39-51), sdr::ReadFull
// (reader.go:72), stream::ReadTransformer (stream/read_transformer.go:45-137),
// stream::ConvertReader / DecimateReader / DownsampleReader / ShiftReader / Gain
// (stream/*.go).  Errors are the reference's sentinels, thrown as hzsdr::Error
// carrying the status code.  Buffers are host memory (a HZSDR_MEM_HOST context),
// as Go slices would be.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hzsdr.h"

namespace hzsdr {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};
struct Eof : std::exception {};

inline void check(hzsdr_ctx *ctx, int rc) {
    if (rc == HZSDR_OK) return;
    std::string m = hzsdr_strerror(rc);
    if (ctx && *hzsdr_last_error(ctx)) m += std::string(": ") + hzsdr_last_error(ctx);
    throw Error(rc, m);
}

// sdr.Samples: a typed view over caller-owned memory (iq.go:59-87)
struct Samples {
    int format = 0;
    void *data = nullptr;
    size_t length = 0;  // IQ samples
    size_t size() const { return length * (size_t)hzsdr_format_size(format); }
    Samples slice(size_t lo, size_t hi) const {
        return Samples{format, (char *)data + lo * (size_t)hzsdr_format_size(format), hi - lo};
    }
};

// sdr.MakeSamples (iq.go:128-141): owning storage + its view
struct Buffer {
    std::vector<unsigned char> bytes;
    Samples view;
    Buffer(int format, size_t n) {
        if (hzsdr_format_size(format) == 0) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
        bytes.assign(n * (size_t)hzsdr_format_size(format), 0);
        view = Samples{format, bytes.data(), n};
    }
};

class Context {
public:
    explicit Context(int device = 0) { check(nullptr, hzsdr_open(device, HZSDR_MEM_HOST, &c_)); }
    ~Context() { if (c_) hzsdr_close(c_); }
    Context(const Context &) = delete;
    hzsdr_ctx *raw() const { return c_; }

    // sdr.ConvertBuffer (conv.go:55)
    size_t ConvertBuffer(Samples dst, Samples src) const {
        size_t n = 0;
        check(c_, hzsdr_convert(c_, dst.format, dst.data, dst.length, src.format, src.data, src.length, &n));
        return n;
    }
    // SamplesC64.Scale / Multiply / Add (iq_c64.go:122-136)
    void Scale(Samples s, float r) const { need_c64(s); check(c_, hzsdr_scale(c_, s.data, s.length, r)); }
    void Multiply(Samples s, float re, float im) const { need_c64(s); check(c_, hzsdr_rotate(c_, s.data, s.length, re, im)); }
    void Add(Samples a, Samples b) const { check(c_, hzsdr_add(c_, a.data, a.length, b.data, b.length, a.data, a.length)); }
    // stream.DecimateBuffer / DownsampleBuffer
    size_t DecimateBuffer(Samples to, Samples from, unsigned factor, int64_t offset) const {
        size_t n = 0;
        check(c_, hzsdr_decimate(c_, to.format, to.data, to.length, from.format, from.data, from.length, factor, offset, &n));
        return n;
    }
    size_t DownsampleBuffer(Samples to, Samples from, unsigned factor, int64_t offset) const {
        size_t n = 0;
        check(c_, hzsdr_downsample(c_, to.format, to.data, to.length, from.format, from.data, from.length, factor, offset, &n));
        return n;
    }

private:
    static void need_c64(const Samples &s) {
        if (s.format != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
    }
    hzsdr_ctx *c_ = nullptr;
};

// sdr.Reader (reader.go:39-51)
struct Reader {
    virtual ~Reader() = default;
    virtual size_t Read(Samples s) = 0;
    virtual int SampleFormat() const = 0;
    virtual unsigned SampleRate() const = 0;
};
using ReaderPtr = std::shared_ptr<Reader>;

// sdr.ReadFull (reader.go:72-117).  Throws Eof with nothing read, Error(short) after a partial read.
inline size_t ReadFull(Reader &r, Samples buf, size_t *partial = nullptr) {
    size_t n = 0;
    while (n < buf.length) {
        try {
            n += r.Read(buf.slice(n, buf.length));
        } catch (const Eof &) {
            if (partial) *partial = n;
            if (n > 0) throw Error(-1, "sdr: unexpected EOF");
            throw;
        }
    }
    return n;
}

// an in-memory source: mock.Sdr's Rx / a written sdr.Pipe in the reference's tests
class BufferReader : public Reader {
public:
    BufferReader(Samples data, unsigned rate, size_t max_read = 0) : d_(data), rate_(rate), max_(max_read) {}
    size_t Read(Samples s) override {
        if (s.format != d_.format) throw Error(HZSDR_ERR_FORMAT_MISMATCH, hzsdr_strerror(HZSDR_ERR_FORMAT_MISMATCH));
        size_t left = d_.length - pos_;
        if (left == 0) throw Eof();
        size_t n = left < s.length ? left : s.length;
        if (max_ && n > max_) n = max_;
        std::memcpy(s.data, d_.slice(pos_, pos_ + n).data, n * (size_t)hzsdr_format_size(d_.format));
        pos_ += n;
        return n;
    }
    int SampleFormat() const override { return d_.format; }
    unsigned SampleRate() const override { return rate_; }

private:
    Samples d_;
    unsigned rate_;
    size_t max_, pos_ = 0;
};

namespace stream {

constexpr size_t kBlock = 32 * 1024;  // stream/convert.go:43-44

// stream.ReadTransformer (stream/read_transformer.go:45-137), pull-driven
class ReadTransformer : public Reader {
public:
    using Proc = std::function<size_t(Samples in, Samples out)>;
    ReadTransformer(ReaderPtr in, size_t in_len, size_t out_len, int out_fmt, unsigned out_rate, Proc p)
        : in_(std::move(in)), ibuf_(in_->SampleFormat(), in_len), obuf_(out_fmt, out_len), fmt_(out_fmt),
          rate_(out_rate), proc_(std::move(p)) {}
    size_t Read(Samples s) override {
        if (s.format != fmt_) throw Error(HZSDR_ERR_FORMAT_MISMATCH, hzsdr_strerror(HZSDR_ERR_FORMAT_MISMATCH));
        if (avail_ == 0) {
            if (done_) throw Eof();
            try {
                size_t inn = ReadFull(*in_, ibuf_.view);
                avail_ = proc_(ibuf_.view.slice(0, inn), obuf_.view);
                off_ = 0;
            } catch (const Eof &) { done_ = true; throw; }
            catch (const Error &e) { done_ = true; if (e.status == -1) throw Eof(); throw; }
        }
        size_t n = avail_ < s.length ? avail_ : s.length;
        std::memcpy(s.data, obuf_.view.slice(off_, off_ + n).data, n * (size_t)hzsdr_format_size(fmt_));
        off_ += n;
        avail_ -= n;
        return n;
    }
    int SampleFormat() const override { return fmt_; }
    unsigned SampleRate() const override { return rate_; }

private:
    ReaderPtr in_;
    Buffer ibuf_, obuf_;
    int fmt_;
    unsigned rate_;
    Proc proc_;
    size_t avail_ = 0, off_ = 0;
    bool done_ = false;
};

// stream.ConvertReader (stream/convert.go:37-51)
inline ReaderPtr ConvertReader(const Context &x, ReaderPtr in, int to) {
    unsigned rate = in->SampleRate();
    return std::make_shared<ReadTransformer>(std::move(in), kBlock, kBlock, to, rate,
                                             [&x](Samples i, Samples o) { return x.ConvertBuffer(o, i); });
}

// stream.DownsampleReader (stream/downsample.go:47-64)
inline ReaderPtr DownsampleReader(const Context &x, ReaderPtr in, unsigned factor) {
    unsigned rate = in->SampleRate() / factor;
    auto offset = std::make_shared<int64_t>(0);
    return std::make_shared<ReadTransformer>(std::move(in), kBlock, kBlock, HZSDR_FMT_C64, rate,
                                             [&x, factor, offset](Samples i, Samples o) {
                                                 size_t n = x.DownsampleBuffer(o, i, factor, *offset);
                                                 *offset += (int64_t)i.length;
                                                 return n;
                                             });
}

// stream.DecimateReader (stream/decimate.go:34-55)
inline ReaderPtr DecimateReader(const Context &x, ReaderPtr in, unsigned factor) {
    unsigned rate = in->SampleRate() / factor;
    int fmt = in->SampleFormat();
    auto offset = std::make_shared<int64_t>(0);
    return std::make_shared<ReadTransformer>(std::move(in), kBlock, kBlock, fmt, rate,
                                             [&x, factor, offset](Samples i, Samples o) {
                                                 size_t n = x.DecimateBuffer(o, i, factor, *offset);
                                                 *offset += (int64_t)i.length;
                                                 return n;
                                             });
}

// stream.ShiftReader (stream/shifter.go:89-102): the closure's ts lives in an hzsdr_nco
class ShiftReaderImpl : public Reader {
public:
    ShiftReaderImpl(const Context &x, ReaderPtr r, double shift_hz) : x_(x), r_(std::move(r)), shift_(shift_hz) {
        if (r_->SampleFormat() != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
        check(x_.raw(), hzsdr_nco_create(x_.raw(), r_->SampleRate(), &nco_));
    }
    ~ShiftReaderImpl() override { if (nco_) hzsdr_nco_free(nco_); }
    size_t Read(Samples s) override {
        if (s.format != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
        size_t n = r_->Read(s);
        check(x_.raw(), hzsdr_nco_shift(nco_, shift_, s.data, n));
        return n;
    }
    int SampleFormat() const override { return r_->SampleFormat(); }
    unsigned SampleRate() const override { return r_->SampleRate(); }

private:
    const Context &x_;
    ReaderPtr r_;
    double shift_;
    hzsdr_nco *nco_ = nullptr;
};
inline ReaderPtr ShiftReader(const Context &x, ReaderPtr r, double shift_hz) {
    return std::make_shared<ShiftReaderImpl>(x, std::move(r), shift_hz);
}

// stream.Gain (stream/gain.go:30-57)
class GainReader : public Reader {
public:
    GainReader(const Context &x, ReaderPtr r, float v) : x_(x), r_(std::move(r)), v_(v) {}
    size_t Read(Samples s) override {
        size_t n = r_->Read(s);
        x_.Scale(s.slice(0, n), v_);
        return n;
    }
    int SampleFormat() const override { return r_->SampleFormat(); }
    unsigned SampleRate() const override { return r_->SampleRate(); }

private:
    const Context &x_;
    ReaderPtr r_;
    float v_;
};
inline ReaderPtr Gain(const Context &x, ReaderPtr r, float v) { return std::make_shared<GainReader>(x, std::move(r), v); }

}  // namespace stream
}  // namespace hzsdr
