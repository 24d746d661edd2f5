// hzsdr.hpp -- C++ host-side mirror of the reference's Go interfaces for the hot
// path, over the C ABI of include/hzsdr.h (the reference is compiled code and the
// image has no Go toolchain, so the compiled-language host layer is C++; the cgo
// package in INTEGRATION.md is this file transliterated).
//
// Names follow the reference: sdr::Reader (reader.go:39-51), sdr::ReadFull
// (reader.go:72), stream::ReadTransformer (stream/read_transformer.go:45-137),
// stream::ConvertReader / DecimateReader / DownsampleReader / ShiftReader / Gain /
// Multiply / Add / ConvolutionReader / ReadBeamform (stream/*.go), sdr::LookupTable
// (iq_lookup_table.go), fft::Planner / Plan / TransformOnce / Convolve / CrossCorrelate /
// ConvolveFreq (fft/*.go), plus stream::Chain, the fused form the north star adds.
// Errors are the reference's sentinels, thrown as hzsdr::Error
// carrying the status code.  Buffers are host memory (a HZSDR_MEM_HOST context),
// as Go slices would be.
#pragma once
#include <complex>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
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
    // hzsdr_call_count: library calls made on this context so far
    unsigned long long CallCount() const {
        unsigned long long n = 0;
        check(c_, hzsdr_call_count(c_, &n));
        return n;
    }

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

// sdr.Writer (writer.go)
struct Writer {
    virtual ~Writer() = default;
    virtual size_t Write(Samples s) = 0;
    virtual int SampleFormat() const = 0;
    virtual unsigned SampleRate() const = 0;
};
using WriterPtr = std::shared_ptr<Writer>;

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

namespace stream {

constexpr size_t kBlock = 32 * 1024;  // stream/convert.go:43-44

// stream.ConvertReader (stream/convert.go:37-51)
inline ReaderPtr ConvertReader(const Context &x, ReaderPtr in, int to) {
    unsigned rate = in->SampleRate();
    return std::make_shared<ReadTransformer>(std::move(in), kBlock, kBlock, to, rate,
                                             [&x](Samples i, Samples o) { return x.ConvertBuffer(o, i); });
}

// stream.ConvertWriter (stream/convert.go:58-118): a Writer of `input_format` in front of `out`; every Write is
// converted in chunks of 32 Ki samples into a buffer of out's format and handed on whole.  A Write of another format
// throws ErrSampleFormatMismatch before anything is converted; "Conversion mismatch" if a chunk comes back short.
class ConvWriter : public Writer {
public:
    ConvWriter(const Context &x, WriterPtr out, int input_format)
        : x_(x), out_(std::move(out)), fmt_(input_format), buf_(out_->SampleFormat(), kBlock) {}
    size_t Write(Samples in) override {
        if (in.format != fmt_) throw Error(HZSDR_ERR_FORMAT_MISMATCH, hzsdr_strerror(HZSDR_ERR_FORMAT_MISMATCH));
        size_t n = 0;
        for (size_t i = 0; i < in.length; i += kBlock) {
            const size_t ie = i + kBlock > in.length ? in.length : i + kBlock;
            const size_t leng = x_.ConvertBuffer(buf_.view, in.slice(i, ie));
            if (ie - i != leng) throw Error(-1, "ConvertWriter: Conversion mismatch");
            n += out_->Write(buf_.view.slice(0, leng));
        }
        return n;
    }
    int SampleFormat() const override { return fmt_; }
    unsigned SampleRate() const override { return out_->SampleRate(); }

private:
    const Context &x_;
    WriterPtr out_;
    int fmt_;
    Buffer buf_;
};
inline WriterPtr ConvertWriter(const Context &x, WriterPtr out, int input_format) {
    return std::make_shared<ConvWriter>(x, std::move(out), input_format);
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
    // hzsdr_nco_set_ulp1 (go/hip/stream.go Shifter.SetULP1): the factor within one float32 ulp of the reference's
    // instead of bit-identical to it; off by default
    void SetULP1(bool on) { check(x_.raw(), hzsdr_nco_set_ulp1(nco_, on ? 1 : 0)); }

private:
    const Context &x_;
    ReaderPtr r_;
    double shift_;
    hzsdr_nco *nco_ = nullptr;
};
inline std::shared_ptr<ShiftReaderImpl> ShiftReader(const Context &x, ReaderPtr r, double shift_hz) {
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

// sdr.LookupTable (iq_lookup_table.go:98-150)
class LookupTable {
public:
    // NewLookupTable(inputFormat, lookup): `lookup` holds 65536 samples of the output format
    LookupTable(const Context &x, int input_format, Samples lookup) : x_(x), in_(input_format), out_(lookup.format) {
        check(x_.raw(), hzsdr_lut_create(x_.raw(), input_format, lookup.format, lookup.data, lookup.length, &t_));
    }
    ~LookupTable() { if (t_) hzsdr_lut_free(t_); }
    LookupTable(const LookupTable &) = delete;
    size_t Lookup(Samples dst, Samples src) const {
        size_t n = 0;
        check(x_.raw(), hzsdr_lut_lookup(t_, dst.format, dst.data, dst.length, src.format, src.data, src.length, &n));
        return n;
    }
    int InputFormat() const { return in_; }
    int OutputFormat() const { return out_; }

private:
    const Context &x_;
    int in_, out_;
    hzsdr_lut *t_ = nullptr;
};
// LookupTableIdentityU8 / I8 (iq_lookup_table.go:69-90): same bytes, two views
inline Buffer LookupTableIdentity(int format) {
    Buffer b(format, 65536);
    hzsdr_lut_identity(b.bytes.data());
    return b;
}

namespace fft {

enum Direction { Backward = HZSDR_FFT_BACKWARD, Forward = HZSDR_FFT_FORWARD };  // fft/fft.go:30-37

// fft.Plan (fft/fft.go:50-59): aliases the two buffers for its whole life
class Plan {
public:
    Plan(const Context &x, Samples iq, Samples frequency, Direction d) : x_(x) {
        check(x_.raw(), hzsdr_fft_plan(x_.raw(), iq.data, iq.length, frequency.data, frequency.length, (int)d, &p_));
    }
    ~Plan() { Close(); }
    Plan(const Plan &) = delete;
    void Transform() { check(x_.raw(), hzsdr_fft_transform(p_)); }
    void Close() { if (p_) { hzsdr_fft_free(p_); p_ = nullptr; } }

private:
    const Context &x_;
    hzsdr_fft *p_ = nullptr;
};
using PlanPtr = std::unique_ptr<Plan>;
// fft.Planner (fft/fft.go:45-48)
using Planner = std::function<PlanPtr(Samples iq, Samples frequency, Direction)>;
inline Planner NewPlanner(const Context &x) {
    return [&x](Samples iq, Samples f, Direction d) { return PlanPtr(new Plan(x, iq, f, d)); };
}
// fft.TransformOnce (fft/fft.go:64-75)
inline void TransformOnce(const Planner &planner, Samples iq, Samples frequency, Direction d) {
    auto p = planner(iq, frequency, d);
    p->Transform();
    p->Close();
}

// the "func() error" closures of fft/convolution.go as an object
class Closure {
public:
    Closure(const Context &x, hzsdr_conv *c) : x_(x), c_(c) {}
    ~Closure() { if (c_) hzsdr_conv_free(c_); }
    Closure(const Closure &) = delete;
    void operator()() { check(x_.raw(), hzsdr_conv_exec(c_)); }

private:
    const Context &x_;
    hzsdr_conv *c_;
};
using ClosurePtr = std::unique_ptr<Closure>;

inline ClosurePtr two_input(const Context &x, Samples dst, Samples iq1, Samples iq2, int mode) {
    hzsdr_conv *c = nullptr;
    check(x.raw(), hzsdr_convolve_create(x.raw(), dst.data, dst.length, iq1.data, iq1.length, iq2.data, iq2.length, mode, &c));
    return ClosurePtr(new Closure(x, c));
}
// fft.Convolve (fft/convolution.go:97-113) / fft.CrossCorrelate (:119-138).  The
// planner argument of the reference is the context here: the plans live in the library.
inline ClosurePtr Convolve(const Context &x, Samples dst, Samples iq1, Samples iq2) {
    return two_input(x, dst, iq1, iq2, HZSDR_CONV_CONVOLVE);
}
inline ClosurePtr CrossCorrelate(const Context &x, Samples dst, Samples iq1, Samples iq2) {
    return two_input(x, dst, iq1, iq2, HZSDR_CONV_CROSS_CORRELATE);
}
// fft.ConvolveOnce (fft/convolution.go:200-211)
inline void ConvolveOnce(const Context &x, Samples dst, Samples iq1, Samples iq2) { (*Convolve(x, dst, iq1, iq2))(); }
// fft.ConvolveFreq (fft/convolution.go:150-192)
inline ClosurePtr ConvolveFreq(const Context &x, Samples dst, Samples src, Samples freq) {
    hzsdr_conv *c = nullptr;
    check(x.raw(), hzsdr_convolve_freq_create(x.raw(), dst.data, dst.length, src.data, src.length, freq.data, freq.length, &c));
    return ClosurePtr(new Closure(x, c));
}

}  // namespace fft

namespace stream {

// stream.Multiply (stream/multiply.go:27-238): c64 in place, u8 / i8 through the
// rotation table; SetMultiplier is the reference's undocumented setter (:34-36)
class MultiplyReader : public Reader {
public:
    MultiplyReader(const Context &x, ReaderPtr r, float re, float im) : x_(x), r_(std::move(r)), re_(re), im_(im) {
        const int f = r_->SampleFormat();
        if (f == HZSDR_FMT_U8 || f == HZSDR_FMT_I8) check(x_.raw(), hzsdr_rotlut_create(x_.raw(), f, re, im, &t_));
        else if (f != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
    }
    ~MultiplyReader() override { if (t_) hzsdr_rotlut_free(t_); }
    void SetMultiplier(float re, float im) {
        re_ = re;
        im_ = im;
        if (t_) check(x_.raw(), hzsdr_rotlut_set_multiplier(t_, re, im));
    }
    size_t Read(Samples s) override {
        if (s.format != r_->SampleFormat()) throw Error(HZSDR_ERR_FORMAT_MISMATCH, hzsdr_strerror(HZSDR_ERR_FORMAT_MISMATCH));
        size_t n = r_->Read(s);
        if (t_) check(x_.raw(), hzsdr_rotlut_apply(t_, s.data, n));
        else if (!(re_ == 1.0f && im_ == 0.0f)) x_.Multiply(s.slice(0, n), re_, im_);  // multiply.go:59-62
        return n;
    }
    int SampleFormat() const override { return r_->SampleFormat(); }
    unsigned SampleRate() const override { return r_->SampleRate(); }

private:
    const Context &x_;
    ReaderPtr r_;
    float re_, im_;
    hzsdr_rotlut *t_ = nullptr;
};
inline std::shared_ptr<MultiplyReader> Multiply(const Context &x, ReaderPtr r, float re, float im) {
    return std::make_shared<MultiplyReader>(x, std::move(r), re, im);
}

// stream.Add (stream/add.go:41-185)
class AddReader : public Reader {
public:
    AddReader(const Context &x, std::vector<ReaderPtr> rs) : x_(x), rs_(std::move(rs)) {}
    size_t Read(Samples s) override {
        if (failed_) throw Error(failed_, hzsdr_strerror(failed_));
        if (s.format != HZSDR_FMT_C64 && s.format != HZSDR_FMT_I16 && s.format != HZSDR_FMT_I8)
            throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
        std::vector<Buffer> bufs;
        std::vector<const void *> ptrs;
        bufs.reserve(rs_.size());
        for (auto &r : rs_) {
            bufs.emplace_back(s.format, s.length);
            try {
                ReadFull(*r, bufs.back().view);  // add.go:147: short input poisons the reader
            } catch (const Eof &) { failed_ = -1; throw; }
            catch (const Error &e) { failed_ = e.status; throw; }
            ptrs.push_back(bufs.back().view.data);
        }
        check(x_.raw(), hzsdr_sum(x_.raw(), s.format, s.data, ptrs.data(), (int)ptrs.size(), s.length));
        return s.length;
    }
    int SampleFormat() const override { return rs_[0]->SampleFormat(); }
    unsigned SampleRate() const override { return rs_[0]->SampleRate(); }

private:
    const Context &x_;
    std::vector<ReaderPtr> rs_;
    int failed_ = 0;
};
inline ReaderPtr Add(const Context &x, std::vector<ReaderPtr> rs) {
    if (rs.empty()) throw Error(HZSDR_ERR_INVALID_ARGUMENT, "stream.Add: No readers passed");
    if (rs.size() == 1) return rs[0];
    const int f = rs[0]->SampleFormat();
    if (f != HZSDR_FMT_C64 && f != HZSDR_FMT_I16 && f != HZSDR_FMT_I8)
        throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
    for (auto &r : rs) {
        if (r->SampleFormat() != f) throw Error(HZSDR_ERR_INVALID_ARGUMENT, "stream.Add: Readers are not all the same format");
        if (r->SampleRate() != rs[0]->SampleRate()) throw Error(HZSDR_ERR_INVALID_ARGUMENT, "stream.Add: Readers are not all the same rate");
    }
    return std::make_shared<AddReader>(x, std::move(rs));
}

// stream.ConvolutionReader (stream/convolution.go:36-82): `filter` is in the frequency domain
inline ReaderPtr ConvolutionReader(const Context &x, ReaderPtr r, Samples filter) {
    if (r->SampleFormat() != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
    const size_t n = filter.length;
    auto iq = std::make_shared<Buffer>(HZSDR_FMT_C64, n);
    std::shared_ptr<fft::Closure> conv = fft::ConvolveFreq(x, iq->view, iq->view, filter);
    unsigned rate = r->SampleRate();
    return std::make_shared<ReadTransformer>(std::move(r), n, n, HZSDR_FMT_C64, rate, [iq, conv](Samples in, Samples out) {
        std::memcpy(iq->view.data, in.data, in.size());
        (*conv)();
        std::memcpy(out.data, iq->view.data, in.size());
        return in.length;
    });
}

// stream.ReadBeamform / Beamform (stream/beamform.go:131-171).  One fused kernel per
// Read instead of K convert + K multiply + K+1 add passes; same arithmetic order.
class Beamform : public Reader {
public:
    Beamform(const Context &x, std::vector<ReaderPtr> rs, std::vector<std::complex<float>> angles)
        : x_(x), rs_(std::move(rs)), w_(rs_.size(), {1.0f, 0.0f}) {
        for (auto &r : rs_)
            if (r->SampleFormat() != rs_[0]->SampleFormat() || r->SampleRate() != rs_[0]->SampleRate())
                throw Error(HZSDR_ERR_INVALID_ARGUMENT, "stream.Add: Readers are not all the same format / rate");
        SetPhaseAngles(angles);
    }
    // returns false where the reference returns its length error (beamform.go:132-134)
    bool SetPhaseAngles(const std::vector<std::complex<float>> &angles) {
        if (angles.size() != rs_.size()) return false;
        w_ = angles;
        return true;
    }
    size_t Read(Samples s) override {
        if (s.format != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_MISMATCH, hzsdr_strerror(HZSDR_ERR_FORMAT_MISMATCH));
        const int f = rs_[0]->SampleFormat();
        std::vector<Buffer> bufs;
        std::vector<const void *> ptrs;
        bufs.reserve(rs_.size());
        for (auto &r : rs_) {
            bufs.emplace_back(f, s.length);
            ReadFull(*r, bufs.back().view);
            ptrs.push_back(bufs.back().view.data);
        }
        check(x_.raw(), hzsdr_beamform(x_.raw(), s.data, f, ptrs.data(), reinterpret_cast<const float *>(w_.data()),
                                       (int)ptrs.size(), s.length));
        return s.length;
    }
    int SampleFormat() const override { return HZSDR_FMT_C64; }
    unsigned SampleRate() const override { return rs_[0]->SampleRate(); }

private:
    const Context &x_;
    std::vector<ReaderPtr> rs_;
    std::vector<std::complex<float>> w_;
};
inline std::shared_ptr<Beamform> ReadBeamform(const Context &x, std::vector<ReaderPtr> rs, std::vector<std::complex<float>> angles) {
    return std::make_shared<Beamform>(x, std::move(rs), std::move(angles));
}
// stream.BeamformAngles (stream/beamform.go:104-128)
inline std::vector<std::complex<float>> BeamformAngles(double frequency_hz, double angle_deg, const std::vector<double> &distances) {
    std::vector<std::complex<float>> out(distances.size());
    hzsdr_beamform_angles(frequency_hz, angle_deg, distances.data(), (int)distances.size(), reinterpret_cast<float *>(out.data()));
    return out;
}

// The fused form of a reader chain (north_star): ops are appended in reader order,
// Run consumes one resident buffer per call; NCO time and FIR history persist.
class Chain {
public:
    Chain(const Context &x, int src_format, uint64_t sample_rate) : x_(x) { check(x_.raw(), hzsdr_chain_create(x_.raw(), src_format, sample_rate, &c_)); }
    ~Chain() { if (c_) hzsdr_chain_free(c_); }
    Chain(const Chain &) = delete;
    Chain &Shift(double hz) { check(x_.raw(), hzsdr_chain_shift(c_, hz)); return *this; }
    Chain &Gain(float r) { check(x_.raw(), hzsdr_chain_gain(c_, r)); return *this; }
    Chain &Multiply(float re, float im) { check(x_.raw(), hzsdr_chain_rotate(c_, re, im)); return *this; }
    Chain &Decimate(unsigned f) { check(x_.raw(), hzsdr_chain_decimate(c_, f)); return *this; }
    Chain &Downsample(unsigned f) { check(x_.raw(), hzsdr_chain_downsample(c_, f)); return *this; }
    Chain &Convolution(Samples filter, unsigned decimate = 1) { check(x_.raw(), hzsdr_chain_convolution(c_, filter.data, filter.length, decimate)); return *this; }
    Chain &FirDecimate(const std::vector<std::complex<float>> &taps, unsigned d) {
        check(x_.raw(), hzsdr_chain_fir_decimate(c_, reinterpret_cast<const float *>(taps.data()), taps.size(), d));
        return *this;
    }
    // hzsdr_chain_pipeline: consecutive Run calls on the matrix path overlap (the input must be complete at call time)
    Chain &Pipeline(bool on = true) { check(x_.raw(), hzsdr_chain_pipeline(c_, on ? 1 : 0)); return *this; }
    // -> (consumed, produced)
    std::pair<size_t, size_t> Run(Samples in, Samples out) {
        size_t used = 0, made = 0;
        check(x_.raw(), hzsdr_chain_run(c_, in.data, in.length, out.data, out.length, &used, &made));
        return {used, made};
    }
    void Reset() { check(x_.raw(), hzsdr_chain_reset(c_)); }
    // every sample mixed before the FIR (reference order) instead of after it
    Chain &MixInOrder(bool in_order = true) { check(x_.raw(), hzsdr_chain_mix_in_order(c_, in_order ? 1 : 0)); return *this; }
    hzsdr_chain *raw() const { return c_; }
    const Context &context() const { return x_; }

private:
    const Context &x_;
    hzsdr_chain *c_ = nullptr;
};

// The pinned ring in front of a chain: what stream.RingBuffer with an IQBufferAllocator of
// hipHostMalloc memory is to a driver callback (stream/ring.go:48-69, :337-392).
class Ring {
public:
    Ring(Chain &chain, size_t slot_length, int slots) : x_(chain.context()) {
        check(x_.raw(), hzsdr_ring_create(chain.raw(), slot_length, slots, &r_));
        void *base = nullptr;
        size_t n = 0;
        check(x_.raw(), hzsdr_ring_iq_buffer(r_, &base, &n, &slot_length_));
        base_ = base;
        total_ = n;
    }
    ~Ring() { if (r_) hzsdr_ring_free(r_); }
    Ring(const Ring &) = delete;
    // the whole IQBufferAllocator region, in the chain's source format
    Samples IQBuffer(int format) const { return Samples{format, base_, total_}; }
    // write cursor: the next slot's memory
    Samples Acquire(int format, int *slot) {
        void *p = nullptr;
        check(x_.raw(), hzsdr_ring_acquire(r_, slot, &p));
        return Samples{format, p, slot_length_};
    }
    void Submit(int slot, size_t n) { check(x_.raw(), hzsdr_ring_submit(r_, slot, n)); }
    // `count` acquired slots, first_slot the oldest, n samples each, as ONE call of the chain (hzsdr_ring_submit_many)
    void SubmitMany(int first_slot, int count, size_t n) { check(x_.raw(), hzsdr_ring_submit_many(r_, first_slot, count, n)); }
    void Release(int slot) { check(x_.raw(), hzsdr_ring_release(r_, slot)); }  // the acquired slot, unused
    // read cursor: the oldest submitted slot's output (complex64), valid until that slot is resubmitted
    Samples Pop() {
        const void *p = nullptr;
        size_t n = 0;
        check(x_.raw(), hzsdr_ring_pop(r_, &p, &n));
        return Samples{HZSDR_FMT_C64, const_cast<void *>(p), n};
    }
    int InFlight() const { return hzsdr_ring_in_flight(r_); }

private:
    const Context &x_;
    hzsdr_ring *r_ = nullptr;
    void *base_ = nullptr;
    size_t total_ = 0, slot_length_ = 0;
};


// ---- nested Readers that fuse into one chain and read ahead (go/hip/fused.go is this, in Go) ---------------------------
// The constructors of `Fused` do not wrap a Reader that one of them made: they extend its chain -- ConvertReader ->
// ShiftReader -> Gain -> Multiply -> DecimateReader / DownsampleReader / ConvolutionReader collapse into one
// hzsdr_chain, one launch per slot -- and the Reader reads AHEAD: `readahead` Reader blocks of 32 Ki samples per slot of a
// pinned ring.  What the nest means is kept: the same samples bit for bit, block-structured stages hand out whole
// blocks only (a source that ends inside a block loses that partial block, as ReadFull's ErrUnexpectedEOF does in
// read_transformer.go:120-135), pass-through stages hand out whatever the source delivered, the source's error is
// sticky and surfaces behind everything read before it.
class ChainReader : public Reader {
public:
    struct Stage { int kind; double shift; float gain, re, im; };  // 0 Shift, 1 Gain, 2 Multiply
    ChainReader(const Context &x, ReaderPtr src, int readahead) : x_(x), src_(std::move(src)), readahead_(readahead) {
        src_format_ = src_->SampleFormat();
        rate_ = src_->SampleRate();
    }
    bool open() const { return !chain_ && term_ == 0; }
    bool c64_here() const { return src_format_ == HZSDR_FMT_C64 || converted_; }
    bool extend_convert(int to) {
        if (!open() || to != HZSDR_FMT_C64 || !stages_.empty() || src_format_ == HZSDR_FMT_C64 || converted_) return false;
        converted_ = true;
        block_ = lcm(block_, kBlock);
        return true;
    }
    bool extend_stage(const Stage &st) {
        if (!open() || !c64_here()) return false;
        stages_.push_back(st);
        return true;
    }
    // the north-star terminal (kind 4: an N-tap FIR whose output is kept every `factor` samples -- hzsdr_chain_fir_decimate;
    // the reference has no such Reader): `slots` slots in the pinned ring, `group` of them per call of the chain
    bool extend_fir(const std::vector<std::complex<float>> &taps, unsigned factor, int slots, int group) {
        if (!open()) return false;
        if (!c64_here()) converted_ = true;  // (the terminal converts on its way in, as DownsampleReader does)
        term_ = 4, factor_ = factor, taps_ = taps;
        block_ = lcm(block_, factor);
        rate_ /= factor;
        slots_ = std::max(2, slots);
        group_ = std::max(1, std::min(std::min(group, 8), slots_ - 1));
        return true;
    }
    const Chain *chain() const { return chain_.get(); }
    // kind 1 Decimate, 2 Downsample (its own conversion of a raw source included), 3 Convolution
    bool extend_terminal(int kind, unsigned factor, Samples filter, size_t block) {
        if (chain_) return false;
        if (kind == 1 && term_ == 3 && decimate_ == 1 && kBlock % filter_.size() == 0) {  // DecimateReader behind the ConvolutionReader
            decimate_ = factor;
            block_ = lcm(block_, kBlock);
            rate_ /= factor;
            return true;
        }
        if (kind == 2 && open() && !c64_here() && stages_.empty()) converted_ = true;
        if (!open() || !c64_here()) return false;
        term_ = kind, factor_ = factor;
        if (kind == 3) filter_.assign((const std::complex<float> *)filter.data, (const std::complex<float> *)filter.data + filter.length);
        block_ = lcm(block_, block);
        if (kind == 1 || kind == 2) rate_ /= factor;
        return true;
    }
    int SampleFormat() const override { return c64_here() ? HZSDR_FMT_C64 : src_format_; }
    unsigned SampleRate() const override { return rate_; }
    size_t Read(Samples s) override {
        if (s.format != SampleFormat() || s.format != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_MISMATCH, hzsdr_strerror(HZSDR_ERR_FORMAT_MISMATCH));
        if (!chain_) build();
        if (off_ >= pending_.length) {
            // everything but the slot being consumed is in flight -- refilled `group_` slots at a time, ONE call of the chain
            // per group (hzsdr_ring_submit_many): a refill waits until that many slots are free, or nothing is in flight
            for (;;) {
                const int free_slots = slots_ - 1 - inflight_;
                if (free_slots < 1 || (free_slots < group_ && inflight_ > 0)) break;
                if (fill(std::min(group_, free_slots)) == 0) break;
            }
            if (inflight_ == 0) {
                if (err_) std::rethrow_exception(err_);
                throw Eof();
            }
            pending_ = ring_->Pop();
            inflight_--;
            off_ = 0;
            if (pending_.length == 0) return 0;
        }
        const size_t n = std::min(pending_.length - off_, s.length);
        memcpy(s.data, (const char *)pending_.data + 8 * off_, 8 * n);
        off_ += n;
        return n;
    }

private:
    static constexpr size_t kBlock = 32 * 1024;
    static size_t gcd(size_t a, size_t b) { while (b) { const size_t t = a % b; a = b; b = t; } return a; }
    static size_t lcm(size_t a, size_t b) { return a / gcd(a, b) * b; }
    void build() {
        chain_ = std::make_unique<Chain>(x_, src_format_, src_->SampleRate());
        for (const Stage &st : stages_) {
            if (st.kind == 0) chain_->Shift(st.shift);
            else if (st.kind == 1) chain_->Gain(st.gain);
            else chain_->Multiply(st.re, st.im);
        }
        if (term_ == 1) chain_->Decimate(factor_);
        else if (term_ == 2) chain_->Downsample(factor_);
        else if (term_ == 3) chain_->Convolution(Samples{HZSDR_FMT_C64, filter_.data(), filter_.size()}, decimate_);
        else if (term_ == 4) {
            chain_->FirDecimate(taps_, factor_);
            check(x_.raw(), hzsdr_chain_pipeline(chain_->raw(), 1));  // (consecutive calls overlap: the ring says what each call's buffers wait for)
        }
        const size_t unit = block_ > 1 ? block_ : kBlock;
        slot_len_ = std::max<size_t>(1, (size_t)readahead_ * kBlock / unit) * unit;
        ring_ = std::make_unique<Ring>(*chain_, slot_len_, slots_);
    }
    // Read the source into up to `want` slots and submit them: the full ones together, ONE call of the chain, a short last
    // one (the source ended) by itself.  -> slots submitted.  An acquired slot is submitted or released whatever the
    // source throws (the error is sticky and surfaces behind everything read before it).
    int fill(int want) {
        if (err_) return 0;
        int first = -1, full = 0, done = 0;
        auto flush = [&]() {
            if (full) {
                ring_->SubmitMany(first, full, slot_len_);
                inflight_ += full;
                full = 0;
            }
        };
        while (done < want && !err_) {
            int slot = -1;
            Samples iq = ring_->Acquire(src_format_, &slot);
            size_t n = 0;
            int idle = 0;
            try {
                while (n < slot_len_) {
                    const size_t got = src_->Read(iq.slice(n, slot_len_));
                    n += got;
                    idle = got ? 0 : idle + 1;
                    if (idle >= 100) throw Error(HZSDR_ERR_INVALID_ARGUMENT, "multiple Read calls return no data or error");  // io.ErrNoProgress
                }
            } catch (...) {
                err_ = std::current_exception();
            }
            n = n / block_ * block_;
            if (n == slot_len_) {
                if (first < 0 || full == 0) first = slot;
                full++, done++;
                continue;
            }
            flush();
            if (n == 0) {
                ring_->Release(slot);
            } else {
                ring_->Submit(slot, n);
                inflight_++, done++;
            }
            return done;
        }
        flush();
        return done;
    }
    const Context &x_;
    ReaderPtr src_;
    int readahead_, src_format_ = 0;
    unsigned rate_ = 0;
    bool converted_ = false;
    std::vector<Stage> stages_;
    int term_ = 0;
    unsigned factor_ = 1, decimate_ = 1;
    std::vector<std::complex<float>> filter_, taps_;
    int slots_ = 3, group_ = 1;
    size_t block_ = 1, slot_len_ = 0;
    std::unique_ptr<Chain> chain_;
    std::unique_ptr<Ring> ring_;
    Samples pending_{};
    size_t off_ = 0;
    int inflight_ = 0;
    std::exception_ptr err_;
};

// The constructor set that fuses (go/hip: Context.Readers()); every name falls back to the plain constructor above when
// the stage cannot join a chain.
struct Fused {
    const Context &x;
    int readahead = 32;
    template <class How> std::shared_ptr<ChainReader> fused(ReaderPtr r, How how) const {
        if (auto cr = std::dynamic_pointer_cast<ChainReader>(r)) {
            if (how(*cr)) return cr;
        }
        auto cr = std::make_shared<ChainReader>(x, r, readahead);
        if (how(*cr)) return cr;
        return nullptr;
    }
    ReaderPtr ConvertReader(ReaderPtr in, int to) const {
        if (auto f = fused(in, [&](ChainReader &c) { return c.extend_convert(to); })) return f;
        return stream::ConvertReader(x, std::move(in), to);
    }
    ReaderPtr ShiftReader(ReaderPtr r, double hz) const {
        if (r->SampleFormat() != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
        if (auto f = fused(r, [&](ChainReader &c) { return c.extend_stage({0, hz, 0.f, 0.f, 0.f}); })) return f;
        return stream::ShiftReader(x, std::move(r), hz);
    }
    ReaderPtr Gain(ReaderPtr r, float v) const {
        if (r->SampleFormat() == HZSDR_FMT_C64)
            if (auto f = fused(r, [&](ChainReader &c) { return c.extend_stage({1, 0.0, v, 0.f, 0.f}); })) return f;
        return stream::Gain(x, std::move(r), v);
    }
    ReaderPtr Multiply(ReaderPtr r, float re, float im) const {
        if (r->SampleFormat() == HZSDR_FMT_C64 && !(re == 1.0f && im == 0.0f))
            if (auto f = fused(r, [&](ChainReader &c) { return c.extend_stage({2, 0.0, 0.f, re, im}); })) return f;
        return stream::Multiply(x, std::move(r), re, im);
    }
    ReaderPtr DecimateReader(ReaderPtr in, unsigned factor) const {
        if (in->SampleFormat() == HZSDR_FMT_C64 && factor > 0)
            if (auto f = fused(in, [&](ChainReader &c) { return c.extend_terminal(1, factor, Samples{}, 32 * 1024); })) return f;
        return stream::DecimateReader(x, std::move(in), factor);
    }
    ReaderPtr DownsampleReader(ReaderPtr in, unsigned factor) const {
        const int f0 = in->SampleFormat();
        if ((f0 == HZSDR_FMT_C64 || f0 == HZSDR_FMT_U8 || f0 == HZSDR_FMT_I16) && factor > 0)
            if (auto f = fused(in, [&](ChainReader &c) { return c.extend_terminal(2, factor, Samples{}, 32 * 1024); })) return f;
        return stream::DownsampleReader(x, std::move(in), factor);
    }
    ReaderPtr ConvolutionReader(ReaderPtr r, Samples filter) const {
        if (r->SampleFormat() != HZSDR_FMT_C64) throw Error(HZSDR_ERR_FORMAT_UNKNOWN, hzsdr_strerror(HZSDR_ERR_FORMAT_UNKNOWN));
        if (auto f = fused(r, [&](ChainReader &c) { return c.extend_terminal(3, 1, filter, filter.length); })) return f;
        return stream::ConvolutionReader(x, std::move(r), filter);
    }
    // The north-star terminal as a Reader (BASELINE.json north_star; the reference's Downsample is the boxcar,
    // stream/downsample.go:47-64, so name and signature follow DecimateReader's, stream/decimate.go:34): always fused --
    // ConvertReader / ShiftReader / Gain / Multiply in front of it join its chain -- `slots` slots in the pinned ring,
    // `group` of them per call of the chain (hzsdr_ring_submit_many: one launch of the int8 matrix kernel per group).
    std::shared_ptr<ChainReader> FirDecimateReader(ReaderPtr in, const std::vector<std::complex<float>> &taps, unsigned factor, int slots = 5,
                                                   int group = 4) const {
        auto f = fused(in, [&](ChainReader &c) { return c.extend_fir(taps, factor, slots, group); });
        if (!f) throw Error(HZSDR_ERR_INVALID_ARGUMENT, "FirDecimateReader: the stage does not fit the Reader in front of it");
        return f;
    }
};

}  // namespace stream
}  // namespace hzsdr
