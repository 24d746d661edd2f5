// test_host_mirror.cpp -- the reference's own known-answer tests, spelt with the
// C++ host mirror (go-sdr_amd/cxx/hzsdr.hpp) over the C ABI, the way the Go shim
// would run them through cgo.  Exit code 0 = all passed.  Needs an MI355X.
//   g++ -std=c++17 -I. tests/cxx/test_host_mirror.cpp -Lgo-sdr_amd -lhzsdr_hip -Wl,-rpath,$PWD/go-sdr_amd -o build/test_host_mirror
#include <cmath>
#include <complex>
#include <cstdio>

#include "go-sdr_amd/cxx/hzsdr.hpp"

using namespace hzsdr;
using c64 = std::complex<float>;

static int failures = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            failures++;                                                      \
        }                                                                    \
    } while (0)

template <class T> static Samples view(int fmt, std::vector<T> &v, size_t per_sample) {
    return Samples{fmt, v.data(), v.size() / per_sample};
}

int main() {
    Context ctx(0);
    std::printf("backend %s, %s\n", hzsdr_backend(), hzsdr_version());

    {  // iq_u8_test.go:134-168, iq_c64_test.go:38-104
        std::vector<uint8_t> u8 = {255, 0};
        std::vector<int8_t> i8(2);
        EXPECT(ctx.ConvertBuffer(view(HZSDR_FMT_I8, i8, 2), view(HZSDR_FMT_U8, u8, 2)) == 1);
        EXPECT(i8[0] == 127 && i8[1] == -128);
        std::vector<c64> c = {{1.f, -1.f}};
        EXPECT(ctx.ConvertBuffer(view(HZSDR_FMT_I8, i8, 2), view(HZSDR_FMT_C64, c, 1)) == 1);
        EXPECT(i8[0] == 127 && i8[1] == -127);
        std::vector<int16_t> i16(2);
        c = {{-1.f, -1.f}};
        ctx.ConvertBuffer(view(HZSDR_FMT_I16, i16, 2), view(HZSDR_FMT_C64, c, 1));
        EXPECT(i16[0] == -32767 && i16[1] == -32767);
        c = {{0.f, 0.f}};
        ctx.ConvertBuffer(view(HZSDR_FMT_U8, u8, 2), view(HZSDR_FMT_C64, c, 1));
        EXPECT(u8[0] == 127 && u8[1] == 127);
    }
    {  // ErrDstTooSmall, iq_u8.go:104-106
        std::vector<uint8_t> u8(16);
        std::vector<c64> c(4);
        try {
            ctx.ConvertBuffer(view(HZSDR_FMT_C64, c, 1), view(HZSDR_FMT_U8, u8, 2));
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_DST_TOO_SMALL); }
    }
    {  // iq_c64_test.go:110-145 (length 31: odd tails)
        std::vector<c64> a(31, {10.f, 10.f}), b(31, {3.f, 1.f});
        ctx.Scale(view(HZSDR_FMT_C64, a, 1), 0.5f);
        for (auto &v : a) EXPECT(v == c64(5.f, 5.f));
        std::fill(a.begin(), a.end(), c64(10.f, 10.f));
        ctx.Multiply(view(HZSDR_FMT_C64, a, 1), 0.5f, 0.5f);
        for (auto &v : a) EXPECT(v == c64(0.f, 10.f));
        std::fill(a.begin(), a.end(), c64(1.f, 3.f));
        ctx.Add(view(HZSDR_FMT_C64, a, 1), view(HZSDR_FMT_C64, b, 1));
        for (auto &v : a) EXPECT(v == c64(4.f, 4.f));
    }
    {  // mock-style source -> ConvertReader (BASELINE config 1 shape), 1 Mi samples
        const size_t n = 1 << 20;
        std::vector<uint8_t> u8(2 * n);
        for (size_t i = 0; i < 2 * n; i++) u8[i] = (uint8_t)(i * 2654435761u >> 24);
        auto src = std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u8, 2), 2400000u, 10000);
        auto r = stream::ConvertReader(ctx, src, HZSDR_FMT_C64);
        EXPECT(r->SampleFormat() == HZSDR_FMT_C64 && r->SampleRate() == 2400000u);
        std::vector<c64> out(n);
        EXPECT(ReadFull(*r, view(HZSDR_FMT_C64, out, 1)) == n);
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++)
            ok = out[i] == c64((float(u8[2 * i]) - 127.5f) / 127.5f, (float(u8[2 * i + 1]) - 127.5f) / 127.5f);
        EXPECT(ok);
        std::vector<int16_t> wrong(32);  // testutils/reader.go:87-97
        try {
            r->Read(view(HZSDR_FMT_I16, wrong, 2));
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_FORMAT_MISMATCH); }
    }
    {  // stream.ConvertWriter (stream/convert.go:58-118): c64 written in front of a u8 sink, 2 x 32 Ki + 5 samples
       // (three chunks); every byte is the reference's truncating conversion (iq_c64.go:96-100)
        struct Sink : Writer {
            std::vector<uint8_t> got;
            size_t calls = 0;
            size_t Write(Samples s) override {
                EXPECT(s.format == HZSDR_FMT_U8 && s.length <= 32 * 1024);
                const uint8_t *p = (const uint8_t *)s.data;
                got.insert(got.end(), p, p + 2 * s.length);
                calls++;
                return s.length;
            }
            int SampleFormat() const override { return HZSDR_FMT_U8; }
            unsigned SampleRate() const override { return 1800000u; }
        };
        auto sink = std::make_shared<Sink>();
        auto w = stream::ConvertWriter(ctx, sink, HZSDR_FMT_C64);
        EXPECT(w->SampleFormat() == HZSDR_FMT_C64 && w->SampleRate() == 1800000u);
        const size_t n = 2 * 32 * 1024 + 5;
        std::vector<c64> in(n);
        for (size_t i = 0; i < n; i++) in[i] = c64((float)((int)(i * 29 % 255) - 127) / 127.5f, (float)((int)(i * 31 % 255) - 127) / 127.5f);
        EXPECT(w->Write(view(HZSDR_FMT_C64, in, 1)) == n);
        EXPECT(sink->calls == 3 && sink->got.size() == 2 * n);
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++)
            ok = sink->got[2 * i] == (uint8_t)(int)(in[i].real() * 127.5f + 127.5f) && sink->got[2 * i + 1] == (uint8_t)(int)(in[i].imag() * 127.5f + 127.5f);
        EXPECT(ok);
        std::vector<int16_t> wrong(32);
        try {
            w->Write(view(HZSDR_FMT_I16, wrong, 2));
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_FORMAT_MISMATCH); }
    }
    {  // stream/downsample_test.go:59-93: i % 4 pattern, factor 4 -> exactly 1.5+1.5i
        const size_t n = 32 * 1024;
        std::vector<c64> in(n), out(n);
        for (size_t i = 0; i < n; i++) in[i] = c64(float(i % 4), float(i % 4));
        auto r = stream::DownsampleReader(ctx, std::make_shared<BufferReader>(view(HZSDR_FMT_C64, in, 1), 10000u), 4);
        EXPECT(r->SampleRate() == 2500u);
        size_t got = 0;
        try { ReadFull(*r, view(HZSDR_FMT_C64, out, 1), &got); } catch (const Error &) {}
        EXPECT(got == n / 4);
        for (size_t i = 0; i < n / 4; i++) EXPECT(out[i] == c64(1.5f, 1.5f));
    }
    {  // stream/decimate_test.go:131-166
        const size_t n = 32 * 1024;
        std::vector<uint8_t> in(2 * n), out(2 * (n / 10), 9);
        for (size_t i = 0; i < n; i++) in[2 * i] = in[2 * i + 1] = (uint8_t)(i % 10);
        auto r = stream::DecimateReader(ctx, std::make_shared<BufferReader>(view(HZSDR_FMT_U8, in, 2), 10000u), 10);
        EXPECT(r->SampleRate() == 1000u);
        EXPECT(ReadFull(*r, view(HZSDR_FMT_U8, out, 2)) == n / 10);
        for (auto v : out) EXPECT(v == 0);
    }
    {  // stream/shifter_test.go:35-72 through ShiftReader + Gain: +1 kHz, -1 kHz, gain 1
        const size_t n = 1024 * 60;
        std::vector<c64> cw(n), buf(n);
        for (size_t i = 0; i < n; i++) {
            double now = double(i) / 1.8e6;
            cw[i] = c64((float)std::cos(2 * M_PI * now), (float)std::sin(2 * M_PI * now));
        }
        auto src = std::make_shared<BufferReader>(view(HZSDR_FMT_C64, cw, 1), 1800000u, 7000);
        auto r = stream::Gain(ctx, stream::ShiftReader(ctx, stream::ShiftReader(ctx, src, 1000.0), -1000.0), 1.0f);
        EXPECT(ReadFull(*r, view(HZSDR_FMT_C64, buf, 1)) == n);
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++)
            ok = std::fabs((1 + cw[i].real()) - (1 + buf[i].real())) <= 1e-4 * std::fabs(1 + cw[i].real()) + 1e-7 &&
                 std::fabs((1 + cw[i].imag()) - (1 + buf[i].imag())) <= 1e-4 * std::fabs(1 + cw[i].imag()) + 1e-7;
        EXPECT(ok);
    }
    {  // the same round trip with the opt-in <= 1-ulp factor on both readers (Shifter.SetULP1 of the Go package)
        const size_t n = 1024 * 60;
        std::vector<c64> cw(n), buf(n);
        for (size_t i = 0; i < n; i++) {
            double now = double(i) / 1.8e6;
            cw[i] = c64((float)std::cos(2 * M_PI * now), (float)std::sin(2 * M_PI * now));
        }
        auto src = std::make_shared<BufferReader>(view(HZSDR_FMT_C64, cw, 1), 1800000u, 7000);
        auto up = stream::ShiftReader(ctx, src, 1000.0);
        up->SetULP1(true);
        auto down = stream::ShiftReader(ctx, up, -1000.0);
        down->SetULP1(true);
        EXPECT(ReadFull(*down, view(HZSDR_FMT_C64, buf, 1)) == n);
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++)
            ok = std::fabs(cw[i].real() - buf[i].real()) <= 1e-4 && std::fabs(cw[i].imag() - buf[i].imag()) <= 1e-4;
        EXPECT(ok);
    }
    {  // stream/multiply_test.go:36-69 TestRotate: CW(phase pi/2) * (0-1i) == CW(phase 0)
        const size_t n = 1024 * 60;
        std::vector<c64> cw0(n), cw90(n), buf(n);
        for (size_t i = 0; i < n; i++) {
            double now = double(i) / 1.8e6, a = 2 * M_PI * 10.0 * now;
            cw0[i] = c64((float)std::cos(a), (float)std::sin(a));
            cw90[i] = c64((float)std::cos(a + M_PI / 2), (float)std::sin(a + M_PI / 2));
        }
        auto r = stream::Multiply(ctx, std::make_shared<BufferReader>(view(HZSDR_FMT_C64, cw90, 1), 1800000u, 9000), 0.f, -1.f);
        EXPECT(ReadFull(*r, view(HZSDR_FMT_C64, buf, 1)) == n);
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++)
            ok = std::fabs(cw0[i].real() - buf[i].real()) <= 1e-4 * std::fabs(1 + cw0[i].real()) + 1e-7 &&
                 std::fabs(cw0[i].imag() - buf[i].imag()) <= 1e-4 * std::fabs(1 + cw0[i].imag()) + 1e-7;
        EXPECT(ok);
        std::vector<uint8_t> wrong(32);  // multiply.go:47-52
        try {
            r->Read(view(HZSDR_FMT_U8, wrong, 2));
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_FORMAT_MISMATCH); }
        std::vector<int16_t> i16(32);  // multiply.go:85-87: no int16 variant
        try {
            stream::Multiply(ctx, std::make_shared<BufferReader>(view(HZSDR_FMT_I16, i16, 2), 1u), 0.f, -1.f);
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_FORMAT_UNKNOWN); }
    }
    {  // stream/multiply_test.go:71-112 TestRotateU8 and :189-230 TestRotateI8: the table
       // reader equals ConvertBuffer -> Multiply(0-1i) -> ConvertBuffer exactly
        const size_t n = 1024 * 60;
        std::vector<uint8_t> u8(2 * n), refu(2 * n), gotu(2 * n);
        std::vector<int8_t> i8(2 * n), refi(2 * n), goti(2 * n);
        std::vector<c64> c(n);
        uint16_t counter = 0;
        for (size_t i = 0; i < n; i++, counter++) {
            u8[2 * i] = (uint8_t)(counter & 0xFF);
            u8[2 * i + 1] = (uint8_t)((counter & 0xFF00) >> 8);
            i8[2 * i] = (int8_t)(counter & 0xFF);
            i8[2 * i + 1] = (int8_t)(((counter & 0xFF00) >> 8) - 127);
        }
        ctx.ConvertBuffer(view(HZSDR_FMT_C64, c, 1), view(HZSDR_FMT_U8, u8, 2));
        ctx.Multiply(view(HZSDR_FMT_C64, c, 1), 0.f, -1.f);
        ctx.ConvertBuffer(view(HZSDR_FMT_U8, refu, 2), view(HZSDR_FMT_C64, c, 1));
        auto ru = stream::Multiply(ctx, std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u8, 2), 1800000u), 0.f, -1.f);
        EXPECT(ReadFull(*ru, view(HZSDR_FMT_U8, gotu, 2)) == n);
        EXPECT(gotu == refu);
        ctx.ConvertBuffer(view(HZSDR_FMT_C64, c, 1), view(HZSDR_FMT_I8, i8, 2));
        ctx.Multiply(view(HZSDR_FMT_C64, c, 1), 0.f, -1.f);
        ctx.ConvertBuffer(view(HZSDR_FMT_I8, refi, 2), view(HZSDR_FMT_C64, c, 1));
        auto ri = stream::Multiply(ctx, std::make_shared<BufferReader>(view(HZSDR_FMT_I8, i8, 2), 1800000u), 0.f, -1.f);
        EXPECT(ReadFull(*ri, view(HZSDR_FMT_I8, goti, 2)) == n);
        EXPECT(goti == refi);
    }
    {  // stream/add_test.go:34-75 (3 x 10+20i = 30+60i), :77-135 (i8 / i16 10+10 = 20)
        std::vector<c64> in(1000, {10.f, 20.f}), out(1000);
        std::vector<ReaderPtr> rs;
        for (int k = 0; k < 3; k++) rs.push_back(std::make_shared<BufferReader>(view(HZSDR_FMT_C64, in, 1), 10000u, 300));
        auto mix = stream::Add(ctx, rs);
        EXPECT(ReadFull(*mix, view(HZSDR_FMT_C64, out, 1)) == 1000);
        for (auto &v : out) EXPECT(v == c64(30.f, 60.f));
        const size_t n = 1024 * 32;
        std::vector<int8_t> a8(2 * n, 10), o8(2 * n);
        auto m8 = stream::Add(ctx, {std::make_shared<BufferReader>(view(HZSDR_FMT_I8, a8, 2), 0u),
                                    std::make_shared<BufferReader>(view(HZSDR_FMT_I8, a8, 2), 0u)});
        EXPECT(ReadFull(*m8, view(HZSDR_FMT_I8, o8, 2)) == n);
        for (auto v : o8) EXPECT(v == 20);
        std::vector<int16_t> a16(2 * n, 10), o16(2 * n);
        auto m16 = stream::Add(ctx, {std::make_shared<BufferReader>(view(HZSDR_FMT_I16, a16, 2), 0u),
                                     std::make_shared<BufferReader>(view(HZSDR_FMT_I16, a16, 2), 0u)});
        EXPECT(ReadFull(*m16, view(HZSDR_FMT_I16, o16, 2)) == n);
        for (auto v : o16) EXPECT(v == 20);
        std::vector<uint8_t> u(64);  // add.go:55-61: no uint8 adder
        try {
            stream::Add(ctx, {std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u, 2), 0u),
                              std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u, 2), 0u)});
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_FORMAT_UNKNOWN); }
    }
    {  // testutils/fft.go:54-138, the conformance suite every fft.Planner must pass
        auto planner = fft::NewPlanner(ctx);
        const size_t n = 1024;
        const double rate = 1.8e6;
        const double tones[4][2] = {{10.0, 0}, {900000.0, 512}, {450000.0, 256}, {225000.0, 128}};
        auto peak = [](const std::vector<c64> &f) {
            size_t best = 0;
            for (size_t i = 1; i < f.size(); i++)
                if (std::abs(f[i]) > std::abs(f[best])) best = i;
            return best;
        };
        for (auto &t : tones) {
            std::vector<c64> iq(n), freq(n);
            for (size_t i = 0; i < n; i++) {
                double a = 2 * M_PI * t[0] * (double(i) / rate);
                iq[i] = c64((float)std::cos(a), (float)std::sin(a));
            }
            fft::TransformOnce(planner, view(HZSDR_FMT_C64, iq, 1), view(HZSDR_FMT_C64, freq, 1), fft::Forward);
            EXPECT(peak(freq) == (size_t)t[1]);
        }
        for (size_t bin : {5, 10, 127, 522, 242, 415, 825}) {
            std::vector<c64> iq(n), freq(n), back(n);
            freq[bin] = c64(1.f, 1.f);
            fft::TransformOnce(planner, view(HZSDR_FMT_C64, iq, 1), view(HZSDR_FMT_C64, freq, 1), fft::Backward);
            fft::TransformOnce(planner, view(HZSDR_FMT_C64, iq, 1), view(HZSDR_FMT_C64, back, 1), fft::Forward);
            EXPECT(peak(back) == bin);
        }
        std::vector<c64> big(1024), small(128);
        try {
            planner(view(HZSDR_FMT_C64, big, 1), view(HZSDR_FMT_C64, small, 1), fft::Forward);
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_DST_TOO_SMALL); }
        try {
            planner(view(HZSDR_FMT_C64, small, 1), view(HZSDR_FMT_C64, big, 1), fft::Backward);
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_DST_TOO_SMALL); }
    }
    {  // stream/convolution.go:36-82: an all-ones filter in the frequency domain is the
       // identity up to the unnormalised round trip (x N); a ragged last block is dropped
        const size_t n = 1024, blocks = 5;
        std::vector<c64> ones(n, {1.f, 0.f}), in(n * blocks + 100), out(n * blocks);
        for (size_t i = 0; i < in.size(); i++) in[i] = c64(float(int(i % 37) - 18) / 32.f, float(int(i % 11) - 5) / 8.f);
        auto r = stream::ConvolutionReader(ctx, std::make_shared<BufferReader>(view(HZSDR_FMT_C64, in, 1), 2000000u, 700),
                                           view(HZSDR_FMT_C64, ones, 1));
        EXPECT(r->SampleRate() == 2000000u);
        EXPECT(ReadFull(*r, view(HZSDR_FMT_C64, out, 1)) == n * blocks);
        bool ok = true;
        for (size_t i = 0; i < out.size() && ok; i++) ok = std::abs(out[i] - in[i] * float(n)) <= 1e-3f * float(n);
        EXPECT(ok);
        std::vector<c64> one(1);
        try {
            r->Read(view(HZSDR_FMT_C64, one, 1));
            EXPECT(!"no EOF");
        } catch (const Eof &) {}
    }
    {  // fft/convolution.go:97-138: delta convolved with x is x (x N^0: forward-forward-backward
       // is unnormalised by N); cross-correlation of x with itself peaks at lag 0
        const size_t n = 4096;
        std::vector<c64> x(n), d(n), out(n);
        for (size_t i = 0; i < n; i++) x[i] = c64(float(int(i * 7919 % 101) - 50) / 64.f, float(int(i * 104729 % 89) - 44) / 64.f);
        d[3] = c64(1.f, 0.f);
        auto conv = fft::Convolve(ctx, view(HZSDR_FMT_C64, out, 1), view(HZSDR_FMT_C64, x, 1), view(HZSDR_FMT_C64, d, 1));
        (*conv)();
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++) ok = std::abs(out[i] - x[(i + n - 3) % n] * float(n)) <= 2e-3f * float(n);
        EXPECT(ok);
        auto xc = fft::CrossCorrelate(ctx, view(HZSDR_FMT_C64, out, 1), view(HZSDR_FMT_C64, x, 1), view(HZSDR_FMT_C64, x, 1));
        (*xc)();
        int64_t lag = 99;
        check(ctx.raw(), hzsdr_peak_lag(ctx.raw(), out.data(), n, &lag));
        EXPECT(lag == 0);
        std::vector<c64> once(n);  // fft.ConvolveOnce: same result as the planned closure above
        fft::ConvolveOnce(ctx, view(HZSDR_FMT_C64, once, 1), view(HZSDR_FMT_C64, x, 1), view(HZSDR_FMT_C64, d, 1));
        ok = true;
        for (size_t i = 0; i < n && ok; i++) ok = std::abs(once[i] - x[(i + n - 3) % n] * float(n)) <= 2e-3f * float(n);
        EXPECT(ok);
        std::vector<c64> shorter(n / 2);
        try {
            fft::Convolve(ctx, view(HZSDR_FMT_C64, out, 1), view(HZSDR_FMT_C64, x, 1), view(HZSDR_FMT_C64, shorter, 1));
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_LENGTH_MISMATCH || e.status == HZSDR_ERR_DST_TOO_SMALL); }
    }
    {  // stream/beamform.go:148-171 over u8 channels: ConvertReader -> Multiply(w) -> Add,
       // checked against the same three steps made one at a time
        const size_t n = 40000;
        const int K = 4;
        auto w = stream::BeamformAngles(433e6, 30.0, {0.0, 0.1, 0.2, 0.3});
        EXPECT(w[0] == c64(1.f, 0.f));
        std::vector<std::vector<uint8_t>> ch(K, std::vector<uint8_t>(2 * n));
        for (int k = 0; k < K; k++)
            for (size_t i = 0; i < 2 * n; i++) ch[k][i] = (uint8_t)((i * 2654435761u + k * 40503u) >> 13);
        std::vector<ReaderPtr> rs;
        for (int k = 0; k < K; k++) rs.push_back(std::make_shared<BufferReader>(view(HZSDR_FMT_U8, ch[k], 2), 2400000u, 5000));
        auto bf = stream::ReadBeamform(ctx, rs, w);
        EXPECT(bf->SampleFormat() == HZSDR_FMT_C64 && bf->SampleRate() == 2400000u);
        EXPECT(!bf->SetPhaseAngles({c64(1.f, 0.f)}));  // beamform.go:132-134
        std::vector<c64> got(n), acc(n, c64(0.f, 0.f)), tmp(n);
        EXPECT(ReadFull(*bf, view(HZSDR_FMT_C64, got, 1)) == n);
        for (int k = 0; k < K; k++) {
            ctx.ConvertBuffer(view(HZSDR_FMT_C64, tmp, 1), view(HZSDR_FMT_U8, ch[k], 2));
            if (w[k] != c64(1.f, 0.f)) ctx.Multiply(view(HZSDR_FMT_C64, tmp, 1), w[k].real(), w[k].imag());
            ctx.Add(view(HZSDR_FMT_C64, acc, 1), view(HZSDR_FMT_C64, tmp, 1));
        }
        EXPECT(std::memcmp(got.data(), acc.data(), n * sizeof(c64)) == 0);
    }
    {  // sdr.LookupTable (iq_lookup_table.go:98-150): identity table u8 -> u8, then a
       // u8 -> c64 table built with ConvertBuffer equals ConvertBuffer
        Buffer ident = LookupTableIdentity(HZSDR_FMT_U8);
        LookupTable t(ctx, HZSDR_FMT_U8, ident.view);
        std::vector<uint8_t> in(2 * 5000), out(2 * 5000);
        for (size_t i = 0; i < in.size(); i++) in[i] = (uint8_t)(i * 2246822519u >> 11);
        EXPECT(t.Lookup(view(HZSDR_FMT_U8, out, 2), view(HZSDR_FMT_U8, in, 2)) == 5000);
        EXPECT(in == out);
        Buffer tab(HZSDR_FMT_C64, 65536);
        ctx.ConvertBuffer(tab.view, ident.view);
        LookupTable tc(ctx, HZSDR_FMT_U8, tab.view);
        std::vector<c64> a(5000), b(5000);
        tc.Lookup(view(HZSDR_FMT_C64, a, 1), view(HZSDR_FMT_U8, in, 2));
        ctx.ConvertBuffer(view(HZSDR_FMT_C64, b, 1), view(HZSDR_FMT_U8, in, 2));
        EXPECT(std::memcmp(a.data(), b.data(), a.size() * sizeof(c64)) == 0);
        std::vector<c64> tiny(10);
        try {
            tc.Lookup(view(HZSDR_FMT_C64, tiny, 1), view(HZSDR_FMT_U8, in, 2));
            EXPECT(!"no error");
        } catch (const Error &e) { EXPECT(e.status == HZSDR_ERR_DST_TOO_SMALL); }
    }
    {  // the fused chain against the same readers run one by one: u8 -> c64 -> Shift -> Gain
        const size_t n = 1 << 18;
        std::vector<uint8_t> u8(2 * n);
        for (size_t i = 0; i < 2 * n; i++) u8[i] = (uint8_t)(i * 2654435761u >> 17);
        auto src = std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u8, 2), 2400000u);
        auto r = stream::Gain(ctx, stream::ShiftReader(ctx, stream::ConvertReader(ctx, src, HZSDR_FMT_C64), 250000.0), 0.25f);
        std::vector<c64> want(n), got(n);
        EXPECT(ReadFull(*r, view(HZSDR_FMT_C64, want, 1)) == n);
        stream::Chain chain(ctx, HZSDR_FMT_U8, 2400000u);
        chain.Shift(250000.0).Gain(0.25f);
        auto res = chain.Run(view(HZSDR_FMT_U8, u8, 2), view(HZSDR_FMT_C64, got, 1));
        EXPECT(res.first == n && res.second == n);
        EXPECT(std::memcmp(got.data(), want.data(), n * sizeof(c64)) == 0);
    }
    {  // the pinned ring in front of a chain: bit-identical to the synchronous chain, slot by slot
        const size_t slot = 1 << 16, trips = 7;
        const int slots = 3;
        std::vector<uint8_t> u8(2 * slot * trips);
        for (size_t i = 0; i < u8.size(); i++) u8[i] = (uint8_t)(i * 2246822519u >> 15);
        std::vector<c64> taps(257);
        for (size_t k = 0; k < taps.size(); k++) {
            const double t = double(k) - 128.0, s = t == 0 ? 1.0 : std::sin(M_PI * t / 8) / (M_PI * t / 8);
            taps[k] = c64(float(s / 8 * (0.54 - 0.46 * std::cos(2 * M_PI * k / 256.0))), 0.f);
        }
        stream::Chain ref(ctx, HZSDR_FMT_U8, 2400000u), ch(ctx, HZSDR_FMT_U8, 2400000u);
        ref.Shift(-300000.0).FirDecimate(taps, 4);
        ch.Shift(-300000.0).FirDecimate(taps, 4);
        std::vector<c64> want(slot * trips / 4), got;
        for (size_t t = 0; t < trips; t++) {
            auto r = ref.Run(Samples{HZSDR_FMT_U8, u8.data() + 2 * slot * t, slot},
                             Samples{HZSDR_FMT_C64, want.data() + slot / 4 * t, slot / 4});
            EXPECT(r.first == slot && r.second == slot / 4);
        }
        stream::Ring ring(ch, slot, slots);
        EXPECT(ring.IQBuffer(HZSDR_FMT_U8).length == slot * slots);
        auto drain = [&]() {
            Samples o = ring.Pop();
            const c64 *p = (const c64 *)o.data;
            got.insert(got.end(), p, p + o.length);
        };
        for (size_t t = 0; t < trips; t++) {
            if (ring.InFlight() == slots) drain();
            int s = -1;
            Samples in = ring.Acquire(HZSDR_FMT_U8, &s);
            EXPECT(s == int(t % slots));
            std::memcpy(in.data, u8.data() + 2 * slot * t, 2 * slot);
            ring.Submit(s, slot);
        }
        while (ring.InFlight()) drain();
        EXPECT(got.size() == want.size());
        EXPECT(std::memcmp(got.data(), want.data(), want.size() * sizeof(c64)) == 0);
        // and the two mixer orders agree to float32 rounding
        stream::Chain ord(ctx, HZSDR_FMT_U8, 2400000u);
        ord.Shift(-300000.0).FirDecimate(taps, 4).MixInOrder();
        std::vector<c64> o2(slot / 4);
        ord.Run(Samples{HZSDR_FMT_U8, u8.data(), slot}, Samples{HZSDR_FMT_C64, o2.data(), slot / 4});
        float worst = 0.f;
        for (size_t i = 0; i < o2.size(); i++) worst = std::fmax(worst, std::abs(o2[i] - want[i]));
        EXPECT(worst < 2e-6f);
    }
    {  // nested Readers that fuse and read ahead (stream::Fused; go/hip/fused.go): Gain(ShiftReader(ConvertReader(u8 source)))
       // and DecimateReader(Multiply(ConvertReader(..))) against the same nests in the reference's own structure -- the
       // same samples bit for bit, the source's partial last block dropped, far fewer calls of the library
        const size_t n = (1 << 22) + 4321;
        std::vector<uint8_t> u8(2 * n);
        for (size_t i = 0; i < u8.size(); i++) u8[i] = (uint8_t)((i * 2654435761u) >> 23);
        auto drain = [](Reader &r, std::vector<c64> &out) {
            std::vector<c64> buf(50001);
            try {
                for (;;) {
                    const size_t k = r.Read(Samples{HZSDR_FMT_C64, buf.data(), buf.size()});
                    out.insert(out.end(), buf.begin(), buf.begin() + (long)k);
                }
            } catch (const Eof &) {
            }
        };
        const stream::Fused f{ctx, 32};
        for (int nest = 0; nest < 2; nest++) {
            auto src_a = std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u8, 2), 2400000u, 77777);
            auto src_b = std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u8, 2), 2400000u, 12345);
            ReaderPtr plain, fused;
            if (nest == 0) {
                plain = stream::Gain(ctx, stream::ShiftReader(ctx, stream::ConvertReader(ctx, src_a, HZSDR_FMT_C64), 310000.0), 0.5f);
                fused = f.Gain(f.ShiftReader(f.ConvertReader(src_b, HZSDR_FMT_C64), 310000.0), 0.5f);
            } else {
                plain = stream::DecimateReader(ctx, stream::Multiply(ctx, stream::ConvertReader(ctx, src_a, HZSDR_FMT_C64), 0.6f, 0.8f), 5);
                fused = f.DecimateReader(f.Multiply(f.ConvertReader(src_b, HZSDR_FMT_C64), 0.6f, 0.8f), 5);
            }
            EXPECT(std::dynamic_pointer_cast<stream::ChainReader>(fused) != nullptr);
            EXPECT(fused->SampleRate() == plain->SampleRate() && fused->SampleFormat() == HZSDR_FMT_C64);
            std::vector<c64> a, b;
            const unsigned long long c0 = ctx.CallCount();
            drain(*plain, a);
            const unsigned long long c1 = ctx.CallCount();
            drain(*fused, b);
            const unsigned long long c2 = ctx.CallCount();
            const size_t whole = n / 32768 * 32768;
            EXPECT(nest == 0 ? a.size() == whole : (a.size() > 0 && a.size() % (whole / 32768) == 0));  // (whole 32 Ki blocks only)
            EXPECT(a.size() == b.size());
            EXPECT(a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(c64)) == 0);
            EXPECT(c2 - c1 < 40 && c1 - c0 >= whole / 32768);
        }
    }
    {  // the north-star chain as ONE Reader (stream::Fused::FirDecimateReader): ConvertReader -> ShiftReader -> 1024-tap FIR,
       // decimate by 8, over a u8 source; nine ring slots of 2^18 samples, four per call of the chain -- against the same
       // stream through the synchronous chain cut at the slots' boundaries, bit for bit, and far fewer library calls than slots
        const size_t slot = (size_t)1 << 18, n = 13 * slot + 2 * 32768 + 77;  // (13 full slots, a short one, a ragged tail that is dropped)
        const unsigned D = 8;
        std::vector<uint8_t> u8(2 * n);
        for (size_t i = 0; i < u8.size(); i++) u8[i] = (uint8_t)((i * 2246822519u) >> 21);
        std::vector<c64> taps(1024);
        for (int k = 0; k < 1024; k++) {
            const double t = (k - 511.5) / 16.0, w = 0.54 - 0.46 * std::cos(6.283185307179586 * k / 1023.0);
            taps[k] = c64((float)((std::fabs(t) < 1e-12 ? 1.0 : std::sin(3.141592653589793 * t) / (3.141592653589793 * t)) / 16.0 * w), 0.f);
        }
        const stream::Fused f{ctx, 8};  // readahead 8 blocks of 32 Ki = 2^18 samples per slot
        auto src = std::make_shared<BufferReader>(view(HZSDR_FMT_U8, u8, 2), 20000000u, 99991);
        auto rd = f.FirDecimateReader(f.ShiftReader(f.ConvertReader(src, HZSDR_FMT_C64), -2.5e6), taps, D, 9, 4);
        EXPECT(rd->SampleRate() == 20000000u / D && rd->SampleFormat() == HZSDR_FMT_C64);
        std::vector<c64> got, buf(40000);
        const unsigned long long c0 = ctx.CallCount();
        try {
            for (;;) {
                const size_t k = rd->Read(Samples{HZSDR_FMT_C64, buf.data(), buf.size()});
                got.insert(got.end(), buf.begin(), buf.begin() + (long)k);
            }
        } catch (const Eof &) {
        }
        const unsigned long long calls = ctx.CallCount() - c0;
        const size_t whole = (13 * slot + 2 * 32768);
        EXPECT(got.size() == whole / D);
        EXPECT(calls <= 14 + 8 + 12);  // 14 pops, at most 4 + 1 + 1 submits (groups of four), the chain's and the ring's construction
        int kern = 0;
        EXPECT(rd->chain() && hzsdr_chain_last_fir_kernel(rd->chain()->raw(), &kern) == HZSDR_OK && kern == HZSDR_FIR_KERNEL_MATRIX_PASSES);
        stream::Chain ch(ctx, HZSDR_FMT_U8, 20000000u);
        ch.Shift(-2.5e6).FirDecimate(taps, D);
        std::vector<c64> ref(whole / D);
        for (size_t a = 0; a < whole; a += slot) {
            const size_t b = std::min(a + slot, whole);
            const auto r = ch.Run(Samples{HZSDR_FMT_U8, u8.data() + 2 * a, b - a}, Samples{HZSDR_FMT_C64, ref.data() + a / D, (b - a) / D});
            EXPECT(r.first == b - a && r.second == (b - a) / D);
        }
        EXPECT(got.size() == ref.size() && std::memcmp(got.data(), ref.data(), ref.size() * sizeof(c64)) == 0);
    }
    std::printf(failures ? "%d FAILED\n" : "all host-mirror tests passed\n", failures);
    return failures ? 1 : 0;
}
