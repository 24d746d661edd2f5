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
    std::printf(failures ? "%d FAILED\n" : "all host-mirror tests passed\n", failures);
    return failures ? 1 : 0;
}
