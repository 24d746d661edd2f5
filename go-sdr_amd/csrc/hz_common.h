// hz_common.h -- host-side plumbing shared by the C-ABI translation units:
// the context object, HIP error capture, host<->device staging for
// HZSDR_MEM_HOST contexts, and launch geometry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/hzsdr.h"

namespace hz {
// Dynamic LDS above the default needs the kernel's limit raised (to the CU's whole 160 KiB: a later launch of the
// same instantiation may ask for more than the first did), once per (kernel, device); hz_ctx.hip.  Thread-safe.
int raise_dynamic_lds(const void *kernel);
}  // namespace hz

struct hzsdr_ctx {
    int device = 0;
    int memspace = HZSDR_MEM_HOST;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    unsigned long long calls = 0;  // hzsdr_call_count
    std::string last_error;
    // grow-only device scratch slots used to stage host buffers
    struct Slot {
        void *ptr = nullptr;
        size_t cap = 0;
    };
    Slot slots[24];
    // small pinned buffer for per-call parameter tables (pointer lists, ...)
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    // HOST memory space, small calls: one pinned, GPU-visible staging area.  Kernels read
    // and write it directly over PCIe (no DMA launch on either side of the kernel), the
    // CPU copies in and out of it; see Stage.
    void *hstage = nullptr;
    size_t hstage_cap = 0;
    // host ranges that are already pinned and GPU-visible (hzsdr_malloc_pinned, ring slots):
    // buffers inside them are handed to the kernels as they are, no staging at all
    std::vector<std::pair<const char *, size_t>> pinned_ranges;
    // FFT twiddle tables exp(-2 pi i m / N), m < N, keyed by N (device memory)
    std::map<size_t, void *> twiddles;
    // streams that objects of this context own and launch on beside `stream` (pipelined chains: hzsdr_chain_pipeline);
    // hzsdr_synchronize waits for them as well
    std::vector<hipStream_t> side_streams;
};

namespace hz {

constexpr int kThreads = 256;
constexpr size_t kReaderBlock = 32 * 1024;  // stream/convert.go:43-44, decimate.go:41-42

inline int fail(hzsdr_ctx *ctx, int status, const std::string &msg) {
    if (ctx) ctx->last_error = msg;
    return status;
}

inline int hip_fail(hzsdr_ctx *ctx, hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (ctx) ctx->last_error = buf;
    return e == hipErrorOutOfMemory ? HZSDR_ERR_OUT_OF_MEMORY : HZSDR_ERR_HIP;
}

#define HZ_HIP(ctx, call)                                                    \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return hz::hip_fail((ctx), e__, #call, __FILE__, __LINE__); \
    } while (0)

#define HZ_TRY(expr)                    \
    do {                                \
        int rc__ = (expr);              \
        if (rc__ != HZSDR_OK) return rc__; \
    } while (0)

inline int format_size(int f) {
    switch (f) {
    case HZSDR_FMT_U8:
    case HZSDR_FMT_I8: return 2;
    case HZSDR_FMT_I16: return 4;
    case HZSDR_FMT_C64: return 8;
    default: return 0;
    }
}

// Select the context's device on the calling thread (cgo: goroutines migrate
// between OS threads and HIP's current device is thread-local).
inline int enter(hzsdr_ctx *ctx) {
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    ctx->calls++;
    HZ_HIP(ctx, hipSetDevice(ctx->device));
    return HZSDR_OK;
}

// Blocks for a grid-stride launch over `items` work items: enough to fill the
// chip many times over.  The cap was 8 workgroups per CU at first; measured on 2^24-sample
// buffers, 32 and 128 per CU are faster for the streaming kernels (in-place Scale 56 -> 44 us,
// 4-channel Beamform 117 -> 111 us, u8 -> c64 28.3 -> 27.5 us): short loops per thread
// spread the HBM channels better than long ones, and the dispatch cost does not show.
inline unsigned blocks_for(const hzsdr_ctx *ctx, size_t items, int threads = kThreads) {
    size_t b = (items + threads - 1) / threads;
    size_t cap = (size_t)ctx->num_cus * 128;
    if (b > cap) b = cap;
    if (b == 0) b = 1;
    return (unsigned)b;
}

int ensure_slot(hzsdr_ctx *ctx, int slot, size_t bytes);
int ensure_pinned(hzsdr_ctx *ctx, size_t bytes);

// Staging helpers.  In a DEVICE context they return the caller's pointer; in a
// HOST context they return device scratch (copying in when asked) and remember
// what must be copied back.
// Three routes for a HOST-space buffer (picked per buffer):
//   * inside a range the library pinned itself (hzsdr_malloc_pinned, ring slots): the kernel
//     gets the pointer as it is -- zero copies;
//   * up to kZeroCopyMax bytes per call in total: the CPU copies it into / out of the
//     context's pinned staging area and the kernel reads / writes that area over PCIe
//     directly -- no DMA launches, one stream wait.  (A 32 Ki-sample reader block is all
//     latency: two hipMemcpyAsync launches cost more than moving its 320 KB.)
//   * larger: device scratch + hipMemcpyAsync each way, as before.
constexpr size_t kZeroCopyMax = (size_t)2 << 20;

struct Stage {
    hzsdr_ctx *ctx;
    struct Back {
        void *host;
        const void *dev;
        size_t bytes;
        bool cpu;  // copy back with memcpy after the stream wait (staging area) instead of a DMA
    };
    std::vector<Back> backs;
    size_t used = 0;  // bytes of the staging area handed out so far
    explicit Stage(hzsdr_ctx *c) : ctx(c) {}
    bool pinned_by_us(const void *p, size_t bytes) const;
    void *stage_small(size_t bytes);
    bool host() const { return ctx->memspace == HZSDR_MEM_HOST; }
    // read-only input
    int in(int slot, const void *p, size_t bytes, const void **dev);
    // output only (contents need not be uploaded)
    int out(int slot, void *p, size_t bytes, void **dev);
    // read-modify-write
    int inout(int slot, void *p, size_t bytes, void **dev);
    // like out(), but uploads the current contents first (partial writers)
    int out_preserve(int slot, void *p, size_t bytes, void **dev) { return inout(slot, p, bytes, dev); }
    // enqueue the copies back and, for HOST contexts, wait for them
    int finish();
};

}  // namespace hz
