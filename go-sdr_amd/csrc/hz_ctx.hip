// hz_ctx.hip -- context lifecycle, memory, staging (C-ABI: library / context section).
#include "hz_common.h"

#include <mutex>
#include <set>
#include <utility>

namespace hz {
int raise_dynamic_lds(const void *kernel) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> raised;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return HZSDR_ERR_HIP;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.count({kernel, dev})) return HZSDR_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return HZSDR_ERR_HIP;
    raised.insert({kernel, dev});
    return HZSDR_OK;
}
}  // namespace hz

namespace hz {

int ensure_slot(hzsdr_ctx *ctx, int slot, size_t bytes) {
    auto &s = ctx->slots[slot];
    if (bytes <= s.cap) return HZSDR_OK;
    if (s.ptr) {
        // the old buffer may still be in use by enqueued work
        HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HZ_HIP(ctx, hipFree(s.ptr));
        s.ptr = nullptr;
        s.cap = 0;
    }
    size_t cap = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    HZ_HIP(ctx, hipMalloc(&s.ptr, cap));
    s.cap = cap;
    return HZSDR_OK;
}

int ensure_pinned(hzsdr_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pinned_cap) return HZSDR_OK;
    if (ctx->pinned) {
        HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HZ_HIP(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_cap = 0;
    }
    size_t cap = bytes < 4096 ? 4096 : bytes;
    HZ_HIP(ctx, hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
    ctx->pinned_cap = cap;
    return HZSDR_OK;
}

bool Stage::pinned_by_us(const void *p, size_t bytes) const {
    const char *c = (const char *)p;
    for (auto &r : ctx->pinned_ranges)
        if (c >= r.first && c + bytes <= r.first + r.second) return true;
    return false;
}

// `bytes` of the context's pinned staging area (256-byte aligned), or nullptr if the call
// has outgrown the small-call route
void *Stage::stage_small(size_t bytes) {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (used + need > kZeroCopyMax) return nullptr;
    if (!ctx->hstage) {
        if (hipHostMalloc(&ctx->hstage, kZeroCopyMax, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            ctx->hstage = nullptr;
            return nullptr;
        }
        ctx->hstage_cap = kZeroCopyMax;
    }
    void *p = (char *)ctx->hstage + used;
    used += need;
    return p;
}

int Stage::in(int slot, const void *p, size_t bytes, const void **dev) {
    if (!host() || bytes == 0 || pinned_by_us(p, bytes)) {
        *dev = p;
        return HZSDR_OK;
    }
    if (void *s = stage_small(bytes)) {
        memcpy(s, p, bytes);
        *dev = s;
        return HZSDR_OK;
    }
    HZ_TRY(ensure_slot(ctx, slot, bytes));
    HZ_HIP(ctx, hipMemcpyAsync(ctx->slots[slot].ptr, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = ctx->slots[slot].ptr;
    return HZSDR_OK;
}

int Stage::out(int slot, void *p, size_t bytes, void **dev) {
    if (!host() || bytes == 0 || pinned_by_us(p, bytes)) {
        *dev = p;
        return HZSDR_OK;
    }
    if (void *s = stage_small(bytes)) {
        *dev = s;
        backs.push_back({p, s, bytes, true});
        return HZSDR_OK;
    }
    HZ_TRY(ensure_slot(ctx, slot, bytes));
    *dev = ctx->slots[slot].ptr;
    backs.push_back({p, *dev, bytes, false});
    return HZSDR_OK;
}

int Stage::inout(int slot, void *p, size_t bytes, void **dev) {
    if (!host() || bytes == 0 || pinned_by_us(p, bytes)) {
        *dev = p;
        return HZSDR_OK;
    }
    if (void *s = stage_small(bytes)) {
        memcpy(s, p, bytes);
        *dev = s;
        backs.push_back({p, s, bytes, true});
        return HZSDR_OK;
    }
    HZ_TRY(ensure_slot(ctx, slot, bytes));
    HZ_HIP(ctx, hipMemcpyAsync(ctx->slots[slot].ptr, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = ctx->slots[slot].ptr;
    backs.push_back({p, *dev, bytes, false});
    return HZSDR_OK;
}

int Stage::finish() {
    HZ_HIP(ctx, hipGetLastError());
    if (!host()) return HZSDR_OK;
    for (auto &b : backs)
        if (!b.cpu) HZ_HIP(ctx, hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, ctx->stream));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &b : backs)
        if (b.cpu) memcpy(b.host, b.dev, b.bytes);
    return HZSDR_OK;
}

static bool device_is_gfx950(int dev) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace hz

extern "C" {

int hzsdr_device_count(int *count) {
    if (!count) return HZSDR_ERR_INVALID_ARGUMENT;
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return HZSDR_ERR_NO_DEVICE;
    }
    int ok = 0;
    for (int i = 0; i < n; i++)
        if (hz::device_is_gfx950(i)) ok++;
    *count = ok;
    return ok > 0 ? HZSDR_OK : HZSDR_ERR_NO_DEVICE;
}

int hzsdr_open(int device, int memspace, hzsdr_ctx **out) {
    if (!out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (memspace != HZSDR_MEM_HOST && memspace != HZSDR_MEM_DEVICE) return HZSDR_ERR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return HZSDR_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return HZSDR_ERR_INVALID_ARGUMENT;
    // This library carries gfx950 code objects only: fail loudly elsewhere.
    if (!hz::device_is_gfx950(device)) return HZSDR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return HZSDR_ERR_NO_DEVICE;
    hzsdr_ctx *ctx = new hzsdr_ctx();
    ctx->device = device;
    ctx->memspace = memspace;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0)
        ctx->num_cus = p.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return HZSDR_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return HZSDR_OK;
}

int hzsdr_close(hzsdr_ctx *ctx) {
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->slots)
        if (s.ptr) (void)hipFree(s.ptr);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->hstage) (void)hipHostFree(ctx->hstage);
    for (auto &kv : ctx->twiddles) (void)hipFree(kv.second);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return HZSDR_OK;
}

const char *hzsdr_last_error(const hzsdr_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int hzsdr_memspace(const hzsdr_ctx *ctx) { return ctx ? ctx->memspace : -1; }

int hzsdr_set_stream(hzsdr_ctx *ctx, void *hip_stream) {
    HZ_TRY(hz::enter(ctx));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    return HZSDR_OK;
}

int hzsdr_use_own_stream(hzsdr_ctx *ctx) {
    HZ_TRY(hz::enter(ctx));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return HZSDR_OK;
}

void *hzsdr_get_stream(const hzsdr_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int hzsdr_synchronize(hzsdr_ctx *ctx) {
    HZ_TRY(hz::enter(ctx));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t s : ctx->side_streams) HZ_HIP(ctx, hipStreamSynchronize(s));  // (a pipelined chain's last history kernel)
    return HZSDR_OK;
}

int hzsdr_call_count(const hzsdr_ctx *ctx, unsigned long long *calls) {
    if (!ctx || !calls) return HZSDR_ERR_INVALID_ARGUMENT;
    *calls = ctx->calls;
    return HZSDR_OK;
}

int hzsdr_malloc_device(hzsdr_ctx *ctx, size_t bytes, void **out) {
    HZ_TRY(hz::enter(ctx));
    if (!out) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_HIP(ctx, hipMalloc(out, bytes ? bytes : 1));
    return HZSDR_OK;
}

int hzsdr_free_device(hzsdr_ctx *ctx, void *ptr) {
    HZ_TRY(hz::enter(ctx));
    HZ_HIP(ctx, hipFree(ptr));
    return HZSDR_OK;
}

int hzsdr_malloc_pinned(hzsdr_ctx *ctx, size_t bytes, void **out) {
    HZ_TRY(hz::enter(ctx));
    if (!out) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    ctx->pinned_ranges.push_back({(const char *)*out, bytes ? bytes : 1});
    return HZSDR_OK;
}

int hzsdr_free_pinned(hzsdr_ctx *ctx, void *ptr) {
    HZ_TRY(hz::enter(ctx));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a kernel may still be reading it
    for (size_t i = 0; i < ctx->pinned_ranges.size(); i++)
        if (ctx->pinned_ranges[i].first == (const char *)ptr) {
            ctx->pinned_ranges.erase(ctx->pinned_ranges.begin() + (long)i);
            break;
        }
    HZ_HIP(ctx, hipHostFree(ptr));
    return HZSDR_OK;
}

int hzsdr_memcpy_h2d(hzsdr_ctx *ctx, void *dst_device, const void *src_host, size_t bytes) {
    HZ_TRY(hz::enter(ctx));
    HZ_HIP(ctx, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return HZSDR_OK;
}

int hzsdr_memcpy_d2h(hzsdr_ctx *ctx, void *dst_host, const void *src_device, size_t bytes) {
    HZ_TRY(hz::enter(ctx));
    HZ_HIP(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return HZSDR_OK;
}

}  // extern "C"
