// hz_ring.hip -- the pinned ring in front of a chain.
#include "hz_chain_host.h"

// =============================================================================
// Pinned ring in front of a chain: the memory a
// stream.RingBuffer's IQBufferAllocator hands out (stream/ring.go:60-68) is ONE
// hipHostMalloc region of slots * slot_length samples; each submitted slot goes
// upload -> chain kernel -> download on three streams chained by events, so the
// PCIe transfers of neighbouring slots overlap the kernel.
// =============================================================================

struct hzsdr_ring {
    hzsdr_chain *chain = nullptr;
    hzsdr_ctx *ctx = nullptr;
    size_t slot_len = 0, out_len = 0;  // samples per slot in / out
    int nslots = 0;
    hipStream_t s_up = nullptr, s_down = nullptr;
    char *pin_in = nullptr, *pin_out = nullptr;  // nslots * slot bytes each, contiguous
    char *dev_in = nullptr, *dev_out = nullptr;
    struct Slot {
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;
        size_t n_out = 0;
        int state = 0;  // 0 free, 1 acquired, 2 in flight
    };
    std::vector<Slot> slots;
    // next slot to acquire / to submit / to pop: several slots may be acquired ahead of their submission (round 6:
    // hzsdr_ring_submit_many puts them into ONE call of the chain)
    size_t aidx = 0, widx = 0, ridx = 0;
    int inflight = 0;
    size_t in_bytes() const { return slot_len * (size_t)hz::format_size(chain->src_fmt); }
    size_t out_bytes() const { return out_len * 8; }
};

extern "C" {

int hzsdr_ring_free(hzsdr_ring *r) {
    if (!r) return HZSDR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(r->ctx->device);
    if (r->s_up) (void)hipStreamSynchronize(r->s_up);
    (void)hipStreamSynchronize(r->ctx->stream);
    if (r->s_down) (void)hipStreamSynchronize(r->s_down);
    for (auto &s : r->slots) {
        if (s.up) (void)hipEventDestroy(s.up);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.down) (void)hipEventDestroy(s.down);
    }
    for (char *p : {r->pin_in, r->pin_out})
        for (size_t i = 0; p && i < r->ctx->pinned_ranges.size(); i++)
            if (r->ctx->pinned_ranges[i].first == p) {
                r->ctx->pinned_ranges.erase(r->ctx->pinned_ranges.begin() + (long)i);
                break;
            }
    if (r->pin_in) (void)hipHostFree(r->pin_in);
    if (r->pin_out) (void)hipHostFree(r->pin_out);
    if (r->dev_in) (void)hipFree(r->dev_in);
    if (r->dev_out) (void)hipFree(r->dev_out);
    if (r->s_up) (void)hipStreamDestroy(r->s_up);
    if (r->s_down) (void)hipStreamDestroy(r->s_down);
    delete r;
    return HZSDR_OK;
}

int hzsdr_ring_create(hzsdr_chain *c, size_t slot_length, int slots, hzsdr_ring **out) {
    using namespace hz;
    if (!c || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    hzsdr_ctx *ctx = c->ctx;
    if (slots < 2 || slots > 64 || slot_length == 0)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: needs 2..64 slots of at least one sample");
    size_t cons, outn;
    hzsdr_chain_plan(c, slot_length, &cons, &outn);
    if (cons != slot_length)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: slot length must be a whole number of the chain's blocks");
    HZ_TRY(enter(ctx));
    hzsdr_ring *r = new hzsdr_ring();
    r->chain = c;
    r->ctx = ctx;
    r->slot_len = slot_length;
    r->out_len = outn ? outn : 1;
    r->nslots = slots;
    r->slots.resize(slots);
#define HZ_RING(call)                                                         \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) {                                              \
            int rc__ = hip_fail(ctx, e__, #call, __FILE__, __LINE__);         \
            hzsdr_ring_free(r);                                               \
            return rc__;                                                      \
        }                                                                     \
    } while (0)
    HZ_RING(hipStreamCreateWithFlags(&r->s_up, hipStreamNonBlocking));
    HZ_RING(hipStreamCreateWithFlags(&r->s_down, hipStreamNonBlocking));
    HZ_RING(hipHostMalloc((void **)&r->pin_in, r->in_bytes() * slots, hipHostMallocDefault));
    HZ_RING(hipHostMalloc((void **)&r->pin_out, r->out_bytes() * slots, hipHostMallocDefault));
    HZ_RING(hipMalloc((void **)&r->dev_in, r->in_bytes() * slots));
    HZ_RING(hipMalloc((void **)&r->dev_out, r->out_bytes() * slots));
    // ring slots are pinned and GPU-visible: any HOST-space call on them skips its staging
    ctx->pinned_ranges.push_back({r->pin_in, r->in_bytes() * slots});
    ctx->pinned_ranges.push_back({r->pin_out, r->out_bytes() * slots});
    for (auto &s : r->slots) {
        HZ_RING(hipEventCreateWithFlags(&s.up, hipEventDisableTiming));
        HZ_RING(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HZ_RING(hipEventCreateWithFlags(&s.down, hipEventDisableTiming));
    }
#undef HZ_RING
    *out = r;
    return HZSDR_OK;
}

int hzsdr_ring_iq_buffer(const hzsdr_ring *r, void **base, size_t *n_samples, size_t *slot_length) {
    if (!r) return HZSDR_ERR_INVALID_ARGUMENT;
    if (base) *base = r->pin_in;
    if (n_samples) *n_samples = r->slot_len * (size_t)r->nslots;
    if (slot_length) *slot_length = r->slot_len;
    return HZSDR_OK;
}

int hzsdr_ring_acquire(hzsdr_ring *r, int *slot, void **iq) {
    using namespace hz;
    if (!r || !slot) return HZSDR_ERR_INVALID_ARGUMENT;
    const int i = (int)(r->aidx % (size_t)r->nslots);
    if (r->slots[i].state != 0)
        return fail(r->ctx, HZSDR_ERR_DST_TOO_SMALL, "ring: every slot is acquired or in flight (pop first)");  // the overrun case
    r->slots[i].state = 1;
    r->aidx++;
    *slot = i;
    if (iq) *iq = r->pin_in + (size_t)i * r->in_bytes();
    return HZSDR_OK;
}

// `count` acquired slots, oldest first, n samples each: their uploads, ONE call of the chain over all of them where the
// chain has a one-launch form (chain_launch with a CallBatch: the persistent-pass FIR kernel, hzsdr_chain_run_batch's
// rules -- otherwise slot by slot, the same results), their downloads.
static int ring_submit(hzsdr_ring *r, int first, int count, size_t n) {
    using namespace hz;
    hzsdr_ctx *ctx = r->ctx;
    if (count < 1 || count > mm2::kMaxBatch || count > r->nslots) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: 1..8 slots per submit");
    if (first != (int)(r->widx % (size_t)r->nslots))
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: submit of a slot that is not the oldest acquired one");
    for (int j = 0; j < count; j++)
        if (r->slots[(size_t)(first + j) % (size_t)r->nslots].state != 1)
            return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: submit of a slot that is not acquired");
    size_t cons, outn;
    hzsdr_chain_plan(r->chain, n, &cons, &outn);
    if (n == 0 || n > r->slot_len || cons != n)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: a slot must hold a whole number of the chain's blocks");
    HZ_TRY(enter(ctx));
    const size_t fs = (size_t)format_size(r->chain->src_fmt);
    const void *dins[mm2::kMaxBatch];
    void *douts[mm2::kMaxBatch];
    int idx[mm2::kMaxBatch];
    for (int j = 0; j < count; j++) {
        idx[j] = (int)((size_t)(first + j) % (size_t)r->nslots);
        dins[j] = r->dev_in + (size_t)idx[j] * r->in_bytes();
        douts[j] = r->dev_out + (size_t)idx[j] * r->out_bytes();
    }
    // (a slot's previous use was popped -- its download, behind every kernel that read or wrote its device copies,
    // has completed -- before it could be acquired again: the upload stream needs no further wait)
    // Uploads: neighbouring full slots are neighbours in the pinned region and on the device -- one copy per stretch.
    for (int j = 0; j < count;) {
        int e = j + 1;
        while (e < count && n == r->slot_len && idx[e] == idx[e - 1] + 1) e++;
        HZ_HIP(ctx, hipMemcpyAsync((void *)dins[j], r->pin_in + (size_t)idx[j] * r->in_bytes(), (size_t)(e - j - 1) * r->in_bytes() + n * fs,
                                   hipMemcpyHostToDevice, r->s_up));
        j = e;
    }
    auto &last = r->slots[idx[count - 1]];
    HZ_HIP(ctx, hipEventRecord(last.up, r->s_up));
    // The ring knows what the call's buffers wait for -- these uploads, nothing else: the slots' device copies are its
    // own -- so a pipelined chain may overlap the call with the one before (hzsdr_chain_run_after's contract); any
    // other chain's launch waits for the event on the context's stream (pipeline_drain).
    struct Scope {  // (the call's ordering lives in the chain for the launches' duration)
        hzsdr_chain *c;
        ~Scope() { c->relaxed = false, c->ready = nullptr; }
    } scope{r->chain};
    r->chain->relaxed = r->chain->pipelined;  // (the slots' device copies, whatever the context's memory space)
    r->chain->ready = last.up;
    if (!r->chain->relaxed) {
        r->chain->ready = nullptr;
        HZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, last.up, 0));
    }
    double ts;
    int rc = kBatchFallback;
    if (count > 1) {
        const CallBatch cb{dins, douts, (size_t)count, n, outn};
        rc = chain_launch(r->chain, dins[0], n * (size_t)count, douts[0], outn * (size_t)count, &ts, &cb);
        if (rc == HZSDR_OK) r->chain->ts = ts;
    }
    if (rc == kBatchFallback) {
        rc = HZSDR_OK;
        for (int j = 0; j < count && rc == HZSDR_OK; j++) {
            r->chain->relaxed = r->chain->pipelined;
            r->chain->ready = r->chain->relaxed ? last.up : nullptr;
            rc = chain_launch(r->chain, dins[j], n, douts[j], outn, &ts);
            if (rc == HZSDR_OK) r->chain->ts = ts;
        }
    }
    HZ_TRY(rc);
    HZ_HIP(ctx, hipGetLastError());
    HZ_HIP(ctx, hipEventRecord(last.done, ctx->stream));
    HZ_HIP(ctx, hipStreamWaitEvent(r->s_down, last.done, 0));
    for (int j = 0; j < count; j++) {  // (a download and an event per slot: hzsdr_ring_pop hands them out one by one)
        auto &s = r->slots[idx[j]];
        HZ_HIP(ctx, hipMemcpyAsync(r->pin_out + (size_t)idx[j] * r->out_bytes(), douts[j], outn * 8, hipMemcpyDeviceToHost, r->s_down));
        HZ_HIP(ctx, hipEventRecord(s.down, r->s_down));
        s.n_out = outn;
        s.state = 2;
    }
    r->widx += (size_t)count;
    r->inflight += count;
    return HZSDR_OK;
}

int hzsdr_ring_submit(hzsdr_ring *r, int slot, size_t n) {
    if (!r || slot < 0 || slot >= r->nslots) return HZSDR_ERR_INVALID_ARGUMENT;
    return ring_submit(r, slot, 1, n);
}

int hzsdr_ring_submit_many(hzsdr_ring *r, int first_slot, int count, size_t n) {
    if (!r || first_slot < 0 || first_slot >= r->nslots) return HZSDR_ERR_INVALID_ARGUMENT;
    return ring_submit(r, first_slot, count, n);
}

int hzsdr_ring_release(hzsdr_ring *r, int slot) {
    using namespace hz;
    if (!r || slot < 0 || slot >= r->nslots) return HZSDR_ERR_INVALID_ARGUMENT;
    auto &s = r->slots[slot];
    // (the NEWEST acquired slot: what a source that ran dry hands back)
    if (s.state != 1 || r->aidx == r->widx || slot != (int)((r->aidx - 1) % (size_t)r->nslots))
        return fail(r->ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: release of a slot that is not the last acquired one");
    s.state = 0;
    r->aidx--;
    return HZSDR_OK;
}

int hzsdr_ring_pop(hzsdr_ring *r, const void **out, size_t *n_out) {
    using namespace hz;
    if (!r) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = r->ctx;
    const int i = (int)(r->ridx % (size_t)r->nslots);
    auto &s = r->slots[i];
    if (s.state != 2) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: nothing in flight");  // the underrun case
    HZ_TRY(enter(ctx));
    HZ_HIP(ctx, hipEventSynchronize(s.down));
    if (out) *out = r->pin_out + (size_t)i * r->out_bytes();
    if (n_out) *n_out = s.n_out;
    s.state = 0;
    r->ridx++;
    r->inflight--;
    return HZSDR_OK;
}

int hzsdr_ring_in_flight(const hzsdr_ring *r) { return r ? r->inflight : -1; }

}  // extern "C"

