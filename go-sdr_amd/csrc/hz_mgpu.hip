// hz_mgpu.hip -- Beamform sharded over GPUs from ONE process (C-ABI: hzsdr_mgpu_*).
//
// A Go program is one process with one context per GPU; the reference sums its coherent
// channels in one place (stream/beamform.go:148-171 -> stream/add.go:115-119).  Here channel
// c lives on shard owner(c) (contiguous ranges, hzsdr_mgpu_shard_channels), every shard
// weights its own channels locally, and ONE exchange step combines them:
//
//   HZSDR_MGPU_ORDERED  the parity path.  Shard s owns slice s of the output;
//       every shard copies slice s of each of its weighted channels 0 + w_c x_c to shard s
//       (hipMemcpyPeerAsync: xGMI between GPUs, a device-to-device copy when two shards
//       share a GPU), shard s adds the K pieces in channel order from +0 (the sum kernel of
//       hzsdr_sum), and the slices are collected on the destination shard.  Bit-identical to
//       the one-GPU / reference result; every link carries 1/G of a channel.
//   HZSDR_MGPU_RCCL     the fast path.  Per-shard partial sums (hzsdr_beamform_partial) and
//       one ncclReduce(float sum) onto the destination: RCCL's order, so a few ULP from the
//       reference, not bit-identical.  librccl is loaded at run time (dlopen), the library
//       itself keeps depending on libamdhip64 only; ncclCommInitAll needs distinct GPUs.
//
// Streams: one per shard (its context's); cross-shard ordering by events only -- no host
// wait inside hzsdr_mgpu_beamform; hzsdr_mgpu_synchronize waits for every shard.
#include <dlfcn.h>

#include "hz_common.h"

namespace {

// the handful of RCCL entry points used, resolved with dlsym (rccl.h is not needed to build)
typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Reduce;
    }
};
constexpr int kNcclFloat = 7, kNcclSum = 0;  // ncclFloat32, ncclSum (rccl.h enums)

}  // namespace

struct hzsdr_mgpu {
    int g = 0;
    std::vector<int> dev;
    std::vector<hzsdr_ctx *> ctx;
    std::vector<hipEvent_t> ready, done;  // per shard: weighted channels ready / its part of the exchange done
    // grow-only device scratch per shard: weighted channels, received pieces, output slice / partial
    std::vector<void *> wbuf, pbuf, obuf;
    std::vector<size_t> wcap, pcap, ocap;
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::string last_error;
    int pairs_direct = 0, pairs_staged = 0;  // ordered pairs of distinct GPUs: peer access enabled / copies stage through the host
};

namespace hz {

static int mfail(hzsdr_mgpu *m, int status, const std::string &msg) {
    if (m) m->last_error = msg;
    return status;
}

static int grow(hzsdr_mgpu *m, int s, std::vector<void *> &buf, std::vector<size_t> &cap, size_t bytes) {
    if (bytes <= cap[s]) return HZSDR_OK;
    hzsdr_ctx *c = m->ctx[s];
    HZ_TRY(enter(c));
    if (buf[s]) {
        HZ_HIP(c, hipStreamSynchronize(c->stream));
        HZ_HIP(c, hipFree(buf[s]));
        buf[s] = nullptr;
        cap[s] = 0;
    }
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HZ_HIP(c, hipMalloc(&buf[s], want));
    cap[s] = want;
    return HZSDR_OK;
}

}  // namespace hz

extern "C" {

int hzsdr_mgpu_shard_channels(int n_channels, int n_shards, int shard, int *lo, int *hi) {
    if (n_channels < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !lo || !hi) return HZSDR_ERR_INVALID_ARGUMENT;
    *lo = (int)((long)n_channels * shard / n_shards);
    *hi = (int)((long)n_channels * (shard + 1) / n_shards);
    return HZSDR_OK;
}

int hzsdr_mgpu_close(hzsdr_mgpu *m) {
    if (!m) return HZSDR_ERR_INVALID_ARGUMENT;
    for (int s = 0; s < (int)m->ctx.size(); s++) {
        hzsdr_ctx *c = m->ctx[s];
        if (!c) continue;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (s < (int)m->comms.size() && m->comms[s]) (void)m->rccl.CommDestroy(m->comms[s]);
        for (void *p : {m->wbuf[s], m->pbuf[s], m->obuf[s]})
            if (p) (void)hipFree(p);
        if (m->ready[s]) (void)hipEventDestroy(m->ready[s]);
        if (m->done[s]) (void)hipEventDestroy(m->done[s]);
        (void)hzsdr_close(c);
    }
    delete m;
    return HZSDR_OK;
}

int hzsdr_mgpu_open(const int *devices, int n_devices, hzsdr_mgpu **out) {
    if (!out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 16) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_mgpu *m = new hzsdr_mgpu();
    m->g = n_devices;
    m->dev.assign(devices, devices + n_devices);
    m->ctx.assign(n_devices, nullptr);
    m->ready.assign(n_devices, nullptr);
    m->done.assign(n_devices, nullptr);
    m->wbuf.assign(n_devices, nullptr);
    m->pbuf.assign(n_devices, nullptr);
    m->obuf.assign(n_devices, nullptr);
    m->wcap.assign(n_devices, 0);
    m->pcap.assign(n_devices, 0);
    m->ocap.assign(n_devices, 0);
    for (int s = 0; s < n_devices; s++) {
        int rc = hzsdr_open(devices[s], HZSDR_MEM_DEVICE, &m->ctx[s]);
        if (rc != HZSDR_OK) {
            hzsdr_mgpu_close(m);
            return rc;
        }
        // (the events belong to the shard's device, whatever hzsdr_open left current)
        if (hipSetDevice(devices[s]) != hipSuccess ||
            hipEventCreateWithFlags(&m->ready[s], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->done[s], hipEventDisableTiming) != hipSuccess) {
            hzsdr_mgpu_close(m);
            return HZSDR_ERR_HIP;
        }
    }
    // direct xGMI copies between distinct GPUs (an error here only means "already enabled" or
    // "not possible": hipMemcpyPeerAsync then stages through the host)
    for (int a = 0; a < n_devices; a++)
        for (int b = 0; b < n_devices; b++)
            if (devices[a] != devices[b]) {
                int can = 0;
                const hipError_t q = hipDeviceCanAccessPeer(&can, devices[a], devices[b]);
                bool direct = false;
                if (q == hipSuccess && can) {
                    (void)hipSetDevice(devices[a]);
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                    direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                    if (!direct)
                        m->last_error = "peer access " + std::to_string(devices[a]) + " -> " + std::to_string(devices[b]) +
                                        " could not be enabled (" + hipGetErrorString(e) + "): copies between them stage through the host";
                } else {
                    // (said, not swallowed: the exchange still works -- hipMemcpyPeerAsync stages through the host --
                    // but at PCIe speed instead of xGMI's; hzsdr_mgpu_peer_pairs / hzsdr_mgpu_last_error report it)
                    m->last_error = "GPU " + std::to_string(devices[a]) + " cannot access GPU " + std::to_string(devices[b]) +
                                    " directly" + (q != hipSuccess ? std::string(" (") + hipGetErrorString(q) + ")" : std::string()) +
                                    ": copies between them stage through the host";
                }
                (direct ? m->pairs_direct : m->pairs_staged)++;
                (void)hipGetLastError();
            }
    *out = m;
    return HZSDR_OK;
}

int hzsdr_mgpu_shards(const hzsdr_mgpu *m) { return m ? m->g : -1; }

int hzsdr_mgpu_ctx(hzsdr_mgpu *m, int shard, hzsdr_ctx **ctx) {
    if (!m || !ctx || shard < 0 || shard >= m->g) return HZSDR_ERR_INVALID_ARGUMENT;
    *ctx = m->ctx[shard];
    return HZSDR_OK;
}

const char *hzsdr_mgpu_last_error(const hzsdr_mgpu *m) { return m ? m->last_error.c_str() : ""; }

int hzsdr_mgpu_peer_pairs(const hzsdr_mgpu *m, int *direct, int *staged) {
    if (!m) return HZSDR_ERR_INVALID_ARGUMENT;
    if (direct) *direct = m->pairs_direct;
    if (staged) *staged = m->pairs_staged;
    return HZSDR_OK;
}

int hzsdr_mgpu_synchronize(hzsdr_mgpu *m) {
    if (!m) return HZSDR_ERR_INVALID_ARGUMENT;
    for (int s = 0; s < m->g; s++) HZ_TRY(hzsdr_synchronize(m->ctx[s]));
    return HZSDR_OK;
}

int hzsdr_mgpu_beamform(hzsdr_mgpu *m, void *out_c64, int dst_shard, int format, const void *const *channels,
                        const float *weights_c64, int n_channels, size_t n, int mode) {
    using namespace hz;
    if (!m) return HZSDR_ERR_INVALID_ARGUMENT;
    const int G = m->g, K = n_channels;
    if (format_size(format) == 0) return mfail(m, HZSDR_ERR_FORMAT_UNKNOWN, "mgpu beamform: unknown channel format");
    if (K < 1 || K > 16 || !channels || !weights_c64 || dst_shard < 0 || dst_shard >= G || (n && !out_c64))
        return mfail(m, HZSDR_ERR_INVALID_ARGUMENT, "mgpu beamform: 1..16 channels, weights, a destination shard");
    if (mode != HZSDR_MGPU_ORDERED && mode != HZSDR_MGPU_RCCL) return HZSDR_ERR_INVALID_ARGUMENT;
    for (int c = 0; c < K; c++)
        if (!channels[c]) return HZSDR_ERR_INVALID_ARGUMENT;
    if (n == 0) return HZSDR_OK;
    std::vector<int> lo(G), hi(G), owner(K);
    for (int s = 0; s < G; s++) {
        hzsdr_mgpu_shard_channels(K, G, s, &lo[s], &hi[s]);
        for (int c = lo[s]; c < hi[s]; c++) owner[c] = s;
    }
    const size_t fs = (size_t)format_size(format);
    (void)fs;

    if (mode == HZSDR_MGPU_RCCL) {
        if (G > 1) {
            for (int a = 0; a < G; a++)
                for (int b = a + 1; b < G; b++)
                    if (m->dev[a] == m->dev[b])
                        return mfail(m, HZSDR_ERR_INVALID_ARGUMENT, "mgpu beamform: RCCL needs one shard per GPU");
        }
        if (m->comms.empty()) {
            if (!m->rccl.load()) return mfail(m, HZSDR_ERR_HIP, "mgpu beamform: librccl could not be loaded");
            m->comms.assign(G, nullptr);
            int rc = m->rccl.CommInitAll(m->comms.data(), G, m->dev.data());
            if (rc != 0) {
                m->comms.clear();
                return mfail(m, HZSDR_ERR_HIP, std::string("ncclCommInitAll: ") +
                                                   (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "failed"));
            }
        }
        // per-shard partial sums from +0 (a shard without channels contributes zeros)
        for (int s = 0; s < G; s++) {
            HZ_TRY(grow(m, s, m->obuf, m->ocap, n * 8));
            hzsdr_ctx *c = m->ctx[s];
            HZ_TRY(enter(c));
            if (hi[s] > lo[s]) {
                HZ_TRY(hzsdr_beamform_partial(c, m->obuf[s], format, channels + lo[s], weights_c64 + 2 * lo[s],
                                              hi[s] - lo[s], n, 0));
            } else {
                HZ_HIP(c, hipMemsetAsync(m->obuf[s], 0, n * 8, c->stream));
            }
        }
        int rc = m->rccl.GroupStart();
        for (int s = 0; s < G && rc == 0; s++) {
            (void)hipSetDevice(m->dev[s]);
            rc = m->rccl.Reduce(m->obuf[s], s == dst_shard ? out_c64 : m->obuf[s], 2 * n, kNcclFloat, kNcclSum,
                                dst_shard, m->comms[s], m->ctx[s]->stream);
        }
        const int rc2 = m->rccl.GroupEnd();
        if (rc != 0 || rc2 != 0) return mfail(m, HZSDR_ERR_HIP, "mgpu beamform: ncclReduce failed");
        return HZSDR_OK;
    }

    // ---- HZSDR_MGPU_ORDERED ----
    auto slice = [&](int s, size_t *a, size_t *b) {
        *a = n * (size_t)s / (size_t)G;
        *b = n * (size_t)(s + 1) / (size_t)G;
    };
    // 1. every shard: weighted channels 0 + w_c x_c of its own channels (c64, n samples each)
    for (int s = 0; s < G; s++) {
        const int kl = hi[s] - lo[s];
        size_t a, b;
        slice(s, &a, &b);
        HZ_TRY(grow(m, s, m->wbuf, m->wcap, (size_t)(kl ? kl : 1) * n * 8));
        HZ_TRY(grow(m, s, m->pbuf, m->pcap, (size_t)(K - kl ? K - kl : 1) * (b - a ? b - a : 1) * 8));
        HZ_TRY(grow(m, s, m->obuf, m->ocap, (b - a ? b - a : 1) * 8));
        hzsdr_ctx *c = m->ctx[s];
        HZ_TRY(enter(c));
        for (int i = 0; i < kl; i++) {
            const void *one = channels[lo[s] + i];
            HZ_TRY(hzsdr_beamform(c, (char *)m->wbuf[s] + (size_t)i * n * 8, format, &one, weights_c64 + 2 * (lo[s] + i), 1, n));
        }
        HZ_HIP(c, hipEventRecord(m->ready[s], c->stream));
    }
    // 2. the exchange: shard r pushes slice s of each of its weighted channels to shard s
    //    (enqueued on r's stream, behind its weighting); piece (s, c) lands at index
    //    "c counted without s's own channels" of s's piece buffer
    for (int r = 0; r < G; r++) {
        hzsdr_ctx *c = m->ctx[r];
        HZ_TRY(enter(c));
        for (int s = 0; s < G; s++) {
            if (s == r) continue;
            // s's piece buffer may still be being summed by the PREVIOUS call: r's pushes wait
            // for s's stream (ready[s] was recorded behind that sum, and again behind s's
            // weighting of this call)
            HZ_HIP(c, hipStreamWaitEvent(c->stream, m->ready[s], 0));
            size_t a, b;
            slice(s, &a, &b);
            if (b == a) continue;
            for (int ch = lo[r]; ch < hi[r]; ch++) {
                const int slot = ch < lo[s] ? ch : ch - (hi[s] - lo[s]);  // channels of s itself are not stored
                void *dst = (char *)m->pbuf[s] + (size_t)slot * (b - a) * 8;
                const void *src = (const char *)m->wbuf[r] + ((size_t)(ch - lo[r]) * n + a) * 8;
                if (m->dev[r] == m->dev[s])
                    HZ_HIP(c, hipMemcpyAsync(dst, src, (b - a) * 8, hipMemcpyDeviceToDevice, c->stream));
                else
                    HZ_HIP(c, hipMemcpyPeerAsync(dst, m->dev[s], src, m->dev[r], (b - a) * 8, c->stream));
            }
        }
        HZ_HIP(c, hipEventRecord(m->done[r], c->stream));
    }
    // 3. shard s: wait for every sender, add the K pieces in channel order from +0
    for (int s = 0; s < G; s++) {
        size_t a, b;
        slice(s, &a, &b);
        hzsdr_ctx *c = m->ctx[s];
        HZ_TRY(enter(c));
        for (int r = 0; r < G; r++)
            if (r != s) HZ_HIP(c, hipStreamWaitEvent(c->stream, m->done[r], 0));
        if (b == a) continue;
        const void *pieces[16];
        for (int ch = 0; ch < K; ch++) {
            if (owner[ch] == s) {
                pieces[ch] = (const char *)m->wbuf[s] + ((size_t)(ch - lo[s]) * n + a) * 8;
            } else {
                const int slot = ch < lo[s] ? ch : ch - (hi[s] - lo[s]);
                pieces[ch] = (const char *)m->pbuf[s] + (size_t)slot * (b - a) * 8;
            }
        }
        void *dst = s == dst_shard ? (void *)((char *)out_c64 + a * 8) : m->obuf[s];
        HZ_TRY(hzsdr_sum(c, HZSDR_FMT_C64, dst, pieces, K, b - a));
        // 4. collect the slice on the destination shard
        if (s != dst_shard) {
            void *fin = (char *)out_c64 + a * 8;
            if (m->dev[s] == m->dev[dst_shard])
                HZ_HIP(c, hipMemcpyAsync(fin, m->obuf[s], (b - a) * 8, hipMemcpyDeviceToDevice, c->stream));
            else
                HZ_HIP(c, hipMemcpyPeerAsync(fin, m->dev[dst_shard], m->obuf[s], m->dev[s], (b - a) * 8, c->stream));
        }
        HZ_HIP(c, hipEventRecord(m->ready[s], c->stream));  // reused: "slice s delivered"
    }
    // the destination's stream is the one a caller orders later work on: it waits for every slice
    {
        hzsdr_ctx *c = m->ctx[dst_shard];
        HZ_TRY(enter(c));
        for (int s = 0; s < G; s++)
            if (s != dst_shard) HZ_HIP(c, hipStreamWaitEvent(c->stream, m->ready[s], 0));
    }
    return HZSDR_OK;
}

}  // extern "C"
