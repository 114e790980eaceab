// coala_comm.cpp -- the owner-partitioned fetch as ONE native call, split-phase and overlapped on two HIP streams:
//
//   caller's stream : route -> counts all-to-all -> [the one host read] -> ids all-to-all-v -> probe (hits copied; the
//                     requester's OWN shard written straight into `out`) -> fill round 0 -> fill round 1 ... -> un-permute 0, 1 ...
//   comm stream     :                                   wait fill 0 -> rows round 0 -> wait fill 1 -> rows round 1 ...
//
// Round k ships the k-th slice of EVERY peer's segment (xGMI is point-to-point: an all-to-all is bound by its busiest link,
// so every round keeps all links loaded), while the PCIe cold fill of the next slice runs beside it.  Own-shard rows never
// enter the exchange or a staging buffer.  One host synchronisation per minibatch (the 2G counts).
//
// Replaces, fused (paths relative to /root/reference): SSD_GNN_NVSHMEM_Cache::send_requests + read_feature
// (COALA_GNN_Modules/ssd_gnn_cache.cuh:111-174: N x 2 one-sided 8-byte puts, 3 nvshmem_barrier_all, N warp-level row puts,
// peers served one by one on G streams :132-174) and the "nccl" orchestration in
// COALA-GNN-Setup/COALA_GNN/COALA_GNN_Manager.py:143-211 (full-capacity all_to_all of ids, G(G-1) serial send/recv of rows,
// the `j == i` local copy :195-199).
//
// Two transports behind one interface: RCCL (one process per GPU, grouped ncclSend/ncclRecv drive every direct xGMI link at
// once) and an in-process one (G ranks = G host threads of one process, device-to-device copies ordered by events) used by
// single-process drivers and by the parity tests, which run the very same orchestration with G logical ranks on one GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/coala_hip.h"
#include "coala_internal.h"

#define fail coala_fail_
#define HIPCHK COALA_HIPCHK
#define NCCLCHK(expr)                                                                                                   \
    do {                                                                                                                \
        ncclResult_t r_ = (expr);                                                                                       \
        if (r_ != ncclSuccess) return fail(COALA_ECOMM, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

namespace {

constexpr int kMaxRounds = 8;

struct Transport {
    int rank = 0, nranks = 1;
    // coala_comm_set_self_loopback: a rank's OWN segment goes through the transport's send/recv path like a peer's (diagnostics)
    bool self_loopback = false;
    virtual ~Transport() {}
    // recv[p] = element [me] of rank p's send (one int64 per peer, self included)
    virtual int all_to_all_i64(const int64_t* send_dev, int64_t* recv_dev, hipStream_t st) = 0;
    // counts / displacements in elements of elem_bytes bytes; the self segment moves only when include_self
    virtual int all_to_all_v(const void* send, const size_t* scnt, const size_t* sdis, void* recv, const size_t* rcnt,
                             const size_t* rdis, size_t elem_bytes, bool include_self, hipStream_t st) = 0;
    // make the peers fail instead of hang after a local error between collectives; the transport is unusable afterwards
    virtual void abort() = 0;
    // ranks as the transport itself counts them (not the number it was told at creation); negative on failure
    virtual int size() = 0;
};

// ------------------------------------------------------------------------------------------------------------ RCCL
struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    ~RcclTransport() override {
        if (comm) (void)ncclCommDestroy(comm);
    }
    int all_to_all_i64(const int64_t* send_dev, int64_t* recv_dev, hipStream_t st) override {
        NCCLCHK(ncclAllToAll(send_dev, recv_dev, 1, ncclInt64, comm, st));
        return COALA_OK;
    }
    int all_to_all_v(const void* send, const size_t* scnt, const size_t* sdis, void* recv, const size_t* rcnt, const size_t* rdis,
                     size_t elem_bytes, bool include_self, hipStream_t st) override {
        const ncclDataType_t dt = (elem_bytes % 8 == 0) ? ncclInt64 : ncclFloat32;
        const size_t per = (elem_bytes % 8 == 0) ? elem_bytes / 8 : elem_bytes / 4;
        const bool self_by_rccl = include_self && self_loopback; // ncclSend + ncclRecv to oneself inside the group call: legal, and a local copy inside RCCL
        if (include_self && scnt[rank] && !self_by_rccl)
            HIPCHK(hipMemcpyAsync((char*)recv + rdis[rank] * elem_bytes, (const char*)send + sdis[rank] * elem_bytes, scnt[rank] * elem_bytes,
                                  hipMemcpyDeviceToDevice, st));
        NCCLCHK(ncclGroupStart());
        for (int p = 0; p < nranks; ++p) {
            if (p == rank && !self_by_rccl) continue;
            if (scnt[p]) NCCLCHK(ncclSend((const char*)send + sdis[p] * elem_bytes, scnt[p] * per, dt, p, comm, st));
            if (rcnt[p]) NCCLCHK(ncclRecv((char*)recv + rdis[p] * elem_bytes, rcnt[p] * per, dt, p, comm, st));
        }
        NCCLCHK(ncclGroupEnd());
        return COALA_OK;
    }
    void abort() override {
        if (comm) (void)ncclCommAbort(comm);
        comm = nullptr;
    }
    int size() override {
        if (!comm) return fail(COALA_ECOMM, "the RCCL communicator was aborted");
        int n = -1;
        NCCLCHK(ncclCommCount(comm, &n));
        return n;
    }
};

} // namespace

// ------------------------------------------------------------------------------------------------------------ in-process
struct coala_comm_group {
    int nranks = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t phase = 0;
    bool aborted = false;
    int attached = 0;
    int timeout_s = 120; // COALA_INPROC_TIMEOUT_S
    struct Slot {
        const void* send = nullptr;
        const size_t* scnt = nullptr;
        const size_t* sdis = nullptr;
        hipEvent_t ready = nullptr, done = nullptr;
        int device = -1; // the device this rank's buffers live on
    };
    std::vector<Slot> slot;
    // peer_state[a * 64 + b]: may a kernel on device a read device b's memory?  0 unknown, 1 yes (same device, or peer access
    // enabled and verified), 2 no (the runtime's copy engine moves the segment instead)
    std::vector<unsigned char> peer_state = std::vector<unsigned char>(64 * 64, 0);

    int barrier() { // -> 0, or -1 when the group was aborted
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return -1;
        const uint64_t my = phase;
        if (++arrived == nranks) {
            arrived = 0;
            ++phase;
            cv.notify_all();
            return 0;
        }
        // a peer that never arrives (it failed before its collective call, or its thread died) must not hang the others forever
        if (!cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return phase != my || aborted; })) {
            aborted = true;
            cv.notify_all();
        }
        return aborted ? -1 : 0;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
};

namespace {

// Device-to-device segment copy of the in-process transport.  hipMemcpyAsync between two buffers of ONE device goes through a DMA
// engine at ~26 GB/s on this platform (8 logical ranks x 0.9 GB of rows per step of the configs[4] probe: 270 ms); a plain kernel
// moves the same bytes at HBM speed.  16-B accesses when both ends are 16-B aligned, bytes otherwise.
__global__ __launch_bounds__(256) void inproc_copy_kernel(char* __restrict__ dst, const char* __restrict__ src, size_t bytes) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const size_t n16 = bytes / 16;
        for (size_t i = tid; i < n16; i += nthreads) reinterpret_cast<u32x4*>(dst)[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + i);
        for (size_t i = n16 * 16 + tid; i < bytes; i += nthreads) dst[i] = src[i];
    } else {
        for (size_t i = tid; i < bytes; i += nthreads) dst[i] = src[i];
    }
}

// kernel_ok: the source is on this device, or on a peer whose memory this device has mapped (peer_readable below).  Otherwise --
// and for small segments (ids, counts) -- the runtime's copy, which needs no peer mapping.
int inproc_copy(void* dst, const void* src, size_t bytes, hipStream_t st, bool kernel_ok) {
    if (bytes < (64u << 10) || !kernel_ok) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
        return COALA_OK;
    }
    const size_t blocks = (bytes / 16 + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(inproc_copy_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, (char*)dst, (const char*)src, bytes);
    HIPCHK(hipGetLastError());
    return COALA_OK;
}

// May a kernel running on device `mine` read memory of device `theirs`?  Decided once per ordered pair (under the group's mutex):
// hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess; a pair without peer access is served by hipMemcpyAsync.
bool peer_readable(coala_comm_group* g, int mine, int theirs) {
    if (mine == theirs) return true;
    if (mine < 0 || theirs < 0 || mine >= 64 || theirs >= 64) return false;
    std::lock_guard<std::mutex> lk(g->m);
    unsigned char& st = g->peer_state[(size_t)mine * 64 + theirs];
    if (st == 0) {
        int can = 0;
        st = 2;
        if (hipDeviceCanAccessPeer(&can, mine, theirs) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(theirs, 0); // (the calling thread's current device is `mine`)
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) st = 1;
        }
        (void)hipGetLastError();
    }
    return st == 1;
}

struct InprocTransport : Transport {
    coala_comm_group* g = nullptr;
    int device = -1;
    int exchange(const void* send, const size_t* scnt, const size_t* sdis, void* recv, const size_t* rcnt, const size_t* rdis,
                 size_t elem_bytes, bool include_self, hipStream_t st) {
        auto& me = g->slot[rank];
        me.send = send;
        me.scnt = scnt;
        me.sdis = sdis;
        HIPCHK(hipEventRecord(me.ready, st)); // my send buffer is complete at this point of my stream
        if (g->barrier()) return fail(COALA_ECOMM, "in-process group aborted (a peer failed or did not arrive in time)");
        for (int p = 0; p < nranks; ++p) {
            if (p == rank && !include_self) continue;
            const auto& peer = g->slot[p];
            if (peer.scnt[rank] != rcnt[p]) {
                g->abort();
                return fail(COALA_ECOMM, "in-process exchange: rank %d sends %zu elements to rank %d, which expects %zu", p, peer.scnt[rank], rank, rcnt[p]);
            }
            if (!rcnt[p]) continue;
            if (p != rank) HIPCHK(hipStreamWaitEvent(st, peer.ready, 0));
            if (int rc = inproc_copy((char*)recv + rdis[p] * elem_bytes, (const char*)peer.send + peer.sdis[rank] * elem_bytes, rcnt[p] * elem_bytes, st,
                                     peer_readable(g, device, peer.device)))
                return rc;
        }
        HIPCHK(hipEventRecord(me.done, st)); // I have pulled what I need from everybody
        if (g->barrier()) return fail(COALA_ECOMM, "in-process group aborted (a peer failed or did not arrive in time)");
        for (int p = 0; p < nranks; ++p) // nobody reuses its send buffer before every peer has pulled from it
            if (p != rank) HIPCHK(hipStreamWaitEvent(st, g->slot[p].done, 0));
        // no third barrier: a peer that races ahead rewrites its slot only before the NEXT exchange's first barrier, which this
        // rank has to reach too; send / counts are read above, before the second barrier
        return COALA_OK;
    }
    int all_to_all_i64(const int64_t* send_dev, int64_t* recv_dev, hipStream_t st) override {
        std::vector<size_t> one((size_t)nranks, 1), dis((size_t)nranks);
        for (int p = 0; p < nranks; ++p) dis[p] = (size_t)p;
        return exchange(send_dev, one.data(), dis.data(), recv_dev, one.data(), dis.data(), sizeof(int64_t), true, st);
    }
    int all_to_all_v(const void* send, const size_t* scnt, const size_t* sdis, void* recv, const size_t* rcnt, const size_t* rdis,
                     size_t elem_bytes, bool include_self, hipStream_t st) override {
        return exchange(send, scnt, sdis, recv, rcnt, rdis, elem_bytes, include_self, st);
    }
    void abort() override { g->abort(); }
    int size() override { return g->nranks; }
};

int grow(void** p, uint64_t* cap, uint64_t need, size_t elem, hipStream_t st) {
    if (need <= *cap) return COALA_OK;
    HIPCHK(hipStreamSynchronize(st));
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    uint64_t c = 4096;
    while (c < need) c *= 2;
    if (c - need > (c >> 2) && need > (1ull << 20)) c = need + (need >> 3); // large buffers: 12 % head room instead of up to 100 %
    if (hipMalloc(p, c * elem) != hipSuccess) {
        (void)hipGetLastError();
        return fail(COALA_ENOMEM, "hipMalloc(%llu) failed for the exchange workspace", (unsigned long long)(c * elem));
    }
    *cap = c;
    return COALA_OK;
}

} // namespace

struct coala_comm {
    Transport* tr = nullptr;
    int rank = 0, nranks = 1, device = 0;
    int rounds = 2;                 // row-exchange rounds per fetch (COALA_EXCHANGE_ROUNDS; coala_comm_set_rounds)
    hipStream_t cs = nullptr;       // the communication stream of this rank
    hipEvent_t ev_fill[kMaxRounds] = {}, ev_x[kMaxRounds] = {};
    // workspace (device), grown on demand, persistent across steps
    int64_t *node = nullptr, *map = nullptr;
    uint64_t node_cap = 0, map_cap = 0;
    int64_t* recv_ids = nullptr;
    uint64_t recv_cap = 0;
    float *rows_send = nullptr, *rows_recv = nullptr;
    uint64_t rows_send_cap = 0, rows_recv_cap = 0; // in floats
    int64_t* counts_dev = nullptr;                 // [3G+1]: send counts, recv counts, offsets
    int64_t* counts_host = nullptr;                // pinned [2G]
    // last step, for tests / diagnostics
    std::vector<int64_t> last_send, last_recv;
    // profiling of the row exchange (coala_comm_profile)
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_live;
    std::vector<hipEvent_t> prof_pool;
    coala_comm_profile_t prof{};
    bool broken = false;
    bool loopback = false;          // coala_comm_set_self_loopback
    // coala_comm_fetch_events: begin / end events of a bucketed fetch without packets of their own on the caller's stream; a ring of triples
    // (begin on the probe's launch, end on the last fill launch of the caller's stream, end behind the last row round on the communicator's stream)
    int fetch_events = 0;           // 0 off, 1 end events only, 2 begin event too (coala_comm_fetch_events)
    bool plain_events = false;      // development builds, COALA_COMM_PLAIN_EVENTS=1: every event recorded behind its kernel, every round waited for (the round-3 form; tools/dist_packets_probe.py)
    static constexpr int kFetchRing = 2048;
    std::vector<hipEvent_t> fev;    // [3 * kFetchRing], created on first use
    uint64_t fev_calls = 0;
    hipEvent_t last_ev[3] = {nullptr, nullptr, nullptr};
    // count exchanges issued ahead of their fetch (coala_comm_counts_begin): ring of device [2G] + pinned [2G] + event
    static constexpr int kCountsRing = COALA_COUNTS_RING;
    int64_t* ahead_dev = nullptr;   // [kCountsRing][2G]
    int64_t* ahead_host = nullptr;  // pinned [kCountsRing][2G]
    hipEvent_t ahead_ev[kCountsRing] = {};
    uint64_t ahead_calls = 0;
    // The workspaces (recv_ids, rows_send, node/map) are ordered by the stream the fetches are enqueued on.  A caller that moves to
    // another stream keeps that order: the new stream first waits (no host wait) for what the previous fetch enqueued on the old
    // one -- the same rule as the cache handle's follow_stream.
    hipStream_t order_stream = nullptr;
    bool order_set = false;
    hipEvent_t order_ev = nullptr;
};

namespace {

int finish_create(coala_comm* c) {
    c->rank = c->tr->rank;
    c->nranks = c->tr->nranks;
    if (const char* e = getenv("COALA_EXCHANGE_ROUNDS")) {
        const int r = atoi(e);
        if (r >= 1 && r <= kMaxRounds) c->rounds = r;
    }
#ifdef COALA_DEV_KNOBS
    if (const char* e = getenv("COALA_COMM_PLAIN_EVENTS")) c->plain_events = atoi(e) != 0;
#endif
    // The communication stream gets the highest stream priority: HIP keeps priority levels on separate hardware queues, so the row
    // exchange can never be queued behind the cold fill it is meant to run beside (with equal priorities the streams of a process
    // share GPU_MAX_HW_QUEUES = 4 queues in creation order), and its few workgroups are scheduled ahead of the fill's.
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&c->cs, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
        (void)hipGetLastError();
        c->cs = nullptr;
        if (hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking) != hipSuccess) return fail(COALA_EHIP, "hipStreamCreate failed");
    }
    for (int k = 0; k < kMaxRounds; ++k)
        if (hipEventCreateWithFlags(&c->ev_fill[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_x[k], hipEventDisableTiming) != hipSuccess)
            return fail(COALA_EHIP, "hipEventCreate failed");
    if (hipMalloc((void**)&c->counts_dev, (3 * (size_t)c->nranks + 1) * sizeof(int64_t)) != hipSuccess ||
        hipHostMalloc((void**)&c->counts_host, 2 * (size_t)c->nranks * sizeof(int64_t)) != hipSuccess)
        return fail(COALA_ENOMEM, "communicator workspace allocation failed");
    c->last_send.assign(c->nranks, 0);
    c->last_recv.assign(c->nranks, 0);
    const size_t ring_words = (size_t)coala_comm::kCountsRing * 2 * (size_t)c->nranks;
    if (hipMalloc((void**)&c->ahead_dev, ring_words * sizeof(int64_t)) != hipSuccess ||
        hipHostMalloc((void**)&c->ahead_host, ring_words * sizeof(int64_t)) != hipSuccess)
        return fail(COALA_ENOMEM, "communicator workspace allocation failed");
    for (int k = 0; k < coala_comm::kCountsRing; ++k)
        if (hipEventCreateWithFlags(&c->ahead_ev[k], hipEventDisableTiming) != hipSuccess) return fail(COALA_EHIP, "hipEventCreate failed");
    return COALA_OK;
}

int follow_stream(coala_comm* c, hipStream_t s) {
    if (c->order_set && c->order_stream != s) {
        if (!c->order_ev) HIPCHK(hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming));
        // (a stream the caller has destroyed in the meantime has nothing left to wait for)
        if (hipEventRecord(c->order_ev, c->order_stream) == hipSuccess) HIPCHK(hipStreamWaitEvent(s, c->order_ev, 0));
        else (void)hipGetLastError();
    }
    c->order_stream = s;
    c->order_set = true;
    return COALA_OK;
}

void drain_profile(coala_comm* c) {
    for (auto& p : c->prof_live) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) c->prof.rows_ms += ms;
        c->prof_pool.push_back(p.first);
        c->prof_pool.push_back(p.second);
    }
    c->prof_live.clear();
}

hipEvent_t take_timing_event(coala_comm* c) {
    if (!c->prof_pool.empty()) {
        hipEvent_t e = c->prof_pool.back();
        c->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

} // namespace

extern "C" {

int coala_comm_unique_id(void* out_id, size_t cap) {
    if (!out_id || cap < NCCL_UNIQUE_ID_BYTES) return fail(COALA_EINVAL, "id buffer must hold %d bytes", NCCL_UNIQUE_ID_BYTES);
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(out_id, &id, NCCL_UNIQUE_ID_BYTES);
    return COALA_OK;
}

int coala_comm_create(const void* id_bytes, int rank, int nranks, int device, coala_comm_t** out) {
    if (!id_bytes || !out || nranks < 1 || rank < 0 || rank >= nranks || nranks > 64) return fail(COALA_EINVAL, "bad communicator arguments");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    coala_comm* c = new (std::nothrow) coala_comm();
    RcclTransport* t = new (std::nothrow) RcclTransport();
    if (!c || !t) {
        delete c;
        delete t;
        return fail(COALA_ENOMEM, "out of host memory");
    }
    c->tr = t;
    c->device = device;
    t->rank = rank;
    t->nranks = nranks;
    ncclUniqueId id;
    memcpy(&id, id_bytes, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&t->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        coala_comm_destroy(c);
        return fail(COALA_ECOMM, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    {   // what RCCL itself thinks this communicator is: it must agree with what the caller asked for
        int cnt = -1, urank = -1, dev = -1;
        if (ncclCommCount(t->comm, &cnt) != ncclSuccess || ncclCommUserRank(t->comm, &urank) != ncclSuccess ||
            ncclCommCuDevice(t->comm, &dev) != ncclSuccess || cnt != nranks || urank != rank || dev != device) {
            coala_comm_destroy(c);
            return fail(COALA_ECOMM, "RCCL reports %d ranks / rank %d / device %d for a communicator created as %d ranks / rank %d / device %d",
                        cnt, urank, dev, nranks, rank, device);
        }
    }
    if (int rc = finish_create(c)) {
        coala_comm_destroy(c);
        return rc;
    }
    *out = c;
    return COALA_OK;
}

int coala_comm_group_create(int nranks, coala_comm_group_t** out) {
    if (!out || nranks < 1 || nranks > 64) return fail(COALA_EINVAL, "bad group size");
    coala_comm_group* g = new (std::nothrow) coala_comm_group();
    if (!g) return fail(COALA_ENOMEM, "out of host memory");
    g->nranks = nranks;
    g->slot.resize((size_t)nranks);
    if (const char* e = getenv("COALA_INPROC_TIMEOUT_S")) {
        const int t = atoi(e);
        if (t > 0) g->timeout_s = t;
    }
    *out = g;
    return COALA_OK;
}

int coala_comm_group_destroy(coala_comm_group_t* g) {
    if (!g) return COALA_OK;
    if (g->attached) return fail(COALA_EINVAL, "%d communicators of this group are still alive", g->attached);
    delete g;
    return COALA_OK;
}

int coala_comm_create_inproc(coala_comm_group_t* g, int rank, int device, coala_comm_t** out) {
    if (!g || !out || rank < 0 || rank >= g->nranks) return fail(COALA_EINVAL, "bad in-process communicator arguments");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    coala_comm* c = new (std::nothrow) coala_comm();
    InprocTransport* t = new (std::nothrow) InprocTransport();
    if (!c || !t) {
        delete c;
        delete t;
        return fail(COALA_ENOMEM, "out of host memory");
    }
    c->tr = t;
    c->device = device;
    t->g = g;
    t->rank = rank;
    t->nranks = g->nranks;
    t->device = device;
    auto& sl = g->slot[rank];
    sl.device = device;
    if (sl.ready || hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) {
        delete c;
        delete t;
        return fail(COALA_EINVAL, "rank %d of the group is taken, or hipEventCreate failed", rank);
    }
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->attached++;
    }
    if (int rc = finish_create(c)) {
        coala_comm_destroy(c);
        return rc;
    }
    *out = c;
    return COALA_OK;
}

int coala_comm_destroy(coala_comm_t* c) {
    if (!c) return COALA_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    drain_profile(c);
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (auto* t = dynamic_cast<InprocTransport*>(c->tr)) {
        auto& sl = t->g->slot[t->rank];
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.done) (void)hipEventDestroy(sl.done);
        sl.ready = sl.done = nullptr;
        std::lock_guard<std::mutex> lk(t->g->m);
        t->g->attached--;
    }
    delete c->tr;
    for (int k = 0; k < kMaxRounds; ++k) {
        if (c->ev_fill[k]) (void)hipEventDestroy(c->ev_fill[k]);
        if (c->ev_x[k]) (void)hipEventDestroy(c->ev_x[k]);
    }
    if (c->cs) (void)hipStreamDestroy(c->cs);
    if (c->order_ev) (void)hipEventDestroy(c->order_ev);
    for (auto e : c->fev)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < coala_comm::kCountsRing; ++k)
        if (c->ahead_ev[k]) (void)hipEventDestroy(c->ahead_ev[k]);
    if (c->ahead_host) (void)hipHostFree(c->ahead_host);
    void* dev[] = {c->node, c->map, c->recv_ids, c->rows_send, c->rows_recv, c->counts_dev, c->ahead_dev};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    if (c->counts_host) (void)hipHostFree(c->counts_host);
    delete c;
    return COALA_OK;
}

int coala_comm_size(const coala_comm_t* c) { return (c && c->tr) ? c->tr->size() : 0; }

int coala_comm_set_rounds(coala_comm_t* c, int rounds) {
    if (!c || rounds < 1 || rounds > kMaxRounds) return fail(COALA_EINVAL, "rounds must be 1..%d", kMaxRounds);
    c->rounds = rounds;
    return COALA_OK;
}

int coala_comm_get_rounds(const coala_comm_t* c) { return c ? c->rounds : 0; }

int coala_comm_fetch_events(coala_comm_t* c, int enable) {
    if (!c) return fail(COALA_EINVAL, "null communicator");
    if (enable < 0 || enable > 2) return fail(COALA_EINVAL, "enable must be 0, 1 or 2");
    c->fetch_events = enable;
    if (!c->fetch_events) c->last_ev[0] = c->last_ev[1] = c->last_ev[2] = nullptr;
    return COALA_OK;
}

int coala_comm_last_fetch_events(const coala_comm_t* c, void** begin_ev, void** end_ev_stream, void** end_ev_comm) {
    if (!c) return fail(COALA_EINVAL, "null communicator");
    if (begin_ev) *begin_ev = (void*)c->last_ev[0];
    if (end_ev_stream) *end_ev_stream = (void*)c->last_ev[1];
    if (end_ev_comm) *end_ev_comm = (void*)c->last_ev[2];
    return COALA_OK;
}

int coala_comm_set_self_loopback(coala_comm_t* c, int on) {
    if (!c || !c->tr) return fail(COALA_EINVAL, "null communicator");
    c->loopback = on != 0;
    c->tr->self_loopback = c->loopback;
    return COALA_OK;
}

int coala_comm_last_counts(const coala_comm_t* c, int64_t* send, int64_t* recv) {
    if (!c) return fail(COALA_EINVAL, "null communicator");
    for (int p = 0; p < c->nranks; ++p) {
        if (send) send[p] = c->last_send[p];
        if (recv) recv[p] = c->last_recv[p];
    }
    return COALA_OK;
}

int coala_comm_profile(coala_comm_t* c, int enable, coala_comm_profile_t* out, int reset) {
    if (!c) return fail(COALA_EINVAL, "null communicator");
    HIPCHK(hipSetDevice(c->device));
    drain_profile(c);
    if (out) *out = c->prof;
    if (reset) c->prof = coala_comm_profile_t{};
    if (enable >= 0) c->profile = enable != 0;
    return COALA_OK;
}

static int fetch_impl(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n, const int64_t* bucket_counts_dev,
                      const int64_t* counts_host, void* stream);

int coala_cache_fetch_distributed(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n, void* stream) {
    return fetch_impl(h, c, out, idx, n, nullptr, nullptr, stream);
}

int coala_cache_fetch_distributed_bucketed(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n,
                                           const int64_t* counts_dev, void* stream) {
    if (!counts_dev) return fail(COALA_EINVAL, "null bucket counts");
    return fetch_impl(h, c, out, idx, n, counts_dev, nullptr, stream);
}

int coala_comm_counts_begin(coala_comm_t* c, const int64_t* counts_dev, void* stream, int64_t* ticket_out) {
    if (!c || !counts_dev || !ticket_out) return fail(COALA_EINVAL, "null argument");
    if (c->broken) return fail(COALA_ECOMM, "this communicator failed in an earlier fetch and was aborted: destroy it");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int G = c->nranks;
    const int slot = (int)(c->ahead_calls % coala_comm::kCountsRing);
    if (c->ahead_calls >= coala_comm::kCountsRing) HIPCHK(hipEventSynchronize(c->ahead_ev[slot])); // the slot's previous exchange has landed
    int64_t* dev = c->ahead_dev + (size_t)slot * 2 * G;
    int64_t* host = c->ahead_host + (size_t)slot * 2 * G;
    if (hipMemcpyAsync(dev, counts_dev, (size_t)G * sizeof(int64_t), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        c->broken = true; // the peers are about to enter the collective this rank will not join
        c->tr->abort();
        return fail(COALA_EHIP, "count exchange issued ahead: staging the counts failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (int rc = c->tr->all_to_all_i64(dev, dev + G, st)) { // a collective: a failure here strands the peers -> abort
        c->broken = true;
        c->tr->abort();
        return rc;
    }
    // behind the collective a local failure must not leave this rank's ticket counter behind its peers': same path as fetch_impl
    if (hipMemcpyAsync(host, dev, 2 * (size_t)G * sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipEventRecord(c->ahead_ev[slot], st) != hipSuccess) {
        c->broken = true;
        c->tr->abort();
        return fail(COALA_EHIP, "count exchange issued ahead: reading the counts back failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *ticket_out = (int64_t)c->ahead_calls++;
    return COALA_OK;
}

int coala_cache_fetch_distributed_bucketed_ahead(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n, int64_t ticket,
                                                 void* stream) {
    if (!c) return fail(COALA_EINVAL, "null handle");
    if (ticket < 0 || (uint64_t)ticket >= c->ahead_calls || c->ahead_calls - (uint64_t)ticket > (uint64_t)coala_comm::kCountsRing)
        return fail(COALA_EINVAL, "ticket %lld is not one of the last %d count exchanges", (long long)ticket, coala_comm::kCountsRing);
    HIPCHK(hipSetDevice(c->device));
    const int slot = (int)((uint64_t)ticket % coala_comm::kCountsRing);
    HIPCHK(hipEventSynchronize(c->ahead_ev[slot])); // issued a step or two ago: normally complete long since
    return fetch_impl(h, c, out, idx, n, c->ahead_dev + (size_t)slot * 2 * c->nranks, c->ahead_host + (size_t)slot * 2 * c->nranks, stream);
}

// bucket_counts_dev == nullptr: route here.  Otherwise idx is already bucketed by owner: node = idx, the map is the identity,
// the rows of every peer are received straight into `out` and nothing is un-permuted.
// counts_host != nullptr: the 2G counts were exchanged ahead (coala_comm_counts_begin) and are on the host already -- no count
// exchange, no host synchronisation in this call.
static int fetch_impl(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n, const int64_t* bucket_counts_dev,
                      const int64_t* counts_host, void* stream) {
    if (!h || !c) return fail(COALA_EINVAL, "null handle");
    c->last_ev[0] = c->last_ev[1] = c->last_ev[2] = nullptr;
    const bool bucketed = bucket_counts_dev != nullptr;
    if (c->broken) return fail(COALA_ECOMM, "this communicator failed in an earlier fetch and was aborted: destroy it");
    // every check that can fail locally comes BEFORE the first collective: a rank that returns between collectives strands its peers
    if (n < 0 || n > 0x7FFFFFFFll || (n > 0 && (!out || !idx))) return fail(COALA_EINVAL, "bad batch");
    coala_cache_geometry_t geo;
    int rc = coala_cache_geometry(h, &geo);
    if (rc) return rc;
    const int G = c->nranks, me = c->rank;
    const int64_t dim = coala_cache_row_dim(h);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = follow_stream(c, st))) return rc;
    const uint64_t nb = (uint64_t)(n > 0 ? n : 1);
    if (!bucketed) {
        if ((rc = grow((void**)&c->node, &c->node_cap, nb, sizeof(int64_t), st))) return rc;
        if ((rc = grow((void**)&c->map, &c->map_cap, nb, sizeof(int64_t), st))) return rc;
        if ((rc = grow((void**)&c->rows_recv, &c->rows_recv_cap, nb * (uint64_t)dim, sizeof(float), st))) return rc;
    }
    // the usual step receives about as many ids as it sends: pre-size the owner-side buffers too, so that the grow after the
    // counts exchange (the one allocation that sits between collectives) only happens for skewed batches
    if ((rc = grow((void**)&c->recv_ids, &c->recv_cap, nb + (nb >> 2), sizeof(int64_t), st))) return rc;
    if ((rc = grow((void**)&c->rows_send, &c->rows_send_cap, (nb + (nb >> 2)) * (uint64_t)dim, sizeof(float), st))) return rc;
    int64_t* send_cnt = c->counts_dev;
    int64_t* recv_cnt = c->counts_dev + G;
    int64_t* offsets = c->counts_dev + 2 * G; // [G+1]
    // 1. bucket by owner (stable), packed layout -- unless the sampler already delivered the ids that way
    const int64_t* node = c->node;
    float* rows_recv = c->rows_recv;
    if (bucketed) {
        if (!counts_host) HIPCHK(hipMemcpyAsync(send_cnt, bucket_counts_dev, (size_t)G * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
        node = idx;
        rows_recv = out; // bucket order IS the caller's order
    } else if ((rc = coala_cache_route(h, idx, n, G, 0, c->node, c->map, send_cnt, offsets, st))) {
        return rc;
    }

    // from here on a local failure aborts the transport so that the peers error out instead of waiting for this rank
    auto broke = [&](int code) {
        c->broken = true;
        c->tr->abort();
        return code;
    };
    if (!counts_host) {
        // 2. counts: every rank tells every owner how many ids follow
        if ((rc = c->tr->all_to_all_i64(send_cnt, recv_cnt, st))) return broke(rc);
        // 3. the one host read of the step
        if (hipMemcpyAsync(c->counts_host, c->counts_dev, 2 * (size_t)G * sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return broke(fail(COALA_EHIP, "reading the exchange counts failed: %s", hipGetErrorString(hipGetLastError())));
        counts_host = c->counts_host;
    }
    std::vector<size_t> scnt(G), sdis(G), rcnt(G), rdis(G);
    size_t total_recv = 0, acc = 0;
    for (int p = 0; p < G; ++p) {
        scnt[p] = (size_t)counts_host[p];
        rcnt[p] = (size_t)counts_host[G + p];
        sdis[p] = acc;
        rdis[p] = total_recv;
        acc += scnt[p];
        total_recv += rcnt[p];
        c->last_send[p] = (int64_t)scnt[p];
        c->last_recv[p] = (int64_t)rcnt[p];
    }
    if ((int64_t)acc != n) return broke(fail(COALA_ECOMM, "%s %zu ids for a batch of %lld", bucketed ? "the bucket counts sum to" : "route produced", acc, (long long)n));
    if (total_recv > 0x7FFFFFFFull) return broke(fail(COALA_ECOMM, "%zu ids routed to one owner in one step", total_recv));
    const uint64_t tr = total_recv ? total_recv : 1;
    if ((rc = grow((void**)&c->recv_ids, &c->recv_cap, tr, sizeof(int64_t), st))) return broke(rc);
    if ((rc = grow((void**)&c->rows_send, &c->rows_send_cap, tr * (uint64_t)dim, sizeof(float), st))) return broke(rc);
    // 4. ids to their owners (exact sizes); the own bucket is a device copy
    if ((rc = c->tr->all_to_all_v(node, scnt.data(), sdis.data(), c->recv_ids, rcnt.data(), rdis.data(), sizeof(int64_t), true, st)))
        return broke(rc);
    // 5. owner side: ONE batch = the concatenation in source-rank order (DESIGN.md "Determinism contract").  Probe all of it;
    //    the own segment is delivered straight to out[map[..]] -- it never sees rows_send, the exchange or rows_recv.
    //    (self-loopback, diagnostics: no own-shard bypass -- the own segment is served, shipped and un-permuted like a peer's, so that a
    //    communicator of ONE rank drives every line of the row exchange below through the transport)
    const bool loop = c->loopback;
    coala_row_redirect_t rd;
    rd.begin = (int64_t)rdis[me];
    rd.end = loop ? rd.begin : (int64_t)(rdis[me] + rcnt[me]);
    rd.out = bucketed ? out + sdis[me] * (size_t)dim : out; // bucketed: the own bucket sits at its offset, in order
    rd.row_map = bucketed ? nullptr : c->map + sdis[me];
    // Events of this fetch.  Every hipEventRecord on the caller's stream is one more barrier packet between the kernels of a saturated
    // stream (6-12 us each: DESIGN.md section 6), so whatever can ride ON a launch does: the hand-over events of the fill rounds always, and
    // -- opt-in, bucketed fetches: coala_comm_fetch_events -- the begin / end events a caller needs for timing and for its consumer's stream.
    const bool ev_mode = c->fetch_events && bucketed && n > 0 && !c->plain_events;
    hipEvent_t ev_begin = nullptr, ev_end_st = nullptr, ev_end_cs = nullptr;
    if (ev_mode) {
        if (c->fev.empty()) c->fev.assign(3 * (size_t)coala_comm::kFetchRing, nullptr);
        const size_t slot = (size_t)(c->fev_calls % coala_comm::kFetchRing);
        for (size_t k = 3 * slot; k < 3 * slot + 3; ++k)
            if (!c->fev[k] && hipEventCreate(&c->fev[k]) != hipSuccess) return broke(fail(COALA_EHIP, "hipEventCreate failed"));
        ev_begin = c->fetch_events >= 2 ? c->fev[3 * slot] : nullptr;   // (an event on the probe's launch costs that launch ~5 us: only when asked for)
        ev_end_st = c->fev[3 * slot + 1];
        ev_end_cs = c->fev[3 * slot + 2];
    }
    hipEvent_t rode = nullptr;
    if ((rc = coala_serve_probe_redirect_ev_(h, c->rows_send, c->recv_ids, (int64_t)total_recv, &rd, st, c->plain_events ? nullptr : ev_begin, &rode))) return broke(rc);
    if (ev_begin && rode != ev_begin && hipEventRecord(ev_begin, st) != hipSuccess) return broke(fail(COALA_EHIP, "hipEventRecord failed"));
    // 6. rounds: fill slice k of every peer's segment on the caller's stream, ship it on the comm stream while slice k+1 fills
    const int K = (G == 1 && !loop) ? 1 : c->rounds;
    const bool exchange_rows = G > 1 || loop;
    const size_t row_bytes = (size_t)dim * sizeof(float);
    std::vector<int64_t> fb(G), fe(G);
    std::vector<size_t> xs_cnt(G), xs_dis(G), xr_cnt(G), xr_dis(G);
    std::vector<int64_t> sb((size_t)G * K), se((size_t)G * K); // requester-side ranges of rows_recv, per round
    hipEvent_t fill_evs[kMaxRounds] = {};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (c->profile && exchange_rows) {
        if (c->prof_live.size() >= 4096) drain_profile(c);
        t0 = take_timing_event(c);
        t1 = take_timing_event(c);
    }
    for (int k = 0; k < K && total_recv > 0; ++k) {
        int nr = 0;
        for (int p = 0; p < G; ++p) {
            int64_t b, e;
            if (p == me && !loop) { // own segment: nobody waits for it on a link -> last round
                if (k != K - 1) continue;
                b = (int64_t)rdis[p];
                e = (int64_t)(rdis[p] + rcnt[p]);
            } else {
                b = (int64_t)(rdis[p] + rcnt[p] * (size_t)k / (size_t)K);
                e = (int64_t)(rdis[p] + rcnt[p] * (size_t)(k + 1) / (size_t)K);
            }
            if (e > b) { fb[nr] = b; fe[nr] = e; ++nr; }
        }
        // the event the row round k waits for rides on this round's fill launch (a profiling cache handle lends its own event of that launch);
        // the last round's is the fetch's end event on this stream
        const bool last = k == K - 1;
        hipEvent_t fill_ev = (last && ev_end_st) ? ev_end_st : (exchange_rows ? c->ev_fill[k] : nullptr);
        if ((rc = coala_serve_fill_ranges_ev_(h, c->rows_send, c->recv_ids, (int64_t)total_recv, fb.data(), fe.data(), nr, st, c->plain_events ? nullptr : fill_ev, &rode))) return broke(rc);
        if (c->plain_events) rode = nullptr;
        const bool must_be_mine = fill_ev && fill_ev == ev_end_st;    // handed out to the caller: a borrowed event will not do
        if (fill_ev && (rode == nullptr || (must_be_mine && rode != fill_ev)) && hipEventRecord(fill_ev, st) != hipSuccess)
            return broke(fail(COALA_EHIP, "hipEventRecord failed"));
        fill_evs[k] = (rode && !must_be_mine) ? rode : fill_ev;
    }
    if (total_recv == 0) // nothing to serve: the rounds below still run (peers may owe this rank rows)
        for (int k = 0; k < K; ++k) {
            fill_evs[k] = (k == K - 1 && ev_end_st) ? ev_end_st : (exchange_rows ? c->ev_fill[k] : nullptr);
            if (fill_evs[k] && hipEventRecord(fill_evs[k], st) != hipSuccess) return broke(fail(COALA_EHIP, "hipEventRecord failed"));
        }
    hipEvent_t x_evs[kMaxRounds] = {};
    if (exchange_rows) {
        for (int k = 0; k < K; ++k) {
            for (int p = 0; p < G; ++p) {
                const size_t a = rcnt[p] * (size_t)k / (size_t)K, b = rcnt[p] * (size_t)(k + 1) / (size_t)K;     // what I serve to p
                const size_t u = scnt[p] * (size_t)k / (size_t)K, v = scnt[p] * (size_t)(k + 1) / (size_t)K;     // what p serves to me
                const bool skip = p == me && !loop;
                xs_cnt[p] = skip ? 0 : b - a;
                xs_dis[p] = rdis[p] + a;
                xr_cnt[p] = skip ? 0 : v - u;
                xr_dis[p] = sdis[p] + u;
                sb[(size_t)k * G + p] = (int64_t)(sdis[p] + u);
                se[(size_t)k * G + p] = skip ? (int64_t)(sdis[p] + u) : (int64_t)(sdis[p] + v);
            }
            if (hipStreamWaitEvent(c->cs, fill_evs[k], 0) != hipSuccess) return broke(fail(COALA_EHIP, "hipStreamWaitEvent failed"));
            if (k == 0 && t0) (void)hipEventRecord(t0, c->cs);
            if ((rc = c->tr->all_to_all_v(c->rows_send, xs_cnt.data(), xs_dis.data(), rows_recv, xr_cnt.data(), xr_dis.data(), row_bytes, loop, c->cs)))
                return broke(rc);
            x_evs[k] = (k == K - 1 && ev_end_cs) ? ev_end_cs : c->ev_x[k];
            if (hipEventRecord(x_evs[k], c->cs) != hipSuccess) return broke(fail(COALA_EHIP, "hipEventRecord failed"));
        }
        if (t0 && t1) {
            (void)hipEventRecord(t1, c->cs);
            c->prof_live.emplace_back(t0, t1);
            c->prof.calls++;
            c->prof.remote_rows_in += (uint64_t)(n - (int64_t)scnt[me]);
        }
        // 7. un-permute round by round as the rows arrive (bucketed: nothing to un-permute, and the communicator's stream completes its rounds
        //    in order -- the caller's stream waits for the last one only; the workspaces are safe to reuse behind that wait)
        for (int k = (bucketed && !c->plain_events) ? K - 1 : 0; k < K; ++k) {
            if (hipStreamWaitEvent(st, x_evs[k], 0) != hipSuccess) return broke(fail(COALA_EHIP, "hipStreamWaitEvent failed"));
            if (!bucketed && (rc = coala_cache_scatter_ranges(h, out, c->rows_recv, c->map, sb.data() + (size_t)k * G, se.data() + (size_t)k * G, G, st)))
                return broke(rc);
        }
    }
    if (ev_mode) {
        c->last_ev[0] = ev_begin;
        c->last_ev[1] = ev_end_st;
        c->last_ev[2] = exchange_rows ? ev_end_cs : nullptr;
        c->fev_calls++;
    }
    return COALA_OK;
}

} // extern "C"
