// coala_comm.cpp -- the owner-partitioned fetch as ONE native call: route -> RCCL all-to-all (counts) -> all-to-all-v (ids)
// -> owner serve -> all-to-all-v (rows) -> un-permute, on one HIP stream, one host read (the 2G counts) per minibatch.
//
// Replaces, fused (paths relative to /root/reference): SSD_GNN_NVSHMEM_Cache::send_requests + read_feature
// (COALA_GNN_Modules/ssd_gnn_cache.cuh:111-174: N x 2 one-sided 8-byte puts, 3 nvshmem_barrier_all, N warp-level row puts) and
// the "nccl" orchestration in COALA-GNN-Setup/COALA_GNN/COALA_GNN_Manager.py:143-211 (full-capacity all_to_all of ids,
// G(G-1) serial send/recv of rows).  RCCL's ncclAllToAllv drives every direct xGMI link of the GPU at once.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstring>
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

struct coala_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    // workspace (device), grown on demand
    int64_t *node = nullptr, *map = nullptr;
    uint64_t batch_cap = 0;
    int64_t* recv_ids = nullptr;
    uint64_t recv_cap = 0;
    float *rows_send = nullptr, *rows_recv = nullptr;
    uint64_t rows_send_cap = 0, rows_recv_cap = 0; // in floats
    int64_t* counts_dev = nullptr;                 // [3G+1]: send counts, recv counts, offsets
    int64_t* counts_host = nullptr;                // pinned [2G]
    // last step, for tests / diagnostics
    std::vector<int64_t> last_send, last_recv;
};

namespace {
int grow(void** p, uint64_t* cap, uint64_t need, size_t elem, hipStream_t st) {
    if (need <= *cap) return COALA_OK;
    HIPCHK(hipStreamSynchronize(st));
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    uint64_t c = *cap ? *cap : 4096;
    while (c < need) c *= 2;
    HIPCHK(hipMalloc(p, c * elem));
    *cap = c;
    return COALA_OK;
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
    if (!c) return fail(COALA_ENOMEM, "out of host memory");
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    ncclUniqueId id;
    memcpy(&id, id_bytes, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(COALA_ECOMM, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    if (hipMalloc((void**)&c->counts_dev, (3 * (size_t)nranks + 1) * sizeof(int64_t)) != hipSuccess ||
        hipHostMalloc((void**)&c->counts_host, 2 * (size_t)nranks * sizeof(int64_t)) != hipSuccess) {
        coala_comm_destroy(c);
        return fail(COALA_ENOMEM, "communicator workspace allocation failed");
    }
    c->last_send.assign(nranks, 0);
    c->last_recv.assign(nranks, 0);
    *out = c;
    return COALA_OK;
}

int coala_comm_destroy(coala_comm_t* c) {
    if (!c) return COALA_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm) (void)ncclCommDestroy(c->comm);
    void* dev[] = {c->node, c->map, c->recv_ids, c->rows_send, c->rows_recv, c->counts_dev};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    if (c->counts_host) (void)hipHostFree(c->counts_host);
    delete c;
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

int coala_cache_fetch_distributed(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n, void* stream) {
    if (!h || !c) return fail(COALA_EINVAL, "null handle");
    if (n < 0 || (n > 0 && (!out || !idx))) return fail(COALA_EINVAL, "bad batch");
    coala_cache_geometry_t geo;
    int rc = coala_cache_geometry(h, &geo);
    if (rc) return rc;
    const int G = c->nranks;
    const int64_t dim = coala_cache_row_dim(h);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    const uint64_t nb = (uint64_t)(n > 0 ? n : 1);
    if (nb > c->batch_cap) {
        uint64_t cap_node = c->batch_cap, cap_map = c->batch_cap;
        if ((rc = grow((void**)&c->node, &cap_node, nb, sizeof(int64_t), st))) return rc;
        if ((rc = grow((void**)&c->map, &cap_map, nb, sizeof(int64_t), st))) return rc;
        c->batch_cap = cap_node < cap_map ? cap_node : cap_map;
    }
    if ((rc = grow((void**)&c->rows_recv, &c->rows_recv_cap, nb * (uint64_t)dim, sizeof(float), st))) return rc;
    int64_t* send_cnt = c->counts_dev;
    int64_t* recv_cnt = c->counts_dev + G;
    int64_t* offsets = c->counts_dev + 2 * G; // [G+1]
    // 1. bucket by owner (stable), packed layout
    if ((rc = coala_cache_route(h, idx, n, G, 0, c->node, c->map, send_cnt, offsets, st))) return rc;
    // 2. counts: every rank tells every owner how many ids follow
    NCCLCHK(ncclAllToAll(send_cnt, recv_cnt, 1, ncclInt64, c->comm, st));
    // 3. the one host read of the step
    HIPCHK(hipMemcpyAsync(c->counts_host, c->counts_dev, 2 * (size_t)G * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<size_t> scnt(G), sdis(G), rcnt(G), rdis(G);
    size_t total_recv = 0, acc = 0;
    for (int p = 0; p < G; ++p) {
        scnt[p] = (size_t)c->counts_host[p];
        rcnt[p] = (size_t)c->counts_host[G + p];
        sdis[p] = acc;
        rdis[p] = total_recv;
        acc += scnt[p];
        total_recv += rcnt[p];
        c->last_send[p] = (int64_t)scnt[p];
        c->last_recv[p] = (int64_t)rcnt[p];
    }
    if ((int64_t)acc != n) return fail(COALA_ECOMM, "route produced %zu ids for a batch of %lld", acc, (long long)n);
    const uint64_t tr = total_recv ? total_recv : 1;
    if ((rc = grow((void**)&c->recv_ids, &c->recv_cap, tr, sizeof(int64_t), st))) return rc;
    if ((rc = grow((void**)&c->rows_send, &c->rows_send_cap, tr * (uint64_t)dim, sizeof(float), st))) return rc;
    // 4. ids to their owners (exact sizes)
    NCCLCHK(ncclAllToAllv(c->node, scnt.data(), sdis.data(), c->recv_ids, rcnt.data(), rdis.data(), ncclInt64, c->comm, st));
    // 5. owner side: one batch = the concatenation in source-rank order
    if ((rc = coala_cache_serve(h, c->rows_send, c->recv_ids, (int64_t)total_recv, st))) return rc;
    // 6. rows back to the requesters
    for (int p = 0; p < G; ++p) { scnt[p] *= (size_t)dim; sdis[p] *= (size_t)dim; rcnt[p] *= (size_t)dim; rdis[p] *= (size_t)dim; }
    NCCLCHK(ncclAllToAllv(c->rows_send, rcnt.data(), rdis.data(), c->rows_recv, scnt.data(), sdis.data(), ncclFloat, c->comm, st));
    // 7. un-permute into the caller's order
    if ((rc = coala_cache_scatter(h, out, c->rows_recv, c->map, n, st))) return rc;
    return COALA_OK;
}

} // extern "C"
