// coala_sampler.hip -- uniform multi-layer neighbour sampler + block compaction over a CSC graph, for gfx950.
//
// Replaces the third-party step of the hot path: dgl.dataloading.MultiLayerNeighborSampler(fanouts).sample(g, seeds) on a
// CSC graph (call site /root/reference/COALA-GNN-Setup/COALA_GNN/COALA_GNN_DataLoader.py:162, sampler built at
// examples/sbatch_ssd_gnn_train.py:70-72, graph at examples/ssd_gnn_dataloader.py:523).  The arithmetic of that step
// lives in DGL 2.5, which is not under /root/reference: parity is pinned by properties and by the CPU twin in
// oracle/coala_oracle.c (same counter-based RNG), not by DGL's RNG stream.
//
// Contract (per layer, fan-out f, destination nodes dst[0..n_dst)):
//   * node v with in-degree deg = indptr[v+1]-indptr[v]: all in-neighbours when deg <= f, otherwise f distinct positions
//     drawn by Floyd's algorithm with r(j) = splitmix64(key(seed, step, layer, v) + j), t = mulhi64(r, j+1);
//   * source nodes of the block = dst nodes first (in order), then every other sampled neighbour in order of FIRST
//     APPEARANCE in the row-major (d, j) scan -- made deterministic with an atomicMin on the first position + prefix sum,
//     whatever order the hash-table inserts land in;
//   * nbr_local[d*f + j] = index of the j-th sampled neighbour of dst d inside the source list, or -1.
// The whole multi-layer sample is enqueued without a host round trip: layer l+1 reads its dst count from device memory.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <new>
#include <vector>

#include "../../include/coala_hip.h"
#include "coala_internal.h"

#define fail coala_fail_
#define HIPCHK COALA_HIPCHK

namespace {

constexpr long long kEmpty = -1;

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ inline uint64_t sample_key(uint64_t seed, uint64_t step, int layer, uint64_t v) {
    uint64_t h = splitmix64(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(layer + 1)));
    h = splitmix64(h ^ (step * 0xD1B54A32D192ED03ull));
    return splitmix64(h ^ v);
}

__device__ __forceinline__ uint32_t hash_slot(int64_t key, uint32_t mask) { return (uint32_t)splitmix64((uint64_t)key) & mask; }

__device__ __forceinline__ int64_t item_key(const int64_t* dst, const int64_t* nbr, int64_t n_dst, int64_t p) {
    return p < n_dst ? dst[p] : nbr[p - n_dst];
}

// S1+S2 fused, a lane per sampled neighbour: GS lanes (16/32/64 >= fanout+1) work on one destination node.  Lane c < fanout
// draws Floyd's c-th candidate on its own, the duplicate resolution walks c = 0..fanout-1 with one shuffle + one ballot per
// step (bit-identical to the sequential loop of the CPU twin), then every lane loads ITS neighbour and inserts it into the
// hash table; lane `fanout` inserts the destination node itself.  Replaces the thread-per-node sampler followed by a separate
// insert kernel: the index loads and the CAS chains of one node now run side by side instead of one after the other.
template <int GS>
__global__ __launch_bounds__(256) void sample_insert_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                            const int64_t* __restrict__ dst, const int64_t* __restrict__ n_dst_dev,
                                                            int fanout, uint64_t seed, uint64_t step, int layer, int64_t num_nodes,
                                                            int64_t* __restrict__ nbr, long long* __restrict__ keys,
                                                            uint32_t* __restrict__ minpos, uint32_t mask, uint32_t* __restrict__ slot_of_item) {
    constexpr int GPW = 64 / GS; // groups per wave
    const int64_t n_dst = *n_dst_dev;
    const int lane = threadIdx.x & 63;
    const int gl = lane % GS;
    const int gbase = lane - gl;
    const uint64_t gmask = (GS == 64) ? ~0ull : (((1ull << GS) - 1ull) << gbase);
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t d0 = wave * GPW; d0 < n_dst; d0 += n_waves * GPW) { // wave-uniform trip count: ballots below need every lane
        const int64_t d = d0 + lane / GS;
        const bool active = d < n_dst;
        const int64_t v = active ? dst[d] : -1;
        const bool okv = active && v >= 0 && v < num_nodes;
        const int64_t start = okv ? indptr[v] : 0;
        const int64_t deg = okv ? indptr[v + 1] - start : 0;
        // candidate of lane c = gl (Floyd step j = deg - fanout + c)
        const uint64_t key = sample_key(seed, step, layer, (uint64_t)v);
        const int64_t jmine = deg - fanout + gl;
        const int64_t t = (deg > fanout && gl < fanout) ? (int64_t)__umul64hi(splitmix64(key + (uint64_t)gl), (uint64_t)(jmine + 1)) : -1;
        int64_t chosen = -2;
        for (int c = 0; c < fanout; ++c) {
            const int64_t tc = __shfl(t, gbase + c);
            const uint64_t dupm = __ballot(gl < c && chosen == tc) & gmask;
            if (gl == c) chosen = dupm ? (deg - fanout + c) : tc;
        }
        int64_t pick = -1;
        if (gl < fanout) pick = (deg <= fanout) ? (gl < deg ? (int64_t)gl : -1) : chosen;
        const int64_t nb = (okv && pick >= 0) ? indices[start + pick] : kEmpty;
        if (active && gl < fanout) nbr[d * fanout + gl] = nb;
        // ---- hash insert: neighbours at positions n_dst + d*fanout + gl, the node itself at position d
        int64_t k = kEmpty;
        int64_t p = -1;
        if (active && gl < fanout) { k = nb; p = n_dst + d * fanout + gl; }
        else if (active && gl == fanout) { k = v; p = d; }
        if (p >= 0) {
            if (k < 0) {
                slot_of_item[p] = 0xFFFFFFFFu;
            } else {
                uint32_t s = hash_slot(k, mask);
                while (true) {
                    const long long cur = keys[s];
                    if (cur == k) break;
                    if (cur == kEmpty) {
                        const long long old = atomicCAS((unsigned long long*)(keys + s), (unsigned long long)kEmpty, (unsigned long long)k);
                        if (old == kEmpty || old == k) break;
                    }
                    s = (s + 1) & mask;
                }
                atomicMin(minpos + s, (uint32_t)p);
                slot_of_item[p] = s;
            }
        }
    }
}

constexpr int kScanItems = 4;
constexpr int kScanBlock = 256;
constexpr int kScanTile = kScanItems * kScanBlock;

__device__ __forceinline__ uint32_t first_flag(const uint32_t* slot_of_item, const uint32_t* minpos, int64_t p, int64_t n_items) {
    if (p >= n_items) return 0;
    const uint32_t s = slot_of_item[p];
    return (s != 0xFFFFFFFFu && minpos[s] == (uint32_t)p) ? 1u : 0u;
}

// A: per-tile count of first occurrences.
__global__ __launch_bounds__(kScanBlock) void flag_count_kernel(const int64_t* __restrict__ n_dst_dev, int fanout,
                                                                const uint32_t* __restrict__ slot_of_item, const uint32_t* __restrict__ minpos,
                                                                uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t wsum[kScanBlock / 64];
    const int64_t n_items = *n_dst_dev * (fanout + 1);
    const int64_t n_tiles = (n_items + kScanTile - 1) / kScanTile;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t c = 0;
        const int64_t base = tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
        for (int i = 0; i < kScanItems; ++i) c += first_flag(slot_of_item, minpos, base + i, n_items);
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < kScanBlock / 64; ++w) t += wsum[w];
            tile_sums[tile] = t;
        }
        __syncthreads();
    }
}

// B: one block turns tile_sums into exclusive offsets and publishes the number of source nodes.
__global__ __launch_bounds__(1024) void tile_scan_kernel(const int64_t* __restrict__ n_dst_dev, int fanout, uint32_t* __restrict__ tile_sums,
                                                         int64_t* __restrict__ n_src_out) {
    __shared__ uint32_t part[1024];
    const int64_t n_items = *n_dst_dev * (fanout + 1);
    const int64_t n_tiles = (n_items + kScanTile - 1) / kScanTile;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const int64_t per = (n_tiles + nthr - 1) / nthr;
    const int64_t lo = (int64_t)tid * per;
    const int64_t hi = lo + per < n_tiles ? lo + per : n_tiles;
    uint32_t s = 0;
    for (int64_t t = lo; t < hi; ++t) s += tile_sums[t];
    part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < nthr; off <<= 1) {
        const uint32_t v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (int64_t t = lo; t < hi; ++t) {
        const uint32_t c = tile_sums[t];
        tile_sums[t] = run;
        run += c;
    }
    if (tid == nthr - 1) *n_src_out = (int64_t)part[tid];
}

// C: local index of every first occurrence = tile offset + in-tile exclusive scan; writes the source list.
__global__ __launch_bounds__(kScanBlock) void assign_local_kernel(const int64_t* __restrict__ dst, const int64_t* __restrict__ nbr,
                                                                  const int64_t* __restrict__ n_dst_dev, int fanout,
                                                                  const uint32_t* __restrict__ slot_of_item, const uint32_t* __restrict__ minpos,
                                                                  const uint32_t* __restrict__ tile_offsets, uint32_t* __restrict__ local_of_slot,
                                                                  int64_t* __restrict__ src_nodes) {
    __shared__ uint32_t woff[kScanBlock / 64];
    const int64_t n_dst = *n_dst_dev;
    const int64_t n_items = n_dst * (fanout + 1);
    const int64_t n_tiles = (n_items + kScanTile - 1) / kScanTile;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
        uint32_t f[kScanItems];
        uint32_t c = 0;
        for (int i = 0; i < kScanItems; ++i) { f[i] = first_flag(slot_of_item, minpos, base + i, n_items); c += f[i]; }
        uint32_t incl = c; // inclusive scan inside the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) woff[w] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int q = 0; q < w; ++q) wbase += woff[q];
        uint32_t run = tile_offsets[tile] + wbase + incl - c;
        for (int i = 0; i < kScanItems; ++i) {
            if (f[i]) {
                const int64_t p = base + i;
                src_nodes[run] = item_key(dst, nbr, n_dst, p);
                local_of_slot[slot_of_item[p]] = run;
                ++run;
            }
        }
        __syncthreads();
    }
}

// A+B+C in one launch for small layers (<= kSmallItems items): one 1024-thread block walks the items tile by tile with a
// running carry.  Saves two launches and two dependent kernel boundaries per layer where the work is a few microseconds.
constexpr int kSmallItems = 1 << 13; // measured: above ~8k items the three parallel kernels win over one block
__global__ __launch_bounds__(1024) void flag_scan_assign_small_kernel(const int64_t* __restrict__ dst, const int64_t* __restrict__ nbr,
                                                                      const int64_t* __restrict__ n_dst_dev, int fanout,
                                                                      const uint32_t* __restrict__ slot_of_item,
                                                                      const uint32_t* __restrict__ minpos,
                                                                      uint32_t* __restrict__ local_of_slot, int64_t* __restrict__ src_nodes,
                                                                      int64_t* __restrict__ n_src_out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_carry;
    const int64_t n_dst = *n_dst_dev;
    const int64_t n_items = n_dst * (fanout + 1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int IT = 4;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t tile0 = 0; tile0 < n_items; tile0 += 1024 * IT) {
        const int64_t base = tile0 + (int64_t)threadIdx.x * IT;
        uint32_t f[IT];
        uint32_t c = 0;
        for (int i = 0; i < IT; ++i) { f[i] = first_flag(slot_of_item, minpos, base + i, n_items); c += f[i]; }
        uint32_t incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t wbase = s_carry;
        for (int q = 0; q < w; ++q) wbase += wsum[q];
        uint32_t run = wbase + incl - c;
        for (int i = 0; i < IT; ++i) {
            if (f[i]) {
                const int64_t p = base + i;
                src_nodes[run] = item_key(dst, nbr, n_dst, p);
                local_of_slot[slot_of_item[p]] = run;
                ++run;
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = run; // last thread's running index == total so far
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_src_out = (int64_t)s_carry;
}

// S4: neighbour -> local index in the source list.
__global__ __launch_bounds__(256) void relabel_kernel(const int64_t* __restrict__ n_dst_dev, int fanout, const uint32_t* __restrict__ slot_of_item,
                                                      const uint32_t* __restrict__ local_of_slot, int32_t* __restrict__ nbr_local) {
    const int64_t n_dst = *n_dst_dev;
    const int64_t n_nbr = n_dst * fanout;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_nbr; q += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t s = slot_of_item[n_dst + q];
        nbr_local[q] = (s == 0xFFFFFFFFu) ? -1 : (int32_t)local_of_slot[s];
    }
}

int grid1d(int64_t n, int block, int cap) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

} // namespace

struct coala_sampler {
    int device = 0;
    const int64_t* indptr = nullptr;
    const int64_t* indices = nullptr;
    int64_t num_nodes = 0, num_edges = 0;
    // workspace, grown on demand
    int64_t* nbr_global = nullptr;    uint64_t nbr_cap = 0;
    long long* keys = nullptr;        uint32_t* minpos = nullptr; uint32_t* local_of_slot = nullptr; uint64_t table_cap = 0;
    uint32_t* slot_of_item = nullptr; uint64_t item_cap = 0;
    uint32_t* tile_sums = nullptr;    uint64_t tile_cap = 0;
    int64_t* counts_dev = nullptr;    // [kMaxLayers+1] dst/src counts per layer
    int64_t* counts_pinned = nullptr; // pinned host staging for the seed count (H2D) and the per-layer source counts (D2H):
                                      // a ring of 8 slots so that asynchronous calls (n_src_host == NULL) do not overwrite each other
    uint64_t calls = 0;
};

extern "C" {

int coala_sampler_create(int device, const int64_t* indptr, const int64_t* indices, int64_t num_nodes, int64_t num_edges,
                         coala_sampler_t** out) {
    if (!out || !indptr || !indices || num_nodes <= 0 || num_edges < 0) return fail(COALA_EINVAL, "bad sampler arguments");
    HIPCHK(hipSetDevice(device));
    coala_sampler* s = new (std::nothrow) coala_sampler();
    if (!s) return fail(COALA_ENOMEM, "out of host memory");
    s->device = device;
    s->indptr = indptr;
    s->indices = indices;
    s->num_nodes = num_nodes;
    s->num_edges = num_edges;
    if (hipMalloc((void**)&s->counts_dev, (COALA_SAMPLER_MAX_LAYERS + 1) * sizeof(int64_t)) != hipSuccess ||
        hipHostMalloc((void**)&s->counts_pinned, 8 * (COALA_SAMPLER_MAX_LAYERS + 2) * sizeof(int64_t)) != hipSuccess) {
        delete s;
        return fail(COALA_ENOMEM, "hipMalloc failed");
    }
    *out = s;
    return COALA_OK;
}

int coala_sampler_destroy(coala_sampler_t* s) {
    if (!s) return COALA_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {s->nbr_global, s->keys, s->minpos, s->local_of_slot, s->slot_of_item, s->tile_sums, s->counts_dev};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->counts_pinned) (void)hipHostFree(s->counts_pinned);
    delete s;
    return COALA_OK;
}

static int grow(void** p, uint64_t* cap, uint64_t need, size_t elem, hipStream_t st) {
    if (need <= *cap) return COALA_OK;
    HIPCHK(hipStreamSynchronize(st));
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    uint64_t c = *cap ? *cap : 1024;
    while (c < need) c *= 2;
    HIPCHK(hipMalloc(p, c * elem));
    *cap = c;
    return COALA_OK;
}

int coala_sampler_sample(coala_sampler_t* s, const int64_t* seeds, int64_t n_seeds, const int32_t* fanouts, int n_layers,
                         uint64_t seed, uint64_t step, int64_t* const* src_nodes_out, int32_t* const* nbr_local_out,
                         int64_t* n_src_host, void* stream) {
    if (!s || !seeds || !fanouts || !src_nodes_out || !nbr_local_out) return fail(COALA_EINVAL, "null argument");
    if (n_layers < 1 || n_layers > COALA_SAMPLER_MAX_LAYERS) return fail(COALA_EINVAL, "n_layers must be 1..%d", COALA_SAMPLER_MAX_LAYERS);
    if (n_seeds < 0 || n_seeds > 0x7FFFFFFF) return fail(COALA_EINVAL, "bad n_seeds");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(s->device));
    // capacities: layer l has at most cap_l dst nodes and cap_l*(f_l+1) source nodes
    int64_t cap = n_seeds;
    uint64_t max_items = 0, max_nbr = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int f = fanouts[l];
        if (f < 1 || f > 32) return fail(COALA_EINVAL, "fan-out %d outside 1..32", f);
        const uint64_t items = (uint64_t)cap * (uint64_t)(f + 1);
        if (items > 0x7FFFFFFFull) return fail(COALA_EINVAL, "layer %d would hold more than 2^31 items", l);
        if (items > max_items) max_items = items;
        if ((uint64_t)cap * f > max_nbr) max_nbr = (uint64_t)cap * f;
        cap = (int64_t)items;
    }
    uint64_t table = 1024;
    while (table < 2 * max_items) table *= 2;
    int rc;
    if ((rc = grow((void**)&s->nbr_global, &s->nbr_cap, max_nbr ? max_nbr : 1, sizeof(int64_t), st))) return rc;
    if ((rc = grow((void**)&s->slot_of_item, &s->item_cap, max_items ? max_items : 1, sizeof(uint32_t), st))) return rc;
    if ((rc = grow((void**)&s->tile_sums, &s->tile_cap, max_items / kScanTile + 2, sizeof(uint32_t), st))) return rc;
    if (table > s->table_cap) {
        HIPCHK(hipStreamSynchronize(st));
        for (void** p : {(void**)&s->keys, (void**)&s->local_of_slot})
            if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
        HIPCHK(hipMalloc((void**)&s->keys, table * (sizeof(long long) + sizeof(uint32_t)))); // keys + first-position words
        HIPCHK(hipMalloc((void**)&s->local_of_slot, table * sizeof(uint32_t)));
        s->table_cap = table;
    }
    if (n_seeds == 0) {
        for (int l = 0; l < n_layers; ++l)
            if (n_src_host) n_src_host[l] = 0;
        return COALA_OK;
    }
    int64_t* pin = s->counts_pinned + (s->calls++ % 8) * (COALA_SAMPLER_MAX_LAYERS + 2);
    pin[0] = n_seeds; // pinned: a truly asynchronous 8-byte H2D (a pageable source is staged synchronously)
    HIPCHK(hipMemcpyAsync(s->counts_dev, pin, sizeof(int64_t), hipMemcpyHostToDevice, st));
    const int64_t* dst = seeds;
    cap = n_seeds;
    for (int l = 0; l < n_layers; ++l) {
        const int f = fanouts[l];
        const int64_t items_cap = cap * (f + 1);
        uint64_t tbl = 1024;
        while (tbl < 2 * (uint64_t)items_cap) tbl *= 2;
        const uint32_t mask = (uint32_t)(tbl - 1);
        const int64_t* n_dst_dev = s->counts_dev + l;
        // keys (8 B) and first-position words (4 B) of this layer's table sit back to back: one 0xFF fill clears both
        long long* keys = s->keys;
        uint32_t* minpos = reinterpret_cast<uint32_t*>(s->keys + tbl);
        HIPCHK(hipMemsetAsync(keys, 0xFF, tbl * (sizeof(long long) + sizeof(uint32_t)), st));
        if (f < 16)
            hipLaunchKernelGGL(sample_insert_kernel<16>, dim3(grid1d(cap * 16, 256, 8192)), dim3(256), 0, st, s->indptr, s->indices, dst,
                               n_dst_dev, f, seed, step, l, s->num_nodes, s->nbr_global, keys, minpos, mask, s->slot_of_item);
        else if (f < 32)
            hipLaunchKernelGGL(sample_insert_kernel<32>, dim3(grid1d(cap * 32, 256, 8192)), dim3(256), 0, st, s->indptr, s->indices, dst,
                               n_dst_dev, f, seed, step, l, s->num_nodes, s->nbr_global, keys, minpos, mask, s->slot_of_item);
        else
            hipLaunchKernelGGL(sample_insert_kernel<64>, dim3(grid1d(cap * 64, 256, 8192)), dim3(256), 0, st, s->indptr, s->indices, dst,
                               n_dst_dev, f, seed, step, l, s->num_nodes, s->nbr_global, keys, minpos, mask, s->slot_of_item);
        if (items_cap <= kSmallItems) {
            hipLaunchKernelGGL(flag_scan_assign_small_kernel, dim3(1), dim3(1024), 0, st, dst, s->nbr_global, n_dst_dev, f, s->slot_of_item,
                               minpos, s->local_of_slot, src_nodes_out[l], s->counts_dev + l + 1);
        } else {
            const int tiles = grid1d(items_cap, kScanTile, 4096);
            hipLaunchKernelGGL(flag_count_kernel, dim3(tiles), dim3(kScanBlock), 0, st, n_dst_dev, f, s->slot_of_item, minpos, s->tile_sums);
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, n_dst_dev, f, s->tile_sums, s->counts_dev + l + 1);
            hipLaunchKernelGGL(assign_local_kernel, dim3(tiles), dim3(kScanBlock), 0, st, dst, s->nbr_global, n_dst_dev, f, s->slot_of_item,
                               minpos, s->tile_sums, s->local_of_slot, src_nodes_out[l]);
        }
        hipLaunchKernelGGL(relabel_kernel, dim3(grid1d(cap * f, 256, 8192)), dim3(256), 0, st, n_dst_dev, f, s->slot_of_item,
                           s->local_of_slot, nbr_local_out[l]);
        dst = src_nodes_out[l];
        cap = items_cap;
    }
    HIPCHK(hipGetLastError());
    if (n_src_host) { // the one host read of the call: every layer's source count
        HIPCHK(hipMemcpyAsync(pin + 1, s->counts_dev + 1, n_layers * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int l = 0; l < n_layers; ++l) n_src_host[l] = pin[1 + l];
    }
    return COALA_OK;
}

} // extern "C"
