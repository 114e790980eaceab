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
//
// Three launches per layer, nothing else on the stream (round 1: five launches + a memset per layer, a D2H copy and a stream
// synchronisation per call):
//   sample_insert   draw + hash insert;
//   scan_assign     first-occurrence flags -> positions in ONE pass: tiles are handed out by an atomic ticket and chained with a
//                   decoupled look-back over generation-tagged status words (no clearing pass, no second kernel for the tile sums);
//   relabel_clear   neighbour -> local index, and the hash table of the NEXT layer (or of the next call) is cleared alongside.
// Layer l+1 reads its destination count from device memory; the counts also go straight into pinned host memory from the kernel
// that produces them, and the host collects them by waiting on an event behind the last kernel -- never on the stream.
// (Tried and dropped: the whole call as one persistent kernel with grid barriers.  Every barrier needs a device-scope fence, i.e.
// an L2 write-back + invalidate issued by every block; 9 barriers cost more than the 10 kernel boundaries they replaced: 0.186 ms
// against 0.107 ms per 5,5 call.)
#include <hip/hip_runtime.h>

#include <algorithm>
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
constexpr int kBlock = 256;                 // threads per block
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kItems = 4;                   // items per thread in the scan phases
constexpr int kTile = kBlock * kItems;      // items per block tile
constexpr int kMaxTiles = 8192;             // tiles per layer (8.4 M items): status words of the single-pass scan
constexpr int kRing = 8;                    // calls whose counts may be outstanding at once
constexpr int kMaxParts = 64;
constexpr int kRouteTile = 64 * kItems;     // ids per wave step of the bucketing phases

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

__host__ __device__ inline uint32_t table_size(int64_t n_items) { // power of two, at most half full
    uint32_t t = 1024;
    while ((int64_t)t < 2 * n_items) t <<= 1;
    return t;
}

struct Graph {
    const int64_t* indptr;
    const int64_t* indices;
    int64_t num_nodes;
};

struct Table {
    long long* keys;          // [table] hash keys
    uint32_t* minpos;         // [table] first position of the key in the (dst..., neighbours...) item list
    uint32_t* local_of_slot;  // [table] index of the key in the source list
};

__device__ __forceinline__ void clear_table(const Table& t, uint32_t tbl) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tbl; i += gridDim.x * blockDim.x) {
        t.keys[i] = kEmpty;
        t.minpos[i] = 0xFFFFFFFFu;
    }
}

// Sample + insert, a lane per sampled neighbour: GS lanes (16/32/64 >= fanout+1) work on one destination node.  Lane c < fanout
// draws Floyd's c-th candidate on its own, the duplicate resolution walks c = 0..fanout-1 with one shuffle + one ballot per
// step (bit-identical to the sequential loop of the CPU twin), then every lane loads ITS neighbour and inserts it into the
// hash table; lane `fanout` inserts the destination node itself.
template <int GS>
__global__ __launch_bounds__(kBlock) void sample_insert_kernel(Graph g, const int64_t* __restrict__ dst, const int64_t* __restrict__ n_dst_dev,
                                                               int64_t n_dst_value /* used when n_dst_dev is null: the first layer */, int fanout, uint64_t seed, uint64_t step, int layer, int64_t* __restrict__ nbr,
                                                               Table tb, uint32_t* __restrict__ slot_of_item) {
    constexpr int GPW = 64 / GS; // groups per wave
    const int64_t n_dst = n_dst_dev ? *n_dst_dev : n_dst_value;
    const uint32_t mask = table_size(n_dst * (fanout + 1)) - 1;
    const int lane = threadIdx.x & 63;
    const int gl = lane % GS;
    const int gbase = lane - gl;
    const uint64_t gmask = (GS == 64) ? ~0ull : (((1ull << GS) - 1ull) << gbase);
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t d0 = wave * GPW; d0 < n_dst; d0 += n_waves * GPW) { // wave-uniform trip count: ballots below need every lane
        const int64_t d = d0 + lane / GS;
        const bool active = d < n_dst;
        const int64_t v = active ? dst[d] : -1;
        const bool okv = active && v >= 0 && v < g.num_nodes;
        const int64_t start = okv ? g.indptr[v] : 0;
        const int64_t deg = okv ? g.indptr[v + 1] - start : 0;
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
        const int64_t nb = (okv && pick >= 0) ? g.indices[start + pick] : kEmpty;
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
                    const long long cur = tb.keys[s];
                    if (cur == k) break;
                    if (cur == kEmpty) {
                        const long long old = atomicCAS((unsigned long long*)(tb.keys + s), (unsigned long long)kEmpty, (unsigned long long)k);
                        if (old == kEmpty || old == k) break;
                    }
                    s = (s + 1) & mask;
                }
                atomicMin(tb.minpos + s, (uint32_t)p);
                slot_of_item[p] = s;
            }
        }
    }
}

__device__ __forceinline__ uint32_t first_flag(const uint32_t* __restrict__ slot_of_item, const uint32_t* __restrict__ minpos, int64_t p,
                                               int64_t n_items) {
    if (p >= n_items) return 0;
    const uint32_t s = slot_of_item[p];
    return (s != 0xFFFFFFFFu && minpos[s] == (uint32_t)p) ? 1u : 0u;
}

// Single-pass scan: a block takes the next tile with an atomic ticket (so every predecessor of its tile has already started),
// publishes the tile's count as an AGGREGATE, looks back over its predecessors until it meets an INCLUSIVE prefix, publishes its
// own INCLUSIVE prefix and numbers its first occurrences.  Status words carry the launch's generation: nothing to reset.
//   word = gen << 34 | status << 32 | value          status: 1 = aggregate, 2 = inclusive prefix
constexpr unsigned long long kAggregate = 1ull << 32, kInclusive = 2ull << 32;
__global__ __launch_bounds__(kBlock) void scan_assign_kernel(const int64_t* __restrict__ dst, const int64_t* __restrict__ nbr,
                                                             const int64_t* __restrict__ n_dst_dev, int64_t n_dst_value, int fanout,
                                                             const uint32_t* __restrict__ slot_of_item, Table tb, unsigned long long* __restrict__ status,
                                                             unsigned long long* __restrict__ ticket, unsigned long long ticket_base,
                                                             unsigned long long gen, int64_t* __restrict__ src_nodes, int64_t* __restrict__ n_src_dev,
                                                             int64_t* __restrict__ n_src_host) {
    __shared__ uint32_t s_woff[kWavesPerBlock];
    __shared__ unsigned long long s_tile;
    __shared__ uint32_t s_prefix;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t n_dst = n_dst_dev ? *n_dst_dev : n_dst_value;
    const int64_t n_items = n_dst * (fanout + 1);
    const int64_t n_tiles = (n_items + kTile - 1) / kTile;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1ull) - ticket_base;
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    if (tile >= n_tiles) { // launched for the capacity; the block that would own the first unused tile reports an empty layer
        if (tile == 0 && threadIdx.x == 0) { *n_src_dev = 0; *n_src_host = 0; }
        return;
    }
    const int64_t base = tile * kTile + (int64_t)threadIdx.x * kItems;
    uint32_t fl[kItems];
    uint32_t c = 0;
    for (int i = 0; i < kItems; ++i) { fl[i] = first_flag(slot_of_item, tb.minpos, base + i, n_items); c += fl[i]; }
    uint32_t incl = c; // inclusive scan inside the wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_woff[w] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int q = 0; q < kWavesPerBlock; ++q) {
        if (q < w) wbase += s_woff[q];
        total += s_woff[q];
    }
    if (threadIdx.x == 0) {
        const unsigned long long tag = gen << 34;
        uint32_t prefix = 0;
        if (tile > 0) {
            __hip_atomic_store(status + tile, tag | kAggregate | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t t = tile - 1; t >= 0;) { // decoupled look-back
                const unsigned long long v = __hip_atomic_load(status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 34) != gen || !(v & (kAggregate | kInclusive))) { __builtin_amdgcn_s_sleep(1); continue; } // not published yet
                prefix += (uint32_t)v;
                if (v & kInclusive) break;
                --t;
            }
        }
        __hip_atomic_store(status + tile, tag | kInclusive | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = prefix;
        if (tile == n_tiles - 1) { // layer l+1 (and the host) read the number of source nodes from here
            *n_src_dev = (int64_t)(prefix + total);
            *n_src_host = (int64_t)(prefix + total);
        }
    }
    __syncthreads();
    uint32_t run = s_prefix + wbase + incl - c;
    for (int i = 0; i < kItems; ++i) {
        if (fl[i]) {
            const int64_t p = base + i;
            src_nodes[run] = item_key(dst, nbr, n_dst, p);
            tb.local_of_slot[slot_of_item[p]] = run;
            ++run;
        }
    }
}

// neighbour -> local index in the source list; alongside, the hash table for what comes next is cleared (keys / minpos are no
// longer read by this layer): next_items_dev != null -> the next layer's size is read from the device, else next_items (next call)
__global__ __launch_bounds__(kBlock) void relabel_clear_kernel(const int64_t* __restrict__ n_dst_dev, int64_t n_dst_value, int fanout, const uint32_t* __restrict__ slot_of_item,
                                                               Table tb, int32_t* __restrict__ nbr_local, const int64_t* __restrict__ next_n_dst_dev,
                                                               int next_fanout, int64_t next_items) {
    const int64_t n_dst = n_dst_dev ? *n_dst_dev : n_dst_value;
    const int64_t n_nbr = n_dst * fanout;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_nbr; q += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t s = slot_of_item[n_dst + q];
        nbr_local[q] = (s == 0xFFFFFFFFu) ? -1 : (int32_t)tb.local_of_slot[s];
    }
    clear_table(tb, table_size(next_n_dst_dev ? *next_n_dst_dev * (next_fanout + 1) : next_items));
}

// ---------------------------------------------------------------------------------------------------------- owner bucketing
// Stable partition of the input nodes by owner = id % n_parts (same ballot / prefix-sum scheme as the cache's route kernels), then
// the last block is re-indexed through the permutation.
__device__ __forceinline__ uint32_t owner_of(uint64_t id, uint32_t n_parts, int pshift) {
    if (pshift >= 0) return (uint32_t)id & (n_parts - 1);
    if ((id >> 32) == 0) return (uint32_t)id % n_parts;
    return (uint32_t)(id % n_parts);
}

__global__ __launch_bounds__(kBlock) void bucket_count_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ n_src_dev, uint32_t P,
                                                              int pshift, uint32_t* __restrict__ wave_counts) {
    const int lane = threadIdx.x & 63;
    const int64_t n_src = *n_src_dev;
    const int64_t n_wt = (n_src + kRouteTile - 1) / kRouteTile;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t wt = wave; wt < n_wt; wt += n_waves) {
        uint32_t mine = 0; // lane g accumulates the count of owner g
        for (int j = 0; j < kItems; ++j) {
            const int64_t i = wt * kRouteTile + j * 64 + lane;
            const uint32_t o = (i < n_src) ? owner_of((uint64_t)src[i], P, pshift) : 0xFFFFFFFFu;
            for (uint32_t g = 0; g < P; ++g) {
                const uint64_t m = __ballot(o == g);
                if ((uint32_t)lane == g) mine += (uint32_t)__builtin_popcountll(m);
            }
        }
        if ((uint32_t)lane < P) wave_counts[wt * P + lane] = mine;
    }
}

// one block, one wave per owner column: exclusive scan over the wave tiles with a running carry, bucket sizes and bases
__global__ __launch_bounds__(1024) void bucket_scan_kernel(uint32_t* __restrict__ wave_counts, const int64_t* __restrict__ n_src_dev, uint32_t P,
                                                           int64_t* __restrict__ counts_out, int64_t* __restrict__ counts_host, int64_t* __restrict__ bases) {
    __shared__ int64_t totals[kMaxParts];
    const int lane = threadIdx.x & 63;
    const int64_t n_wt = (*n_src_dev + kRouteTile - 1) / kRouteTile;
    for (uint32_t g = threadIdx.x >> 6; g < P; g += blockDim.x >> 6) {
        uint32_t carry = 0;
        for (int64_t t0 = 0; t0 < n_wt; t0 += 64) {
            const int64_t t = t0 + lane;
            const uint32_t c = (t < n_wt) ? wave_counts[t * P + g] : 0u;
            uint32_t incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if (t < n_wt) wave_counts[t * P + g] = carry + incl - c;
            carry += __shfl(incl, 63);
        }
        if (lane == 0) totals[g] = (int64_t)carry;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        for (uint32_t g = 0; g < P; ++g) {
            counts_out[g] = totals[g];
            counts_host[g] = totals[g];
            bases[g] = acc;
            acc += totals[g];
        }
    }
}

__global__ __launch_bounds__(kBlock) void bucket_scatter_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ n_src_dev, uint32_t P,
                                                                int pshift, const uint32_t* __restrict__ wave_offsets, const int64_t* __restrict__ bases,
                                                                int64_t* __restrict__ bucketed, uint32_t* __restrict__ new_of_old) {
    const int lane = threadIdx.x & 63;
    const int64_t n_src = *n_src_dev;
    const int64_t n_wt = (n_src + kRouteTile - 1) / kRouteTile;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t wt = wave; wt < n_wt; wt += n_waves) {
        int64_t off = 0; // lane g: next free slot of bucket g for this wave tile
        if ((uint32_t)lane < P) off = bases[lane] + (int64_t)wave_offsets[wt * P + lane];
        for (int j = 0; j < kItems; ++j) {
            const int64_t i = wt * kRouteTile + j * 64 + lane;
            const bool valid = i < n_src;
            const int64_t id = valid ? src[i] : 0;
            const uint32_t o = valid ? owner_of((uint64_t)id, P, pshift) : 0xFFFFFFFFu;
            int64_t dest = -1;
            for (uint32_t g = 0; g < P; ++g) {
                const uint64_t m = __ballot(o == g);
                const int64_t bg = __shfl(off, (int)g);
                if (o == g) dest = bg + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                if ((uint32_t)lane == g) off += __builtin_popcountll(m);
            }
            if (valid) {
                bucketed[dest] = id;
                new_of_old[i] = (uint32_t)dest;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void bucket_reindex_kernel(const int64_t* __restrict__ n_dst_dev, int64_t n_dst_value, int fanout,
                                                                const uint32_t* __restrict__ new_of_old, int32_t* __restrict__ nbr_local,
                                                                int32_t* __restrict__ dst_in_src) {
    const int64_t n_dst = n_dst_dev ? *n_dst_dev : n_dst_value;
    const int64_t n_nbr = n_dst * fanout;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_nbr; q += (int64_t)gridDim.x * blockDim.x) {
        const int32_t o = nbr_local[q];
        if (o >= 0) nbr_local[q] = (int32_t)new_of_old[o];
    }
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_dst; d += (int64_t)gridDim.x * blockDim.x)
        dst_in_src[d] = (int32_t)new_of_old[d]; // the dst nodes are the first n_dst entries of the unbucketed list
}

// ---------------------------------------------------------------------------------------------------------- block ops
// The one dense-side primitive a consumer of these blocks needs (DGL's SAGEConv "mean" reduces to it): out[d] = mean of the rows
// h_src[nbr[d, j]] over the valid j.  One wave per destination row, 16-B accesses, the neighbour indices read once per wave.
// Replaces gather -> mask -> sum -> divide in eager torch (four passes over a [n_dst, fanout, dim] intermediate).
template <int VEC>
__global__ __launch_bounds__(kBlock) void mean_aggregate_kernel(const int32_t* __restrict__ nbr, const float* __restrict__ h_src,
                                                                float* __restrict__ out, int64_t n_dst, int fanout, int dim) {
    typedef float vf __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int units = dim / VEC;
    for (int64_t d = wave; d < n_dst; d += n_waves) {
        const int32_t mine = lane < fanout ? nbr[d * fanout + lane] : -1; // fan-out <= 32: one load per wave
        const int cnt = __builtin_popcountll(__ballot(mine >= 0));
        const float inv = cnt ? 1.0f / (float)cnt : 0.0f;
        for (int u0 = 0; u0 < units; u0 += 64) {
            const int u = u0 + lane;
            vf acc = vf(0.0f);
            for (int j = 0; j < fanout; ++j) {
                const int32_t idx = __shfl(mine, j);
                if (idx >= 0 && u < units) acc += *reinterpret_cast<const vf*>(h_src + (int64_t)idx * dim + (int64_t)u * VEC);
            }
            if (u < units) *reinterpret_cast<vf*>(out + d * dim + (int64_t)u * VEC) = acc * inv;
        }
    }
}

// grad_src[nbr[d, j]] += grad_out[d] / cnt[d]   (grad_src zeroed by the caller; hardware float atomics: summation order varies)
__global__ __launch_bounds__(kBlock) void mean_aggregate_backward_kernel(const int32_t* __restrict__ nbr, const float* __restrict__ grad_out,
                                                                         float* __restrict__ grad_src, int64_t n_dst, int fanout, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t d = wave; d < n_dst; d += n_waves) {
        const int32_t mine = lane < fanout ? nbr[d * fanout + lane] : -1;
        const int cnt = __builtin_popcountll(__ballot(mine >= 0));
        if (!cnt) continue;
        const float inv = 1.0f / (float)cnt;
        for (int c = lane; c < dim; c += 64) {
            const float g = grad_out[d * dim + c] * inv;
            for (int j = 0; j < fanout; ++j) {
                const int32_t idx = __shfl(mine, j);
                if (idx >= 0) unsafeAtomicAdd(grad_src + (int64_t)idx * dim + c, g);
            }
        }
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
    Graph g{};
    int64_t num_edges = 0;
    // workspace, grown on demand
    int64_t* nbr_global = nullptr;    uint64_t nbr_cap = 0;
    Table tb{};                       uint64_t table_cap = 0;
    uint64_t clean_items = 0;         // the table is clean for a first layer of at most this many items (0 = unknown)
    uint32_t* slot_of_item = nullptr; uint64_t item_cap = 0;
    uint32_t* wave_counts = nullptr;  uint64_t wc_cap = 0;
    uint32_t* new_of_old = nullptr;   uint64_t noo_cap = 0;
    unsigned long long* status = nullptr;  // [kMaxTiles] look-back status words (generation-tagged, never reset)
    unsigned long long* ticket = nullptr;  // tile ticket counter, monotonic across launches
    unsigned long long ticket_total = 0, scan_gen = 0;
    int64_t* counts_dev = nullptr;         // [kMaxLayers + 1] source counts of the call in flight; then [kMaxParts] bucket bases
    // pinned host ring: per call [kMaxLayers source counts][kMaxParts bucket sizes], and an event recorded behind the last kernel
    int64_t* counts_pinned = nullptr; // host pointer
    int64_t* counts_pinned_dev = nullptr;
    hipEvent_t done[kRing] = {};
    int ring_layers[kRing] = {};
    int ring_parts[kRing] = {};
    uint64_t calls = 0;
    hipStream_t last_stream = nullptr; // stream of the previous call
};

namespace {
constexpr int kSlot = COALA_SAMPLER_MAX_LAYERS + kMaxParts; // int64 words per ring slot

int grow(void** p, uint64_t* cap, uint64_t need, size_t elem, hipStream_t st) {
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

int ilog2_exact(uint64_t v) {
    if (v == 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1ull << s) != v) ++s;
    return s;
}
} // namespace

extern "C" {

int coala_sampler_create(int device, const int64_t* indptr, const int64_t* indices, int64_t num_nodes, int64_t num_edges,
                         coala_sampler_t** out) {
    if (!out || !indptr || !indices || num_nodes <= 0 || num_edges < 0) return fail(COALA_EINVAL, "bad sampler arguments");
    HIPCHK(hipSetDevice(device));
    coala_sampler* s = new (std::nothrow) coala_sampler();
    if (!s) return fail(COALA_ENOMEM, "out of host memory");
    s->device = device;
    s->g = Graph{indptr, indices, num_nodes};
    s->num_edges = num_edges;
    bool ok = hipMalloc((void**)&s->status, kMaxTiles * sizeof(unsigned long long)) == hipSuccess &&
              hipMemset(s->status, 0, kMaxTiles * sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void**)&s->ticket, sizeof(unsigned long long)) == hipSuccess &&
              hipMemset(s->ticket, 0, sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void**)&s->counts_dev, (COALA_SAMPLER_MAX_LAYERS + 1 + kMaxParts) * sizeof(int64_t)) == hipSuccess &&
              hipHostMalloc((void**)&s->counts_pinned, kRing * kSlot * sizeof(int64_t), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void**)&s->counts_pinned_dev, s->counts_pinned, 0) == hipSuccess;
    for (int i = 0; i < kRing && ok; ++i) ok = hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok || hipDeviceSynchronize() != hipSuccess) {
        coala_sampler_destroy(s);
        return fail(COALA_ENOMEM, "sampler set-up failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = s;
    return COALA_OK;
}

int coala_sampler_destroy(coala_sampler_t* s) {
    if (!s) return COALA_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {s->nbr_global, s->tb.keys, s->tb.local_of_slot, s->slot_of_item, s->wave_counts, s->new_of_old, s->status, s->ticket, s->counts_dev};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->counts_pinned) (void)hipHostFree(s->counts_pinned);
    for (int i = 0; i < kRing; ++i)
        if (s->done[i]) (void)hipEventDestroy(s->done[i]);
    delete s;
    return COALA_OK;
}

int coala_block_mean_aggregate(int device, const int32_t* nbr, const float* h_src, float* out, int64_t n_dst, int fanout, int dim, void* stream) {
    if (n_dst < 0 || fanout < 1 || fanout > 32 || dim < 1) return fail(COALA_EINVAL, "bad block shape (fan-out 1..32)");
    if (n_dst == 0) return COALA_OK;
    if (!nbr || !h_src || !out) return fail(COALA_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(device));
    const dim3 grid(grid1d(n_dst * 64, kBlock, 8192)), blk(kBlock);
    const bool v4 = dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(h_src) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    if (v4) hipLaunchKernelGGL(mean_aggregate_kernel<4>, grid, blk, 0, (hipStream_t)stream, nbr, h_src, out, n_dst, fanout, dim);
    else hipLaunchKernelGGL(mean_aggregate_kernel<1>, grid, blk, 0, (hipStream_t)stream, nbr, h_src, out, n_dst, fanout, dim);
    HIPCHK(hipGetLastError());
    return COALA_OK;
}

int coala_block_mean_aggregate_backward(int device, const int32_t* nbr, const float* grad_out, float* grad_src, int64_t n_dst, int fanout,
                                        int dim, void* stream) {
    if (n_dst < 0 || fanout < 1 || fanout > 32 || dim < 1) return fail(COALA_EINVAL, "bad block shape (fan-out 1..32)");
    if (n_dst == 0) return COALA_OK;
    if (!nbr || !grad_out || !grad_src) return fail(COALA_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(device));
    hipLaunchKernelGGL(mean_aggregate_backward_kernel, dim3(grid1d(n_dst * 64, kBlock, 8192)), dim3(kBlock), 0, (hipStream_t)stream, nbr, grad_out,
                       grad_src, n_dst, fanout, dim);
    HIPCHK(hipGetLastError());
    return COALA_OK;
}

int coala_sampler_wait(coala_sampler_t* s, int64_t ticket, int64_t* n_src_host, int64_t* bucket_counts_host) {
    if (!s) return fail(COALA_EINVAL, "null sampler");
    if (ticket < 0 || (uint64_t)ticket >= s->calls || s->calls - (uint64_t)ticket > kRing)
        return fail(COALA_EINVAL, "ticket %lld is not one of the last %d calls", (long long)ticket, kRing);
    HIPCHK(hipSetDevice(s->device));
    const int slot = (int)((uint64_t)ticket % kRing);
    HIPCHK(hipEventSynchronize(s->done[slot])); // the kernels behind this event stored the counts into pinned host memory
    const int64_t* pin = s->counts_pinned + (size_t)slot * kSlot;
    if (n_src_host)
        for (int l = 0; l < s->ring_layers[slot]; ++l) n_src_host[l] = pin[l];
    if (bucket_counts_host)
        for (int g = 0; g < s->ring_parts[slot]; ++g) bucket_counts_host[g] = pin[COALA_SAMPLER_MAX_LAYERS + g];
    return COALA_OK;
}

int coala_sampler_sample(coala_sampler_t* s, const int64_t* seeds, int64_t n_seeds, const int32_t* fanouts, int n_layers,
                         uint64_t seed, uint64_t step, int64_t* const* src_nodes_out, int32_t* const* nbr_local_out,
                         int64_t* n_src_host, const coala_sampler_bucketing_t* bucketing, int64_t* ticket_out, void* stream) {
    if (!s || (!seeds && n_seeds > 0) || !fanouts || !src_nodes_out || !nbr_local_out) return fail(COALA_EINVAL, "null argument");
    if (n_layers < 1 || n_layers > COALA_SAMPLER_MAX_LAYERS) return fail(COALA_EINVAL, "n_layers must be 1..%d", COALA_SAMPLER_MAX_LAYERS);
    if (n_seeds < 0 || n_seeds > 0x7FFFFFFF) return fail(COALA_EINVAL, "bad n_seeds");
    const int n_parts = bucketing ? bucketing->n_parts : 0;
    if (n_parts < 0 || n_parts > kMaxParts) return fail(COALA_EINVAL, "bucketing: n_parts must be 0..%d", kMaxParts);
    if (n_parts > 0 && (!bucketing->bucketed_nodes || !bucketing->counts || !bucketing->dst_in_src)) return fail(COALA_EINVAL, "bucketing: null buffer");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(s->device));
    // the handle's scratch (hash table, scan state) is ordered by the stream of its calls: a caller that moves to another stream
    // first waits there for the previous call's last kernel
    if (s->calls > 0 && s->last_stream != st) HIPCHK(hipStreamWaitEvent(st, s->done[(s->calls - 1) % kRing], 0));
    s->last_stream = st;
    // capacities: layer l has at most cap_l dst nodes and cap_l*(f_l+1) source nodes
    int64_t cap = n_seeds;
    uint64_t max_items = 0, max_nbr = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int f = fanouts[l];
        if (f < 1 || f > 32) return fail(COALA_EINVAL, "fan-out %d outside 1..32", f);
        const uint64_t items = (uint64_t)cap * (uint64_t)(f + 1);
        if (items > (uint64_t)kMaxTiles * kTile) return fail(COALA_EINVAL, "layer %d would hold %llu items (limit %d)", l, (unsigned long long)items, kMaxTiles * kTile);
        if (items > max_items) max_items = items;
        if ((uint64_t)cap * f > max_nbr) max_nbr = (uint64_t)cap * f;
        cap = (int64_t)items;
    }
    const uint64_t table = table_size((int64_t)max_items);
    const uint64_t wave_tiles = ((uint64_t)cap + kRouteTile - 1) / kRouteTile;
    int rc;
    if ((rc = grow((void**)&s->nbr_global, &s->nbr_cap, max_nbr ? max_nbr : 1, sizeof(int64_t), st))) return rc;
    if ((rc = grow((void**)&s->slot_of_item, &s->item_cap, max_items ? max_items : 1, sizeof(uint32_t), st))) return rc;
    if (n_parts > 0) {
        if ((rc = grow((void**)&s->wave_counts, &s->wc_cap, (wave_tiles + 1) * (uint64_t)n_parts, sizeof(uint32_t), st))) return rc;
        if ((rc = grow((void**)&s->new_of_old, &s->noo_cap, (uint64_t)cap ? (uint64_t)cap : 1, sizeof(uint32_t), st))) return rc;
    }
    if (table > s->table_cap) {
        HIPCHK(hipStreamSynchronize(st));
        for (void** p : {(void**)&s->tb.keys, (void**)&s->tb.local_of_slot})
            if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
        HIPCHK(hipMalloc((void**)&s->tb.keys, table * (sizeof(long long) + sizeof(uint32_t)))); // keys, then the first-position words
        HIPCHK(hipMalloc((void**)&s->tb.local_of_slot, table * sizeof(uint32_t)));
        s->table_cap = table;
        s->tb.minpos = reinterpret_cast<uint32_t*>(s->tb.keys + table);
        s->clean_items = 0;
    }
    const uint64_t ticket = s->calls;
    const int slot = (int)(ticket % kRing);
    if (ticket >= kRing) HIPCHK(hipEventSynchronize(s->done[slot])); // the ring slot's previous user has finished writing it
    int64_t* pin = s->counts_pinned + (size_t)slot * kSlot;
    int64_t* pin_dev = s->counts_pinned_dev + (size_t)slot * kSlot;
    s->ring_layers[slot] = n_layers;
    s->ring_parts[slot] = n_parts;
    if (n_seeds == 0) {
        for (int l = 0; l < n_layers; ++l) pin[l] = 0;
        for (int g = 0; g < n_parts; ++g) pin[COALA_SAMPLER_MAX_LAYERS + g] = 0;
        if (n_parts > 0) HIPCHK(hipMemsetAsync(bucketing->counts, 0, (size_t)n_parts * sizeof(int64_t), st));
    } else {
        // the first layer's table: normally left clean by the previous call's last kernel
        const uint64_t items0 = (uint64_t)n_seeds * (uint64_t)(fanouts[0] + 1);
        if (s->clean_items == 0 || table_size((int64_t)items0) > table_size((int64_t)s->clean_items)) {
            const uint32_t t0 = table_size((int64_t)items0);
            HIPCHK(hipMemsetAsync(s->tb.keys, 0xFF, (size_t)t0 * sizeof(long long), st));
            HIPCHK(hipMemsetAsync(s->tb.minpos, 0xFF, (size_t)t0 * sizeof(uint32_t), st));
        }
        const int64_t* n_dst_dev = nullptr;            // first layer: the seed count travels as a kernel argument
        const int64_t* dst = seeds;
        cap = n_seeds;
        for (int l = 0; l < n_layers; ++l) {
            const int f = fanouts[l];
            const int64_t items_cap = cap * (f + 1);
            int64_t* n_src_dev = s->counts_dev + l + 1;
            const dim3 gs(grid1d(cap * (f < 16 ? 16 : f < 32 ? 32 : 64), kBlock, 8192)), blk(kBlock);
            if (f < 16)
                hipLaunchKernelGGL(sample_insert_kernel<16>, gs, blk, 0, st, s->g, dst, n_dst_dev, n_seeds, f, seed, step, l, s->nbr_global, s->tb, s->slot_of_item);
            else if (f < 32)
                hipLaunchKernelGGL(sample_insert_kernel<32>, gs, blk, 0, st, s->g, dst, n_dst_dev, n_seeds, f, seed, step, l, s->nbr_global, s->tb, s->slot_of_item);
            else
                hipLaunchKernelGGL(sample_insert_kernel<64>, gs, blk, 0, st, s->g, dst, n_dst_dev, n_seeds, f, seed, step, l, s->nbr_global, s->tb, s->slot_of_item);
            const int tiles = grid1d(items_cap, kTile, kMaxTiles);
            if ((++s->scan_gen & 0x3FFFFFFFull) == 0) { // 2^30 scans: the generation tag wraps -> clear the status words once
                HIPCHK(hipMemsetAsync(s->status, 0, kMaxTiles * sizeof(unsigned long long), st));
                s->scan_gen++;
            }
            hipLaunchKernelGGL(scan_assign_kernel, dim3(tiles), blk, 0, st, dst, s->nbr_global, n_dst_dev, n_seeds, f, s->slot_of_item, s->tb, s->status,
                               s->ticket, s->ticket_total, s->scan_gen & 0x3FFFFFFFull, src_nodes_out[l], n_src_dev, pin_dev + l);
            s->ticket_total += (unsigned long long)tiles;
            const bool last = l + 1 == n_layers;
            const int64_t clear_cap = last ? (int64_t)table_size((int64_t)items0) : (int64_t)table_size(items_cap * (fanouts[l + 1] + 1));
            hipLaunchKernelGGL(relabel_clear_kernel, dim3(grid1d(std::max<int64_t>(cap * f, clear_cap), kBlock, 4096)), blk, 0, st, n_dst_dev, n_seeds, f,
                               s->slot_of_item, s->tb, nbr_local_out[l], last ? (const int64_t*)nullptr : (const int64_t*)n_src_dev,
                               last ? 0 : fanouts[l + 1], (int64_t)items0);
            if (last && n_parts > 0) {
                const uint32_t P = (uint32_t)n_parts;
                const int pshift = ilog2_exact((uint64_t)n_parts);
                int64_t* bases = s->counts_dev + COALA_SAMPLER_MAX_LAYERS + 1;
                const dim3 gw(grid1d(((items_cap + kRouteTile - 1) / kRouteTile) * 64, kBlock, 4096));
                hipLaunchKernelGGL(bucket_count_kernel, gw, blk, 0, st, src_nodes_out[l], n_src_dev, P, pshift, s->wave_counts);
                hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(64 * (n_parts < 16 ? n_parts : 16)), 0, st, s->wave_counts, n_src_dev, P,
                                   bucketing->counts, pin_dev + COALA_SAMPLER_MAX_LAYERS, bases);
                hipLaunchKernelGGL(bucket_scatter_kernel, gw, blk, 0, st, src_nodes_out[l], n_src_dev, P, pshift, s->wave_counts, bases,
                                   bucketing->bucketed_nodes, s->new_of_old);
                hipLaunchKernelGGL(bucket_reindex_kernel, dim3(grid1d(cap * f, kBlock, 4096)), blk, 0, st, n_dst_dev, n_seeds, f, s->new_of_old,
                                   nbr_local_out[l], bucketing->dst_in_src);
            }
            dst = src_nodes_out[l];
            n_dst_dev = n_src_dev;
            cap = items_cap;
        }
        s->clean_items = items0;
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(s->done[slot], st));
    s->calls++;
    if (ticket_out) *ticket_out = (int64_t)ticket;
    if (n_src_host) return coala_sampler_wait(s, (int64_t)ticket, n_src_host, nullptr);
    return COALA_OK;
}

} // extern "C"
