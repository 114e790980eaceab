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
// ONE launch per call: a persistent kernel of at most one 256-thread block per CU walks through every phase of every layer
// (clear table -> sample + insert -> count first occurrences -> assign -> relabel, then the optional owner bucketing of the
// input nodes) with grid-wide barriers in between, layer l+1 reading its destination count from what layer l just produced.
// Round 1 issued 10 launches per 2-layer call (and a blocking read of the counts); the kernel boundaries, not the work, were
// the cost.  The counts are stored straight into pinned host memory by the kernel; the host waits for them on an event.
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
constexpr int kMaxTiles = 4096;             // tiles per layer the in-LDS scan of the tile sums can hold (4.19 M items)
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

struct SampleArgs {
    const int64_t* indptr;
    const int64_t* indices;
    int64_t num_nodes;
    const int64_t* seeds;
    int64_t n_seeds;
    int32_t n_layers;
    int32_t fanout[COALA_SAMPLER_MAX_LAYERS];
    int64_t* src_out[COALA_SAMPLER_MAX_LAYERS];
    int32_t* nbr_local_out[COALA_SAMPLER_MAX_LAYERS];
    uint64_t seed, step;
    // workspace
    int64_t* nbr;             // [max cap_l * f_l] sampled neighbours of the current layer (global ids, -1 padded)
    long long* keys;          // [table] hash keys
    uint32_t* minpos;         // [table] first position of the key in the (dst..., neighbours...) item list
    uint32_t* local_of_slot;  // [table] index of the key in the source list
    uint32_t* slot_of_item;   // [max items]
    uint32_t* tile_sums;      // [max(kMaxTiles, wave tiles * n_parts)]
    int64_t* counts_host;     // device alias of pinned host memory: [n_layers] source counts, then [n_parts] bucket sizes
    unsigned long long* barrier;  // monotonic arrival counter of the grid barrier
    unsigned long long barrier_base;
    int* error;               // device alias of a pinned host flag: 1 = a grid barrier timed out
    // optional: the input nodes of the LAST layer bucketed by owner = id % n_parts (stable), blocks re-indexed accordingly
    int32_t n_parts;
    int32_t pshift;           // log2(n_parts) or -1
    int64_t* bucketed;        // [cap_L]
    int64_t* bucket_counts;   // device [n_parts]
    int32_t* dst_in_src;      // [cap_{L-1}] position of dst d of the last block inside `bucketed`
    uint32_t* new_of_old;     // [cap_L] workspace
};

// Grid-wide barrier of a kernel whose blocks are all resident (grid <= one block per CU).  Bounded: if a block of the grid never
// arrives (it faulted), the others give up after ~2 s, raise the error flag and leave -- the grid always drains.
__device__ __forceinline__ bool grid_sync(const SampleArgs& a, unsigned long long& target, int* s_ok) {
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
        __threadfence(); // this block's writes are visible device-wide before it reports in
        atomicAdd(a.barrier, 1ull);
        int ok = 1;
        long spins = 0;
        while (__hip_atomic_load(a.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 0xFFF) == 0 && (spins > (1L << 22) || __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))) {
                __hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
        }
        __threadfence();
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

__device__ __forceinline__ void clear_table(const SampleArgs& a, uint32_t tbl) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < tbl; i += gridDim.x * kBlock) {
        a.keys[i] = kEmpty;
        a.minpos[i] = 0xFFFFFFFFu;
    }
}

// Sample + insert, a lane per sampled neighbour: GS lanes (16/32/64 >= fanout+1) work on one destination node.  Lane c < fanout
// draws Floyd's c-th candidate on its own, the duplicate resolution walks c = 0..fanout-1 with one shuffle + one ballot per
// step (bit-identical to the sequential loop of the CPU twin), then every lane loads ITS neighbour and inserts it into the
// hash table; lane `fanout` inserts the destination node itself.
template <int GS>
__device__ __forceinline__ void phase_sample_insert(const SampleArgs& a, const int64_t* __restrict__ dst, int64_t n_dst, int fanout, int layer,
                                                    uint32_t mask) {
    constexpr int GPW = 64 / GS; // groups per wave
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
        const bool okv = active && v >= 0 && v < a.num_nodes;
        const int64_t start = okv ? a.indptr[v] : 0;
        const int64_t deg = okv ? a.indptr[v + 1] - start : 0;
        // candidate of lane c = gl (Floyd step j = deg - fanout + c)
        const uint64_t key = sample_key(a.seed, a.step, layer, (uint64_t)v);
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
        const int64_t nb = (okv && pick >= 0) ? a.indices[start + pick] : kEmpty;
        if (active && gl < fanout) a.nbr[d * fanout + gl] = nb;
        // ---- hash insert: neighbours at positions n_dst + d*fanout + gl, the node itself at position d
        int64_t k = kEmpty;
        int64_t p = -1;
        if (active && gl < fanout) { k = nb; p = n_dst + d * fanout + gl; }
        else if (active && gl == fanout) { k = v; p = d; }
        if (p >= 0) {
            if (k < 0) {
                a.slot_of_item[p] = 0xFFFFFFFFu;
            } else {
                uint32_t s = hash_slot(k, mask);
                while (true) {
                    const long long cur = a.keys[s];
                    if (cur == k) break;
                    if (cur == kEmpty) {
                        const long long old = atomicCAS((unsigned long long*)(a.keys + s), (unsigned long long)kEmpty, (unsigned long long)k);
                        if (old == kEmpty || old == k) break;
                    }
                    s = (s + 1) & mask;
                }
                atomicMin(a.minpos + s, (uint32_t)p);
                a.slot_of_item[p] = s;
            }
        }
    }
}

__device__ __forceinline__ uint32_t first_flag(const SampleArgs& a, int64_t p, int64_t n_items) {
    if (p >= n_items) return 0;
    const uint32_t s = a.slot_of_item[p];
    return (s != 0xFFFFFFFFu && a.minpos[s] == (uint32_t)p) ? 1u : 0u;
}

// exclusive scan of s_vals[0..n) in place (n <= kMaxTiles), every thread of the block; returns the total
__device__ __forceinline__ uint32_t block_scan_inplace(uint32_t* s_vals, int n, uint32_t* s_part) {
    const int per = (n + kBlock - 1) / kBlock;
    const int lo = threadIdx.x * per, hi = min(lo + per, n);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += s_vals[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        const uint32_t v = (threadIdx.x >= (unsigned)off) ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) {
        const uint32_t c = s_vals[i];
        s_vals[i] = run;
        run += c;
    }
    const uint32_t total = s_part[kBlock - 1];
    __syncthreads();
    return total;
}

__device__ __forceinline__ uint32_t owner_of(uint64_t id, uint32_t n_parts, int pshift) {
    if (pshift >= 0) return (uint32_t)id & (n_parts - 1);
    if ((id >> 32) == 0) return (uint32_t)id % n_parts;
    return (uint32_t)(id % n_parts);
}

__global__ __launch_bounds__(kBlock) void sample_layers_kernel(SampleArgs a) {
    __shared__ uint32_t s_tiles[kMaxTiles];
    __shared__ uint32_t s_part[kBlock];
    __shared__ uint32_t s_woff[kWavesPerBlock];
    __shared__ int s_ok;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long target = a.barrier_base;
    const int64_t* dst = a.seeds;
    int64_t n_dst = a.n_seeds;
    clear_table(a, table_size(n_dst * (a.fanout[0] + 1)));
    if (!grid_sync(a, target, &s_ok)) return;
    for (int l = 0; l < a.n_layers; ++l) {
        const int f = a.fanout[l];
        const int64_t n_items = n_dst * (f + 1);
        const uint32_t mask = table_size(n_items) - 1;
        // ---- sample + insert
        if (f < 16) phase_sample_insert<16>(a, dst, n_dst, f, l, mask);
        else if (f < 32) phase_sample_insert<32>(a, dst, n_dst, f, l, mask);
        else phase_sample_insert<64>(a, dst, n_dst, f, l, mask);
        if (!grid_sync(a, target, &s_ok)) return;
        // ---- count first occurrences per tile
        const int n_tiles = (int)((n_items + kTile - 1) / kTile);
        for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            const int64_t base = (int64_t)tile * kTile + (int64_t)threadIdx.x * kItems;
            uint32_t c = 0;
            for (int i = 0; i < kItems; ++i) c += first_flag(a, base + i, n_items);
            for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
            if (lane == 0) s_woff[w] = c;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t t = 0;
                for (int q = 0; q < kWavesPerBlock; ++q) t += s_woff[q];
                a.tile_sums[tile] = t;
            }
            __syncthreads();
        }
        if (!grid_sync(a, target, &s_ok)) return;
        // ---- assign: every block scans the tile sums in LDS, then numbers the first occurrences of its own tiles
        for (int i = threadIdx.x; i < n_tiles; i += kBlock) s_tiles[i] = a.tile_sums[i];
        __syncthreads();
        const int64_t n_src = (int64_t)block_scan_inplace(s_tiles, n_tiles, s_part);
        int64_t* __restrict__ src = a.src_out[l];
        for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            const int64_t base = (int64_t)tile * kTile + (int64_t)threadIdx.x * kItems;
            uint32_t fl[kItems];
            uint32_t c = 0;
            for (int i = 0; i < kItems; ++i) { fl[i] = first_flag(a, base + i, n_items); c += fl[i]; }
            uint32_t incl = c; // inclusive scan inside the wave
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if (lane == 63) s_woff[w] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (int q = 0; q < w; ++q) wbase += s_woff[q];
            uint32_t run = s_tiles[tile] + wbase + incl - c;
            for (int i = 0; i < kItems; ++i) {
                if (fl[i]) {
                    const int64_t p = base + i;
                    src[run] = item_key(dst, a.nbr, n_dst, p);
                    a.local_of_slot[a.slot_of_item[p]] = run;
                    ++run;
                }
            }
            __syncthreads();
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) a.counts_host[l] = n_src;
        if (!grid_sync(a, target, &s_ok)) return;
        // ---- relabel: neighbour -> index in the source list; the table of the next layer is cleared alongside (keys / minpos are
        //      no longer read in this layer)
        {
            const int64_t n_nbr = n_dst * f;
            int32_t* __restrict__ loc = a.nbr_local_out[l];
            for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n_nbr; q += (int64_t)gridDim.x * kBlock) {
                const uint32_t s = a.slot_of_item[n_dst + q];
                loc[q] = (s == 0xFFFFFFFFu) ? -1 : (int32_t)a.local_of_slot[s];
            }
            if (l + 1 < a.n_layers) clear_table(a, table_size(n_src * (a.fanout[l + 1] + 1)));
        }
        const bool last = l + 1 == a.n_layers;
        if (last && a.n_parts <= 0) return; // nothing after the last relabel reads what other blocks wrote
        if (!grid_sync(a, target, &s_ok)) return;
        if (last) {
            // ---- owner bucketing of the input nodes (stable inside each bucket): per wave tile of 256 ids, per-owner counts by
            //      ballots -> per-owner scan over the wave tiles -> scatter -> re-index the last block
            const uint32_t P = (uint32_t)a.n_parts;
            const int64_t n_wt = (n_src + kRouteTile - 1) / kRouteTile;
            const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + w;
            const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
            for (int64_t wt = wave; wt < n_wt; wt += n_waves) {
                uint32_t mine = 0; // lane g accumulates the count of owner g
                for (int j = 0; j < kItems; ++j) {
                    const int64_t i = wt * kRouteTile + j * 64 + lane;
                    const uint32_t o = (i < n_src) ? owner_of((uint64_t)src[i], P, a.pshift) : 0xFFFFFFFFu;
                    for (uint32_t g = 0; g < P; ++g) {
                        const uint64_t m = __ballot(o == g);
                        if ((uint32_t)lane == g) mine += (uint32_t)__builtin_popcountll(m);
                    }
                }
                if ((uint32_t)lane < P) a.tile_sums[wt * P + lane] = mine;
            }
            if (!grid_sync(a, target, &s_ok)) return;
            // one wave per owner column: exclusive scan over the wave tiles with a running carry; lane 0 publishes the bucket size
            for (uint32_t g = (uint32_t)wave; g < P; g += (uint32_t)n_waves) {
                uint32_t carry = 0;
                for (int64_t t0 = 0; t0 < n_wt; t0 += 64) {
                    const int64_t t = t0 + lane;
                    const uint32_t c = (t < n_wt) ? a.tile_sums[t * P + g] : 0u;
                    uint32_t incl = c;
                    for (int off = 1; off < 64; off <<= 1) {
                        const uint32_t v = __shfl_up(incl, off);
                        if (lane >= off) incl += v;
                    }
                    if (t < n_wt) a.tile_sums[t * P + g] = carry + incl - c;
                    carry += __shfl(incl, 63);
                }
                if (lane == 0) {
                    a.bucket_counts[g] = (int64_t)carry;
                    a.counts_host[a.n_layers + g] = (int64_t)carry;
                }
            }
            if (!grid_sync(a, target, &s_ok)) return;
            for (int64_t wt = wave; wt < n_wt; wt += n_waves) {
                int64_t off = 0; // lane g: next free slot of bucket g for this wave tile
                if ((uint32_t)lane < P) {
                    int64_t b = 0;
                    for (uint32_t g = 0; g < (uint32_t)lane; ++g) b += a.bucket_counts[g];
                    off = b + (int64_t)a.tile_sums[wt * P + lane];
                }
                for (int j = 0; j < kItems; ++j) {
                    const int64_t i = wt * kRouteTile + j * 64 + lane;
                    const bool valid = i < n_src;
                    const int64_t id = valid ? src[i] : 0;
                    const uint32_t o = valid ? owner_of((uint64_t)id, P, a.pshift) : 0xFFFFFFFFu;
                    int64_t dest = -1;
                    for (uint32_t g = 0; g < P; ++g) {
                        const uint64_t m = __ballot(o == g);
                        const int64_t bg = __shfl(off, (int)g);
                        if (o == g) dest = bg + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                        if ((uint32_t)lane == g) off += __builtin_popcountll(m);
                    }
                    if (valid) {
                        a.bucketed[dest] = id;
                        a.new_of_old[i] = (uint32_t)dest;
                    }
                }
            }
            if (!grid_sync(a, target, &s_ok)) return;
            const int64_t n_nbr = n_dst * f;
            int32_t* __restrict__ loc = a.nbr_local_out[l];
            for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n_nbr; q += (int64_t)gridDim.x * kBlock) {
                const int32_t o = loc[q];
                if (o >= 0) loc[q] = (int32_t)a.new_of_old[o];
            }
            for (int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x; d < n_dst; d += (int64_t)gridDim.x * kBlock)
                a.dst_in_src[d] = (int32_t)a.new_of_old[d]; // the dst nodes are the first n_dst entries of the unbucketed list
            return;
        }
        dst = src;
        n_dst = n_src;
    }
}

} // namespace

struct coala_sampler {
    int device = 0;
    const int64_t* indptr = nullptr;
    const int64_t* indices = nullptr;
    int64_t num_nodes = 0, num_edges = 0;
    int max_grid = 0;                 // blocks of the persistent kernel: at most one per CU
    // workspace, grown on demand
    int64_t* nbr_global = nullptr;    uint64_t nbr_cap = 0;
    long long* keys = nullptr;        uint32_t* minpos = nullptr; uint32_t* local_of_slot = nullptr; uint64_t table_cap = 0;
    uint32_t* slot_of_item = nullptr; uint64_t item_cap = 0;
    uint32_t* tile_sums = nullptr;    uint64_t tile_cap = 0;
    uint32_t* new_of_old = nullptr;   uint64_t noo_cap = 0;
    unsigned long long* barrier = nullptr; // device counter, monotonic across calls
    unsigned long long barrier_total = 0;
    // pinned host ring: per call [kMaxLayers source counts][kMaxParts bucket sizes], an error flag, and an event recorded behind the kernel
    int64_t* counts_pinned = nullptr; // host pointer
    int64_t* counts_pinned_dev = nullptr;
    int* error_pinned = nullptr;
    int* error_pinned_dev = nullptr;
    hipEvent_t done[kRing] = {};
    int ring_layers[kRing] = {};
    int ring_parts[kRing] = {};
    uint64_t calls = 0;
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
    s->indptr = indptr;
    s->indices = indices;
    s->num_nodes = num_nodes;
    s->num_edges = num_edges;
    hipDeviceProp_t prop;
    int per_cu = 0;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sample_layers_kernel, kBlock, 0) != hipSuccess || per_cu < 1) {
        delete s;
        return fail(COALA_EHIP, "cannot size the sampler's persistent kernel: %s", hipGetErrorString(hipGetLastError()));
    }
    // One block per CU at most: every block of the grid is resident at once (the grid barriers rely on it) with room to spare for
    // whatever else runs on the GPU -- including the persistent kernel of another sampler (another process on the same GPU).
    s->max_grid = prop.multiProcessorCount;
    bool ok = hipMalloc((void**)&s->barrier, sizeof(unsigned long long)) == hipSuccess &&
              hipMemset(s->barrier, 0, sizeof(unsigned long long)) == hipSuccess &&
              hipHostMalloc((void**)&s->counts_pinned, kRing * kSlot * sizeof(int64_t), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void**)&s->counts_pinned_dev, s->counts_pinned, 0) == hipSuccess &&
              hipHostMalloc((void**)&s->error_pinned, sizeof(int), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void**)&s->error_pinned_dev, s->error_pinned, 0) == hipSuccess;
    if (ok) {
        *s->error_pinned = 0;
        for (int i = 0; i < kRing && ok; ++i) ok = hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming) == hipSuccess;
    }
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
    void* ptrs[] = {s->nbr_global, s->keys, s->local_of_slot, s->slot_of_item, s->tile_sums, s->new_of_old, s->barrier};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->counts_pinned) (void)hipHostFree(s->counts_pinned);
    if (s->error_pinned) (void)hipHostFree(s->error_pinned);
    for (int i = 0; i < kRing; ++i)
        if (s->done[i]) (void)hipEventDestroy(s->done[i]);
    delete s;
    return COALA_OK;
}

int coala_sampler_wait(coala_sampler_t* s, int64_t ticket, int64_t* n_src_host, int64_t* bucket_counts_host) {
    if (!s) return fail(COALA_EINVAL, "null sampler");
    if (ticket < 0 || (uint64_t)ticket >= s->calls || s->calls - (uint64_t)ticket > kRing)
        return fail(COALA_EINVAL, "ticket %lld is not one of the last %d calls", (long long)ticket, kRing);
    HIPCHK(hipSetDevice(s->device));
    const int slot = (int)((uint64_t)ticket % kRing);
    HIPCHK(hipEventSynchronize(s->done[slot])); // the kernel behind this event stored the counts into pinned host memory
    if (*s->error_pinned) return fail(COALA_EHIP, "the sampler kernel gave up at a grid barrier (a block of its grid never arrived)");
    const int64_t* pin = s->counts_pinned + (size_t)slot * kSlot;
    if (n_src_host)
        for (int l = 0; l < s->ring_layers[slot]; ++l) n_src_host[l] = pin[l];
    if (bucket_counts_host)
        for (int g = 0; g < s->ring_parts[slot]; ++g) bucket_counts_host[g] = pin[s->ring_layers[slot] + g];
    return COALA_OK;
}

int coala_sampler_sample(coala_sampler_t* s, const int64_t* seeds, int64_t n_seeds, const int32_t* fanouts, int n_layers,
                         uint64_t seed, uint64_t step, int64_t* const* src_nodes_out, int32_t* const* nbr_local_out,
                         int64_t* n_src_host, const coala_sampler_bucketing_t* bucketing, int64_t* ticket_out, void* stream) {
    if (!s || !seeds || !fanouts || !src_nodes_out || !nbr_local_out) return fail(COALA_EINVAL, "null argument");
    if (n_layers < 1 || n_layers > COALA_SAMPLER_MAX_LAYERS) return fail(COALA_EINVAL, "n_layers must be 1..%d", COALA_SAMPLER_MAX_LAYERS);
    if (n_seeds < 0 || n_seeds > 0x7FFFFFFF) return fail(COALA_EINVAL, "bad n_seeds");
    const int n_parts = bucketing ? bucketing->n_parts : 0;
    if (n_parts < 0 || n_parts > kMaxParts) return fail(COALA_EINVAL, "bucketing: n_parts must be 0..%d", kMaxParts);
    if (n_parts > 0 && (!bucketing->bucketed_nodes || !bucketing->counts || !bucketing->dst_in_src)) return fail(COALA_EINVAL, "bucketing: null buffer");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(s->device));
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
    if ((rc = grow((void**)&s->tile_sums, &s->tile_cap, std::max<uint64_t>(kMaxTiles, (wave_tiles + 1) * (uint64_t)(n_parts > 0 ? n_parts : 1)), sizeof(uint32_t), st))) return rc;
    if (n_parts > 0 && (rc = grow((void**)&s->new_of_old, &s->noo_cap, (uint64_t)cap ? (uint64_t)cap : 1, sizeof(uint32_t), st))) return rc;
    if (table > s->table_cap) {
        HIPCHK(hipStreamSynchronize(st));
        for (void** p : {(void**)&s->keys, (void**)&s->local_of_slot})
            if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
        HIPCHK(hipMalloc((void**)&s->keys, table * (sizeof(long long) + sizeof(uint32_t)))); // keys + first-position words
        HIPCHK(hipMalloc((void**)&s->local_of_slot, table * sizeof(uint32_t)));
        s->table_cap = table;
        s->minpos = reinterpret_cast<uint32_t*>(s->keys + table);
    }
    const uint64_t ticket = s->calls;
    const int slot = (int)(ticket % kRing);
    if (ticket >= kRing) HIPCHK(hipEventSynchronize(s->done[slot])); // the ring slot's previous user has finished writing it
    int64_t* pin = s->counts_pinned + (size_t)slot * kSlot;
    s->ring_layers[slot] = n_layers;
    s->ring_parts[slot] = n_parts;
    if (n_seeds == 0) {
        for (int l = 0; l < n_layers; ++l) pin[l] = 0;
        for (int g = 0; g < n_parts; ++g) pin[n_layers + g] = 0;
        if (n_parts > 0) HIPCHK(hipMemsetAsync(bucketing->counts, 0, (size_t)n_parts * sizeof(int64_t), st));
    } else {
        SampleArgs a{};
        a.indptr = s->indptr;
        a.indices = s->indices;
        a.num_nodes = s->num_nodes;
        a.seeds = seeds;
        a.n_seeds = n_seeds;
        a.n_layers = n_layers;
        for (int l = 0; l < n_layers; ++l) {
            a.fanout[l] = fanouts[l];
            a.src_out[l] = src_nodes_out[l];
            a.nbr_local_out[l] = nbr_local_out[l];
        }
        a.seed = seed;
        a.step = step;
        a.nbr = s->nbr_global;
        a.keys = s->keys;
        a.minpos = s->minpos;
        a.local_of_slot = s->local_of_slot;
        a.slot_of_item = s->slot_of_item;
        a.tile_sums = s->tile_sums;
        a.counts_host = s->counts_pinned_dev + (size_t)slot * kSlot;
        a.barrier = s->barrier;
        a.barrier_base = s->barrier_total;
        a.error = s->error_pinned_dev;
        a.n_parts = n_parts;
        a.pshift = n_parts > 0 ? ilog2_exact((uint64_t)n_parts) : -1;
        if (n_parts > 0) {
            a.bucketed = bucketing->bucketed_nodes;
            a.bucket_counts = bucketing->counts;
            a.dst_in_src = bucketing->dst_in_src;
            a.new_of_old = s->new_of_old;
        }
        // grid: enough waves for the widest sampling phase (16 lanes per destination node), at most one block per CU
        const int64_t want = (cap / (int64_t)(fanouts[n_layers - 1] + 1) * 16 + kBlock * 2 - 1) / (kBlock * 2);
        const int grid = (int)std::max<int64_t>(16, std::min<int64_t>(s->max_grid, want));
        const int barriers = 1 + 4 * n_layers - (n_parts > 0 ? 0 : 1) + (n_parts > 0 ? 3 : 0);
        s->barrier_total += (unsigned long long)grid * (unsigned long long)barriers;
        hipLaunchKernelGGL(sample_layers_kernel, dim3(grid), dim3(kBlock), 0, st, a);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(s->done[slot], st));
    s->calls++;
    if (ticket_out) *ticket_out = (int64_t)ticket;
    if (n_src_host) return coala_sampler_wait(s, (int64_t)ticket, n_src_host, nullptr);
    return COALA_OK;
}

} // extern "C"
