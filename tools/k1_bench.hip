// tools/k1_bench.hip -- A/B harness for the probe+gather kernel (development tool, not part of the product or tests).
// Includes the product source so that the kernels in its anonymous namespace can be launched with other grids and
// compared with experimental variants in one process (interleaved timing, hipEvents).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/k1_bench.hip coala-gnn_amd/csrc/coala_host.cpp -o tools/k1_bench -lrt
//   tools/k1_bench [rows_in_table] [n] [dim] [hit_percent]
#include "../coala-gnn_amd/csrc/coala_cache.hip"

#include <algorithm>
#include <functional>
#include <numeric>
#include <random>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

namespace {

// ---------------------------------------------------------------- experimental variant: software-pipelined, optional nt
template <int CD, bool NT, int PASSES_, int EXTRA = 0 /* 1: valid/ok/bad ballots + bad zero rows  2: + miss push and row_state */>
__global__ __launch_bounds__(1024) void probe_gather_v2(CacheDev c, const int64_t* __restrict__ idx, float* __restrict__ out,
                                                        int64_t n, uint32_t gen) {
    // all-hit fast path only (no miss bookkeeping): measures what the data movement alone can reach
    constexpr int R = PASSES_;
    using V = vfloat4;
    constexpr int VPL = CD / 4 / 64;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t n_chunks = (n + R - 1) / R;
    int64_t chunk = wave;
    if (chunk >= n_chunks) return;
    // prologue: ids + tags of the first chunk
    auto load_id = [&](int64_t ch) -> uint64_t {
        const int64_t i_l = ch * R + (lane >> 4);
        return ((lane >> 4) < R && i_l < n) ? (uint64_t)idx[i_l] : 0xFFFFFFFFFFFFFFFFull;
    };
    uint64_t id = load_id(chunk);
    uint64_t id_next = (chunk + n_waves < n_chunks) ? load_id(chunk + n_waves) : 0xFFFFFFFFFFFFFFFFull;
    bool ok = id < c.num_rows;
    uint64_t set = ok ? set_of(c, id) : 0;
    vu64x2 kk = {kEmptyKey, kEmptyKey};
    if (ok) kk = *reinterpret_cast<const vu64x2*>(c.keys + set * COALA_WAYS + (lane & 15) * 2);
    for (; chunk < n_chunks; chunk += n_waves) {
        const int64_t base = chunk * R;
        const uint64_t m0 = __ballot(ok && kk.x == id);
        const uint64_t m1 = __ballot(ok && kk.y == id);
        const uint64_t okm = EXTRA ? __ballot(ok) : 0;
        const uint64_t vm = EXTRA ? __ballot(id != 0xFFFFFFFFFFFFFFFFull) : 0;
        uint32_t slot[R];
        uint32_t hitmask = 0, missmask = 0, badmask = 0;
        uint64_t my_set = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const uint32_t a = (uint32_t)(m0 >> (16 * q)) & 0xFFFFu, b = (uint32_t)(m1 >> (16 * q)) & 0xFFFFu, mm = a | b;
            const uint64_t set_q = readlane64(set, 16 * q);
            uint32_t way = 0;
            if (mm) { const int j = __builtin_ctz(mm); way = 2 * j + (((a >> j) & 1) ? 0 : 1); hitmask |= 1u << q; }
            else if (EXTRA && ((okm >> (16 * q)) & 1)) missmask |= 1u << q;
            else if (EXTRA && ((vm >> (16 * q)) & 1)) badmask |= 1u << q;
            slot[q] = (uint32_t)(set_q * COALA_WAYS) + way;
            if (EXTRA >= 2 && lane == q) my_set = set_q;
        }
        const bool i_miss = EXTRA >= 2 && lane < R && ((missmask >> lane) & 1);
        unsigned long long prev = 0;
        if (i_miss) {
            const unsigned long long tag = ((unsigned long long)gen << 32) | (unsigned long long)(base + lane + 1);
            prev = atomicExch(reinterpret_cast<unsigned long long*>(c.set_head + my_set), tag);
            atomicAdd(c.set_cnt + my_set, 1u);
        }
        // id two chunks ahead
        const int64_t ch2 = chunk + 2 * n_waves;
        uint64_t id_next2 = (ch2 < n_chunks) ? load_id(ch2) : 0xFFFFFFFFFFFFFFFFull;
        V val[R][VPL];
#pragma unroll
        for (int p = 0; p < R; ++p) {
            const V* src = reinterpret_cast<const V*>(c.lines + (uint64_t)slot[p] * CD);
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if ((hitmask >> p) & 1) val[p][v] = NT ? __builtin_nontemporal_load(src + v * 64 + lane) : src[v * 64 + lane];
        }
        // tags of the next chunk (its id was requested one iteration ago)
        id = id_next;
        id_next = id_next2;
        ok = id < c.num_rows;
        set = ok ? set_of(c, id) : 0;
        kk = vu64x2{kEmptyKey, kEmptyKey};
        if (ok) kk = *reinterpret_cast<const vu64x2*>(c.keys + set * COALA_WAYS + (lane & 15) * 2);
#pragma unroll
        for (int p = 0; p < R; ++p) {
            V* dst = reinterpret_cast<V*>(out + (base + p) * (int64_t)c.dim);
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if ((hitmask >> p) & 1) {
                    if (NT) __builtin_nontemporal_store(val[p][v], dst + v * 64 + lane);
                    else dst[v * 64 + lane] = val[p][v];
                } else if (EXTRA && ((badmask >> p) & 1)) dst[v * 64 + lane] = V(0.0f);
        }
        if (EXTRA >= 2) {
            if (i_miss) c.miss_link[base + lane] = kLinkMiss | (((uint32_t)(prev >> 32) == gen) ? (uint32_t)prev : 0u);
            else if (lane < R && ((badmask >> lane) & 1)) c.miss_link[base + lane] = kLinkBad;
        }
    }
}

// ---------------------------------------------------------------- experimental variant: rows staged through LDS by LDS-DMA
// (global_load_lds: per-lane source address, wave-contiguous LDS destination), then ds_read_b128 -> global_store.
template <int ROWS, bool NT>
__global__ __launch_bounds__(128) void probe_gather_lds(CacheDev c, const int64_t* __restrict__ idx, float* __restrict__ out,
                                                        int64_t n, uint32_t gen) {
    constexpr int CD = 1024;
    __shared__ __attribute__((aligned(16))) vfloat4 stage[2][ROWS][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 2 + w;
    const int64_t n_waves = (int64_t)gridDim.x * 2;
    const int64_t n_chunks = (n + ROWS - 1) / ROWS;
    auto load_id = [&](int64_t ch) -> uint64_t {
        const int64_t i_l = ch * ROWS + (lane >> 4);
        return (ch < n_chunks && (lane >> 4) < ROWS && i_l < n) ? (uint64_t)idx[i_l] : 0xFFFFFFFFFFFFFFFFull;
    };
    int64_t chunk = wave;
    uint64_t id = load_id(chunk), id_next = load_id(chunk + n_waves);
    bool ok = id < c.num_rows;
    uint64_t set = ok ? set_of(c, id) : 0;
    vu64x2 kk = {kEmptyKey, kEmptyKey};
    if (ok) kk = *reinterpret_cast<const vu64x2*>(c.keys + set * COALA_WAYS + (lane & 15) * 2);
    for (; chunk < n_chunks; chunk += n_waves) {
        const int64_t base = chunk * ROWS;
        const uint64_t m0 = __ballot(ok && kk.x == id), m1 = __ballot(ok && kk.y == id);
        uint32_t slot[ROWS];
        uint32_t hitmask = 0;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const uint32_t a = (uint32_t)(m0 >> (16 * q)) & 0xFFFFu, b = (uint32_t)(m1 >> (16 * q)) & 0xFFFFu, mm = a | b;
            const uint64_t set_q = readlane64(set, 16 * q);
            uint32_t way = 0;
            if (mm) { const int j = __builtin_ctz(mm); way = 2 * j + (((a >> j) & 1) ? 0 : 1); hitmask |= 1u << q; }
            slot[q] = (uint32_t)(set_q * COALA_WAYS) + way;
        }
        const uint64_t id_next2 = load_id(chunk + 2 * n_waves);
#pragma unroll
        for (int p = 0; p < ROWS; ++p) {
            const char* src = reinterpret_cast<const char*>(c.lines + (uint64_t)slot[p] * CD);
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if ((hitmask >> p) & 1)
                    __builtin_amdgcn_global_load_lds((const void*)(src + (v * 64 + lane) * 16),
                                                     (void __attribute__((address_space(3)))*)&stage[w][p][v * 64], 16, 0, NT ? 2 : 0);
        }
        id = id_next; id_next = id_next2;
        ok = id < c.num_rows;
        set = ok ? set_of(c, id) : 0;
        kk = vu64x2{kEmptyKey, kEmptyKey};
        if (ok) kk = *reinterpret_cast<const vu64x2*>(c.keys + set * COALA_WAYS + (lane & 15) * 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // LDS-DMA landed (also the next tags)
#pragma unroll
        for (int p = 0; p < ROWS; ++p) {
            vfloat4* dst = reinterpret_cast<vfloat4*>(out + (base + p) * (int64_t)c.dim);
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if ((hitmask >> p) & 1) {
                    const vfloat4 x = stage[w][p][v * 64 + lane];
                    if (NT) __builtin_nontemporal_store(x, dst + v * 64 + lane);
                    else dst[v * 64 + lane] = x;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS reads done before the next chunk's DMA overwrites the slice
    }
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
};

} // namespace

#include <functional>

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? strtoull(argv[1], 0, 10) : 2000000ull;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 36864;
    const int dim = argc > 3 ? atoi(argv[3]) : 1024;
    const int hit_pct = argc > 4 ? atoi(argv[4]) : 100;
    CK(hipSetDevice(0));
    float* cold = nullptr;
    CK(hipMalloc(&cold, rows * (uint64_t)dim * 4));
    CK(hipMemset(cold, 0x3c, rows * (uint64_t)dim * 4));
    coala_cache_config_t cfg{};
    cfg.device = 0; cfg.dim = dim; cfg.cache_mb = 4096; cfg.n_gpus = 1; cfg.rank = 0; cfg.flags = COALA_FLAG_SYNC;
    cfg.cold_table = cold; cfg.num_rows = rows; cfg.max_batch = (uint64_t)n;
    coala_cache_t* h = nullptr;
    if (coala_cache_create(&cfg, &h)) { printf("create failed: %s\n", coala_last_error()); return 1; }
    std::vector<int64_t> perm(rows);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937_64 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int64_t> warm(perm.begin(), perm.begin() + n);
    std::vector<int64_t> ids = warm;
    const int64_t n_hit = n * hit_pct / 100;
    for (int64_t i = n_hit; i < n; ++i) ids[i] = perm[n + i]; // never cached
    std::shuffle(ids.begin(), ids.end(), rng);
    int64_t *d_warm, *d_ids;
    float* out;
    CK(hipMalloc(&d_warm, n * 8)); CK(hipMalloc(&d_ids, n * 8)); CK(hipMalloc(&out, (uint64_t)n * dim * 4));
    CK(hipMemcpy(d_warm, warm.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ids, ids.data(), n * 8, hipMemcpyHostToDevice));
    if (coala_cache_read_feature(h, out, d_warm, n, nullptr)) { printf("warm failed: %s\n", coala_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const double alg_bytes = (double)n * 264.0 + (double)n_hit * 2.0 * dim * 4.0;

    std::vector<Variant> vs;
    static uint32_t fake_gen = 100;
    auto add_prod = [&](int grid, int block, int np = 4) {
        vs.push_back({"prod P" + std::to_string(np) + " g" + std::to_string(grid) + " b" + std::to_string(block), [=](hipStream_t s) {
            // K1 alone: with misses it leaves marks/chains behind that K2 would consume; harmless for timing (fresh generation)
            ++fake_gen;
            if (dim == 1024 && np == 4) hipLaunchKernelGGL((probe_gather_kernel<1024, 4, 4, false, 0>), dim3(grid), dim3(block), 0, s, d_ids, out, n, fake_gen, (uint32_t)grid, h->d, Redirect{0, 0, nullptr, nullptr});
            else if (dim == 1024 && np == 6) hipLaunchKernelGGL((probe_gather_kernel<1024, 4, 4, true, 1>), dim3(grid), dim3(block), 0, s, d_ids, out, n, fake_gen, (uint32_t)grid, h->d, Redirect{0, 0, nullptr, nullptr});
            else if (dim == 1024 && np == 5) hipLaunchKernelGGL((probe_gather_kernel<1024, 4, 4, true>), dim3(grid), dim3(block), 0, s, d_ids, out, n, fake_gen, (uint32_t)grid, h->d, Redirect{0, 0, nullptr, nullptr});
            else if (dim == 128 && np == 5) hipLaunchKernelGGL((probe_gather_kernel<128, 4, 4, true>), dim3(grid), dim3(block), 0, s, d_ids, out, n, fake_gen, (uint32_t)grid, h->d, Redirect{0, 0, nullptr, nullptr});
            else if (dim == 1024 && np == 2) hipLaunchKernelGGL((probe_gather_kernel<1024, 4, 2>), dim3(grid), dim3(block), 0, s, d_ids, out, n, fake_gen, (uint32_t)grid, h->d, Redirect{0, 0, nullptr, nullptr});
            else if (dim == 128 && np == 4) hipLaunchKernelGGL((probe_gather_kernel<128, 4, 4>), dim3(grid), dim3(block), 0, s, d_ids, out, n, fake_gen, (uint32_t)grid, h->d, Redirect{0, 0, nullptr, nullptr});
            else if (dim == 128 && np == 2) hipLaunchKernelGGL((probe_gather_kernel<128, 4, 2>), dim3(grid), dim3(block), 0, s, d_ids, out, n, fake_gen, (uint32_t)grid, h->d, Redirect{0, 0, nullptr, nullptr});
        }});
    };
    add_prod(2048, 128, 4); add_prod(2048, 128, 5); add_prod(4096, 128, 5); add_prod(8192, 128, 5); add_prod(16384, 128, 5); add_prod(2048, 128, 6); add_prod(8192, 128, 6);
    if (hit_pct == 100) {
        if (dim == 1024) {
            auto add_v2 = [&](const char* nm, auto kern, int grid, int block) {
                vs.push_back({std::string(nm) + " g" + std::to_string(grid) + " b" + std::to_string(block),
                              [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, s, h->d, d_ids, out, n, 1u); }});
            };
            for (int g : {2048, 8192}) {
                add_v2("v3 lds-dma+nt R4", probe_gather_lds<4, true>, g, 128);
            }
            for (auto gb : {std::pair<int, int>{2048, 128}}) {
                add_v2("v2 pipe+nt R4", probe_gather_v2<1024, true, 4>, gb.first, gb.second);
                add_v2("v2+classify", probe_gather_v2<1024, true, 4, 1>, gb.first, gb.second);
                add_v2("v2+classify+miss", probe_gather_v2<1024, true, 4, 2>, gb.first, gb.second);
            }
        }
    }
    // with misses the full call would insert them and turn every later launch into all hits: time it only for all-hit runs
    if (hit_pct == 100) vs.push_back({"product read_feature (K1+K2)", [=](hipStream_t s) { coala_cache_read_feature(h, out, d_ids, n, s); }});

    const int reps = 30;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<double> best(vs.size(), 1e30), sum(vs.size(), 0);
    for (int r = 0; r < reps + 2; ++r)
        for (size_t v = 0; v < vs.size(); ++v) { // interleaved A/B
            CK(hipEventRecord(a, st));
            vs[v].launch(st);
            CK(hipEventRecord(b, st));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (r >= 2) { best[v] = std::min(best[v], (double)ms); sum[v] += ms; }
        }
    CK(hipGetLastError());
    printf("n=%lld dim=%d hit%%=%d alg_bytes=%.1f MB\n", (long long)n, dim, hit_pct, alg_bytes / 1e6);
    for (size_t v = 0; v < vs.size(); ++v)
        printf("%-40s avg %8.2f us  min %8.2f us   %7.1f GB/s (avg)  %5.1f%% of 8 TB/s\n", vs[v].name.c_str(), sum[v] / reps * 1e3,
               best[v] * 1e3, alg_bytes / (sum[v] / reps * 1e-3) / 1e9, alg_bytes / (sum[v] / reps * 1e-3) / 1e9 / 80.0);
    return 0;
}
