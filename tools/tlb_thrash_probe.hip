// tlb_thrash_probe.hip -- why does a gather of 512-B lines out of a 16 GiB cache take ~3x as long inside K1 (right behind K2's cold fill) as in
// tools/line_stride_probe (the same gather, repeated on its own)?  Hypothesis: address translation.  K2 reads ~100 k random rows of a pinned host
// table (4-KiB host pages) between two K1 launches; every XCD has its own translation cache, and 16 GiB is 8,192 pages of 2 MiB per XCD.
// Development tool (round 3), not part of the product.
//
//   gather      : every half-wave reads one 512-B line (32 lanes x 16 B), 4 line pairs in flight per wave, line list precomputed
//   gather_xcd  : the list is bucketed by the top 3 bits of the line index (2 GiB address ranges); blocks b, b+8, b+16 ... (one XCD under the
//                 observed round-robin placement) read bucket b % 8 only -> an XCD translates 1/8 of the footprint
//   thrash_host : N random 512-B rows of a pinned host buffer (what K2's cold fill does to the translation caches)
//   thrash_dev  : N random 512-B lines of ANOTHER 16 GiB device buffer
//
//   hipcc --offload-arch=gfx950 -O3 tools/tlb_thrash_probe.hip -o tools/tlb_thrash_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef unsigned int vu32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(128) void gather(const vu32x4* __restrict__ buf, const uint32_t* __restrict__ line_of, int64_t n_lines, uint32_t* out) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, l_in = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 6);
    const int64_t base = wave * 8;
    if (base >= n_lines) return;
    vu32x4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t i = base + 2 * p + sub;
        const uint32_t line = i < n_lines ? line_of[i] : 0;
        v[p] = __builtin_nontemporal_load(buf + (uint64_t)line * 32 + l_in);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) acc += v[p].y;
    if (acc == 0x12345678u) out[0] = acc;
}

// the same gather with the lines written out as consecutive 512-B rows (what K1's hit path does once a row's slot is known)
template <bool NT>
__global__ __launch_bounds__(128) void gather_copy(const vu32x4* __restrict__ buf, const uint32_t* __restrict__ line_of, int64_t n_lines, vu32x4* __restrict__ dst) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, l_in = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 6);
    const int64_t base = wave * 8;
    if (base >= n_lines) return;
    vu32x4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t i = base + 2 * p + sub;
        const uint32_t line = i < n_lines ? line_of[i] : 0;
        v[p] = __builtin_nontemporal_load(buf + (uint64_t)line * 32 + l_in);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t i = base + 2 * p + sub;
        if (i < n_lines) {
            if (NT) __builtin_nontemporal_store(v[p], dst + i * 32 + l_in);
            else dst[i * 32 + l_in] = v[p];
        }
    }
}

// bucket k of the list = lines with (line >> shift) == k, at [off[k], off[k+1]); blocks with blockIdx % 8 == k walk bucket k
__global__ __launch_bounds__(128) void gather_xcd(const vu32x4* __restrict__ buf, const uint32_t* __restrict__ line_of, const uint32_t* __restrict__ off,
                                                  uint32_t* out, uint32_t* xcc_of_block) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, l_in = lane & 31;
    const uint32_t k = blockIdx.x & 7, local = blockIdx.x >> 3, per = gridDim.x >> 3;
    if (xcc_of_block && threadIdx.x == 0) {
        uint32_t x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_of_block[blockIdx.x] = x & 15;
    }
    const uint32_t b0 = off[k], b1 = off[k + 1];
    uint32_t acc = 0;
    for (uint32_t base = b0 + (local * 2 + (threadIdx.x >> 6)) * 8; base < b1; base += per * 2 * 8) {
        vu32x4 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t i = base + 2 * p + sub;
            const bool ok = i < b1;
            const uint32_t line = ok ? line_of[i] : 0;
            if (ok) v[p] = __builtin_nontemporal_load(buf + (uint64_t)line * 32 + l_in);
            else v[p] = vu32x4(0u);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) acc += v[p].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void thrash(const vu32x4* __restrict__ buf, const uint32_t* __restrict__ row_of, int64_t n_rows, uint32_t* out) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, l_in = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    uint32_t acc = 0;
    for (int64_t base = wave * 8; base < n_rows; base += n_waves * 8) {
        vu32x4 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t i = base + 2 * p + sub;
            const uint32_t row = i < n_rows ? row_of[i] : 0;
            v[p] = __builtin_nontemporal_load(buf + (uint64_t)row * 32 + l_in);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) acc += v[p].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void empty_kernel() {}

// plain streaming copy (16 B per lane, 4 loads in flight per lane, nontemporal loads, plain stores): what the memory system gives a read + write kernel with
// perfectly sequential addresses -- the ceiling of K1's all-hit launch (151 MB of lines in, 151 MB of rows out)
__global__ __launch_bounds__(256) void stream_copy(const vu32x4* __restrict__ src, vu32x4* __restrict__ dst, size_t n_vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        vu32x4 v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
        vu32x4 v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
        dst[i] = v0; dst[i + stride] = v1; dst[i + 2 * stride] = v2; dst[i + 3 * stride] = v3;
    }
    for (; i < n_vec; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

// latency: ONE wave walks a chain of random 512-B lines (the first dword of a line names the next one); ns per dependent hop
__global__ void chain_init(vu32x4* buf, const uint32_t* seq, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[(uint64_t)seq[i] * 32].x = seq[(i + 1) % n];
}
__global__ void chain_walk(const vu32x4* __restrict__ buf, uint32_t start, int hops, uint32_t* out) {
    const int lane = threadIdx.x & 31;
    uint32_t line = start, acc = 0;
    for (int h = 0; h < hops; ++h) {
        const vu32x4 v = __builtin_nontemporal_load(buf + (uint64_t)line * 32 + lane);
        acc += v.y;
        line = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.x);
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (threadIdx.x == 0) out[1] = line;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Stat { double mean, mn; };
template <typename F, typename G>
static Stat timed(int reps, G&& before, F&& launch) {
    double sum = 0, mn = 1e30;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    for (int r = 0; r < reps; ++r) {
        before(r);
        CHK(hipEventRecord(a));
        launch(r);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        sum += ms * 1e3;
        mn = std::min(mn, (double)ms * 1e3);
    }
    CHK(hipEventDestroy(a));
    CHK(hipEventDestroy(b));
    return {sum / reps, mn};
}

int main() {
    const uint64_t bytes = 16ull << 30, n_total = bytes / 512;
    const int REPS = 10;
    vu32x4 *buf, *buf2, *host;
    uint32_t* out;
    CHK(hipMalloc((void**)&buf, bytes));
    CHK(hipMemset(buf, 1, bytes));
    CHK(hipMalloc((void**)&buf2, bytes));
    CHK(hipMemset(buf2, 1, bytes));
    CHK(hipHostMalloc((void**)&host, bytes, hipHostMallocDefault));
    for (uint64_t i = 0; i < bytes / 16; i += 256) ((uint32_t*)host)[i * 4] = 1; // touch every 4-KiB page
    CHK(hipMalloc((void**)&out, 64));
    const int64_t n_thrash = 120000;
    std::vector<uint32_t> h(n_thrash);
    uint32_t* d_thrash[REPS];
    for (int r = 0; r < REPS; ++r) {
        for (auto& x : h) x = (uint32_t)(rnd() % n_total);
        CHK(hipMalloc((void**)&d_thrash[r], n_thrash * 4));
        CHK(hipMemcpy(d_thrash[r], h.data(), n_thrash * 4, hipMemcpyHostToDevice));
    }
    printf("# tools/tlb_thrash_probe: gather of N random 512-B lines out of 16 GiB (hipEvent bracket included), mean / min over %d repetitions, fresh lines every repetition\n", REPS);
    {
        auto s = timed(REPS, [](int) {}, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); });
        printf("empty kernel in the bracket                                   : %6.1f / %6.1f us\n", s.mean, s.mn);
    }
    {   // dependent-load latency on this box, idle chip: what a latency-bound kernel (K1 on short lines) is made of
        const int n_chain = 65536;
        std::vector<uint32_t> seq(n_chain);
        for (auto& x : seq) x = (uint32_t)(rnd() % n_total);
        uint32_t* d_seq;
        CHK(hipMalloc((void**)&d_seq, n_chain * 4));
        CHK(hipMemcpy(d_seq, seq.data(), n_chain * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(chain_init, dim3(n_chain / 256), dim3(256), 0, 0, buf, d_seq, n_chain);
        CHK(hipDeviceSynchronize());
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: the chain's lines pushed out of L2 / Infinity Cache first (600 MB of other reads); pass 1: walked again right away (8 MB: cache-resident)
            if (pass == 0) for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, 0, buf2, d_thrash[r], n_thrash, out);
            auto s = timed(1, [](int) {}, [&](int) { hipLaunchKernelGGL(chain_walk, dim3(1), dim3(64), 0, 0, buf, seq[0], n_chain, out); });
            printf("dependent chain of %d random 512-B lines of the 16 GiB buffer, one wave, %s: %6.1f ns per hop\n", n_chain, pass ? "walked again (cache-resident)" : "from HBM                     ", (s.mn - 6.0) * 1e3 / n_chain);
        }
        CHK(hipFree(d_seq));
    }
    {   // streaming copy of 151 MB (one all-hit launch of 36,864 x 4 KiB), sources rotating over 5 regions of the 16 GiB buffer, destinations over 1 or 3 buffers
        const size_t cbytes = (size_t)36864 * 4096, n_vec = cbytes / 16;
        vu32x4* dsts[3];
        for (auto& d : dsts) CHK(hipMalloc((void**)&d, cbytes));
        for (unsigned grid : {2048u, 8192u}) {
            for (int nd : {1, 3}) {
                auto s = timed(REPS, [](int) {}, [&](int r) { hipLaunchKernelGGL(stream_copy, dim3(grid), dim3(256), 0, 0, buf + (size_t)(r % 5) * (1ull << 26), dsts[r % nd], n_vec); });
                printf("streaming copy of %.0f MB, %u blocks, %d destination buffer(s) in rotation: %6.1f / %6.1f us  = %6.0f GB/s read + written (min, bracket of ~6 us included)\n",
                       cbytes / 1e6, grid, nd, s.mean, s.mn, 2.0 * cbytes / (s.mn * 1e-6) / 1e9);
            }
        }
        for (auto& d : dsts) CHK(hipFree(d));
    }
    for (int64_t n : {(int64_t)43008, (int64_t)196608}) {
        uint32_t *d_list[REPS], *d_blist[REPS], *d_off[REPS], *d_xcc;
        std::vector<uint32_t> l(n), bl(n), off(9);
        const unsigned grid = (unsigned)((n / 8 + 1) / 2 + 1);
        const unsigned grid_x = ((grid + 7) / 8) * 8;
        CHK(hipMalloc((void**)&d_xcc, grid_x * 4));
        for (int r = 0; r < REPS; ++r) {
            for (auto& x : l) x = (uint32_t)(rnd() % n_total);
            // stable bucketing by the top 3 bits of the line index (n_total = 2^25 lines)
            std::fill(off.begin(), off.end(), 0u);
            for (auto x : l) off[(x >> 22) + 1]++;
            for (int k = 0; k < 8; ++k) off[k + 1] += off[k];
            std::vector<uint32_t> cur(off.begin(), off.end() - 1);
            for (auto x : l) bl[cur[x >> 22]++] = x;
            CHK(hipMalloc((void**)&d_list[r], n * 4));
            CHK(hipMalloc((void**)&d_blist[r], n * 4));
            CHK(hipMalloc((void**)&d_off[r], 9 * 4));
            CHK(hipMemcpy(d_list[r], l.data(), n * 4, hipMemcpyHostToDevice));
            CHK(hipMemcpy(d_blist[r], bl.data(), n * 4, hipMemcpyHostToDevice));
            CHK(hipMemcpy(d_off[r], off.data(), 9 * 4, hipMemcpyHostToDevice));
        }
        auto none = [](int) {};
        auto th_host = [&](int r) { hipLaunchKernelGGL(thrash, dim3(64), dim3(256), 0, 0, host, d_thrash[r], n_thrash, out); };
        auto th_dev = [&](int r) { hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, 0, buf2, d_thrash[r], n_thrash, out); };
        auto g_plain = [&](int r) { hipLaunchKernelGGL(gather, dim3(grid), dim3(128), 0, 0, buf, d_list[r], n, out); };
        auto g_same = [&](int) { hipLaunchKernelGGL(gather, dim3(grid), dim3(128), 0, 0, buf, d_list[0], n, out); };
        auto g_bucketed_list = [&](int r) { hipLaunchKernelGGL(gather, dim3(grid), dim3(128), 0, 0, buf, d_blist[r], n, out); };
        auto g_xcd = [&](int r) { hipLaunchKernelGGL(gather_xcd, dim3(grid_x), dim3(128), 0, 0, buf, d_blist[r], d_off[r], out, (uint32_t*)nullptr); };
        auto g_xcd_small = [&](int r) { hipLaunchKernelGGL(gather_xcd, dim3(2048), dim3(128), 0, 0, buf, d_blist[r], d_off[r], out, (uint32_t*)nullptr); };
        struct { const char* name; Stat s; } rows[] = {
            {"same list every time (translations warm)", timed(REPS, none, g_same)},
            {"fresh list", timed(REPS, none, g_plain)},
            {"fresh list, behind 120 k random 512-B reads of pinned host", timed(REPS, th_host, g_plain)},
            {"fresh list, behind 120 k random 512-B reads of another 16 GiB", timed(REPS, th_dev, g_plain)},
            {"fresh list sorted into 8 address ranges, plain grid", timed(REPS, none, g_bucketed_list)},
            {"  .. behind the host reads", timed(REPS, th_host, g_bucketed_list)},
            {"XCD-partitioned (block b reads range b % 8), one chunk/wave", timed(REPS, none, g_xcd)},
            {"  .. behind the host reads", timed(REPS, th_host, g_xcd)},
            {"  .. behind the device reads", timed(REPS, th_dev, g_xcd)},
            {"XCD-partitioned, 2048 blocks looping", timed(REPS, none, g_xcd_small)},
            {"  .. behind the host reads", timed(REPS, th_host, g_xcd_small)},
        };
        for (auto& r : rows) printf("%7lld lines  %-62s: %6.1f / %6.1f us\n", (long long)n, r.name, r.s.mean, r.s.mn);
        {   // gather + write: one output buffer reused against three in rotation (the Infinity Cache holds 256 MiB)
            vu32x4* dsts[3];
            for (auto& d : dsts) CHK(hipMalloc((void**)&d, (size_t)n * 512));
            auto c1 = [&](int r) { hipLaunchKernelGGL(gather_copy<false>, dim3(grid), dim3(128), 0, 0, buf, d_list[r], n, dsts[0]); };
            auto c3 = [&](int r) { hipLaunchKernelGGL(gather_copy<false>, dim3(grid), dim3(128), 0, 0, buf, d_list[r], n, dsts[r % 3]); };
            auto c3nt = [&](int r) { hipLaunchKernelGGL(gather_copy<true>, dim3(grid), dim3(128), 0, 0, buf, d_list[r], n, dsts[r % 3]); };
            struct { const char* name; Stat s; } rows2[] = {
                {"gather + write rows, ONE output buffer", timed(REPS, none, c1)},
                {"gather + write rows, three output buffers in rotation", timed(REPS, none, c3)},
                {"  .. behind the host reads", timed(REPS, th_host, c3)},
                {"  .. nontemporal stores", timed(REPS, none, c3nt)},
                {"gather + write rows, ONE output buffer (again)", timed(REPS, none, c1)},
            };
            for (auto& r : rows2) printf("%7lld lines  %-62s: %6.1f / %6.1f us\n", (long long)n, r.name, r.s.mean, r.s.mn);
            for (auto& d : dsts) CHK(hipFree(d));
        }
        // does blockIdx % 8 name an XCD?
        hipLaunchKernelGGL(gather_xcd, dim3(grid_x), dim3(128), 0, 0, buf, d_blist[0], d_off[0], out, d_xcc);
        std::vector<uint32_t> xcc(grid_x);
        CHK(hipMemcpy(xcc.data(), d_xcc, grid_x * 4, hipMemcpyDeviceToHost));
        unsigned agree = 0;
        for (unsigned b = 0; b < grid_x; ++b) agree += xcc[b] == xcc[b & 7];
        printf("%7lld lines  blocks whose XCC_ID equals that of block (b %% 8): %u of %u; XCC of blocks 0..7: %u %u %u %u %u %u %u %u\n", (long long)n, agree, grid_x,
               xcc[0], xcc[1], xcc[2], xcc[3], xcc[4], xcc[5], xcc[6], xcc[7]);
        for (int r = 0; r < REPS; ++r) {
            CHK(hipFree(d_list[r]));
            CHK(hipFree(d_blist[r]));
            CHK(hipFree(d_off[r]));
        }
        CHK(hipFree(d_xcc));
    }
    return 0;
}
