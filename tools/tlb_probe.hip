// tlb_probe.hip -- what a random access costs on MI355X as the footprint grows from inside the Infinity Cache to a 16-64 GiB table
// (a 16 GiB cache shard: BASELINE configs[3] / [4]).  Development tool, not part of the product.
//
//   latency : ONE wave, a dependent chain of 64-lane x 16-B loads (1 KiB contiguous, at a random 1-KiB-aligned offset taken from the
//             previous load's data) -> ns per hop = memory + address-translation latency at that footprint
//   rate    : the whole chip, every wave issues `ILP` independent random 1-KiB loads per step from a precomputed offset list
//             -> GB/s = what a latency-tolerant random gather can reach at that footprint (translation throughput included)
//
//   hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o tools/tlb_probe && tools/tlb_probe
#include <hip/hip_runtime.h>

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

// every 1-KiB block of the buffer starts with the index of the next block of a random cycle (lane 0's first dword)
__global__ void chase_kernel(const vu32x4* __restrict__ buf, uint32_t start, int hops, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    uint32_t blk = start;
    uint32_t acc = 0;
    for (int h = 0; h < hops; ++h) {
        const vu32x4 v = buf[(uint64_t)blk * 64 + lane];
        acc += v.y;
        blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.x);
    }
    if (acc == 0x12345678u) out[0] = acc + blk;
    if (lane == 0) out[1] = blk;
}

template <int ILP>
__global__ __launch_bounds__(256) void rate_kernel(const vu32x4* __restrict__ buf, const uint32_t* __restrict__ offs, int64_t n_loads, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    uint32_t acc = 0;
    for (int64_t i = wave * ILP; i + ILP <= n_loads; i += n_waves * ILP) {
        vu32x4 v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) v[k] = __builtin_nontemporal_load(buf + (uint64_t)offs[i + k] * 64 + lane);
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc += v[k].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void fill_cycle(vu32x4* buf, const uint32_t* next, uint64_t n_blocks) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_blocks) buf[b * 64] = vu32x4{next[b], 1u, 2u, 3u};
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const double sizes_gb[] = {0.125, 1, 4, 16, 64};
    uint32_t* out;
    CHK(hipMalloc((void**)&out, 64));
    printf("# tools/tlb_probe: random 1-KiB blocks (64 lanes x 16 B) of a buffer of the given size\n");
    for (double gb : sizes_gb) {
        if (argc > 1 && atof(argv[1]) > 0 && gb > atof(argv[1])) break;
        const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
        const uint64_t n_blocks = bytes / 1024;
        vu32x4* buf = nullptr;
        if (hipMalloc((void**)&buf, bytes) != hipSuccess) { printf("%7.3f GiB: allocation failed\n", gb); break; }
        CHK(hipMemset(buf, 0, bytes));
        // a random cycle over a SAMPLE of the blocks (the chase touches 4096 hops; a full permutation of 64 M blocks is not needed)
        const int hops = 4096;
        std::vector<uint32_t> ring(hops);
        for (auto& r : ring) r = (uint32_t)(rnd() % n_blocks);
        std::vector<vu32x4> head(1);
        for (int h = 0; h < hops; ++h) {
            const vu32x4 v{ring[(h + 1) % hops], 1u, 2u, 3u};
            CHK(hipMemcpy(buf + (uint64_t)ring[h] * 64, &v, sizeof(v), hipMemcpyHostToDevice));
        }
        hipEvent_t a, b;
        CHK(hipEventCreate(&a));
        CHK(hipEventCreate(&b));
        float ms = 0.f, best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { // the second and third pass re-walk the same 4096 blocks: 4 MB, cached lines but NOT cached translations beyond the TLB's reach
            CHK(hipEventRecord(a));
            hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(64), 0, 0, buf, ring[0], hops, out);
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            CHK(hipEventElapsedTime(&ms, a, b));
            if (rep == 0) printf("%7.3f GiB: dependent chain, first walk  %7.1f ns per hop", gb, ms * 1e6 / hops);
            else if (ms < best) best = ms;
        }
        printf("   re-walk %7.1f ns per hop\n", best * 1e6 / hops);
        // rate: 2 M random blocks (2 GiB of traffic), offsets on the device
        const int64_t n_loads = 1 << 21;
        std::vector<uint32_t> offs(n_loads);
        for (auto& o : offs) o = (uint32_t)(rnd() % n_blocks);
        uint32_t* d_offs;
        CHK(hipMalloc((void**)&d_offs, n_loads * 4));
        CHK(hipMemcpy(d_offs, offs.data(), n_loads * 4, hipMemcpyHostToDevice));
        for (int ilp : {1, 4}) {
            for (int blocks : {2048, 8192}) {
                best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHK(hipEventRecord(a));
                    if (ilp == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, d_offs, n_loads, out);
                    else hipLaunchKernelGGL(rate_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, d_offs, n_loads, out);
                    CHK(hipEventRecord(b));
                    CHK(hipEventSynchronize(b));
                    CHK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                printf("             rate: %d blocks x 4 waves, %d loads in flight per wave: %8.1f us for 2 GiB = %7.1f GB/s\n", blocks, ilp, best * 1e3,
                       (double)n_loads * 1024 / (best * 1e-3) / 1e9);
            }
        }
        // small launch: 72 k random blocks, one per wave (the shape of a 72 k-row probe pass)
        for (int64_t n_small : {(int64_t)73728, (int64_t)294912}) {
            best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHK(hipEventRecord(a));
                hipLaunchKernelGGL(rate_kernel<1>, dim3((unsigned)(n_small / 4)), dim3(256), 0, 0, buf, d_offs + rep * n_small, n_small, out);
                CHK(hipEventRecord(b));
                CHK(hipEventSynchronize(b));
                CHK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("             one load per wave, %7lld waves: %7.1f us\n", (long long)n_small, best * 1e3);
        }
        CHK(hipFree(d_offs));
        CHK(hipFree(buf));
    }
    return 0;
}
