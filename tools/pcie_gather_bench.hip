// tools/pcie_gather_bench.hip -- development tool: how fast can random 4 KiB rows be pulled zero-copy from pinned host memory?
// Variants of the cold-fill data movement (rows in flight per wave, nontemporal or not, waves per CU), against hipMemcpy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pcie_gather_bench.hip -o tools/pcie_gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int ROWS, int MODE> // MODE 0 plain, 1 nontemporal load, 2 sc1 (system-scope-ish) via builtin atomic load? keep 0/1
__global__ __launch_bounds__(256) void gather_rows(const float* __restrict__ host, const int64_t* __restrict__ ids, float* __restrict__ out, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t base = wave * ROWS; base < n; base += n_waves * ROWS) {
        vf4 v[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = base + r;
            const vf4* src = reinterpret_cast<const vf4*>(host + (i < n ? ids[i] : 0) * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i < n) v[r][k] = MODE == 1 ? __builtin_nontemporal_load(src + k * 64 + lane) : src[k * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = base + r;
            vf4* dst = reinterpret_cast<vf4*>(out + i * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i < n) __builtin_nontemporal_store(v[r][k], dst + k * 64 + lane);
        }
    }
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 2000000, n = argc > 2 ? atoll(argv[2]) : 19300;
    CK(hipSetDevice(0));
    float* host; CK(hipHostMalloc((void**)&host, rows * 4096, hipHostMallocMapped | hipHostMallocPortable));
    for (int64_t i = 0; i < rows * 1024; i += 1024) host[i] = (float)i;
    float* hostd; CK(hipHostGetDevicePointer((void**)&hostd, host, 0));
    std::vector<int64_t> perm(rows); std::iota(perm.begin(), perm.end(), 0);
    std::mt19937_64 rng(3); std::shuffle(perm.begin(), perm.end(), rng);
    int64_t* ids; CK(hipMalloc(&ids, n * 8)); CK(hipMemcpy(ids, perm.data(), n * 8, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, n * 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char* name, auto launch) {
        double best = 1e30, sum = 0; const int reps = 12;
        for (int r = 0; r < reps + 2; ++r) {
            // fresh ids each time so that nothing is served from a cache
            CK(hipMemcpy(ids, perm.data() + ((r + 1) * n) % (rows - n), n * 8, hipMemcpyHostToDevice));
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r >= 2) { best = std::min(best, (double)ms); sum += ms; }
        }
        printf("%-34s avg %8.1f us  best %8.1f us  %6.2f GB/s (avg) %6.2f (best)\n", name, sum / reps * 1e3, best * 1e3, n * 4096.0 / (sum / reps * 1e-3) / 1e9, n * 4096.0 / (best * 1e-3) / 1e9);
    };
    time("hipMemcpy H2D contiguous", [&] { CK(hipMemcpyAsync(out, host, n * 4096, hipMemcpyHostToDevice, 0)); });
#define V(R, M, G) time("rows/wave " #R " mode " #M " grid " #G, [&] { hipLaunchKernelGGL((gather_rows<R, M>), dim3(G), dim3(256), 0, 0, hostd, ids, out, n); });
    V(4, 1, 2048) V(4, 0, 2048) V(1, 1, 2048) V(2, 1, 2048) V(8, 1, 1024) V(4, 1, 512) V(4, 1, 256) V(4, 1, 4096) V(1, 1, 8192) V(2, 1, 4096) V(8, 0, 512)
    return 0;
}
