// line_stride_probe.hip -- do 512-B lines that all sit at the SAME offset of a 16-KiB block (way 0 of every set of a set-major cache:
// address = set x 16 KiB + way x 512 B) read slower than 512-B lines spread uniformly?  Development tool (round 3).
//
// Every half-wave reads one 512-B line (32 lanes x 16 B), 4 line pairs in flight per wave, from a 16 GiB buffer:
//   spread   : line index uniform in [0, 32 Mi)                                    -> any 512-B-aligned address
//   way0     : line index = 32 x (uniform set)            (set-major, always way 0) -> addresses that are multiples of 16 KiB
//   ways0-3  : line index = 32 x set + (0..3)
//   waymajor : line index = way x num_sets + set with way 0                         -> the first 512 MiB of the buffer, densely
//   hipcc --offload-arch=gfx950 -O3 tools/line_stride_probe.hip -o tools/line_stride_probe
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

__global__ __launch_bounds__(128) void read_lines(const vu32x4* __restrict__ buf, const uint32_t* __restrict__ line_of, int64_t n_lines, uint32_t* out) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, l_in = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 6);
    const int64_t base = wave * 8; // 8 lines per wave: 4 instructions of two lines
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

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main() {
    const uint64_t bytes = 16ull << 30, n_total = bytes / 512, num_sets = n_total / 32;
    vu32x4* buf;
    uint32_t *d_lines, *out;
    CHK(hipMalloc((void**)&buf, bytes));
    CHK(hipMemset(buf, 1, bytes));
    CHK(hipMalloc((void**)&out, 64));
    printf("# tools/line_stride_probe: 512-B lines of a 16 GiB buffer, one line per half-wave, 8 lines per wave, one launch per row count\n");
    for (int64_t n : {(int64_t)43008, (int64_t)196608, (int64_t)1081344}) {
        CHK(hipMalloc((void**)&d_lines, n * 4));
        std::vector<uint32_t> h(n);
        const char* names[] = {"spread", "way0", "ways0-3", "waymajor"};
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                for (auto& x : h) {
                    const uint64_t set = rnd() % num_sets;
                    x = mode == 0 ? (uint32_t)(rnd() % n_total) : mode == 1 ? (uint32_t)(set * 32) : mode == 2 ? (uint32_t)(set * 32 + rnd() % 4) : (uint32_t)set;
                }
                CHK(hipMemcpy(d_lines, h.data(), n * 4, hipMemcpyHostToDevice));
                hipEvent_t a, b;
                CHK(hipEventCreate(&a));
                CHK(hipEventCreate(&b));
                CHK(hipEventRecord(a));
                hipLaunchKernelGGL(read_lines, dim3((unsigned)((n / 8 + 1) / 2 + 1)), dim3(128), 0, 0, buf, d_lines, n, out);
                CHK(hipEventRecord(b));
                CHK(hipEventSynchronize(b));
                float ms;
                CHK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
                CHK(hipEventDestroy(a));
                CHK(hipEventDestroy(b));
            }
            printf("%8lld lines %-9s: %8.1f us  = %7.1f GB/s\n", (long long)n, names[mode], best * 1e3, (double)n * 512 / (best * 1e-3) / 1e9);
        }
        CHK(hipFree(d_lines));
    }
    return 0;
}
