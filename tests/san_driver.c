/* Driver for the sanitizer run of the CPU oracle (tests/test_sanitize_cpu.py): ASan + UBSan, CPU build only
 * (GPU sanitizers are not available on the pool). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../oracle/coala_oracle.h"

static unsigned long long rng_state = 88172645463325252ull;
static unsigned long long rnd(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

int main(void) {
    const int dim = 100, rows = 5000, G = 3;
    float* feat = (float*)malloc(sizeof(float) * rows * dim);
    orc_fill_features(feat, 0, rows, dim, 7);
    int64_t* color = (int64_t*)malloc(sizeof(int64_t) * rows);
    for (int i = 0; i < rows; ++i) color[i] = (int64_t)(rnd() % 9);
    for (int sched = 0; sched < 2; ++sched) {
        orc_cache* c = orc_cache_create(1, dim, feat, rows, color, 8, 1, 0, 0);
        for (int b = 0; b < 6; ++b) {
            int64_t n = 1 + (int64_t)(rnd() % 3000);
            int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * n);
            float* out = (float*)malloc(sizeof(float) * n * dim);
            for (int64_t i = 0; i < n; ++i) idx[i] = (int64_t)(rnd() % rows);
            orc_read_feature(c, idx, n, out, sched);
            for (int64_t i = 0; i < n; ++i)
                if (memcmp(out + i * dim, feat + idx[i] * dim, sizeof(float) * dim)) { printf("row mismatch\n"); return 1; }
            free(idx); free(out);
        }
        if (c->hit_cnt + c->miss_cnt == 0) return 1;
        orc_cache_destroy(c);
    }
    { /* collective step */
        orc_cache* caches[3];
        int64_t n[3]; const int64_t* idx[3]; float* out[3];
        for (int g = 0; g < G; ++g) caches[g] = orc_cache_create(1, dim, feat, rows, NULL, 0, G, 1, 0);
        for (int step = 0; step < 3; ++step) {
            for (int g = 0; g < G; ++g) {
                n[g] = (g == 1 && step == 1) ? 0 : 1 + (int64_t)(rnd() % 2000);
                int64_t* p = (int64_t*)malloc(sizeof(int64_t) * (n[g] ? n[g] : 1));
                for (int64_t i = 0; i < n[g]; ++i) p[i] = (int64_t)(rnd() % rows);
                idx[g] = p;
                out[g] = (float*)malloc(sizeof(float) * (n[g] ? n[g] : 1) * dim);
            }
            orc_dist_fetch(caches, G, idx, n, out, ORC_SCHED_HITS_FIRST);
            for (int g = 0; g < G; ++g) {
                for (int64_t i = 0; i < n[g]; ++i)
                    if (memcmp(out[g] + i * dim, feat + idx[g][i] * dim, sizeof(float) * dim)) { printf("dist mismatch\n"); return 1; }
                free((void*)idx[g]); free(out[g]);
            }
        }
        for (int g = 0; g < G; ++g) orc_cache_destroy(caches[g]);
    }
    { /* sampler twin + compaction */
        const int64_t nn = 2000;
        int64_t* indptr = (int64_t*)malloc(sizeof(int64_t) * (nn + 1));
        indptr[0] = 0;
        for (int64_t i = 0; i < nn; ++i) indptr[i + 1] = indptr[i] + (int64_t)(rnd() % 20);
        int64_t* indices = (int64_t*)malloc(sizeof(int64_t) * (indptr[nn] ? indptr[nn] : 1));
        for (int64_t e = 0; e < indptr[nn]; ++e) indices[e] = (int64_t)(rnd() % nn);
        int64_t dst[64];
        for (int i = 0; i < 64; ++i) dst[i] = i * 31;
        int64_t* nbr = (int64_t*)malloc(sizeof(int64_t) * 64 * 7);
        orc_sample_layer(indptr, indices, nn, dst, 64, 7, 5, 2, 0, nbr);
        int64_t* src = (int64_t*)malloc(sizeof(int64_t) * 64 * 8);
        int32_t* local = (int32_t*)malloc(sizeof(int32_t) * 64 * 7);
        int64_t ns = orc_compact_block(dst, 64, nbr, 7, src, local);
        if (ns < 64) return 1;
        free(indptr); free(indices); free(nbr); free(src); free(local);
    }
    { /* .npy header */
        const char hdr[] = "\x93NUMPY\x01\x00\x46\x00{'descr': '<i8', 'fortran_order': False, 'shape': (6, 10), }          \n";
        int64_t shape[2]; int nd; size_t off; char descr[8];
        if (orc_npy_parse(hdr, sizeof(hdr) - 1, 2, shape, &nd, &off, descr, sizeof(descr)) != 0 || nd != 2 || shape[1] != 10) return 1;
        orc_npy_parse(hdr, 9, 2, shape, &nd, &off, descr, sizeof(descr)); /* truncated: must not read past the end */
    }
    free(feat); free(color);
    printf("sanitized run ok\n");
    return 0;
}
