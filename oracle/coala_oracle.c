/*
 * coala_oracle.c -- CPU restatement (plain C) of the COALA-GNN feature-cache path.
 * TEST INFRASTRUCTURE ONLY; see coala_oracle.h for the rules and the "parity unpinned" statement.
 * Citations are relative to /root/reference/COALA_GNN_Modules unless they name another directory.
 */
#include "coala_oracle.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <string.h>

/* ---------------------------------------------------------------- geometry */

int orc_cache_dim(int dim) { /* ssd_gnn_cache.cuh:34-44 */
    if (dim <= 128) return 128;
    if (dim <= 256) return 256;
    if (dim <= 512) return 512;
    if (dim <= 1024) return 1024;
    return -1; /* reference: throw std::runtime_error */
}

uint64_t orc_num_sets(uint64_t cache_mb, int cache_dim) { /* ssd_gnn_cache.cuh:96-97 */
    uint64_t page_size = (uint64_t)cache_dim * sizeof(float); /* ssd_gnn_cache.cuh:47 */
    uint64_t num_pages = cache_mb * 1024ull * 1024ull / page_size;
    return num_pages / ORC_WAYS;
}

orc_cache* orc_cache_create(uint64_t cache_mb, int dim, const float* feat, uint64_t num_rows,
                            const int64_t* node_color, int num_colors, int n_gpus, int distributed,
                            int tag_only) {
    int cd = orc_cache_dim(dim);
    if (cd < 0) return NULL;
    orc_cache* c = (orc_cache*)calloc(1, sizeof(orc_cache));
    if (!c) return NULL;
    c->num_ways = ORC_WAYS;
    c->cache_dim = (uint32_t)cd;
    c->dim = (uint32_t)dim;
    c->num_sets = orc_num_sets(cache_mb, cd);
    if (c->num_sets == 0) { free(c); return NULL; }
    uint64_t slots = c->num_sets * c->num_ways;
    /* isolated_cache.h:541-552 : locks/set_cnt zeroed, keys filled with 0xFF */
    c->keys = (uint64_t*)malloc(slots * sizeof(uint64_t));
    c->set_cnt = (uint32_t*)calloc(c->num_sets, sizeof(uint32_t));
    c->color_meta = (uint64_t*)calloc(slots, sizeof(uint64_t)); /* isolated_cache.h:562-563 */
    c->num_colors = num_colors;
    c->color_counters = (int32_t*)calloc((size_t)num_colors + 1, sizeof(int32_t)); /* :558-560 */
    if (!tag_only) c->lines = (float*)malloc(slots * (uint64_t)cd * sizeof(float)); /* :568-570 */
    if (!c->keys || !c->set_cnt || !c->color_meta || !c->color_counters || (!tag_only && !c->lines)) {
        orc_cache_destroy(c);
        return NULL;
    }
    memset(c->keys, 0xFF, slots * sizeof(uint64_t));
    c->feat = feat;
    c->num_rows = num_rows;
    c->node_color = node_color;
    c->n_gpus = n_gpus > 0 ? n_gpus : 1;
    c->distributed = distributed;
    return c;
}

void orc_cache_destroy(orc_cache* c) {
    if (!c) return;
    free(c->keys); free(c->set_cnt); free(c->color_meta); free(c->color_counters); free(c->lines);
    free(c);
}

uint64_t orc_set_id(const orc_cache* c, uint64_t id) {
    if (c->distributed) return (id / (uint64_t)c->n_gpus) % c->num_sets; /* isolated_cache.h:191-195; nvshmem_cache.h:191-196,347 */
    return id % c->num_sets;                                               /* isolated_cache.h:183-187 */
}

unsigned orc_search_ways(const orc_cache* c, uint64_t id, uint64_t set) { /* isolated_cache.h:145-174 */
    const uint64_t* ways = c->keys + set * c->num_ways;
    /* lanes stride the ways; the ballot picks the lowest matching lane == lowest matching way */
    for (unsigned w = 0; w < c->num_ways; ++w)
        if (ways[w] == id) return w;
    return c->num_ways;
}

/* the part of get_data after the lookup failed: isolated_cache.h:417-474 */
static void orc_miss(orc_cache* c, uint64_t id, uint64_t set, float* out) {
    uint64_t set_off = set * c->num_ways;
    uint32_t way = (c->set_cnt[set]++) % c->num_ways;             /* :197-210 round_robin_evict */
    uint64_t slot = set_off + way;
    if (c->node_color) {                                          /* color_track_ (always true in the reference) */
        c->color_counters[c->color_meta[slot]] -= 1;              /* :427-429 (colour 0 goes negative on first touch) */
    }
    c->keys[slot] = id;                                           /* :434 */
    if (c->node_color) {
        int64_t color = c->node_color[id];                        /* :437 */
        c->color_meta[slot] = (uint64_t)color;                    /* :438 */
        c->color_counters[color] += 1;                            /* :439-441 */
    }
    const float* src = c->feat + id * (uint64_t)c->dim;           /* :323-331 with host stride = dim (SURVEY 3.3 fix) */
    if (c->lines) {
        float* line = c->lines + slot * (uint64_t)c->cache_dim;
        memcpy(line, src, (size_t)c->dim * sizeof(float));        /* :449-452 fill the line */
        if (out) memcpy(out, line, (size_t)c->dim * sizeof(float)); /* :464-465 line -> output */
    } else if (out) {
        memcpy(out, src, (size_t)c->dim * sizeof(float));
    }
    c->miss_cnt++;                                                /* :471-472 */
}

static void orc_hit(orc_cache* c, uint64_t set, unsigned way, uint64_t id, float* out) { /* isolated_cache.h:366-406 */
    uint64_t slot = set * c->num_ways + way;
    if (out) {
        if (c->lines) memcpy(out, c->lines + slot * (uint64_t)c->cache_dim, (size_t)c->dim * sizeof(float)); /* :384-385 */
        else memcpy(out, c->feat + id * (uint64_t)c->dim, (size_t)c->dim * sizeof(float));
    }
    c->hit_cnt++;                                                 /* :402-403 */
}

int orc_get_data(orc_cache* c, uint64_t id, float* out) { /* isolated_cache.h:335-475 */
    uint64_t set = orc_set_id(c, id);
    unsigned way = orc_search_ways(c, id, set);               /* :366 ; seqlocks are no-ops for one warp */
    if (way < c->num_ways) { orc_hit(c, set, way, id, out); return 1; }
    orc_miss(c, id, set, out);
    return 0;
}

void orc_read_feature(orc_cache* c, const int64_t* idx, int64_t n, float* out, int schedule) {
    /* cache_kernel.cu:59-77 : warp i handles index[i] and writes out + i*dim */
    if (schedule == ORC_SCHED_SEQUENTIAL) {
        for (int64_t i = 0; i < n; ++i) orc_get_data(c, (uint64_t)idx[i], out ? out + i * (int64_t)c->dim : NULL);
        return;
    }
    /* ORC_SCHED_HITS_FIRST: one legal interleaving of the n independent warps -- every warp whose lookup succeeds on
     * the pre-batch table runs to completion first, then the remaining warps run their miss path in batch order. */
    unsigned char* missed = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        uint64_t id = (uint64_t)idx[i];
        uint64_t set = orc_set_id(c, id);
        unsigned way = orc_search_ways(c, id, set);
        missed[i] = (way >= c->num_ways);
        if (!missed[i]) orc_hit(c, set, way, id, out ? out + i * (int64_t)c->dim : NULL);
    }
    for (int64_t i = 0; i < n; ++i) {
        if (!missed[i]) continue;
        uint64_t id = (uint64_t)idx[i];
        /* the miss path never re-probes (isolated_cache.h:417ff): a duplicate id inside one batch inserts twice */
        orc_miss(c, id, orc_set_id(c, id), out ? out + i * (int64_t)c->dim : NULL);
    }
    free(missed);
}

/* ---------------------------------------------------------------- owner-partitioned (nccl / nvshmem) path */

void orc_split_node_list(const int64_t* idx, int64_t n, int64_t* node, int64_t* map, int64_t* counter,
                         int local_size, int64_t max_sample) { /* cache_kernel.cu:79-91 */
    for (int g = 0; g < local_size; ++g) counter[g] = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t cur = idx[i];
        int64_t gpu = cur % local_size;                         /* :86 */
        int64_t enq = counter[gpu]++;                           /* :87 (atomicAdd; stable order is our contract) */
        node[max_sample * gpu + enq] = cur;                     /* :88 */
        map[max_sample * gpu + enq] = i;                        /* :89 */
    }
}

void orc_map_feat_data(float* out, const float* src, const int64_t* map, int64_t n, int dim) { /* cache_kernel.cu:129-137 */
    for (int64_t r = 0; r < n; ++r) memcpy(out + map[r] * (int64_t)dim, src + r * (int64_t)dim, (size_t)dim * sizeof(float));
}

void orc_dist_fetch(orc_cache** caches, int G, const int64_t* const* idx, const int64_t* n, float* const* out,
                    int schedule) {
    /* COALA_GNN_Manager.py:143-211 : split -> all_to_all ids -> serve -> send/recv rows -> remap */
    int64_t max_n = 1;
    for (int g = 0; g < G; ++g) if (n[g] > max_n) max_n = n[g];
    int64_t** node = (int64_t**)malloc(sizeof(int64_t*) * G);
    int64_t** map = (int64_t**)malloc(sizeof(int64_t*) * G);
    int64_t** cnt = (int64_t**)malloc(sizeof(int64_t*) * G);
    for (int s = 0; s < G; ++s) {
        node[s] = (int64_t*)malloc(sizeof(int64_t) * G * max_n);
        map[s] = (int64_t*)malloc(sizeof(int64_t) * G * max_n);
        cnt[s] = (int64_t*)malloc(sizeof(int64_t) * G);
        orc_split_node_list(idx[s], n[s], node[s], map[s], cnt[s], G, max_n);
    }
    for (int o = 0; o < G; ++o) { /* owner o serves concat_s bucket[s -> o] as one batch */
        int64_t total = 0;
        for (int s = 0; s < G; ++s) total += cnt[s][o];
        int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * (total > 0 ? total : 1));
        int dim = (int)caches[o]->dim;
        int want = 0; /* tag-only callers pass no output arrays: counters and tags only */
        for (int s = 0; out && s < G; ++s) want |= out[s] != NULL;
        float* rows = want ? (float*)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1) * dim) : NULL;
        int64_t off = 0;
        for (int s = 0; s < G; ++s) { memcpy(ids + off, node[s] + max_n * o, sizeof(int64_t) * cnt[s][o]); off += cnt[s][o]; }
        orc_read_feature(caches[o], ids, total, rows, schedule); /* cache_kernel.cu:93-111 with the distributed set index */
        off = 0;
        for (int s = 0; s < G; ++s) { /* rows travel back to s and are un-permuted: cache_kernel.cu:129-137 */
            if (out && out[s]) orc_map_feat_data(out[s], rows + off * dim, map[s] + max_n * o, cnt[s][o], dim);
            off += cnt[s][o];
        }
        free(ids); free(rows);
    }
    for (int s = 0; s < G; ++s) { free(node[s]); free(map[s]); free(cnt[s]); }
    free(node); free(map); free(cnt);
}

/* ---------------------------------------------------------------- node distributor */

void orc_distribute_node_with_affinity(const int64_t* items, uint64_t offset, int global_batch_size,
                                       int domain_batch_size, int node_id, int num_nodes,
                                       const int64_t* color, const int64_t* topk, const double* score, int topk_k,
                                       const int32_t* const* meta, int64_t* out) {
    /* node_distributor_pybind.cuh:150-222 */
    int* bucket_len = (int*)calloc((size_t)num_nodes, sizeof(int));      /* :160 */
    for (int64_t i = 0; i < global_batch_size; ++i) {                    /* :167 */
        int64_t id = items[i + (int64_t)offset];                         /* :168 */
        int64_t node_color = color[id];                                  /* :172 */
        int cur_max_part = 0;                                            /* :173 */
        double max_score = -1.0;                                         /* :174 */
        for (int j = 0; j < num_nodes; ++j) {                            /* :176 */
            const int32_t* meta_ptr = meta[j];
            double cur_score = 0;
            if (node_color != 0) {                                       /* :183-186 */
                for (int k = 0; k < topk_k; ++k) {                       /* :187 */
                    int64_t neigh_color = topk[(node_color - 1) * topk_k + k];
                    double neigh_affinity = score[(node_color - 1) * topk_k + k];
                    if (neigh_color != 0) {                              /* :190 */
                        if (meta_ptr[neigh_color] == 0) continue;        /* :191-192 */
                        double neigh_score = (double)meta_ptr[neigh_color];
                        cur_score += neigh_score * neigh_affinity;       /* :195 */
                    }
                }
            }
            if (bucket_len[j] == domain_batch_size) cur_score = -1.0;    /* :207-208 */
            if (cur_score > max_score) { cur_max_part = j; max_score = cur_score; } /* :210-213 strict > : first max wins */
        }
        if (cur_max_part == node_id) out[bucket_len[cur_max_part]] = id; /* :216-219 */
        bucket_len[cur_max_part] += 1;                                   /* :220 */
    }
    free(bucket_len);
}

/* ---------------------------------------------------------------- .npy header */

static const char* find_str(const char* hay, size_t hlen, const char* needle) {
    size_t nlen = strlen(needle);
    if (nlen > hlen) return NULL;
    for (size_t i = 0; i + nlen <= hlen; ++i)
        if (memcmp(hay + i, needle, nlen) == 0) return hay + i;
    return NULL;
}

int orc_npy_parse(const char* buf, size_t len, int want_dim, int64_t* shape, int* ndim_out, size_t* data_off,
                  char* descr, size_t descr_cap) {
    /* node_distributor_pybind.cuh:37-109 */
    if (len < 10 || memcmp(buf, "\x93NUMPY", 6) != 0) return -1;       /* :43-46 */
    unsigned major = (unsigned char)buf[6];                              /* :48 */
    size_t pos = 8;
    uint32_t header_len = 0;
    if (major == 1) { uint16_t h16; memcpy(&h16, buf + pos, 2); header_len = h16; pos += 2; }        /* :55-60 */
    else if (major == 2) { if (len < 12) return -1; memcpy(&header_len, buf + pos, 4); pos += 4; }  /* :61-64 */
    else return -2;                                                      /* :65-67 */
    if (pos + header_len > len) return -1;
    const char* hdr = buf + pos;
    *data_off = pos + header_len;                                        /* :71-72 */
    *ndim_out = 0;
    if (want_dim != 1 && want_dim != 2) return -3;                       /* :97-99 */
    /* shape regexes :75-76 -- 1-D: 'shape':\s?\((\d+),?\)   2-D: 'shape':\s?\((\d+),\s?(\d+)\) */
    const char* p = find_str(hdr, header_len, "'shape':");
    if (p) {
        const char* end = hdr + header_len;
        p += 8;
        if (p < end && isspace((unsigned char)*p)) ++p;
        if (p < end && *p == '(') {
            ++p;
            int64_t v[2] = {0, 0};
            int nd = 0;
            const char* q = p;
            if (q < end && isdigit((unsigned char)*q)) {
                while (q < end && isdigit((unsigned char)*q)) v[0] = v[0] * 10 + (*q++ - '0');
                nd = 1;
                if (want_dim == 1) {
                    if (q < end && *q == ',') ++q;
                    if (q < end && *q == ')') { shape[0] = v[0]; *ndim_out = 1; }   /* :83-87 (else: shape stays empty) */
                } else {
                    if (q < end && *q == ',') {
                        ++q;
                        if (q < end && isspace((unsigned char)*q)) ++q;
                        if (q < end && isdigit((unsigned char)*q)) {
                            while (q < end && isdigit((unsigned char)*q)) v[1] = v[1] * 10 + (*q++ - '0');
                            if (q < end && *q == ')') { shape[0] = v[0]; shape[1] = v[1]; *ndim_out = 2; } /* :89-96 */
                        }
                    }
                }
            }
            (void)nd;
        }
    }
    /* dtype regex :77 'descr':\s*'(.*?)' */
    if (descr && descr_cap) descr[0] = 0;
    p = find_str(hdr, header_len, "'descr':");
    if (p && descr && descr_cap) {
        const char* end = hdr + header_len;
        p += 8;
        while (p < end && isspace((unsigned char)*p)) ++p;
        if (p < end && *p == '\'') {
            ++p;
            size_t k = 0;
            while (p < end && *p != '\'' && k + 1 < descr_cap) descr[k++] = *p++;
            descr[k] = 0;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- synthetic features */

float orc_feat_value(uint64_t row, uint32_t col, uint32_t seed) {
    uint32_t u = (uint32_t)(row * 0x9E3779B1ull) + col * 0x85EBCA77u + seed;
    return (float)(u >> 8) * (1.0f / 16777216.0f);
}

void orc_fill_features(float* dst, uint64_t row0, uint64_t nrows, uint32_t dim, uint32_t seed) {
    for (uint64_t r = 0; r < nrows; ++r)
        for (uint32_t c = 0; c < dim; ++c) dst[r * dim + c] = orc_feat_value(row0 + r, c, seed);
}

/* ---------------------------------------------------------------- neighbour sampler twin (contract of coala_sampler.hip) */

static uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static uint64_t orc_sample_key(uint64_t seed, uint64_t step, int layer, uint64_t v) {
    uint64_t h = orc_splitmix64(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(layer + 1)));
    h = orc_splitmix64(h ^ (step * 0xD1B54A32D192ED03ull));
    return orc_splitmix64(h ^ v);
}

static void orc_sample_row(const int64_t* indptr, const int64_t* indices, int64_t num_nodes, int64_t v, int fanout, uint64_t seed,
                           uint64_t step, int layer, int64_t* row) {
    if (v < 0 || v >= num_nodes) { for (int j = 0; j < fanout; ++j) row[j] = -1; return; }
    int64_t start = indptr[v], deg = indptr[v + 1] - start;
    if (deg <= fanout) { for (int j = 0; j < fanout; ++j) row[j] = j < deg ? indices[start + j] : -1; return; }
    uint64_t key = orc_sample_key(seed, step, layer, (uint64_t)v);
    int64_t chosen[32];
    int c = 0;
    for (int64_t j = deg - fanout; j < deg; ++j) { /* Floyd's subset sampling */
        uint64_t r = orc_splitmix64(key + (uint64_t)c);
        int64_t t = (int64_t)(((unsigned __int128)r * (unsigned __int128)(uint64_t)(j + 1)) >> 64);
        int dup = 0;
        for (int q = 0; q < c; ++q) dup |= (chosen[q] == t);
        if (dup) t = j;
        chosen[c++] = t;
    }
    for (int j = 0; j < fanout; ++j) row[j] = indices[start + chosen[j]];
}

void orc_sample_layer(const int64_t* indptr, const int64_t* indices, int64_t num_nodes, const int64_t* dst, int64_t n_dst,
                      int fanout, uint64_t seed, uint64_t step, int layer, int64_t* nbr) {
    for (int64_t d = 0; d < n_dst; ++d) orc_sample_row(indptr, indices, num_nodes, dst[d], fanout, seed, step, layer, nbr + d * fanout);
}

/* The same layer with the destination nodes spread over `threads` host threads (OpenMP): the draw of a node depends only on
 * (seed, step, layer, node), so the result is identical to orc_sample_layer's for any thread count.  bench.py's cpu_baseline times
 * it beside the one-core call (the CPU sampler on all host cores).  Returns the number of threads OpenMP gave. */
int orc_sample_layer_mt(const int64_t* indptr, const int64_t* indices, int64_t num_nodes, const int64_t* dst, int64_t n_dst,
                        int fanout, uint64_t seed, uint64_t step, int layer, int64_t* nbr, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static, 512)
        for (int64_t d = 0; d < n_dst; ++d) orc_sample_row(indptr, indices, num_nodes, dst[d], fanout, seed, step, layer, nbr + d * fanout);
    }
#else
    (void)threads;
    orc_sample_layer(indptr, indices, num_nodes, dst, n_dst, fanout, seed, step, layer, nbr);
#endif
    return used;
}

/* CPU feature gather on `threads` host threads (OpenMP): out[i, :] = table[idx[i], :], one memcpy per row -- what the reference's CPU path
 * does with `feat[input_nodes]` on a row-major fp32 host tensor before the copy to the GPU (BASELINE.json configs[0]; DGL / torch index_select).
 * For bench.py's cpu_baseline leg only (the gather on the box's host cores).  Rows are dealt out in blocks of 64 so
 * that a thread streams whole rows and neighbouring threads do not share output cache lines.  Returns the number of threads OpenMP gave. */
int orc_gather_rows_mt(const float* table, int64_t dim, const int64_t* idx, int64_t n, float* out, int threads) {
    int used = 1;
    const size_t row = (size_t)dim * sizeof(float);
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static, 64)
        for (int64_t i = 0; i < n; ++i) memcpy(out + (size_t)i * dim, table + (size_t)idx[i] * dim, row);
    }
#else
    (void)threads;
    for (int64_t i = 0; i < n; ++i) memcpy(out + (size_t)i * dim, table + (size_t)idx[i] * dim, row);
#endif
    return used;
}

typedef struct { int64_t key; int32_t val; } orc_slot;

int64_t orc_compact_block(const int64_t* dst, int64_t n_dst, const int64_t* nbr, int fanout, int64_t* src_out, int32_t* local) {
    int64_t n_items = n_dst * (fanout + 1);
    uint64_t cap = 16;
    while (cap < 2 * (uint64_t)(n_items > 0 ? n_items : 1)) cap *= 2;
    orc_slot* tab = (orc_slot*)malloc(cap * sizeof(orc_slot));
    for (uint64_t i = 0; i < cap; ++i) tab[i].key = -1;
    int64_t n_src = 0;
    for (int64_t p = 0; p < n_items; ++p) { /* sequential scan == first-appearance order */
        int64_t k = p < n_dst ? dst[p] : nbr[p - n_dst];
        int32_t loc = -1;
        if (k >= 0) {
            uint64_t s = orc_splitmix64((uint64_t)k) & (cap - 1);
            while (tab[s].key != -1 && tab[s].key != k) s = (s + 1) & (cap - 1);
            if (tab[s].key == -1) { tab[s].key = k; tab[s].val = (int32_t)n_src; src_out[n_src++] = k; }
            loc = tab[s].val;
        }
        if (p >= n_dst) local[p - n_dst] = loc;
    }
    free(tab);
    return n_src;
}

/* The same compaction on `threads` host threads (OpenMP), same result: (1) every item inserts its key into an open-addressing table
 * (CAS on the key word) and lowers the slot's "first position" with an atomic min; (2) an item is a first appearance iff the slot's
 * first position is its own; (3) a two-level prefix sum over the first-appearance flags numbers them in position order -- which IS
 * the sequential scan's order; (4) every neighbour item reads its key's number.  Mirrors what coala_sampler.hip does on the GPU. */
int64_t orc_compact_block_mt(const int64_t* dst, int64_t n_dst, const int64_t* nbr, int fanout, int64_t* src_out, int32_t* local, int threads) {
#ifndef _OPENMP
    (void)threads;
    return orc_compact_block(dst, n_dst, nbr, fanout, src_out, local);
#else
    const int64_t n_items = n_dst * (fanout + 1);
    if (threads < 1) threads = 1;
    uint64_t cap = 16;
    while (cap < 2 * (uint64_t)(n_items > 0 ? n_items : 1)) cap *= 2;
    int64_t* tkey = (int64_t*)malloc(cap * sizeof(int64_t));
    int64_t* tpos = (int64_t*)malloc(cap * sizeof(int64_t)); /* first position, then the local number */
    uint32_t* slot_of = (uint32_t*)malloc((size_t)(n_items > 0 ? n_items : 1) * sizeof(uint32_t));
    int64_t* part = (int64_t*)calloc((size_t)threads + 1, sizeof(int64_t));
    int64_t n_src = 0;
#pragma omp parallel num_threads(threads)
    {
        const int nt = omp_get_num_threads(), me = omp_get_thread_num();
#pragma omp for schedule(static)
        for (uint64_t i = 0; i < cap; ++i) { tkey[i] = -1; tpos[i] = INT64_MAX; }
#pragma omp for schedule(static)
        for (int64_t p = 0; p < n_items; ++p) {
            const int64_t k = p < n_dst ? dst[p] : nbr[p - n_dst];
            if (k < 0) { slot_of[p] = 0xFFFFFFFFu; continue; }
            uint64_t sl = orc_splitmix64((uint64_t)k) & (cap - 1);
            for (;;) {
                int64_t cur = __atomic_load_n(&tkey[sl], __ATOMIC_RELAXED);
                if (cur == -1) {
                    int64_t expect = -1;
                    if (__atomic_compare_exchange_n(&tkey[sl], &expect, k, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                    cur = expect;
                }
                if (cur == k) break;
                sl = (sl + 1) & (cap - 1);
            }
            slot_of[p] = (uint32_t)sl;
            int64_t seen = __atomic_load_n(&tpos[sl], __ATOMIC_RELAXED);
            while (p < seen && !__atomic_compare_exchange_n(&tpos[sl], &seen, p, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
        /* first appearances, numbered in position order: per-thread counts over contiguous ranges, then an exclusive scan */
        const int64_t lo = n_items * me / nt, hi = n_items * (me + 1) / nt;
        int64_t mine = 0;
        for (int64_t p = lo; p < hi; ++p) mine += (slot_of[p] != 0xFFFFFFFFu && tpos[slot_of[p]] == p);
        part[me + 1] = mine;
#pragma omp barrier
#pragma omp single
        {
            for (int t = 0; t < nt; ++t) part[t + 1] += part[t];
            n_src = part[nt];
        }
        int64_t next = part[me];
        for (int64_t p = lo; p < hi; ++p)
            if (slot_of[p] != 0xFFFFFFFFu && tpos[slot_of[p]] == p) src_out[next++] = p < n_dst ? dst[p] : nbr[p - n_dst];
#pragma omp barrier
        /* the slot now carries the local number instead of the first position (every first appearance owns its slot) */
        next = part[me];
        for (int64_t p = lo; p < hi; ++p) {
            const uint32_t sl = slot_of[p];
            if (sl != 0xFFFFFFFFu && tpos[sl] == p) tpos[sl] = -(next++) - 1; /* negative: cannot be mistaken for a position */
        }
#pragma omp barrier
#pragma omp for schedule(static)
        for (int64_t p = n_dst; p < n_items; ++p) local[p - n_dst] = slot_of[p] == 0xFFFFFFFFu ? -1 : (int32_t)(-(tpos[slot_of[p]] + 1));
    }
    free(tkey); free(tpos); free(slot_of); free(part);
    return n_src;
#endif
}
