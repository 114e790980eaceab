// coala_coloring.cpp -- offline graph colouring + colour-affinity tables (host C++), C ABI coala_coloring_*.
//
// Replaces Graph_Coloring (COALA_GNN_Modules/graph_coloring.h:15-68, graph_coloring.cpp) used by
// examples/color_info_gen/generate_color_data.py:11-68 to write color.npy / topk.npy / score.npy.
// Parity for this row IS pinned by the reference: tests/golden/coloring_*.npz are produced by the reference's own
// source compiled in place (oracle/ref_build.py) and this implementation must reproduce them (colours bit-exact; top-k
// rows equal up to the order of exactly tied scores, which the reference leaves to std::sort / unordered_map order).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/coala_hip.h"
#include "coala_internal.h"

#define fail coala_fail_

struct coala_coloring {
    uint64_t num_nodes = 0;
    uint64_t num_colored = 0;
    uint64_t color_counter = 1;            // graph_coloring.h:18
    int max_hop = 10;                       // graph_coloring.h:20
    float sampling_rate = 0.005f;           // graph_coloring.h:21 (a float in the reference: the products below round alike)
    int topk = 10;                          // graph_coloring.h:23
    const int64_t* indptr = nullptr;
    const int64_t* indices = nullptr;
    int64_t* color = nullptr;
    int64_t* topk_color = nullptr;
    double* topk_affinity = nullptr;
    std::vector<uint16_t> hop;              // graph_coloring.cpp:324 (malloc'ed, never initialised there; zeroed here)
    std::vector<uint8_t> is_train;
    std::vector<std::pair<uint64_t, uint64_t>> buf[2]; // (node, colour) work lists used as stacks
};

namespace {

// graph_coloring.cpp:75-160 (optimized = true) and :43-72 (optimized = false)
void bfs_color(coala_coloring* g, bool optimized) {
    int hop = 0;
    for (; hop < g->max_hop; ++hop) {
        const int cur = hop % 2, next = (hop + 1) % 2;
        if (optimized && hop == 0) { // :121-137 uncoloured TRAINING in-neighbours of the seeds join the seed's colour first
            const size_t initial = g->buf[0].size();
            for (size_t i = 0; i < initial; ++i) {
                const uint64_t node = g->buf[0][i].first, col = g->buf[0][i].second;
                for (int64_t e = g->indptr[node]; e < g->indptr[node + 1]; ++e) {
                    const uint64_t nb = (uint64_t)g->indices[e];
                    if (g->is_train[nb] && g->color[nb] == 0) g->buf[0].emplace_back(nb, col);
                }
            }
        }
        auto& st = g->buf[cur];
        while (!st.empty()) { // :140-152 LIFO
            const auto pr = st.back();
            st.pop_back();
            if (g->color[pr.first] == 0) {
                g->color[pr.first] = (int64_t)pr.second;
                if (optimized) g->hop[pr.first] = (uint16_t)(hop + 1);
                g->num_colored++;
                for (int64_t e = g->indptr[pr.first]; e < g->indptr[pr.first + 1]; ++e) // :22-28
                    g->buf[next].emplace_back((uint64_t)g->indices[e], pr.second);
            }
        }
    }
    // :154-155 / :70-71 -- the reference flushes buffer (hop+1)%2 with hop == max_hop, i.e. the list that was just drained,
    // so whatever the last hop queued stays uncoloured.  Reproduced: same buffer index, same effect.
    auto& fl = g->buf[(hop + 1) % 2];
    while (!fl.empty()) {
        const auto pr = fl.back();
        fl.pop_back();
        if (g->color[pr.first] == 0) {
            g->color[pr.first] = (int64_t)pr.second;
            if (!optimized) g->hop[pr.first] = (uint16_t)(hop + 2); // cpu_flush_buffer<true>(next, hop+1) :30-41
            g->num_colored++;
        }
    }
}

template <typename T>
std::vector<std::pair<uint64_t, T>> top_k(const std::unordered_map<uint64_t, T>& m, size_t k) { // :165-181
    std::vector<std::pair<uint64_t, T>> v(m.begin(), m.end());
    std::sort(v.begin(), v.end(), [](const std::pair<uint64_t, T>& a, const std::pair<uint64_t, T>& b) {
        return a.second != b.second ? a.second > b.second : a.first < b.first; // ties: smaller colour first (reference: unspecified)
    });
    if (v.size() > k) v.resize(k);
    return v;
}

int check_ready(const coala_coloring* g, bool need_topk) {
    if (!g) return fail(COALA_EINVAL, "null handle");
    if (!g->indptr || !g->indices) return fail(COALA_EINVAL, "set_adj_csc was not called");
    if (!g->color) return fail(COALA_EINVAL, "set_color_buffer was not called");
    if (need_topk && !g->topk_color) return fail(COALA_EINVAL, "set_topk_color_buffer was not called");
    return COALA_OK;
}

} // namespace

extern "C" {

int coala_coloring_create(uint64_t num_nodes, coala_coloring_t** out) { // graph_coloring.cpp:317-326
    if (!out || num_nodes == 0) return fail(COALA_EINVAL, "bad arguments");
    coala_coloring* g = new (std::nothrow) coala_coloring();
    if (!g) return fail(COALA_ENOMEM, "out of host memory");
    g->num_nodes = num_nodes;
    g->hop.assign(num_nodes, 0);
    g->is_train.assign(num_nodes, 0);
    *out = g;
    return COALA_OK;
}

int coala_coloring_destroy(coala_coloring_t* g) {
    delete g;
    return COALA_OK;
}

int coala_coloring_set_adj_csc(coala_coloring_t* g, const int64_t* indptr, const int64_t* indices) { // :297-300
    if (!g || !indptr || !indices) return fail(COALA_EINVAL, "null argument");
    g->indptr = indptr;
    g->indices = indices;
    return COALA_OK;
}
int coala_coloring_set_color_buffer(coala_coloring_t* g, int64_t* color) { // :285-287
    if (!g || !color) return fail(COALA_EINVAL, "null argument");
    g->color = color;
    return COALA_OK;
}
int coala_coloring_set_topk_buffers(coala_coloring_t* g, int64_t* topk_color, double* topk_affinity, int topk) { // :289-295
    if (!g || topk < 1) return fail(COALA_EINVAL, "bad argument");
    if (topk_color) g->topk_color = topk_color;
    if (topk_affinity) g->topk_affinity = topk_affinity;
    g->topk = topk;
    return COALA_OK;
}

// cpu_color_graph_optimized (:108-160).  seed 1 == the reference's never-seeded rand() in a fresh process.
int coala_coloring_color_optimized(coala_coloring_t* g, const int64_t* train, uint64_t n_train, unsigned seed) {
    int rc = check_ready(g, false);
    if (rc) return rc;
    if (!train || n_train == 0) return fail(COALA_EINVAL, "no training nodes");
    for (uint64_t i = 0; i < n_train; ++i) { // :100-106
        if (train[i] < 0 || (uint64_t)train[i] >= g->num_nodes) return fail(COALA_ERANGE, "training node %lld out of range", (long long)train[i]);
        g->is_train[train[i]] = 1;
    }
    srand(seed);
    const double frac = std::min(20.0, (double)g->num_nodes / (double)n_train); // :77
    const double rate = g->sampling_rate * frac;                                  // :78 (float * double)
    for (uint64_t k = 0; k < n_train; ++k) {                                      // :81-92
        const int64_t i = train[k];
        if (g->color[i] == 0) {
            const float samp = static_cast<float>(rand()) / static_cast<float>(RAND_MAX); // float / int in the reference: same value
            if (samp <= rate) {
                g->buf[0].emplace_back((uint64_t)i, g->color_counter);
                g->color_counter++;
            }
        }
    }
    bfs_color(g, true);
    return COALA_OK;
}

// cpu_color_graph (:43-72): seeds sampled from ALL nodes at the base rate
int coala_coloring_color_all(coala_coloring_t* g, unsigned seed) {
    int rc = check_ready(g, false);
    if (rc) return rc;
    srand(seed);
    for (uint64_t i = 0; i < g->num_nodes; ++i) { // :3-16
        if (g->color[i] == 0) {
            const float samp = static_cast<float>(rand()) / static_cast<float>(RAND_MAX); // float / int in the reference: same value
            if (samp <= g->sampling_rate) {
                g->buf[0].emplace_back(i, g->color_counter);
                g->color_counter++;
            }
        }
    }
    bfs_color(g, false);
    return COALA_OK;
}

uint64_t coala_coloring_num_color(const coala_coloring_t* g) { return (!g || g->color_counter == 0) ? 0 : g->color_counter - 1; } // :303-306
uint64_t coala_coloring_num_color_node(const coala_coloring_t* g) { return g ? g->num_colored : 0; }                             // :308-311

// cpu_calculate_color_affinity (:254-294) when with_affinity != 0, cpu_count_nearest_color_less_memory (:213-247) otherwise.
// Like the reference, colours 1 .. num_colors-1 get a row; the LAST colour's row is left as the caller initialised it
// (the loop runs c in [0, num_c) and writes row c-1: SURVEY.md appendix A.9).
int coala_coloring_topk(coala_coloring_t* g, int with_affinity) {
    int rc = check_ready(g, true);
    if (rc) return rc;
    const uint64_t num_c = g->color_counter - 1;
    std::vector<std::vector<uint64_t>> nodes_of(num_c + 1);
    for (uint64_t n = 0; n < g->num_nodes; ++n) {
        const int64_t c = g->color[n];
        if (c < 0 || (uint64_t)c > num_c) return fail(COALA_ERANGE, "colour %lld of node %llu outside [0,%llu]", (long long)c, (unsigned long long)n, (unsigned long long)num_c);
        if (c != 0) nodes_of[c].push_back(n);
    }
    for (uint64_t c = 1; c < num_c; ++c) { // c == 0 has no nodes and would write row -1 in the reference
        std::unordered_map<uint64_t, double> aff;
        std::unordered_map<uint64_t, uint64_t> cnt;
        double neigh_count = 0;
        for (const uint64_t node : nodes_of[c]) {
            neigh_count += (double)(g->indptr[node + 1] - g->indptr[node]);
            for (int64_t e = g->indptr[node]; e < g->indptr[node + 1]; ++e) {
                const uint64_t nb = (uint64_t)g->indices[e];
                const uint64_t ncol = (uint64_t)g->color[nb];
                if (ncol != 0 && ncol != c) {
                    if (with_affinity) aff[ncol] += std::exp(-0.5 * (double)(int)g->hop[nb]); // score_func :250-252
                    else cnt[ncol] += 1;
                }
            }
        }
        if (with_affinity) {
            const auto tk = top_k<double>(aff, (size_t)g->topk);
            for (size_t i = 0; i < tk.size(); ++i) {
                g->topk_color[(c - 1) * g->topk + i] = (int64_t)tk[i].first;
                if (g->topk_affinity) g->topk_affinity[(c - 1) * g->topk + i] = tk[i].second / neigh_count;
            }
        } else {
            const auto tk = top_k<uint64_t>(cnt, (size_t)g->topk);
            for (size_t i = 0; i < tk.size(); ++i) g->topk_color[(c - 1) * g->topk + i] = (int64_t)tk[i].first;
        }
    }
    return COALA_OK;
}

// cpu_count_nearest_color (:183-209): edge counts between colours over ALL nodes; colour 0 (which the reference would write
// to row -1, out of bounds) is skipped, and here every colour 1..num_colors gets its row.
int coala_coloring_nearest(coala_coloring_t* g) {
    int rc = check_ready(g, true);
    if (rc) return rc;
    std::unordered_map<uint64_t, std::unordered_map<uint64_t, uint64_t>> conn;
    for (uint64_t node = 0; node < g->num_nodes; ++node) {
        const uint64_t c = (uint64_t)g->color[node];
        for (int64_t e = g->indptr[node]; e < g->indptr[node + 1]; ++e) {
            const uint64_t ncol = (uint64_t)g->color[g->indices[e]];
            if (ncol != 0 && c != ncol) conn[c][ncol] += 1;
        }
    }
    for (const auto& kv : conn) {
        if (kv.first == 0) continue;
        const auto tk = top_k<uint64_t>(kv.second, (size_t)g->topk);
        for (size_t i = 0; i < tk.size(); ++i) g->topk_color[(kv.first - 1) * g->topk + i] = (int64_t)tk[i].first;
    }
    return COALA_OK;
}

} // extern "C"
