// oracle/ref_coloring_wrap.cpp -- C entry points around the REFERENCE's own Graph_Coloring class, compiled together with
// /root/reference/COALA_GNN_Modules/graph_coloring.cpp where it lies (oracle/ref_build.py).  TEST INFRASTRUCTURE ONLY: it
// exists to generate golden vectors for the colouring row from the real reference.  Nothing of
// the reference is copied here: this file only calls its public methods (graph_coloring.h:15-68).
#include "graph_coloring.h"

#include <cstdlib>

extern "C" {

void* ref_gc_create(uint64_t num_nodes, int topk) {
    Graph_Coloring* g = new Graph_Coloring(num_nodes);
    g->topk = topk; // graph_coloring.h:23 (public member, default 10)
    return g;
}

void ref_gc_destroy(void* h) { delete static_cast<Graph_Coloring*>(h); }

// examples/color_info_gen/generate_color_data.py:20-37 : set_adj_csc, set_color_buffer, cpu_color_graph_optimized
void ref_gc_color(void* h, const uint64_t* indptr, const uint64_t* indices, uint64_t* color_buf, const int64_t* train,
                  uint64_t n_train, unsigned seed) {
    Graph_Coloring* g = static_cast<Graph_Coloring*>(h);
    srand(seed); // the reference never seeds: glibc's rand() then starts as after srand(1)
    g->set_adj_csc((uint64_t)indptr, (uint64_t)indices);
    g->set_color_buffer((uint64_t)color_buf);
    g->cpu_color_graph_optimized((uint64_t)train, n_train);
}

uint64_t ref_gc_num_color(void* h) { return static_cast<Graph_Coloring*>(h)->get_num_color(); }
uint64_t ref_gc_num_color_node(void* h) { return static_cast<Graph_Coloring*>(h)->get_num_color_node(); }

// generate_color_data.py:45-54 : set_topk_color_buffer, set_topk_affinity_buffer, cpu_calculate_color_affinity
void ref_gc_affinity(void* h, uint64_t* topk_color, double* topk_affinity) {
    Graph_Coloring* g = static_cast<Graph_Coloring*>(h);
    g->set_topk_color_buffer((uint64_t)topk_color);
    g->set_topk_affinity_buffer((uint64_t)topk_affinity);
    g->cpu_calculate_color_affinity();
}

} // extern "C"
