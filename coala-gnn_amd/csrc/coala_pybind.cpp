// coala_pybind.cpp -- the compiled binding a maintainer of the reference would ship in place of
// COALA_GNN_Modules/COALA_GNN_Pybind.cu:27-79: the same seven classes, constructor argument order and method names, on top of the
// C ABI of libcoala_hip.so (include/coala_hip.h).  Pointers still cross as integers, exactly as in the reference.  Every call
// that touches the GPU or blocks releases the GIL.  Extras next to the reference surface are keyword arguments with defaults
// (num_rows, rank, flags, stream, ...), so the reference's own call sites work unchanged.
//
// Built by coala-gnn_amd/build.py (host C++ only; links libcoala_hip.so) into COALA_GNN_Pybind/_coala_pybind*.so; the package
// COALA_GNN_Pybind uses it for its per-step calls and falls back to its ctypes table when the module is absent.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/coala_hip.h"

namespace py = pybind11;

namespace {

void check(int rc) {
    if (rc != COALA_OK) {
        const char* m = coala_last_error();
        throw std::runtime_error(std::string("libcoala_hip: ") + (m ? m : "") + " (code " + std::to_string(rc) + ")");
    }
}
template <typename T> T* ptr(uint64_t p) { return reinterpret_cast<T*>(static_cast<uintptr_t>(p)); }
void* stream_of(uint64_t s) { return reinterpret_cast<void*>(static_cast<uintptr_t>(s)); }

// shared_UVA.cuh:26-115
struct SharedUVAManager {
    coala_shm_t* h = nullptr;
    bool creator = false;
    SharedUVAManager(const std::string& path, int64_t size, int node, int64_t /*global_comm*/, int64_t /*local_comm*/, int local_rank, int device) {
        creator = local_rank == 0;
        py::gil_scoped_release nogil;
        check(coala_shm_open(path.c_str(), (uint64_t)size, creator ? 1 : 0, device < 0 ? local_rank : device, &h));
        (void)node;
    }
    ~SharedUVAManager() { cleanup(); }
    uint64_t get_host_ptr() const { return (uint64_t)(uintptr_t)coala_shm_host_ptr(h); }
    uint64_t get_device_ptr() const { return (uint64_t)(uintptr_t)coala_shm_device_ptr(h); }
    void cleanup() {
        if (h) coala_shm_close(h, creator ? 1 : 0);
        h = nullptr;
    }
};

// ssd_gnn_cache.cuh:10-55
struct SSD_GNN_SSD_Controllers {
    uint32_t n_ctrls, cudaDevice;
    uint64_t num_elements, offset;
    int dim, cache_dim, page_size;
    bool SSD_SIM;
    SSD_GNN_SSD_Controllers(uint32_t n, uint32_t /*p_size: overwritten in the reference too, :47*/, uint64_t n_elems, uint64_t read_off, uint32_t device,
                            int feat_dim, bool sim)
        : n_ctrls(n), cudaDevice(device), num_elements(n_elems), offset(read_off), dim(feat_dim), SSD_SIM(sim) {
        cache_dim = coala_cache_dim(feat_dim);
        if (cache_dim < 0) check(cache_dim); // "Only Feature Embedding Size less than 8KB is supported"
        page_size = cache_dim * 4;
    }
};

// node_distributor_pybind.cuh:112-238
struct Node_distributor_pybind {
    coala_distributor_t* h = nullptr;
    Node_distributor_pybind(uint64_t items, int n_nodes) { check(coala_distributor_create_plain(ptr<const int64_t>(items), n_nodes, &h)); }
    Node_distributor_pybind(uint64_t items, int node_id, int batch, int local_size, int n_nodes, const std::string& color, const std::string& topk,
                            const std::string& score) {
        check(coala_distributor_create(ptr<const int64_t>(items), node_id, batch, local_size, n_nodes, color.c_str(), topk.c_str(), score.c_str(), &h));
    }
    ~Node_distributor_pybind() { coala_distributor_destroy(h); }
    void distribute_node_with_affinity(uint64_t offset, uint64_t out, const std::vector<uint64_t>& meta) {
        std::vector<const int32_t*> m;
        for (uint64_t p : meta) m.push_back(ptr<const int32_t>(p));
        py::gil_scoped_release nogil;
        check(coala_distributor_assign(h, offset, ptr<int64_t>(out), m.data(), (int)m.size()));
    }
    int get_num_colors() const { return coala_distributor_num_colors(h); }
    uint64_t get_color_buffer_ptr() const { return (uint64_t)(uintptr_t)coala_distributor_color_ptr(h); }
    int64_t get_num_color_entries() const { return coala_distributor_num_color_entries(h); }
};

// ssd_gnn_cache.cuh:58-196 / 201-371: one handle type, two constructors' worth of flags
struct Cache {
    coala_cache_t* h = nullptr;
    int dim = 0, num_color = 0, global_rank = 0, local_rank = 0, n_gpus = 1;
    py::object keepalive; // the distributor object (the reference borrows its colour array: ssd_gnn_cache.cuh:91,234)
    Cache(const SSD_GNN_SSD_Controllers& c, py::object distributor, int g_rank, int n_gpus_, uint64_t cache_mb, uint64_t sim_buf, bool distributed,
          uint64_t num_rows, int rank, uint32_t extra_flags, uint64_t max_batch) {
        if (!sim_buf) throw std::runtime_error("sim_buf is 0: the NVMe/BaM storage tier is out of scope on this platform; pass the pinned-host feature table");
        coala_cache_config_t cfg{};
        cfg.device = (int32_t)c.cudaDevice;
        cfg.dim = c.dim;
        cfg.cache_mb = cache_mb;
        cfg.n_gpus = n_gpus_;
        cfg.rank = rank < 0 ? (int)(c.cudaDevice % (uint32_t)(n_gpus_ > 0 ? n_gpus_ : 1)) : rank;
        cfg.global_rank = g_rank;
        cfg.flags = extra_flags | (distributed ? COALA_FLAG_DISTRIBUTED : 0u);
        cfg.cold_table = ptr<const float>(sim_buf);
        int64_t entries = 0;
        if (!distributor.is_none()) {
            auto& d = distributor.cast<Node_distributor_pybind&>();
            cfg.node_color = coala_distributor_color_ptr(d.h);
            cfg.num_colors = coala_distributor_num_colors(d.h);
            entries = coala_distributor_num_color_entries(d.h);
            keepalive = distributor;
        }
        cfg.num_rows = num_rows ? num_rows : (uint64_t)entries;
        if (!cfg.num_rows) throw std::runtime_error("number of feature rows unknown: pass num_rows= or a distributor built with a colour file");
        cfg.max_batch = max_batch;
        dim = c.dim;
        num_color = cfg.num_colors;
        global_rank = g_rank;
        local_rank = cfg.rank;
        n_gpus = n_gpus_;
        py::gil_scoped_release nogil;
        check(coala_cache_create(&cfg, &h));
    }
    ~Cache() { close(); }
    void close() {
        if (h) {
            py::gil_scoped_release nogil;
            coala_cache_destroy(h);
        }
        h = nullptr;
    }
    uint64_t handle() const { return (uint64_t)(uintptr_t)h; }
    void read_feature(uint64_t out, uint64_t idx, int64_t n, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_read_feature(h, ptr<float>(out), ptr<const int64_t>(idx), n, stream_of(stream)));
    }
    void serve(uint64_t out, uint64_t ids, int64_t n, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_serve(h, ptr<float>(out), ptr<const int64_t>(ids), n, stream_of(stream)));
    }
    void get_cache_data(uint64_t dst, int n_entries, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_color_counts(h, ptr<int32_t>(dst), n_entries < 0 ? num_color : n_entries, stream_of(stream)));
    }
    py::tuple stats(bool reset, uint64_t stream) {
        uint64_t hit = 0, miss = 0, bad = 0;
        {
            py::gil_scoped_release nogil;
            check(coala_cache_stats(h, &hit, &miss, &bad, reset ? 1 : 0, stream_of(stream)));
        }
        return py::make_tuple(hit, miss, bad);
    }
    void print_stats(uint64_t stream) { // isolated_cache.h:132-141 (prints, then resets)
        py::tuple s = stats(true, stream);
        const uint64_t hit = s[0].cast<uint64_t>(), miss = s[1].cast<uint64_t>();
        py::print("Global Rank:", global_rank, "Local Rank:", local_rank, "hit count:", hit, "miss count:", miss);
        py::print("Global Rank:", global_rank, "Local Rank:", local_rank, " GPU hit ratio:", hit + miss ? (double)hit / (double)(hit + miss) : 0.0);
    }
    // ssd_gnn_cache.cuh:283-295 ([G][max_sample] layout)
    void split_node_list(uint64_t idx, int64_t n, uint64_t node, uint64_t map, uint64_t counter, int local_size, int max_sample, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_route(h, ptr<const int64_t>(idx), n, local_size, max_sample, ptr<int64_t>(node), ptr<int64_t>(map), ptr<int64_t>(counter), nullptr,
                                stream_of(stream)));
    }
    // ssd_gnn_cache.cuh:297-325: one batch per peer buffer, in peer order (one batch when the buffers are contiguous)
    void nccl_get_feature(const std::vector<uint64_t>& idx_list, const std::vector<uint64_t>& ret_list, const std::vector<int64_t>& sizes, int local_size,
                          int /*max_sample*/, uint64_t stream) {
        bool contiguous = true;
        int64_t total = 0;
        for (int i = 0; i < local_size; ++i) {
            if (i + 1 < local_size && (idx_list[i + 1] != idx_list[i] + (uint64_t)sizes[i] * 8 || ret_list[i + 1] != ret_list[i] + (uint64_t)sizes[i] * dim * 4))
                contiguous = false;
            total += sizes[i];
        }
        py::gil_scoped_release nogil;
        if (contiguous && local_size > 0) {
            check(coala_cache_serve(h, ptr<float>(ret_list[0]), ptr<const int64_t>(idx_list[0]), total, stream_of(stream)));
        } else {
            for (int i = 0; i < local_size; ++i) check(coala_cache_serve(h, ptr<float>(ret_list[i]), ptr<const int64_t>(idx_list[i]), sizes[i], stream_of(stream)));
        }
    }
    // ssd_gnn_cache.cuh:327-356
    void map_feat_data(uint64_t out, const std::vector<uint64_t>& ret_list, uint64_t meta, const std::vector<int64_t>& sizes, int local_size, int max_sample,
                       uint64_t stream) {
        py::gil_scoped_release nogil;
        for (int i = 0; i < local_size; ++i)
            check(coala_cache_scatter(h, ptr<float>(out), ptr<const float>(ret_list[i]), ptr<const int64_t>(meta + (uint64_t)i * (uint64_t)max_sample * 8), sizes[i],
                                      stream_of(stream)));
    }
};
struct Isolated_Cache : Cache {
    using Cache::Cache;
};
struct SSD_GNN_NVSHMEM_Cache : Cache {
    using Cache::Cache;
    // ssd_gnn_cache.cuh:111-174.  NVSHMEM's device-initiated transport does not exist on this platform: the Python package attaches
    // an exchange object (RCCL all-to-all-v) and forwards these two calls to it.
    py::object exchange;
    void send_requests(uint64_t idx, int64_t n, uint64_t req, int64_t max_index) {
        if (exchange.is_none() || !exchange) throw std::runtime_error("SSD_GNN_NVSHMEM_Cache.send_requests: no exchange attached (COALA_GNN_Manager installs the RCCL all-to-all-v exchange)");
        exchange.attr("send_requests")(py::cast(this), idx, n, req, max_index);
    }
    void read_feature_x(uint64_t out, uint64_t req, int64_t max_index) {
        if (exchange.is_none() || !exchange) throw std::runtime_error("SSD_GNN_NVSHMEM_Cache.read_feature: no exchange attached");
        exchange.attr("read_feature")(py::cast(this), out, req, max_index);
    }
};

// nvshmem_manager.cuh:9-54: the "symmetric heap" is ordinary device-visible memory here (RCCL needs none); pinned host memory mapped
// into the device keeps this module free of a HIP dependency
struct NVSHMEM_Manager {
    int local_rank;
    std::vector<std::pair<uint64_t, void*>> bufs;
    NVSHMEM_Manager(int64_t /*local_comm*/, int rank) : local_rank(rank) {}
    ~NVSHMEM_Manager() { finalize(); }
    uint64_t allocate(int64_t size) {
        if (size <= 0) throw std::runtime_error("NVSHMEM_Manager.allocate: size must be positive");
        void *hp = nullptr, *dp = nullptr;
        check(coala_pinned_alloc((uint64_t)size, local_rank, &hp, &dp));
        bufs.emplace_back((uint64_t)(uintptr_t)dp, hp);
        return (uint64_t)(uintptr_t)dp;
    }
    void free(uint64_t p) {
        for (auto it = bufs.begin(); it != bufs.end(); ++it)
            if (it->first == p) {
                coala_pinned_free(it->second);
                bufs.erase(it);
                return;
            }
    }
    void finalize() {
        for (auto& b : bufs) coala_pinned_free(b.second);
        bufs.clear();
    }
};

// graph_coloring.h:15-68
struct Graph_Coloring {
    coala_coloring_t* h = nullptr;
    int topk;
    unsigned seed;
    Graph_Coloring(uint64_t n, int topk_, unsigned seed_) : topk(topk_), seed(seed_) { check(coala_coloring_create(n, &h)); }
    ~Graph_Coloring() { coala_coloring_destroy(h); }
    void set_adj_csc(uint64_t indptr, uint64_t indices) { check(coala_coloring_set_adj_csc(h, ptr<const int64_t>(indptr), ptr<const int64_t>(indices))); }
    void set_color_buffer(uint64_t p) { check(coala_coloring_set_color_buffer(h, ptr<int64_t>(p))); }
    void set_topk_color_buffer(uint64_t p) { check(coala_coloring_set_topk_buffers(h, ptr<int64_t>(p), nullptr, topk)); }
    void set_topk_affinity_buffer(uint64_t p) { check(coala_coloring_set_topk_buffers(h, nullptr, ptr<double>(p), topk)); }
    void cpu_color_graph() { py::gil_scoped_release nogil; check(coala_coloring_color_all(h, seed)); }
    void cpu_color_graph_optimized(uint64_t train, uint64_t n) { py::gil_scoped_release nogil; check(coala_coloring_color_optimized(h, ptr<const int64_t>(train), n, seed)); }
    void cpu_count_nearest_color() { py::gil_scoped_release nogil; check(coala_coloring_nearest(h)); }
    void cpu_count_nearest_color_less_memory() { py::gil_scoped_release nogil; check(coala_coloring_topk(h, 0)); }
    void cpu_calculate_color_affinity() { py::gil_scoped_release nogil; check(coala_coloring_topk(h, 1)); }
    uint64_t get_num_color() const { return coala_coloring_num_color(h); }
    uint64_t get_num_color_node() const { return coala_coloring_num_color_node(h); }
};

template <typename C>
py::class_<C> bind_cache(py::module_& m, const char* name, bool distributed) {
    py::class_<C> cls(m, name);
    cls.def(py::init([distributed](const SSD_GNN_SSD_Controllers& c, py::object nd, int g_rank, int n_gpus, uint64_t cache_mb, uint64_t sim_b, uint64_t num_rows,
                                   int rank, bool sync, bool profile, bool cold_partitioned, uint64_t max_batch) {
                const uint32_t flags = (sync ? COALA_FLAG_SYNC : 0u) | (profile ? COALA_FLAG_PROFILE : 0u) | (cold_partitioned ? COALA_FLAG_COLD_PARTITIONED : 0u);
                return new C(c, nd, g_rank, n_gpus, cache_mb, sim_b, distributed, num_rows, rank, flags, max_batch);
            }),
            py::arg("SSD_Controllers"), py::arg("node_distributer"), py::arg("g_rank"), py::arg("n_gpus"), py::arg("cache_size"), py::arg("sim_b"),
            py::arg("num_rows") = 0, py::arg("rank") = -1, py::arg("sync") = true, py::arg("profile") = false, py::arg("cold_partitioned") = false,
            py::arg("max_batch") = 0)
        .def("get_cache_data", &C::get_cache_data, py::arg("ret_i_ptr"), py::arg("n_entries") = -1, py::arg("stream") = 0)
        .def("print_stats", &C::print_stats, py::arg("stream") = 0)
        .def("stats", &C::stats, py::arg("reset") = false, py::arg("stream") = 0)
        .def("serve", &C::serve, py::arg("out"), py::arg("ids"), py::arg("n"), py::arg("stream") = 0)
        .def("split_node_list", &C::split_node_list, py::arg("i_index_ptr"), py::arg("index_size"), py::arg("i_node_tensor"), py::arg("i_map_tensor"),
             py::arg("i_counter_tensor"), py::arg("local_size"), py::arg("max_sample"), py::arg("stream") = 0)
        .def("nccl_get_feature", &C::nccl_get_feature, py::arg("i_index_ptr_list"), py::arg("i_return_ptr_list"), py::arg("index_size_list"), py::arg("local_size"),
             py::arg("max_sample"), py::arg("stream") = 0)
        .def("map_feat_data", &C::map_feat_data, py::arg("i_return_ptr"), py::arg("i_return_ptr_list"), py::arg("i_meta_buffer"), py::arg("index_size_list"),
             py::arg("local_size"), py::arg("max_sample"), py::arg("stream") = 0)
        .def("handle", &C::handle)
        .def("close", &C::close)
        .def_readonly("dim", &C::dim)
        .def_readonly("num_color", &C::num_color);
    return cls;
}

} // namespace

PYBIND11_MODULE(_coala_pybind, m) {
    m.doc() = "COALA_GNN_Pybind on the MI355X C ABI (libcoala_hip.so): compiled binding, GIL released around every native call";
    m.def("abi_version", &coala_abi_version);
    m.def("last_error", []() { return std::string(coala_last_error() ? coala_last_error() : ""); });

    py::class_<SharedUVAManager>(m, "SharedUVAManager")
        .def(py::init<const std::string&, int64_t, int, int64_t, int64_t, int, int>(), py::arg("path"), py::arg("shm_size"), py::arg("node") = 0,
             py::arg("global_comm_ptr") = 0, py::arg("local_comm_ptr") = 0, py::arg("local_rank") = 0, py::arg("device") = -1)
        .def("get_host_ptr", &SharedUVAManager::get_host_ptr)
        .def("get_device_ptr", &SharedUVAManager::get_device_ptr)
        .def("cleanup", &SharedUVAManager::cleanup);

    py::class_<SSD_GNN_SSD_Controllers>(m, "SSD_GNN_SSD_Controllers")
        .def(py::init<uint32_t, uint32_t, uint64_t, uint64_t, uint32_t, int, bool>())
        .def_readonly("cache_dim", &SSD_GNN_SSD_Controllers::cache_dim)
        .def_readonly("page_size", &SSD_GNN_SSD_Controllers::page_size)
        .def_readonly("cudaDevice", &SSD_GNN_SSD_Controllers::cudaDevice)
        .def_readonly("dim", &SSD_GNN_SSD_Controllers::dim);

    py::class_<Node_distributor_pybind>(m, "Node_distributor_pybind")
        .def(py::init<uint64_t, int>())
        .def(py::init<uint64_t, int, int, int, int, const std::string&, const std::string&, const std::string&>())
        .def("distribute_node_with_affinity", &Node_distributor_pybind::distribute_node_with_affinity)
        .def("get_num_colors", &Node_distributor_pybind::get_num_colors)
        .def("get_color_buffer_ptr", &Node_distributor_pybind::get_color_buffer_ptr)
        .def("get_num_color_entries", &Node_distributor_pybind::get_num_color_entries);

    bind_cache<Isolated_Cache>(m, "Isolated_Cache", false)
        .def("read_feature", &Isolated_Cache::read_feature, py::arg("i_return_tensor_ptr"), py::arg("i_index_ptr"), py::arg("max_index"), py::arg("stream") = 0);
    bind_cache<SSD_GNN_NVSHMEM_Cache>(m, "SSD_GNN_NVSHMEM_Cache", true)
        .def("attach_exchange", [](SSD_GNN_NVSHMEM_Cache& c, py::object x) { c.exchange = x; })
        .def("send_requests", &SSD_GNN_NVSHMEM_Cache::send_requests)
        .def("read_feature", &SSD_GNN_NVSHMEM_Cache::read_feature_x);

    py::class_<NVSHMEM_Manager>(m, "NVSHMEM_Manager")
        .def(py::init<int64_t, int>(), py::arg("local_comm_ptr") = 0, py::arg("local_rank") = 0)
        .def("allocate", &NVSHMEM_Manager::allocate)
        .def("free", &NVSHMEM_Manager::free)
        .def("finalize", &NVSHMEM_Manager::finalize);

    py::class_<Graph_Coloring>(m, "Graph_Coloring")
        .def(py::init<uint64_t, int, unsigned>(), py::arg("num_nodes"), py::arg("topk") = 10, py::arg("seed") = 1)
        .def("cpu_color_graph", &Graph_Coloring::cpu_color_graph)
        .def("cpu_color_graph_optimized", &Graph_Coloring::cpu_color_graph_optimized)
        .def("cpu_count_nearest_color", &Graph_Coloring::cpu_count_nearest_color)
        .def("cpu_count_nearest_color_less_memory", &Graph_Coloring::cpu_count_nearest_color_less_memory)
        .def("cpu_calculate_color_affinity", &Graph_Coloring::cpu_calculate_color_affinity)
        .def("set_color_buffer", &Graph_Coloring::set_color_buffer)
        .def("set_topk_color_buffer", &Graph_Coloring::set_topk_color_buffer)
        .def("set_topk_affinity_buffer", &Graph_Coloring::set_topk_affinity_buffer)
        .def("set_adj_csc", &Graph_Coloring::set_adj_csc)
        .def("get_num_color_node", &Graph_Coloring::get_num_color_node)
        .def("get_num_color", &Graph_Coloring::get_num_color);

    // the per-step calls of the Python package, flat (handles and pointers as integers), GIL released
    m.def("cache_read_feature", [](uint64_t h, uint64_t out, uint64_t idx, int64_t n, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_read_feature(ptr<coala_cache_t>(h), ptr<float>(out), ptr<const int64_t>(idx), n, stream_of(stream)));
    });
    m.def("cache_serve", [](uint64_t h, uint64_t out, uint64_t ids, int64_t n, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_serve(ptr<coala_cache_t>(h), ptr<float>(out), ptr<const int64_t>(ids), n, stream_of(stream)));
    });
    m.def("cache_fetch_distributed", [](uint64_t h, uint64_t c, uint64_t out, uint64_t idx, int64_t n, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_fetch_distributed(ptr<coala_cache_t>(h), ptr<coala_comm_t>(c), ptr<float>(out), ptr<const int64_t>(idx), n, stream_of(stream)));
    });
    m.def("cache_fetch_distributed_bucketed", [](uint64_t h, uint64_t c, uint64_t out, uint64_t idx, int64_t n, uint64_t counts, uint64_t stream) {
        py::gil_scoped_release nogil;
        check(coala_cache_fetch_distributed_bucketed(ptr<coala_cache_t>(h), ptr<coala_comm_t>(c), ptr<float>(out), ptr<const int64_t>(idx), n,
                                                     ptr<const int64_t>(counts), stream_of(stream)));
    });
}
