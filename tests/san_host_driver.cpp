// AddressSanitizer / UBSan driver for the PRODUCT's host-side C++ (coala_host.cpp: .npy parser, node distributor;
// coala_coloring.cpp: the colouring tool) -- compiled with g++ and the sanitizers by tests/test_sanitize_cpu.py, no GPU needed
// (nothing here touches the device).  Inputs include the malformed ones the parsers must refuse.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/coala_hip.h"

static int fails = 0;
#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #c, coala_last_error()); \
            ++fails;                                                     \
        }                                                                \
    } while (0)

static std::string npy(const std::string& descr, const std::string& shape, int version, const void* data, size_t bytes, bool fortran = false) {
    std::string dict = "{'descr': '" + descr + "', 'fortran_order': " + (fortran ? "True" : "False") + ", 'shape': (" + shape + "), }";
    const size_t pre = version == 1 ? 10 : 12;
    while ((pre + dict.size() + 1) % 64) dict += ' ';
    dict += '\n';
    std::string out = "\x93NUMPY";
    out += (char)version;
    out += (char)0;
    const uint32_t hl = (uint32_t)dict.size();
    out += (char)(hl & 0xFF);
    out += (char)((hl >> 8) & 0xFF);
    if (version != 1) {
        out += (char)((hl >> 16) & 0xFF);
        out += (char)((hl >> 24) & 0xFF);
    }
    out += dict;
    out.append((const char*)data, bytes);
    return out;
}

static void write_file(const std::string& path, const std::string& bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    fwrite(bytes.data(), 1, bytes.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    // ---------------------------------------------------------------- .npy parser: good headers, then everything that can be wrong
    {
        std::vector<int64_t> v(12);
        for (size_t i = 0; i < v.size(); ++i) v[i] = (int64_t)i;
        int64_t shape[2] = {0, 0};
        int nd = 0;
        size_t off = 0;
        char descr[16];
        for (int ver = 1; ver <= 2; ++ver) {
            std::string a = npy("<i8", "12,", ver, v.data(), v.size() * 8);
            CHECK(coala_npy_parse(a.data(), a.size(), 1, shape, &nd, &off, descr, sizeof descr) == COALA_OK && nd == 1 && shape[0] == 12);
            CHECK(off % 64 == 0 && !memcmp(a.data() + off, v.data(), 96) && !strcmp(descr, "<i8"));
            std::string b = npy("<f8", "3, 4", ver, v.data(), v.size() * 8);
            CHECK(coala_npy_parse(b.data(), b.size(), 2, shape, &nd, &off, descr, sizeof descr) == COALA_OK && nd == 2 && shape[0] == 3 && shape[1] == 4);
            CHECK(coala_npy_parse(b.data(), b.size(), 1, shape, &nd, &off, descr, sizeof descr) == COALA_OK && nd == 0);  // other rank: empty shape
            for (size_t cut = 0; cut < 80 && cut < a.size(); cut += 3)                                     // truncated buffers of every length
                (void)coala_npy_parse(a.data(), cut, 1, shape, &nd, &off, descr, sizeof descr);
        }
        std::string bad = npy("<i8", "12,", 1, v.data(), 96);
        bad[0] = 'X';
        CHECK(coala_npy_parse(bad.data(), bad.size(), 1, shape, &nd, &off, descr, sizeof descr) != COALA_OK);   // magic
        bad = npy("<i8", "12,", 1, v.data(), 96);
        bad[8] = (char)0xFF; bad[9] = (char)0xFF;                                                              // header length past the buffer
        CHECK(coala_npy_parse(bad.data(), bad.size(), 1, shape, &nd, &off, descr, sizeof descr) != COALA_OK);
        bad = npy("<i8", "12,", 3, v.data(), 96);
        (void)coala_npy_parse(bad.data(), bad.size(), 1, shape, &nd, &off, descr, sizeof descr);                // unknown version
        bad = npy("<i8", "99999999999999999999999999,", 1, v.data(), 96);
        (void)coala_npy_parse(bad.data(), bad.size(), 1, shape, &nd, &off, descr, sizeof descr);                // a dimension that overflows
        bad = npy("<i8", "", 1, v.data(), 96);
        (void)coala_npy_parse(bad.data(), bad.size(), 1, shape, &nd, &off, descr, sizeof descr);                // 0-d
        char tiny[4];
        bad = npy("<i8", "12,", 1, v.data(), 96);
        (void)coala_npy_parse(bad.data(), bad.size(), 1, shape, &nd, &off, tiny, sizeof tiny);                  // descr buffer too small for "<i8"? (3 chars + NUL fits)
        (void)coala_npy_parse(bad.data(), bad.size(), 1, shape, &nd, &off, tiny, 2);
        CHECK(coala_npy_parse(nullptr, 0, 1, shape, &nd, &off, descr, sizeof descr) != COALA_OK);
        // seeded mutations of good headers (both versions, both ranks): whatever comes back, no out-of-bounds access and no overflow
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (int it = 0; it < 20000; ++it) {
            std::string m = npy(it & 1 ? "<i8" : "<f8", it & 2 ? "12," : "3, 4", 1 + ((it >> 2) & 1), v.data(), 96);
            const int edits = 1 + (it % 4);
            for (int e = 0; e < edits; ++e) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                const size_t at = (size_t)(x % 140) % m.size();
                const int kind = (int)((x >> 32) % 4);
                if (kind == 0) m[at] = (char)(x >> 40);
                else if (kind == 1) m[at] = "0123456789(),' :"[(x >> 40) % 16];
                else if (kind == 2) m.erase(at, 1 + (x >> 50) % 5);
                else m.insert(at, std::string(1 + (x >> 50) % 3, "9(,' "[(x >> 44) % 5]));
            }
            const size_t cut = (it % 7 == 0) ? (size_t)(x % (m.size() + 1)) : m.size();
            (void)coala_npy_parse(m.data(), cut, 1 + (it & 1), shape, &nd, &off, descr, 1 + (size_t)(x % 15));
        }
    }
    // ---------------------------------------------------------------- node distributor: good files, then a topk entry out of range
    {
        const int n_items = 4000, num_colors = 9, topk = 3, batch = 16, local = 2, nodes = 2;
        std::vector<int64_t> color(n_items), tk((num_colors + 1) * topk), items(batch * local * nodes * 3);
        std::vector<double> sc((num_colors + 1) * topk);
        for (int i = 0; i < n_items; ++i) color[i] = (i * 7) % (num_colors + 1);
        for (size_t i = 0; i < tk.size(); ++i) { tk[i] = (int64_t)((i * 5) % (num_colors + 1)); sc[i] = 1.0 / (double)(1 + i % topk); }
        for (size_t i = 0; i < items.size(); ++i) items[i] = (int64_t)((i * 37) % n_items);
        const std::string cf = tmp + "/san_color.npy", tf = tmp + "/san_topk.npy", sf = tmp + "/san_score.npy";
        write_file(cf, npy("<i8", std::to_string(n_items) + ",", 1, color.data(), color.size() * 8));
        write_file(tf, npy("<i8", std::to_string(num_colors + 1) + ", " + std::to_string(topk), 1, tk.data(), tk.size() * 8));
        write_file(sf, npy("<f8", std::to_string(num_colors + 1) + ", " + std::to_string(topk), 1, sc.data(), sc.size() * 8));
        coala_distributor_t* d = nullptr;
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) == COALA_OK && d);
        if (d) {
            const int entries = (int)coala_distributor_num_color_entries(d);
            CHECK(entries >= coala_distributor_num_colors(d) && coala_distributor_color_ptr(d) != nullptr);
            std::vector<int32_t> m0(entries, 0), m1(entries, 3);
            const int32_t* meta[2] = {m0.data(), m1.data()};
            std::vector<int64_t> out(batch * local);
            for (int step = 0; step < 3; ++step)
                CHECK(coala_distributor_assign(d, (uint64_t)step * batch * local * nodes, out.data(), meta, 2) == COALA_OK);
            CHECK(coala_distributor_assign(d, 0, out.data(), meta, 1) != COALA_OK);       // one counter array for two domains
            CHECK(coala_distributor_assign(d, 0, nullptr, meta, 2) != COALA_OK);
            CHECK(coala_distributor_destroy(d) == COALA_OK);
        }
        tk[4] = num_colors + 50;                                                          // a neighbour colour that indexes past the counters
        write_file(tf, npy("<i8", std::to_string(num_colors + 1) + ", " + std::to_string(topk), 1, tk.data(), tk.size() * 8));
        d = nullptr;
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) != COALA_OK && d == nullptr);
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, (tmp + "/missing.npy").c_str(), tf.c_str(), sf.c_str(), &d) != COALA_OK);
        write_file(tf, "\x93NUMPY garbage");                                              // a file that is not a .npy at all
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) != COALA_OK);
        // files whose headers promise more than they hold, or disagree with each other
        tk[4] = 1;
        write_file(tf, npy("<i8", std::to_string(num_colors + 1) + ", " + std::to_string(topk), 1, tk.data(), tk.size() * 8));
        write_file(cf, npy("<i8", std::to_string(n_items) + ",", 1, color.data(), 100 * 8));                      // 4000 promised, 100 present
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) != COALA_OK);
        write_file(cf, npy("<i4", std::to_string(n_items) + ",", 1, color.data(), color.size() * 4));             // wrong element type
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) != COALA_OK);
        write_file(cf, npy("<i8", std::to_string(n_items) + ",", 1, color.data(), color.size() * 8));
        write_file(sf, npy("<f8", std::to_string(num_colors + 1) + ", " + std::to_string(topk - 1), 1, sc.data(), (num_colors + 1) * (topk - 1) * 8));
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) != COALA_OK);   // topk [10,3] vs score [10,2]
        write_file(sf, npy("<f8", std::to_string(num_colors + 1) + ", " + std::to_string(topk), 1, sc.data(), sc.size() * 8));
        for (int64_t bad_color : {(int64_t)-4, (int64_t)num_colors + 2}) {                                        // a colour outside [0, rows of topk]: refused when a step meets it
            color[17] = bad_color;
            write_file(cf, npy("<i8", std::to_string(n_items) + ",", 1, color.data(), color.size() * 8));
            items[3] = 17;
            d = nullptr;
            CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) == COALA_OK && d);
            if (d) {
                const int entries = (int)coala_distributor_num_color_entries(d);
                std::vector<int32_t> m0(entries, 0), m1(entries, 0);
                const int32_t* meta[2] = {m0.data(), m1.data()};
                std::vector<int64_t> out(batch * local);
                CHECK(coala_distributor_assign(d, 0, out.data(), meta, 2) != COALA_OK);
                CHECK(coala_distributor_destroy(d) == COALA_OK);
            }
            items[3] = (3 * 37) % n_items;
        }
        color[17] = 3;
        write_file(cf, npy("<i8", std::to_string(n_items) + ",", 1, color.data(), color.size() * 8));
        items[5] = n_items + 123;                                                                                // an item id past the colour array
        d = nullptr;
        CHECK(coala_distributor_create(items.data(), 0, batch, local, nodes, cf.c_str(), tf.c_str(), sf.c_str(), &d) == COALA_OK && d);
        if (d) {
            const int entries = (int)coala_distributor_num_color_entries(d);
            std::vector<int32_t> m0(entries, 0), m1(entries, 0);
            const int32_t* meta[2] = {m0.data(), m1.data()};
            std::vector<int64_t> out(batch * local);
            CHECK(coala_distributor_assign(d, 0, out.data(), meta, 2) != COALA_OK);
            CHECK(coala_distributor_destroy(d) == COALA_OK);
        }
        items[5] = 5;
        CHECK(coala_distributor_create_plain(items.data(), nodes, &d) == COALA_OK);
        CHECK(coala_distributor_destroy(d) == COALA_OK);
        remove(cf.c_str()); remove(tf.c_str()); remove(sf.c_str());
    }
    // ---------------------------------------------------------------- colouring: a small graph through every entry point
    {
        const int64_t n = 3000;
        std::vector<int64_t> indptr(n + 1, 0), indices;
        uint64_t x = 88172645463325252ull;
        for (int64_t v = 0; v < n; ++v) {
            const int deg = (int)(v % 7);                               // includes isolated nodes
            for (int k = 0; k < deg; ++k) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                indices.push_back((int64_t)(x % (uint64_t)n));
            }
            indptr[v + 1] = (int64_t)indices.size();
        }
        std::vector<int64_t> train(n / 2);
        for (size_t i = 0; i < train.size(); ++i) train[i] = (int64_t)(i * 2);
        for (int mode = 0; mode < 2; ++mode) {
            coala_coloring_t* g = nullptr;
            CHECK(coala_coloring_create((uint64_t)n, &g) == COALA_OK && g);
            std::vector<int64_t> color(n, 0);
            CHECK(coala_coloring_set_adj_csc(g, indptr.data(), indices.data()) == COALA_OK);
            CHECK(coala_coloring_set_color_buffer(g, color.data()) == COALA_OK);
            if (mode == 0) CHECK(coala_coloring_color_optimized(g, train.data(), train.size(), 1) == COALA_OK);
            else CHECK(coala_coloring_color_all(g, 1) == COALA_OK);
            const uint64_t nc = coala_coloring_num_color(g);
            CHECK(nc > 0 && coala_coloring_num_color_node(g) <= (uint64_t)n);
            const int topk = 5;
            std::vector<int64_t> tk(nc * topk, 0);
            std::vector<double> sc(nc * topk, 0.0);
            CHECK(coala_coloring_set_topk_buffers(g, tk.data(), sc.data(), topk) == COALA_OK);
            CHECK(coala_coloring_topk(g, mode) == COALA_OK);
            CHECK(coala_coloring_nearest(g) == COALA_OK);
            for (auto c : color) CHECK(c >= 0 && (uint64_t)c <= nc);
            for (auto t : tk) CHECK(t >= 0 && (uint64_t)t <= nc);
            CHECK(coala_coloring_destroy(g) == COALA_OK);
        }
        coala_coloring_t* g = nullptr;
        CHECK(coala_coloring_create(10, &g) == COALA_OK);
        CHECK(coala_coloring_color_all(g, 1) != COALA_OK);                                // no graph, no buffer set
        CHECK(coala_coloring_destroy(g) == COALA_OK);
    }
    if (fails) {
        fprintf(stderr, "%d checks failed\n", fails);
        return 1;
    }
    printf("sanitized host run ok\n");
    return 0;
}
