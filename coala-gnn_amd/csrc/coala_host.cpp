// coala_host.cpp -- host-side pieces of libcoala_hip.so: error channel, shared pinned-host regions, .npy reader and the
// colour-affinity node distributor.  C ABI in include/coala_hip.h.
//
// Replaces (paths relative to /root/reference/COALA_GNN_Modules):
//   shared_UVA.cuh:26-115              SharedUVAManager           -> coala_shm_*
//   node_distributor_pybind.cuh:11-109 load_file_to_memory / parse_numpy_file -> coala_npy_parse + file loader
//   node_distributor_pybind.cuh:112-238 Node_distributor_pybind    -> coala_distributor_*
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <new>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../include/coala_hip.h"
#include "coala_internal.h"

namespace {
thread_local char g_err[512];
}

int coala_fail_(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define fail coala_fail_
#define HIPCHK COALA_HIPCHK

// ------------------------------------------------------------------------------------------------ shared pinned host
struct coala_shm {
    std::string name;
    uint64_t bytes = 0;
    int fd = -1;
    void* host = nullptr;
    void* dev = nullptr;
    bool registered = false;
    bool creator = false;
    int device = 0;
};

// ------------------------------------------------------------------------------------------------ distributor
struct coala_distributor {
    const int64_t* items = nullptr;
    int node_id = 0, num_nodes = 1, local_size = 1;
    int batch_size = 0, domain_batch_size = 0, global_batch_size = 0;
    bool use_color = false;
    std::vector<char> color_file, topk_file, score_file; // raw file images (the reference keeps them in pinned memory)
    const int64_t* color = nullptr;
    int64_t num_color_entries = 0;
    const int64_t* topk = nullptr;
    const double* score = nullptr;
    int num_colors = 0;
    int topk_k = 0;
};

namespace {

int load_file(const char* path, std::vector<char>& buf) { // node_distributor_pybind.cuh:11-35
    FILE* f = fopen(path, "rb");
    if (!f) return fail(COALA_EIO, "Unable to open file: %s", path);
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return fail(COALA_EIO, "seek failed: %s", path); }
    long sz = ftell(f);
    if (sz < 0) { fclose(f); return fail(COALA_EIO, "tell failed: %s", path); }
    rewind(f);
    buf.resize((size_t)sz);
    if (sz > 0 && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        return fail(COALA_EIO, "Error reading the file: %s", path);
    }
    fclose(f);
    return COALA_OK;
}

const char* find_in(const char* hay, size_t hlen, const char* needle) {
    const size_t nlen = strlen(needle);
    if (nlen > hlen) return nullptr;
    for (size_t i = 0; i + nlen <= hlen; ++i)
        if (memcmp(hay + i, needle, nlen) == 0) return hay + i;
    return nullptr;
}
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

} // namespace

extern "C" {

const char* coala_last_error(void) { return g_err; }
int coala_abi_version(void) { return 4; }

// ------------------------------------------------------------------------------------------------ shm
int coala_shm_open(const char* name, uint64_t bytes, int is_creator, int device, coala_shm_t** out) {
    if (!name || !out || bytes == 0) return fail(COALA_EINVAL, "bad shm arguments");
    *out = nullptr;
    HIPCHK(hipSetDevice(device)); // shared_UVA.cuh:67
    coala_shm* s = new (std::nothrow) coala_shm();
    if (!s) return fail(COALA_ENOMEM, "out of host memory");
    s->name = name;
    s->bytes = bytes;
    s->creator = is_creator != 0;
    s->device = device;
    int rc = COALA_OK;
    do {
        if (is_creator) { // shared_UVA.cuh:69-76
            // a fresh object, never the leftovers of a crashed run: the reference memsets the mapping (shared_UVA.cuh:99); a
            // new POSIX shm object is zero-filled by the kernel, so unlink any stale one and create exclusively
            (void)shm_unlink(name);
            s->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, S_IRUSR | S_IWUSR);
            if (s->fd < 0) { rc = fail(COALA_EIO, "shm_open(%s) failed: %s", name, strerror(errno)); break; }
            if (ftruncate(s->fd, (off_t)bytes) != 0) { rc = fail(COALA_EIO, "ftruncate(%llu) failed: %s", (unsigned long long)bytes, strerror(errno)); break; }
        } else { // shared_UVA.cuh:78-81
            s->fd = shm_open(name, O_RDWR, S_IRUSR | S_IWUSR);
            if (s->fd < 0) { rc = fail(COALA_EIO, "shm_open(%s) failed: %s", name, strerror(errno)); break; }
            struct stat st;
            if (fstat(s->fd, &st) != 0 || (uint64_t)st.st_size < bytes) { rc = fail(COALA_EIO, "shm %s is smaller than %llu bytes", name, (unsigned long long)bytes); break; }
        }
        s->host = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, s->fd, 0); // shared_UVA.cuh:83-87
        if (s->host == MAP_FAILED) { s->host = nullptr; rc = fail(COALA_EIO, "mmap failed: %s", strerror(errno)); break; }
        hipError_t e = hipHostRegister(s->host, bytes, hipHostRegisterMapped); // shared_UVA.cuh:89-93
        if (e != hipSuccess) { rc = fail(COALA_EHIP, "hipHostRegister(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); break; }
        s->registered = true;
        e = hipHostGetDevicePointer(&s->dev, s->host, 0); // shared_UVA.cuh:94-98
        if (e != hipSuccess) { rc = fail(COALA_EHIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e)); break; }
        // The reference memsets the whole mapping from every rank (shared_UVA.cuh:99); the creator made a fresh object above,
        // which the kernel zero-fills page by page on first touch.
    } while (0);
    if (rc != COALA_OK) {
        coala_shm_close(s, is_creator);
        return rc;
    }
    *out = s;
    return COALA_OK;
}

void* coala_shm_host_ptr(const coala_shm_t* s) { return s ? s->host : nullptr; }
void* coala_shm_device_ptr(const coala_shm_t* s) { return s ? s->dev : nullptr; }

int coala_shm_close(coala_shm_t* s, int unlink_it) { // shared_UVA.cuh:102-110
    if (!s) return COALA_OK;
    if (s->host) {
        if (s->registered) (void)hipHostUnregister(s->host);
        munmap(s->host, s->bytes);
    }
    if (s->fd >= 0) close(s->fd);
    if (unlink_it) shm_unlink(s->name.c_str());
    delete s;
    return COALA_OK;
}

int coala_pinned_alloc(uint64_t bytes, int device, void** host_ptr, void** device_ptr) {
    if (!host_ptr || !device_ptr || bytes == 0) return fail(COALA_EINVAL, "bad pinned_alloc arguments");
    HIPCHK(hipSetDevice(device));
    void* hp = nullptr;
    hipError_t e = hipHostMalloc(&hp, bytes, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return fail(COALA_ENOMEM, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    void* dp = nullptr;
    e = hipHostGetDevicePointer(&dp, hp, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(hp);
        return fail(COALA_EHIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
    }
    *host_ptr = hp;
    *device_ptr = dp;
    return COALA_OK;
}

int coala_pinned_free(void* host_ptr) {
    if (!host_ptr) return COALA_OK;
    HIPCHK(hipHostFree(host_ptr));
    return COALA_OK;
}

int coala_device_pci_bus_id(int device, char* out, size_t cap) {
    if (!out || cap < 16) return fail(COALA_EINVAL, "the bus id needs a buffer of at least 16 bytes");
    HIPCHK(hipDeviceGetPCIBusId(out, (int)cap, device));
    return COALA_OK;
}

// ------------------------------------------------------------------------------------------------ .npy
// decimal digits -> *v; false when the value does not fit an int64 (a hostile or damaged header: the reference's std::stoll would throw)
static bool read_dim(const char*& p, const char* end, int64_t* v) {
    int64_t x = 0;
    while (p < end && is_digit(*p)) {
        const int d = *p++ - '0';
        if (x > (INT64_MAX - d) / 10) return false;
        x = x * 10 + d;
    }
    *v = x;
    return true;
}

int coala_npy_parse(const char* buf, size_t len, int want_dim, int64_t* shape, int* ndim_out, size_t* data_off,
                    char* descr, size_t descr_cap) {
    // node_distributor_pybind.cuh:37-109
    if (!buf || !shape || !ndim_out || !data_off) return fail(COALA_EINVAL, "null argument");
    if (len < 10 || memcmp(buf, "\x93NUMPY", 6) != 0) return fail(COALA_EFORMAT, "Not a valid .npy file.");
    const unsigned major = (unsigned char)buf[6];
    size_t pos = 8;
    uint32_t hlen = 0;
    if (major == 1) {
        uint16_t h16;
        memcpy(&h16, buf + pos, 2);
        hlen = h16;
        pos += 2;
    } else if (major == 2) {
        if (len < 12) return fail(COALA_EFORMAT, "Not a valid .npy file.");
        memcpy(&hlen, buf + pos, 4);
        pos += 4;
    } else {
        return fail(COALA_EFORMAT, "Unsupported .npy file version: %u", major);
    }
    if (pos + hlen > len) return fail(COALA_EFORMAT, "truncated .npy header");
    if (want_dim != 1 && want_dim != 2) return fail(COALA_EINVAL, "Unsupported dimension for .npy file");
    const char* hdr = buf + pos;
    const char* end = hdr + hlen;
    *data_off = pos + hlen;
    *ndim_out = 0;
    if (const char* p = find_in(hdr, hlen, "'shape':")) {
        p += 8;
        if (p < end && is_space(*p)) ++p; // \s?
        if (p < end && *p == '(') {
            ++p;
            if (p < end && is_digit(*p)) {
                int64_t v0 = 0, v1 = 0;
                if (!read_dim(p, end, &v0)) return fail(COALA_EFORMAT, "a dimension in the .npy header does not fit 63 bits");
                if (want_dim == 1) { // 'shape':\s?\((\d+),?\)
                    if (p < end && *p == ',') ++p;
                    if (p < end && *p == ')') { shape[0] = v0; *ndim_out = 1; }
                } else if (p < end && *p == ',') { // 'shape':\s?\((\d+),\s?(\d+)\)
                    ++p;
                    if (p < end && is_space(*p)) ++p;
                    if (p < end && is_digit(*p)) {
                        if (!read_dim(p, end, &v1)) return fail(COALA_EFORMAT, "a dimension in the .npy header does not fit 63 bits");
                        if (p < end && *p == ')') { shape[0] = v0; shape[1] = v1; *ndim_out = 2; }
                    }
                }
            }
        }
    }
    if (descr && descr_cap) {
        descr[0] = 0;
        if (const char* p = find_in(hdr, hlen, "'descr':")) { // 'descr':\s*'(.*?)'
            p += 8;
            while (p < end && is_space(*p)) ++p;
            if (p < end && *p == '\'') {
                ++p;
                size_t k = 0;
                while (p < end && *p != '\'' && k + 1 < descr_cap) descr[k++] = *p++;
                descr[k] = 0;
            }
        }
    }
    return COALA_OK;
}

// ------------------------------------------------------------------------------------------------ distributor
int coala_distributor_create_plain(const int64_t* items, int num_nodes, coala_distributor_t** out) {
    if (!out) return fail(COALA_EINVAL, "null argument");
    coala_distributor* d = new (std::nothrow) coala_distributor();
    if (!d) return fail(COALA_ENOMEM, "out of host memory");
    d->items = items;
    d->num_nodes = num_nodes;
    *out = d;
    return COALA_OK;
}

static int load_npy(const char* path, int want_dim, const char* want_descr, std::vector<char>& img, int64_t* shape, int* nd,
                    size_t* off) {
    int rc = load_file(path, img);
    if (rc) return rc;
    char descr[16];
    rc = coala_npy_parse(img.data(), img.size(), want_dim, shape, nd, off, descr, sizeof(descr));
    if (rc) return rc;
    // the reference asserts "<i8" or "<f8" (node_distributor_pybind.cuh:100-103); be stricter: the exact one we read as
    if (strcmp(descr, want_descr) != 0) return fail(COALA_EFORMAT, "%s: dtype '%s', expected '%s'", path, descr, want_descr);
    return COALA_OK;
}

int coala_distributor_create(const int64_t* items, int node_id, int batch_size, int local_size, int num_nodes,
                             const char* color_file, const char* topk_file, const char* score_file,
                             coala_distributor_t** out) {
    if (!out || !items || !color_file || !topk_file || !score_file) return fail(COALA_EINVAL, "null argument");
    if (batch_size <= 0 || local_size <= 0 || num_nodes <= 0 || node_id < 0 || node_id >= num_nodes)
        return fail(COALA_EINVAL, "bad distributor geometry");
    *out = nullptr;
    coala_distributor* d = new (std::nothrow) coala_distributor();
    if (!d) return fail(COALA_ENOMEM, "out of host memory");
    d->items = items;
    d->node_id = node_id;
    d->batch_size = batch_size;
    d->local_size = local_size;
    d->num_nodes = num_nodes;
    d->use_color = true;
    int64_t shape[2] = {0, 0};
    int nd = 0;
    size_t off = 0;
    int rc = COALA_OK;
    do {
        if ((rc = load_npy(color_file, 1, "<i8", d->color_file, shape, &nd, &off))) break; // :141
        if (nd != 1) { rc = fail(COALA_EFORMAT, "%s: expected a 1-D array", color_file); break; }
        if (off + (size_t)shape[0] * 8 > d->color_file.size()) { rc = fail(COALA_EFORMAT, "%s: truncated payload", color_file); break; }
        d->color = reinterpret_cast<const int64_t*>(d->color_file.data() + off);
        d->num_color_entries = shape[0];
        if ((rc = load_npy(topk_file, 2, "<i8", d->topk_file, shape, &nd, &off))) break;   // :142
        if (nd != 2) { rc = fail(COALA_EFORMAT, "%s: expected a 2-D array", topk_file); break; }
        if (off + (size_t)shape[0] * (size_t)shape[1] * 8 > d->topk_file.size()) { rc = fail(COALA_EFORMAT, "%s: truncated payload", topk_file); break; }
        d->topk = reinterpret_cast<const int64_t*>(d->topk_file.data() + off);
        d->num_colors = (int)shape[0]; // :143
        d->topk_k = (int)shape[1];
        // The reference parses score.npy with the 1-D regex although it is 2-D, so its shape vector stays empty and only
        // the payload pointer is used (:144).  We parse it as 2-D and check it against topk.
        if ((rc = load_npy(score_file, 2, "<f8", d->score_file, shape, &nd, &off))) break;
        if (nd != 2 || shape[0] != d->num_colors || shape[1] != d->topk_k) { rc = fail(COALA_EFORMAT, "%s: shape does not match %s", score_file, topk_file); break; }
        if (off + (size_t)shape[0] * (size_t)shape[1] * 8 > d->score_file.size()) { rc = fail(COALA_EFORMAT, "%s: truncated payload", score_file); break; }
        d->score = reinterpret_cast<const double*>(d->score_file.data() + off);
        // every neighbour colour indexes the per-domain counter arrays (num_colors + 1 entries) in assign(): validate once
        // here, so that a malformed or mismatched topk file is a COALA_EFORMAT and not an out-of-bounds read per step
        for (int64_t k = 0; k < (int64_t)d->num_colors * d->topk_k; ++k)
            if (d->topk[k] < 0 || d->topk[k] > d->num_colors) {
                rc = fail(COALA_EFORMAT, "%s: entry %lld is colour %lld, outside [0, %d]", topk_file, (long long)k, (long long)d->topk[k], d->num_colors);
                break;
            }
        if (rc) break;
        d->domain_batch_size = batch_size * local_size;       // :146
        d->global_batch_size = d->domain_batch_size * num_nodes; // :147
    } while (0);
    if (rc) { delete d; return rc; }
    *out = d;
    return COALA_OK;
}

int coala_distributor_destroy(coala_distributor_t* d) {
    delete d;
    return COALA_OK;
}

int coala_distributor_num_colors(const coala_distributor_t* d) { return d ? d->num_colors : 0; }
const int64_t* coala_distributor_color_ptr(const coala_distributor_t* d) { return d ? d->color : nullptr; }
int64_t coala_distributor_num_color_entries(const coala_distributor_t* d) { return d ? d->num_color_entries : 0; }

int coala_distributor_assign(const coala_distributor_t* d, uint64_t offset, int64_t* out, const int32_t* const* meta,
                             int n_meta) {
    // node_distributor_pybind.cuh:150-222
    if (!d || !out || !meta) return fail(COALA_EINVAL, "null argument");
    if (!d->use_color || !d->score || !d->topk || !d->color) // :153-156 (the reference prints and returns)
        return fail(COALA_EINVAL, "Node_distributor_pybind is not created with color information.");
    if (n_meta < d->num_nodes) return fail(COALA_EINVAL, "meta list has %d entries, need %d", n_meta, d->num_nodes);
    const int num_nodes = d->num_nodes;
    const int topk = d->topk_k;
    std::vector<int> bucket_len((size_t)num_nodes, 0);
    for (int64_t i = 0; i < d->global_batch_size; ++i) {
        const int64_t id = d->items[i + (int64_t)offset];
        if (id < 0 || id >= d->num_color_entries) return fail(COALA_ERANGE, "item %lld outside the colour table", (long long)id);
        const int64_t node_color = d->color[id];
        if (node_color < 0 || node_color > d->num_colors) return fail(COALA_ERANGE, "colour %lld outside [0,%d]", (long long)node_color, d->num_colors);
        int best = 0;
        double best_score = -1.0;
        for (int j = 0; j < num_nodes; ++j) {
            const int32_t* cnt = meta[j];
            double cur = 0.0;
            if (node_color != 0) {
                const int64_t* tk = d->topk + (node_color - 1) * topk;
                const double* sc = d->score + (node_color - 1) * topk;
                for (int k = 0; k < topk; ++k) {
                    const int64_t nc = tk[k];
                    if (nc == 0) continue;
                    const int32_t held = cnt[nc];
                    if (held == 0) continue;
                    cur += (double)held * sc[k]; // same order of fp64 operations as the reference (:195)
                }
            }
            if (bucket_len[j] == d->domain_batch_size) cur = -1.0;
            if (cur > best_score) { best = j; best_score = cur; }
        }
        if (best == d->node_id && bucket_len[best] < d->domain_batch_size) out[bucket_len[best]] = id;
        bucket_len[best] += 1;
    }
    return COALA_OK;
}

} // extern "C"
