/*
 * coala_hip.h -- C ABI of libcoala_hip.so: the MI355X (gfx950) feature-cache / minibatch-assembly path.
 *
 * This is the drop-in boundary for the hot path of jeongminpark417/COALA-GNN.  Every entry point names the
 * reference interface it replaces (paths relative to /root/reference).  Plain pointers and sizes only: no torch,
 * pybind or C++ types cross this boundary.  All functions return 0 on success or a negative COALA_E* code;
 * coala_last_error() returns a thread-local message for the last failure.  Nothing here ever calls exit().
 *
 * Streams: `stream` is a hipStream_t passed as void*.  Work is enqueued on it and the call returns without
 * synchronising, unless the handle was created with COALA_FLAG_SYNC (the reference's behaviour: every native call
 * ends in cudaDeviceSynchronize, COALA_GNN_Modules/ssd_gnn_cache.cuh:266) in which case the call returns after the
 * stream has drained.  A cache or sampler handle keeps its own work in program order across streams: a call on another
 * stream than the handle's previous call first waits there (hipStreamWaitEvent, no host wait) for that call's work.  One handle
 * is driven by one host thread at a time.
 */
#ifndef COALA_HIP_H
#define COALA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COALA_OK 0
#define COALA_EINVAL (-1)   /* bad argument                                    */
#define COALA_EHIP (-2)     /* a HIP runtime call failed                       */
#define COALA_ENOMEM (-3)   /* allocation failed                               */
#define COALA_EIO (-4)      /* file / shm failure                              */
#define COALA_EFORMAT (-5)  /* malformed .npy                                  */
#define COALA_ERANGE (-6)   /* an index was outside [0, num_rows)              */
#define COALA_ECOMM (-7)    /* an RCCL call failed                             */

#define COALA_WAYS 32u /* COALA_GNN_Modules/ssd_gnn_cache.cuh:61,204 */
#define COALA_COUNTS_RING 8 /* count exchanges issued ahead (coala_comm_counts_begin) whose tickets stay valid */

#define COALA_FLAG_SYNC 1u        /* synchronise the stream before returning (reference semantics)            */
#define COALA_FLAG_DISTRIBUTED 2u /* set = (id / n_gpus) % sets   (nvshmem_cache.h:191-196,347) instead of id % sets */
#define COALA_FLAG_PROFILE 4u     /* record hipEvents around the probe+gather kernel (see coala_cache_profile) */
#define COALA_FLAG_TAG64 16u      /* keep the reference's 64-bit tags (a set = 32 x 8 B = two 128-B lines, isolated_cache.h:552) even when
                                     num_rows < 2^32.  Default: 32-bit tags whenever every id fits -- a set is then ONE 128-B line, half
                                     the probe bytes; coala_cache_dump widens them, so the visible table state is the same.            */
#define COALA_FLAG_COLD_PARTITIONED 8u /* cold_table holds only this owner's rows: row k = node id k*n_gpus + rank, i.e. the
                                          cold row of id is id / n_gpus.  An owner of the partitioned cache never reads any
                                          other row, so each GPU pins 1/n_gpus of the table next to its own PCIe link instead
                                          of all GPUs mapping one shared copy (shared_UVA.cuh:42-100).                        */

const char* coala_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int coala_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Cache geometry: replaces SSD_GNN_SSD_Controllers (COALA_GNN_Modules/ssd_gnn_cache.cuh:10-55).
 * cache_dim = 128/256/512/1024 for dim <= 128/256/512/1024; COALA_EINVAL for dim > 1024 (reference throws).
 * ------------------------------------------------------------------------------------------------------------ */
int coala_cache_dim(int dim);
/* num_sets = (cache_mb * 2^20 / (cache_dim*4)) / 32   (ssd_gnn_cache.cuh:96-97,239-240) */
uint64_t coala_cache_num_sets(uint64_t cache_mb, int cache_dim);

typedef struct coala_cache coala_cache_t; /* opaque */

typedef struct coala_cache_config {
    int32_t device;            /* HIP device ordinal (SSD_GNN_SSD_Controllers.cudaDevice)                              */
    int32_t dim;               /* floats per row of the cold table and of every output row                            */
    uint64_t cache_mb;         /* cache capacity in MiB (Isolated_Cache ctor `cache_size`, ssd_gnn_cache.cuh:227)      */
    int32_t n_gpus;            /* GPUs sharing the owner-partitioned cache (1 = isolated)                             */
    int32_t rank;              /* this GPU's rank in [0, n_gpus)                                                      */
    int32_t global_rank;       /* only used in print_stats lines (isolated_cache.h:136-138)                           */
    uint32_t flags;            /* COALA_FLAG_*                                                                        */
    const float* cold_table;   /* device-visible pointer to fp32 [num_rows, dim]: pinned host (zero-copy) or HBM.      */
                               /* Replaces `sim_buf` (ssd_gnn_cache.cuh:227; isolated_cache.h:323-331).  Row stride is */
                               /* `dim` floats (the reference strides by cache_dim: SURVEY.md section 3.3, defect 2).  */
    uint64_t num_rows;         /* rows of cold_table; ids outside [0,num_rows) are rejected (COALA_ERANGE)            */
    const int64_t* node_color; /* HOST pointer to int64[num_rows] colours (Node_distributor_pybind::color_buffer_ptr,  */
                               /* node_distributor_pybind.cuh:226-229) or NULL: copied to the device at creation      */
    int32_t num_colors;        /* colours are 0..num_colors inclusive (0 = uncoloured); num_colors+1 counters are kept */
    int32_t reserved;
    uint64_t max_batch;        /* rows per call to pre-size scratch for (0 = grow on demand)                          */
} coala_cache_config_t;

/* Replaces Isolated_Cache / SSD_GNN_NVSHMEM_Cache ctors (ssd_gnn_cache.cuh:84-109,227-252) and
 * Isolated_cache_handle / NVSHMEM_cache_handle ctors (isolated_cache.h:520-636, nvshmem_cache.h:525-640). */
int coala_cache_create(const coala_cache_config_t* cfg, coala_cache_t** out);
/* Replaces the destructors (isolated_cache.h:638-653, ssd_gnn_cache.cuh:366-369). */
int coala_cache_destroy(coala_cache_t* h);

typedef struct coala_cache_geometry {
    uint64_t num_sets;
    uint32_t num_ways;
    uint32_t cache_dim;   /* floats per line */
    uint64_t line_bytes;  /* cache_dim * 4   */
    uint64_t table_bytes; /* HBM bytes held by lines + tags + metadata */
    uint32_t tag_set_bytes; /* bytes of one set's 32 tags as stored: 128 (32-bit tags) or 256 (COALA_FLAG_TAG64 / num_rows >= 2^32) */
    uint32_t reserved;
} coala_cache_geometry_t;
int coala_cache_geometry(const coala_cache_t* h, coala_cache_geometry_t* out);

/* out[i, 0:dim] = cache(idx[i]) for i in [0, n).  Replaces Isolated_Cache::read_feature (ssd_gnn_cache.cuh:255-268),
 * Isolated_read_feature_kernel (cache_kernel.cu:59-77) and Isolated_cache_d_t::get_data (isolated_cache.h:335-475).
 * With COALA_FLAG_DISTRIBUTED the set index is the distributed one (see coala_cache_serve).
 * `out` fp32 [n, dim] and `idx` int64 [n] are device pointers.  Batch-synchronous, deterministic replacement:
 * every probe sees the pre-call table; misses then take ways (set_cnt + k) % 32 in order of position. */
int coala_cache_read_feature(coala_cache_t* h, float* out, const int64_t* idx, int64_t n, void* stream);

/* Owner-side serve of the partitioned cache: same as coala_cache_read_feature but always with the distributed set index
 * set = (id / n_gpus) % sets, whatever the handle's flags.  Replaces Isolated_Cache::nccl_get_feature
 * (ssd_gnn_cache.cuh:297-325: get_data(id, out, local_size, true)) and SSD_GNN_NVSHMEM_Cache::read_feature
 * (ssd_gnn_cache.cuh:132-174) minus the transport.  ids = concatenation, in source-rank order, of the ids routed to
 * this owner; out = packed fp32 [n, dim] in the same order. */
int coala_cache_serve(coala_cache_t* h, float* out, const int64_t* ids, int64_t n, void* stream);

/* The same serve in two phases, for callers that overlap the cold fill with the exchange of rows that are already in place
 * (ssd_gnn_cache.cuh:132-174 serves and ships peer by peer on separate streams): serve_probe classifies the WHOLE batch and
 * copies the hits; serve_fill completes the positions [begin, end) -- ranking still spans the whole batch, so any set of
 * fills that covers [0, n) exactly once, in any order, leaves table, counters and rows exactly as one coala_cache_serve does.
 * Same out / ids / n in every call of one batch.  The batch stays OPEN until its fills have covered [0, n): until then any
 * new probe (read_feature, serve, serve_probe*) and any fill that overlaps an earlier one returns COALA_EINVAL;
 * coala_cache_serve_abort drops an open batch (its misses stay uncached, rows of unfilled positions are undefined). */
int coala_cache_serve_probe(coala_cache_t* h, float* out, const int64_t* ids, int64_t n, void* stream);
int coala_cache_serve_fill(coala_cache_t* h, float* out, const int64_t* ids, int64_t n, int64_t begin, int64_t end, void* stream);
/* One fill over a union of disjoint position ranges (HOST arrays begins/ends of n_ranges entries): one kernel launch per 64
 * ranges.  A row exchange split into rounds fills "the k-th slice of every peer's segment" with one call per round. */
int coala_cache_serve_fill_ranges(coala_cache_t* h, float* out, const int64_t* ids, int64_t n, const int64_t* begins,
                                  const int64_t* ends, int n_ranges, void* stream);
int coala_cache_serve_abort(coala_cache_t* h, void* stream);

/* serve_probe with part of the batch delivered elsewhere: rows at batch positions [begin, end) are written to
 * redirect->out[row_map[pos - begin], 0:dim] (row_map: device int64[end-begin], NULL = row pos - begin) instead of
 * out[pos, 0:dim], by the probe and by every later fill of the batch.  This is how the requester's OWN shard of a
 * distributed fetch goes straight into the caller's tensor (the reference's `j == i` local copy,
 * COALA-GNN-Setup/COALA_GNN/COALA_GNN_Manager.py:195-199) while staying part of the owner's one batch per step.
 * `out` may be NULL when the redirect covers the whole batch.
 * Every row_map entry must lie in [0, 2^31): the probe carries a row's destination as a 32-bit value (a batch never has more
 * rows than that: n <= 2^31 - 1 is checked).  The entries live in device memory, so the library cannot check them on the host: a
 * value outside that range addresses the wrong row.  The library's own caller (coala_cache_fetch_distributed) passes positions
 * of the batch, which are below n by construction. */
typedef struct coala_row_redirect {
    int64_t begin, end;
    float* out;
    const int64_t* row_map;
} coala_row_redirect_t;
int coala_cache_serve_probe_redirect(coala_cache_t* h, float* out, const int64_t* ids, int64_t n, const coala_row_redirect_t* redirect,
                                     void* stream);

/* Bucket idx by owner = id % n_parts, stable inside each bucket.  Replaces Isolated_Cache::split_node_list
 * (ssd_gnn_cache.cuh:283-295) / nccl_split_node_list_kernel (cache_kernel.cu:79-91) and the routing half of
 * NVSHMEM_send_requests_kernel (cache_kernel.cu:4-17).
 *   node_out, map_out : int64 device buffers.  bucket_stride > 0: bucket g starts at g*bucket_stride (the reference's
 *                       [G][max_sample] layout).  bucket_stride == 0: buckets are packed back to back (all-to-all-v
 *                       send layout) and offsets_out[g] (int64[n_parts+1], device) receives the bucket starts.
 *   counts_out        : int64[n_parts] device. */
int coala_cache_route(coala_cache_t* h, const int64_t* idx, int64_t n, int n_parts, int64_t bucket_stride,
                      int64_t* node_out, int64_t* map_out, int64_t* counts_out, int64_t* offsets_out, void* stream);

/* out[map[r], 0:dim] = src[r, 0:dim] for r in [0, n).  Replaces Isolated_Cache::map_feat_data
 * (ssd_gnn_cache.cuh:327-356) / nccl_gather_feature_kernel + block_memcpy (cache_kernel.cu:113-137). */
int coala_cache_scatter(coala_cache_t* h, float* out, const float* src, const int64_t* map, int64_t n, void* stream);
/* The same for the rows r of a union of disjoint ranges (HOST arrays): what one round of a split row exchange delivered. */
int coala_cache_scatter_ranges(coala_cache_t* h, float* out, const float* src, const int64_t* map, const int64_t* begins,
                               const int64_t* ends, int n_ranges, void* stream);

/* Floats per row of the cold table / output (cfg.dim). */
int64_t coala_cache_row_dim(const coala_cache_t* h);

/* ------------------------------------------------------------------------------------------------------------
 * Native fused fetch of the owner-partitioned cache over RCCL (one process per GPU, one communicator per cache group).
 * Replaces, in one call: SSD_GNN_NVSHMEM_Cache::send_requests + read_feature (ssd_gnn_cache.cuh:111-174) and the "nccl"
 * orchestration of COALA_GNN_Manager.fetch_feature (COALA-GNN-Setup/COALA_GNN/COALA_GNN_Manager.py:143-211):
 * route -> all-to-all(counts) -> one host read -> all-to-all-v(ids) -> probe (own shard straight into `out`) ->
 * { cold fill of slice k  ||  all-to-all-v(rows of slice k-1) on a second stream } -> un-permute.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct coala_comm coala_comm_t;
/* ncclGetUniqueId: called by ONE rank; the 128 bytes are handed to the others by any side channel (torch.distributed). */
int coala_comm_unique_id(void* out_id, size_t cap);
/* ncclCommInitRank: collective over the nranks processes of the group (one process per GPU, RCCL over xGMI). */
int coala_comm_create(const void* id_bytes, int rank, int nranks, int device, coala_comm_t** out);
int coala_comm_destroy(coala_comm_t* c);
/* Ranks of the communicator AS THE TRANSPORT SEES THEM: ncclCommCount for RCCL (and ncclCommUserRank is checked against `rank`
 * at creation), the group size for the in-process transport.  Negative on failure. */
int coala_comm_size(const coala_comm_t* c);
/* In-process transport: the nranks ranks of a group are host THREADS of one process (one communicator each, any mix of
 * devices, several ranks per device allowed); ids and rows move with device-to-device copies ordered by events -- a copy kernel
 * on the receiver's device when the sender's memory is on the same device or on a peer whose access could be enabled
 * (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess, once per device pair), the runtime's hipMemcpyAsync otherwise.  Same orchestration as over RCCL -- it is how the parity tests run G logical ranks on one GPU, and how a
 * single-process multi-GPU driver would use the partitioned cache.  The group outlives its communicators. */
typedef struct coala_comm_group coala_comm_group_t;
int coala_comm_group_create(int nranks, coala_comm_group_t** out);
int coala_comm_group_destroy(coala_comm_group_t* g);
int coala_comm_create_inproc(coala_comm_group_t* g, int rank, int device, coala_comm_t** out);
/* Row-exchange rounds per fetch, 1..8 (default 2, or COALA_EXCHANGE_ROUNDS): round k ships the k-th slice of every peer's
 * segment while the cold fill of slice k+1 runs.  Must be the same on every rank. */
int coala_comm_set_rounds(coala_comm_t* c, int rounds);
int coala_comm_get_rounds(const coala_comm_t* c); /* 0 for a null handle */
/* Diagnostics (off by default; the same setting on every rank): with on != 0 a rank's OWN segment takes the road of a peer's -- no
 * own-shard bypass, its rows are served into the staging buffer, shipped in rounds on the communicator's stream and un-permuted, and
 * the RCCL transport moves it with ncclSend + ncclRecv to ITSELF inside the same group call as the peers' segments instead of a local
 * copy.  Delivered rows and cache state are the same.  It exists so that a communicator of one rank -- all a one-GPU box can create
 * over RCCL -- executes every line of the exchange (counts, datatypes, displacements, both streams) on the real transport. */
int coala_comm_set_self_loopback(coala_comm_t* c, int on);
/* per-peer id counts of the last fetch (host int64[nranks] each, either may be NULL) */
int coala_comm_last_counts(const coala_comm_t* c, int64_t* send, int64_t* recv);
/* out[i, 0:dim] = row of idx[i], wherever its owner (idx[i] % nranks) is.  Collective: every rank of the communicator calls
 * it once per step (n may be 0).  Split-phase: ids go out, the owner probes its ONE batch per step (the concatenation of what
 * it received, in source-rank order), the requester's own shard lands directly in `out`, and the rows of the other peers come
 * back in rounds on the communicator's own stream while the owner's cold fill of the next round runs on `stream`.
 * Synchronises `stream` once, in the middle (the counts); everything after that is enqueued, and `stream` is ordered behind
 * the communicator's stream on return.  Error behaviour: argument and allocation checks happen before the first collective; a
 * failure after it aborts the transport (ncclCommAbort) so that the peers fail too instead of waiting -- the communicator
 * then only accepts coala_comm_destroy. */
int coala_cache_fetch_distributed(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n, void* stream);
/* The same for ids that arrive ALREADY bucketed by owner (idx = bucket 0 | bucket 1 | ..., counts_dev = device int64[nranks]
 * bucket sizes summing to n -- what coala_sampler_sample delivers with `bucketing`): no routing pass, no un-permute, the rows
 * of owner p are received straight into out[offset of bucket p ...] and the own bucket is gathered in place.
 * out[i] = row of idx[i] as above. */
int coala_cache_fetch_distributed_bucketed(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n,
                                           const int64_t* counts_dev, void* stream);
/* The bucketed fetch without its host synchronisation (opt-in): the count exchange is split off and issued AHEAD -- typically right
 * behind the sampler that produced counts_dev, on the sampler's stream, one or two steps before the fetch -- and the fetch then
 * finds both count vectors on the host.  coala_comm_counts_begin is a collective (counts all-to-all + a copy to pinned memory +
 * an event; no host wait; up to 8 may be outstanding); every rank issues its calls on one communicator in the same order.
 * coala_cache_fetch_distributed_bucketed_ahead(ticket) waits for that exchange's event (normally long complete) and runs the rest of
 * the sequence -- ids, probe, fill rounds beside row rounds -- fully stream-ordered.  Rows and cache state equal the plain call's. */
int coala_comm_counts_begin(coala_comm_t* c, const int64_t* counts_dev, void* stream, int64_t* ticket_out);
int coala_cache_fetch_distributed_bucketed_ahead(coala_cache_t* h, coala_comm_t* c, float* out, const int64_t* idx, int64_t n,
                                                 int64_t ticket, void* stream);
/* ------------------------------------------------------------------------------------------------------------
 * Completion and timing of a read WITHOUT packets of their own.  With coala_cache_fetch_events(h, 1) every coala_cache_read_feature
 * attaches a begin event to its first kernel launch and an end event to its last one (hipExtLaunchKernelGGL: the events ride on the
 * two dispatch packets).  coala_cache_last_fetch_events hands out the pair of the most recent call -- owned by the handle, valid for
 * the next 2048 calls, NULL when that call launched nothing (n = 0), when the handle profiles (COALA_FLAG_PROFILE uses the dispatches'
 * event slots itself) or for the split-phase serve calls: the caller then records an event of its own.  A consumer on another stream
 * waits with coala_stream_wait_event (hipStreamWaitEvent); coala_event_elapsed_ms gives begin -> end in milliseconds (wait = 0:
 * returns 1, no error, while the end event has not completed).  What it costs, by the kernels' own timestamps (profiles/r04_handover.txt):
 * plain launches hand over from the fill of one read to the probe of the next without a gap; these riding events 14 us per read (an
 * attached event makes its kernel wait for, and be waited for by, its neighbours); ONE hipEventRecord behind the read 6 us; a recorded
 * timing pair + completion event 15 us.  So: the cheapest completion signal is one recorded event, and this interface is for per-read
 * TIMING (what COALA_FLAG_PROFILE does per kernel).  The reference synchronises the device after every call instead
 * (ssd_gnn_cache.cuh:266).
 * ------------------------------------------------------------------------------------------------------------ */
int coala_cache_fetch_events(coala_cache_t* h, int enable);
int coala_cache_last_fetch_events(const coala_cache_t* h, void** begin_ev, void** end_ev);
int coala_stream_wait_event(void* stream, void* event);
int coala_event_elapsed_ms(void* begin_ev, void* end_ev, int wait, float* ms_out);

/* The same for a fetch over a communicator (opt-in; bucketed fetches).  enable = 1: a bucketed fetch hands out two END events -- the one its
 * last fill launch on the caller's stream carries anyway (the hand-over to the row round), and one recorded behind the last row round on
 * the communicator's own stream -- so that a consumer's stream can wait for the rows without a completion event recorded on the caller's
 * stream (the rows are complete once BOTH have completed).  enable = 2: additionally a BEGIN event on the probe's launch, for a timer
 * (begin -> end_ev_comm); an event on a launch costs that launch about 5 us (profiles/r04_handover.txt), so a caller that samples its
 * timing asks for it on the sampled fetches only.  coala_comm_last_fetch_events hands out the three (owned by the communicator, valid for
 * 2048 fetches; begin_ev NULL with enable = 1; end_ev_comm NULL for a communicator of one rank; all NULL after a routed fetch or an empty
 * batch: the caller then records its own).  Independently of this switch the fetch puts its internal hand-over events (fill of round k ->
 * row round k) on the fill launches and waits for the last row round only. */
int coala_comm_fetch_events(coala_comm_t* c, int enable);
int coala_comm_last_fetch_events(const coala_comm_t* c, void** begin_ev, void** end_ev_stream, void** end_ev_comm);

/* Timing of the row exchange (all rounds of a fetch, HIP events on the communicator's stream; includes any wait for the fill of
 * a later round): enable = 1 / 0 switches it, -1 leaves it; out (nullable) receives the totals since the last reset. */
typedef struct coala_comm_profile {
    double rows_ms;          /* summed duration of the row exchange */
    uint64_t calls;          /* fetches timed */
    uint64_t remote_rows_in; /* rows received from other ranks */
} coala_comm_profile_t;
int coala_comm_profile(coala_comm_t* c, int enable, coala_comm_profile_t* out, int reset);

/* Copy the colour occupancy counters to HOST memory dst[0 .. n_entries).  Replaces get_cache_data
 * (ssd_gnn_cache.cuh:176-186,270-280).  The reference copies num_colors entries; pass num_colors+1 to also get the
 * last colour.  Synchronises `stream`. */
int coala_cache_color_counts(coala_cache_t* h, int32_t* dst, int32_t n_entries, void* stream);
/* The same snapshot in two halves: _async enqueues the copy at this point of `stream` and returns without waiting; _finish (any
 * thread) waits for that copy only and delivers dst[0 .. n_entries).  One snapshot may be pending per handle. */
int coala_cache_color_counts_async(coala_cache_t* h, int32_t n_entries, void* stream);
int coala_cache_color_counts_finish(coala_cache_t* h, int32_t* dst, int32_t n_entries);

/* hit / miss counters since the last reset.  Replaces print_stats_kernel / print_stats (cache_kernel.cu:139-143,
 * isolated_cache.h:132-141), which print and reset.  Synchronises `stream`.  range_errors counts rejected ids. */
int coala_cache_stats(coala_cache_t* h, uint64_t* hit, uint64_t* miss, uint64_t* range_errors, int reset, void* stream);

/* Debug / test access to the table (device -> host copies; synchronising).  Any pointer may be NULL.
 * keys: u64[sets*32]; set_cnt: u32[sets]; color_meta: u32[sets*32]. */
int coala_cache_dump(coala_cache_t* h, uint64_t* keys, uint32_t* set_cnt, uint32_t* color_meta, void* stream);

/* With COALA_FLAG_PROFILE: accumulated hipEvent time (ms) and launch count of the probe+gather kernel and of the
 * cold-fill kernel since the last reset; rows_* are the rows each processed.  Synchronises the recorded events. */
typedef struct coala_cache_profile {
    double gather_ms;
    uint64_t gather_launches;
    uint64_t gather_rows;   /* rows probed */
    uint64_t gather_hits;   /* rows copied from HBM lines */
    double fill_ms;
    uint64_t fill_launches;
    uint64_t fill_rows;
    double event_overhead_us; /* median elapsed time of an EMPTY hipEvent bracket on the same stream: what every bracketed */
                              /* launch above includes on top of the kernel itself                                        */
} coala_cache_profile_t;
int coala_cache_profile(coala_cache_t* h, coala_cache_profile_t* out, int reset);

/* ------------------------------------------------------------------------------------------------------------
 * Offline graph colouring + colour-affinity tables (host only).  Replaces Graph_Coloring
 * (COALA_GNN_Modules/graph_coloring.h:15-68, graph_coloring.cpp) driven by examples/color_info_gen/generate_color_data.py.
 * All buffers are HOST pointers owned by the caller, int64 (the reference aliases int64 torch tensors as uint64).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct coala_coloring coala_coloring_t;
int coala_coloring_create(uint64_t num_nodes, coala_coloring_t** out);                                  /* Graph_Coloring(u64) */
int coala_coloring_destroy(coala_coloring_t* g);
int coala_coloring_set_adj_csc(coala_coloring_t* g, const int64_t* indptr, const int64_t* indices);     /* set_adj_csc */
int coala_coloring_set_color_buffer(coala_coloring_t* g, int64_t* color);                               /* set_color_buffer (zeroed [N]) */
/* set_topk_color_buffer / set_topk_affinity_buffer: [num_colors*topk]; either pointer may be NULL to keep the current one */
int coala_coloring_set_topk_buffers(coala_coloring_t* g, int64_t* topk_color, double* topk_affinity, int topk);
/* cpu_color_graph_optimized(train_ptr, n): seeds sampled from the training nodes with glibc rand(); seed 1 reproduces the
 * reference (which never calls srand) */
int coala_coloring_color_optimized(coala_coloring_t* g, const int64_t* train, uint64_t n_train, unsigned seed);
int coala_coloring_color_all(coala_coloring_t* g, unsigned seed);                                       /* cpu_color_graph */
uint64_t coala_coloring_num_color(const coala_coloring_t* g);                                           /* get_num_color */
uint64_t coala_coloring_num_color_node(const coala_coloring_t* g);                                      /* get_num_color_node */
/* with_affinity=1: cpu_calculate_color_affinity; 0: cpu_count_nearest_color_less_memory */
int coala_coloring_topk(coala_coloring_t* g, int with_affinity);
int coala_coloring_nearest(coala_coloring_t* g);                                                        /* cpu_count_nearest_color */

/* ------------------------------------------------------------------------------------------------------------
 * Neighbour sampler + block compaction over a CSC graph resident in device-visible memory (HBM, or pinned host).
 * Replaces the DGL call on the hot path: graph_sampler.sample(g, seeds) with
 * dgl.dataloading.MultiLayerNeighborSampler(fanouts) (COALA-GNN-Setup/COALA_GNN/COALA_GNN_DataLoader.py:162,
 * examples/sbatch_ssd_gnn_train.py:70-72; graph: examples/ssd_gnn_dataloader.py:523).  DGL's arithmetic is not under
 * /root/reference; the contract (coala_sampler.hip header) is pinned by properties and by the CPU twin in oracle/.
 * ------------------------------------------------------------------------------------------------------------ */
#define COALA_SAMPLER_MAX_LAYERS 8
typedef struct coala_sampler coala_sampler_t;
/* indptr int64[num_nodes+1], indices int64[num_edges]: device-visible, borrowed for the sampler's lifetime. */
int coala_sampler_create(int device, const int64_t* indptr, const int64_t* indices, int64_t num_nodes, int64_t num_edges,
                         coala_sampler_t** out);
int coala_sampler_destroy(coala_sampler_t* s);
/* Sample n_layers layers starting from `seeds` (device int64[n_seeds]).  fanouts[l] is the fan-out of the l-th SAMPLED
 * layer (DGL walks reversed(fanouts): pass them already reversed).  Layer l's destination nodes are layer l-1's source
 * nodes.  Outputs, all device buffers owned by the caller, for cap_0 = n_seeds, cap_{l+1} = cap_l*(fanouts[l]+1):
 *   src_nodes_out[l] : int64[cap_{l+1}]        source (input) nodes of block l: its dst nodes first, then new ones
 *   nbr_local_out[l] : int32[cap_l*fanouts[l]] row d holds the local indices of dst d's sampled neighbours, -1 padded
 *   n_src_host[l]    : HOST int64, number of source nodes of block l.  Non-NULL: the call returns when the counts are there (it
 *                      waits on an event behind the sampler's one kernel, not on the stream).  NULL: the call only enqueues;
 *                      collect the counts later with coala_sampler_wait(ticket) -- up to 8 calls may be outstanding.
 * The whole multi-layer sample is ONE kernel launch (a persistent kernel with grid barriers between its phases).
 * Randomness: counter-based, keyed by (seed, step, layer, node id): same arguments, same sample.
 *
 * bucketing (nullable): additionally deliver the input nodes of the LAST layer bucketed by owner = id % n_parts, stable inside
 * each bucket -- the layout the owner-partitioned fetch sends (coala_cache_fetch_distributed_bucketed: no routing pass, rows are
 * received straight into the output tensor).  With it nbr_local_out[n_layers-1] indexes `bucketed_nodes` and dst_in_src[d] is
 * the position of the block's d-th destination node in it (the "dst nodes first" convention cannot hold for a bucketed list);
 * src_nodes_out[n_layers-1] still receives the unbucketed first-appearance list. */
typedef struct coala_sampler_bucketing {
    int32_t n_parts;         /* owners (1..64); 0 = off                                  */
    int32_t reserved;
    int64_t* bucketed_nodes; /* device int64[cap_L]                                      */
    int64_t* counts;         /* device int64[n_parts]: bucket sizes                      */
    int32_t* dst_in_src;     /* device int32[cap_{L-1}]                                  */
} coala_sampler_bucketing_t;
int coala_sampler_sample(coala_sampler_t* s, const int64_t* seeds, int64_t n_seeds, const int32_t* fanouts, int n_layers,
                         uint64_t seed, uint64_t step, int64_t* const* src_nodes_out, int32_t* const* nbr_local_out,
                         int64_t* n_src_host, const coala_sampler_bucketing_t* bucketing, int64_t* ticket_out, void* stream);
/* Counts of an earlier call (its ticket): n_src_host[n_layers] and, when it bucketed, bucket_counts_host[n_parts] (either NULL). */
int coala_sampler_wait(coala_sampler_t* s, int64_t ticket, int64_t* n_src_host, int64_t* bucket_counts_host);

/* Block op for the consumer of these blocks (the native Block objects stand where DGL blocks stand in
 * examples/sbatch_ssd_gnn_train.py:138-141; dgl.nn.SAGEConv's "mean" reduces to this): out[d, :] = mean over the valid j of
 * h_src[nbr[d, j], :]; nbr int32 [n_dst, fanout] (-1 padded, fan-out <= 32), fp32 rows of `dim` floats.  The backward adds
 * grad_out[d] / count(d) into grad_src[nbr[d, j]] with hardware float atomics (grad_src zeroed by the caller). */
int coala_block_mean_aggregate(int device, const int32_t* nbr, const float* h_src, float* out, int64_t n_dst, int fanout, int dim, void* stream);
int coala_block_mean_aggregate_backward(int device, const int32_t* nbr, const float* grad_out, float* grad_src, int64_t n_dst, int fanout,
                                        int dim, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Shared pinned-host ("UVA") region.  Replaces SharedUVAManager (COALA_GNN_Modules/shared_UVA.cuh:26-115):
 * creator shm_open+ftruncate, everybody mmap + hipHostRegister + hipHostGetDevicePointer.  The MPI barrier between
 * create and open (shared_UVA.cuh:76,79) is the caller's job (torch.distributed barrier): is_creator selects the role.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct coala_shm coala_shm_t;
int coala_shm_open(const char* name, uint64_t bytes, int is_creator, int device, coala_shm_t** out);
void* coala_shm_host_ptr(const coala_shm_t* s);   /* SharedUVAManager::get_host_ptr   */
void* coala_shm_device_ptr(const coala_shm_t* s); /* SharedUVAManager::get_device_ptr */
int coala_shm_close(coala_shm_t* s, int unlink);  /* SharedUVAManager::cleanup        */

/* Plain pinned host allocation visible to the device (private cold tier; hipHostMalloc mapped). */
int coala_pinned_alloc(uint64_t bytes, int device, void** host_ptr, void** device_ptr);
int coala_pinned_free(void* host_ptr);
/* "dddd:bb:dd.f" of HIP device `device` (hipDeviceGetPCIBusId; initialises the runtime).  The host side places a rank's threads
 * and its cold-tier shard on that GPU's NUMA node (COALA_GNN/numa.py reads /sys/bus/pci/devices/<id>/numa_node); the reference
 * leaves the placement of its one shared segment to chance (COALA_GNN_Modules/shared_UVA.cuh:60-100). */
int coala_device_pci_bus_id(int device, char* out, size_t cap);

/* ------------------------------------------------------------------------------------------------------------
 * .npy reader and node distributor.  Replace parse_numpy_file / load_file_to_memory
 * (COALA_GNN_Modules/node_distributor_pybind.cuh:11-109) and Node_distributor_pybind (:112-238).
 * ------------------------------------------------------------------------------------------------------------ */
/* Parse a .npy v1/v2 header in memory.  want_dim is 1 or 2 (the reference's regex choice); when the stored shape has a
 * different rank, *ndim_out is 0 and shape is untouched (the reference leaves its vector empty).  descr receives e.g. "<i8". */
int coala_npy_parse(const char* buf, size_t len, int want_dim, int64_t* shape, int* ndim_out, size_t* data_off,
                    char* descr, size_t descr_cap);

typedef struct coala_distributor coala_distributor_t;
/* Node_distributor_pybind(u64 items, int n_nodes)  (node_distributor_pybind.cuh:133-136) */
int coala_distributor_create_plain(const int64_t* items, int num_nodes, coala_distributor_t** out);
/* Node_distributor_pybind(u64 items, int node_id, int batch, int local_size, int n_nodes, color, topk, score) (:138-148) */
int coala_distributor_create(const int64_t* items, int node_id, int batch_size, int local_size, int num_nodes,
                             const char* color_file, const char* topk_file, const char* score_file,
                             coala_distributor_t** out);
int coala_distributor_destroy(coala_distributor_t* d);
int coala_distributor_num_colors(const coala_distributor_t* d);          /* get_num_colors (:224-226) */
const int64_t* coala_distributor_color_ptr(const coala_distributor_t* d); /* get_color_buffer_ptr (:228-231) */
int64_t coala_distributor_num_color_entries(const coala_distributor_t* d);
/* distribute_node_with_affinity(u64 offset, u64 out, list[u64] meta)  (:150-222).  meta[j] -> int32 counters of domain j,
 * indexed by colour (num_colors+1 entries).  out -> int64[batch_size*local_size].  Re-entrant per handle. */
int coala_distributor_assign(const coala_distributor_t* d, uint64_t offset, int64_t* out, const int32_t* const* meta,
                             int n_meta);

#ifdef __cplusplus
}
#endif
#endif /* COALA_HIP_H */
