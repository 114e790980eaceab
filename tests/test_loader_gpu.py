"""End-to-end on one MI355X: shared pinned region across two processes, and the COALA_GNN_DataLoader iterator feeding a
tiny GraphSAGE step (the consumer contract of examples/sbatch_ssd_gnn_train.py:129-145)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from _util import ColorFiles

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_shared_uva_two_processes_one_gpu():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_shm_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and f"rank {r} ok" in out, out[-3000:]


@pytest.mark.parametrize("backend,method,prefetch", [("isolated", "node_color", 0), ("nvshmem", "baseline", 0), ("isolated", "baseline", 2),
                                                     ("nccl", "node_color", 1), ("nvshmem", "baseline", 2)])
def test_dataloader_epoch_and_sage_step(hiplib, oracle, tmp_path, backend, method, prefetch):
    import torch
    from COALA_GNN import COALA_GNN_DataLoader, MPI_Comm_Manager, Node_Distributor, SSD_INFO
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import alloc_pinned_table, block_colors, feature_rows_torch, powerlaw_csc
    torch.manual_seed(0)
    n_nodes, dim, batch, fan = 20000, 128, 64, [5, 5]
    table = alloc_pinned_table(n_nodes, dim, seed=3, device=0)
    indptr, indices = powerlaw_csc(n_nodes, 8.0, seed=1, device="cuda")
    labels = (torch.arange(n_nodes, device="cuda") * 7) % 5
    color, tk, sc, ncol = block_colors(n_nodes, nodes_per_color=512)
    files = ColorFiles(tmp_path, color, tk, sc)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group(backend)
    train_ids = torch.randperm(int(0.6 * n_nodes), generator=torch.Generator().manual_seed(0))[:64 * 12]
    nd = Node_Distributor(comm, train_ids, batch, files.color_file, files.topk_file, files.score_file, parsing_method=method)
    sampler = NeighborSampler(fan, seed=5)
    g = sampler.make_graph(indptr, indices, ndata={"labels": labels})
    loader = COALA_GNN_DataLoader(SSD_INFO(1, dim * 4, 1024, 0), nd, g, sampler, batch, dim, fan, 4, "cuda:0", refresh_counter=3,
                                  cache_backend=backend, sim_buf=table, num_rows=n_nodes, prefetch=prefetch)
    assert loader.total_count == 12 - 1  # COALA_GNN_DataLoader.py:141
    w1 = torch.nn.Linear(2 * dim, 32).cuda()
    w2 = torch.nn.Linear(64, 5).cuda()
    opt = torch.optim.Adam(list(w1.parameters()) + list(w2.parameters()), lr=1e-2)
    for epoch in range(2):
        steps, seen = 0, []
        for input_nodes, seeds, blocks, feat in loader:
            assert feat.shape == (input_nodes.numel(), dim) and feat.is_cuda
            assert torch.equal(feat, feature_rows_torch(input_nodes, dim, 3))      # the cache is a transparent gather
            assert torch.equal(blocks[0].src_nodes, input_nodes) and blocks[-1].num_dst == batch
            batch_labels = blocks[-1].dstdata["labels"]
            blocks = [b.int().to("cuda:0") for b in blocks]
            h = feat
            h = torch.relu(w1(torch.cat([h[: blocks[0].num_dst], blocks[0].mean_aggregate(h)], 1)))   # SAGE-mean layer 1
            h = w2(torch.cat([h[: blocks[1].num_dst], blocks[1].mean_aggregate(h)], 1))               # layer 2
            loss = torch.nn.functional.cross_entropy(h, batch_labels.view(-1))
            opt.zero_grad(); loss.backward(); opt.step()
            assert torch.isfinite(loss)
            seen.append(seeds.cpu())
            steps += 1
        assert steps == 11
        assert torch.equal(torch.cat(seen), train_ids[: 11 * batch])  # one domain: contiguous striping in order
    loader.print_stats()
    hit, miss, bad = loader.COALA_GNN_Manager.COALA_GNN_Cache.stats()
    assert bad == 0
    cc = np.zeros(ncol + 1, dtype=np.int32)
    loader.COALA_GNN_Manager.get_cache_data(cc.ctypes.data, ncol + 1)
    geo = loader.COALA_GNN_Manager.COALA_GNN_Cache.geometry()
    keys, _, _ = loader.COALA_GNN_Manager.COALA_GNN_Cache.dump()
    assert cc[1:].sum() == int((keys != np.uint64(0xFFFFFFFFFFFFFFFF)).sum())  # colour occupancy == live lines
    assert cc[0] == -cc[1:].sum()                                                 # colour 0 absorbed the first-touch decrements
    del loader
    table.close()


def test_shared_csc_dataset_from_npy(hiplib, oracle, tmp_path):
    """Row f-4: the reference's on-disk layout (csc_indptr/csc_indices/node_feat/node_label_19 .npy) -> shm cold tier + HBM CSC."""
    import torch
    from COALA_GNN import MPI_Comm_Manager
    from COALA_GNN.datasets import SharedCSCDataset
    from COALA_GNN.sampler import NeighborSampler
    rng = np.random.default_rng(0)
    n, dim = 3000, 64
    deg = rng.integers(1, 9, size=n)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    indices = rng.integers(0, n, size=int(indptr[-1])).astype(np.int64)
    feat = oracle.make_features(n, dim, seed=6)
    np.save(tmp_path / "csc_indptr.npy", indptr); np.save(tmp_path / "csc_indices.npy", indices)
    np.save(tmp_path / "node_feat.npy", feat); np.save(tmp_path / "node_label_19.npy", (np.arange(n) % 19).astype(np.int64))
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    ds = SharedCSCDataset(str(tmp_path), comm, "cuda:0", shm_name=f"/coala_ds_test_{os.getpid()}")
    g = ds[0]
    assert (g.num_nodes, g.num_edges, ds.dim) == (n, len(indices), dim)
    assert int(g.ndata["train_mask"].sum()) == int(0.6 * n)
    assert np.array_equal(ds.feat_data[::37].cpu().numpy(), feat[::37])
    P = hiplib
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = P.Isolated_Cache(ctrl, None, 0, 1, 1, ds.feat_data.data_ptr(), num_rows=n)
    inp, _, blocks = NeighborSampler([3, 3], seed=1).sample(g, torch.arange(0, 300, device="cuda"))
    out = torch.empty((inp.numel(), dim), dtype=torch.float32, device="cuda")
    cache.read_feature(out.data_ptr(), inp.data_ptr(), inp.numel())
    assert np.array_equal(out.cpu().numpy(), feat[inp.cpu().numpy()])
    assert torch.equal(blocks[-1].dstdata["labels"], torch.arange(0, 300, device="cuda") % 19)
    cache.close()
    ds.close()


@pytest.mark.parametrize("cold_tier", ["private", "partitioned"])
def test_shared_csc_dataset_private_cold_tiers(hiplib, oracle, tmp_path, cold_tier):
    """The faster cold-tier kinds of SharedCSCDataset (hipHostMalloc instead of shm + hipHostRegister): the whole table private to the
    rank, or only the rows the rank owns (here: rank 1 of a 3-rank machine) behind a partitioned cache shard."""
    import torch
    from COALA_GNN.datasets import SharedCSCDataset
    rng = np.random.default_rng(1)
    n, dim = 2999, 128
    deg = rng.integers(1, 6, size=n)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    np.save(tmp_path / "csc_indptr.npy", indptr); np.save(tmp_path / "csc_indices.npy", rng.integers(0, n, size=int(indptr[-1])).astype(np.int64))
    feat = oracle.make_features(n, dim, seed=9)
    np.save(tmp_path / "node_feat.npy", feat); np.save(tmp_path / "node_label_19.npy", (np.arange(n) % 19).astype(np.int64))

    class Comm:     # the topology a rank of a 3-GPU machine would see (no process group needed for these tiers)
        node_id, local_rank, local_size, device_index = 0, (1 if cold_tier == "partitioned" else 0), (3 if cold_tier == "partitioned" else 1), 0
    ds = SharedCSCDataset(str(tmp_path), Comm(), "cuda:0", cold_tier=cold_tier)
    G, r = Comm.local_size, Comm.local_rank
    assert ds.cold_partitioned == (cold_tier == "partitioned") and ds.feat_data.rows == len(range(r, n, G))
    assert np.array_equal(ds.feat_data.array, feat[r::G])
    P = hiplib
    ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
    cache = P.Isolated_Cache(ctrl, None, r, G, 1, ds.feat_data.data_ptr(), num_rows=n, rank=r, cold_partitioned=ds.cold_partitioned)
    ids = np.arange(r, n, G)[::3].astype(np.int64)            # ids this owner serves
    out = torch.empty((len(ids), dim), dtype=torch.float32, device="cuda")
    (cache.serve if G > 1 else cache.read_feature)(out.data_ptr(), torch.from_numpy(ids).cuda().data_ptr(), len(ids))
    assert out.cpu().numpy().tobytes() == feat[ids].tobytes()
    cache.close()
    ds.close()


def test_example_training_script_runs():
    """examples/train_synthetic.py: the reference's training loop on the API mirror, end to end (colouring tool included)."""
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_synthetic.py"), "--nodes", "60000", "--dim", "64",
                          "--batch_size", "256", "--epochs", "2", "--cache_size", "4", "--prefetch", "1"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("Epoch Time:") == 2 and "GPU hit ratio:" in out.stdout and "Aggregation time:" in out.stdout
    assert "final loss" in out.stdout and "Test Acc" in out.stdout      # training, then the evaluation over the test nodes through a second loader


def test_loader_modes_are_equivalent(hiplib, oracle, tmp_path):
    """The reference's blocking __next__ (sync_fetch=True), the stream-ordered default and the prefetching producer deliver the
    same batches in the same order and leave the same tag table, cursors and counters: they differ in scheduling only."""
    import torch
    from COALA_GNN import COALA_GNN_DataLoader, MPI_Comm_Manager, Node_Distributor, SSD_INFO
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import alloc_pinned_table, block_colors, powerlaw_csc
    n_nodes, dim, batch, fan = 30000, 256, 128, [5, 5]
    table = alloc_pinned_table(n_nodes, dim, seed=7, device=0)
    indptr, indices = powerlaw_csc(n_nodes, 8.0, seed=2, device="cuda")
    color, tk, sc, ncol = block_colors(n_nodes, nodes_per_color=512)
    files = ColorFiles(tmp_path, color, tk, sc)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    train_ids = torch.randperm(int(0.6 * n_nodes), generator=torch.Generator().manual_seed(3))[:batch * 25]
    results = []
    for kw in ({"sync_fetch": True}, {}, {"prefetch": 2}):
        nd = Node_Distributor(comm, train_ids, batch, files.color_file, files.topk_file, files.score_file, parsing_method="node_color")
        sampler = NeighborSampler(fan, seed=9)
        g = sampler.make_graph(indptr, indices)
        loader = COALA_GNN_DataLoader(SSD_INFO(1, dim * 4, 1024, 0), nd, g, sampler, batch, dim, fan, 2, "cuda:0", refresh_counter=4,
                                      cache_backend="isolated", sim_buf=table, num_rows=n_nodes, **kw)
        seen = []
        for input_nodes, seeds, blocks, feat in loader:
            # a consumer that only uses the stream contract: work ordered on the current stream
            seen.append((input_nodes.clone(), seeds.clone(), feat.sum(dim=1), blocks[0].nbr.clone()))
        torch.cuda.synchronize()
        cache = loader.COALA_GNN_Manager.COALA_GNN_Cache
        cc = np.zeros(ncol + 1, dtype=np.int32)
        cache.get_cache_data(cc.ctypes.data, ncol + 1)
        results.append((seen, cache.stats(), cache.dump(), cc, loader.COALA_GNN_Manager.get_aggregate_time()))
        del loader, nd
    ref = results[0]
    assert len(ref[0]) == 24 and ref[4] > 0
    for other in results[1:]:
        assert len(other[0]) == len(ref[0]) and other[1] == ref[1] and other[4] > 0
        for (a_in, a_seed, a_sum, a_nbr), (b_in, b_seed, b_sum, b_nbr) in zip(ref[0], other[0]):
            assert torch.equal(a_in, b_in) and torch.equal(a_seed, b_seed) and torch.equal(a_sum, b_sum) and torch.equal(a_nbr, b_nbr)
        for x, y in zip(ref[2], other[2]):
            assert np.array_equal(x, y)
        assert np.array_equal(ref[3], other[3])
    table.close()


def test_prefetching_loader_can_be_abandoned_mid_epoch(hiplib, oracle, tmp_path):
    """A consumer that leaves the loop early must not leave the producer thread blocked on its queue: close() (or dropping the
    loader) returns promptly and the process can go on using the GPU."""
    import time
    import torch
    from COALA_GNN import COALA_GNN_DataLoader, MPI_Comm_Manager, Node_Distributor, SSD_INFO
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import alloc_pinned_table, block_colors, powerlaw_csc
    n_nodes, dim, batch, fan = 20000, 128, 64, [5, 5]
    table = alloc_pinned_table(n_nodes, dim, seed=3, device=0)
    indptr, indices = powerlaw_csc(n_nodes, 8.0, seed=1, device="cuda")
    color, tk, sc, _ = block_colors(n_nodes, nodes_per_color=512)
    files = ColorFiles(tmp_path, color, tk, sc)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    train_ids = torch.randperm(int(0.6 * n_nodes), generator=torch.Generator().manual_seed(0))[:64 * 40]
    nd = Node_Distributor(comm, train_ids, batch, files.color_file, files.topk_file, files.score_file, parsing_method="baseline")
    sampler = NeighborSampler(fan, seed=5)
    g = sampler.make_graph(indptr, indices)
    loader = COALA_GNN_DataLoader(SSD_INFO(1, dim * 4, 1024, 0), nd, g, sampler, batch, dim, fan, 4, "cuda:0",
                                  cache_backend="isolated", sim_buf=table, num_rows=n_nodes, prefetch=2)
    for step, (input_nodes, seeds, blocks, feat) in enumerate(loader):
        if step == 3:
            break
    time.sleep(0.3)                      # the producer has filled its queue and is waiting
    producer = loader._producer
    assert producer is not None and producer.is_alive()
    t0 = time.time()
    loader.close()
    assert time.time() - t0 < 5.0 and not producer.is_alive()
    del loader
    torch.cuda.synchronize()
    table.close()


def test_nvshmem_backend_prefetch_with_slow_consumer(hiplib, oracle, tmp_path):
    """with the "nvshmem" backend and a prefetching producer, a consumer whose kernels are still reading step t's rows
    while the producer fetches step t+3 must never see those rows overwritten (the round-1 output ring could be).  The consumer
    here keeps a long-running kernel between receiving the rows and reading them, and never synchronises per step."""
    import torch
    from COALA_GNN import COALA_GNN_DataLoader, MPI_Comm_Manager, Node_Distributor, SSD_INFO
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import alloc_pinned_table, block_colors, feature_rows_torch, powerlaw_csc
    n_nodes, dim, batch, fan = 30000, 256, 128, [5, 5]
    table = alloc_pinned_table(n_nodes, dim, seed=7, device=0)
    indptr, indices = powerlaw_csc(n_nodes, 8.0, seed=2, device="cuda")
    color, tk, sc, _ = block_colors(n_nodes, nodes_per_color=512)
    files = ColorFiles(tmp_path, color, tk, sc)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("nvshmem")
    train_ids = torch.randperm(int(0.6 * n_nodes), generator=torch.Generator().manual_seed(3))[:batch * 21]
    nd = Node_Distributor(comm, train_ids, batch, files.color_file, files.topk_file, files.score_file, parsing_method="baseline")
    sampler = NeighborSampler(fan, seed=9)
    g = sampler.make_graph(indptr, indices)
    loader = COALA_GNN_DataLoader(SSD_INFO(1, dim * 4, 1024, 0), nd, g, sampler, batch, dim, fan, 2, "cuda:0",
                                  cache_backend="nvshmem", sim_buf=table, num_rows=n_nodes, prefetch=2)
    busy = torch.rand((4096, 4096), device="cuda")
    ok = []
    for input_nodes, seeds, blocks, feat in loader:
        for _ in range(6):                       # ~ milliseconds of consumer work queued BEFORE the rows are read
            busy = (busy @ busy).clamp_(0, 1)
        ok.append(torch.equal(feat, feature_rows_torch(input_nodes, dim, 7)))   # enqueued behind the busy work, no host sync
    torch.cuda.synchronize()
    assert len(ok) == 20 and all(ok)
    del loader
    table.close()


def test_shm_creator_never_reuses_a_stale_segment(hiplib):
    """a POSIX shm object left behind by a crashed run must not leak its old contents into a new run."""
    import ctypes as C
    import mmap
    P = hiplib
    name = f"/coala_stale_test_{os.getpid()}"
    size = 1 << 20
    fd = os.open("/dev/shm" + name, os.O_CREAT | os.O_RDWR, 0o600)   # the "crashed run": junk left in the segment
    os.ftruncate(fd, size)
    m = mmap.mmap(fd, size)
    m[:] = b"\xAB" * size
    m.close()
    os.close(fd)
    mgr = P.SharedUVAManager(name, size, 0, 0, 0, local_rank=0, device=0)
    buf = np.frombuffer((C.c_ubyte * size).from_address(mgr.get_host_ptr()), dtype=np.uint8)
    assert not buf.any()
    mgr.cleanup()
    assert not os.path.exists("/dev/shm" + name)


@pytest.mark.parametrize("steps", [1, 2, 3])
def test_pipelined_loader_short_epochs(hiplib, oracle, tmp_path, steps):
    """Epochs shorter than the depth of the default loader's pipeline (sample two steps ahead, fetch one ahead), two epochs in a
    row: the same batches as the blocking loader, StopIteration exactly after `steps` items, and the second epoch starts clean."""
    import torch
    from COALA_GNN import COALA_GNN_DataLoader, MPI_Comm_Manager, Node_Distributor, SSD_INFO
    from COALA_GNN.sampler import NeighborSampler
    from COALA_GNN.synthetic import alloc_pinned_table, block_colors, feature_rows_torch, powerlaw_csc
    n_nodes, dim, batch, fan = 8000, 64, 32, [3, 3]
    table = alloc_pinned_table(n_nodes, dim, seed=2, device=0)
    indptr, indices = powerlaw_csc(n_nodes, 6.0, seed=5, device="cuda")
    color, tk, sc, ncol = block_colors(n_nodes, nodes_per_color=256)
    files = ColorFiles(tmp_path, color, tk, sc)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    train_ids = torch.randperm(int(0.6 * n_nodes), generator=torch.Generator().manual_seed(4))[:batch * (steps + 1)]
    seen = {}
    for name, kw in (("blocking", {"sync_fetch": True}), ("pipelined", {})):
        nd = Node_Distributor(comm, train_ids, batch, files.color_file, files.topk_file, files.score_file, parsing_method="baseline")
        sampler = NeighborSampler(fan, seed=1)
        g = sampler.make_graph(indptr, indices)
        loader = COALA_GNN_DataLoader(SSD_INFO(1, dim * 4, 1024, 0), nd, g, sampler, batch, dim, fan, 1, "cuda:0", refresh_counter=2,
                                      cache_backend="isolated", sim_buf=table, num_rows=n_nodes, **kw)
        assert len(loader) == steps
        got = []
        for epoch in range(2):
            n = 0
            for input_nodes, seeds, blocks, feat in loader:
                assert torch.equal(feat, feature_rows_torch(input_nodes, dim, 2))
                got.append((epoch, input_nodes.clone(), seeds.clone()))
                n += 1
            assert n == steps
        torch.cuda.synchronize()
        seen[name] = (got, loader.COALA_GNN_Manager.COALA_GNN_Cache.stats())
        del loader, nd
    a, b = seen["blocking"], seen["pipelined"]
    assert a[1] == b[1] and len(a[0]) == len(b[0]) == 2 * steps
    for (ea, ia, sa), (eb, ib, sb) in zip(a[0], b[0]):
        assert ea == eb and torch.equal(ia, ib) and torch.equal(sa, sb)
    table.close()


def test_shared_csc_dataset_from_edge_index(hiplib, oracle, tmp_path):
    """Row f-4: a dataset directory that only holds the edge list (edge_index.npy, IGB layout) -- the conversion the reference does
    with DGL runs on the GPU when the dataset is opened; same graph as from the preprocessed csc_*.npy files.  Then the
    conversion alone at 60 M edges on the GPU: a valid CSC whose columns keep the order of the edge list."""
    import torch
    from COALA_GNN import MPI_Comm_Manager
    from COALA_GNN.datasets import SharedCSCDataset, csc_from_edge_index
    rng = np.random.default_rng(1)
    n, e, dim = 4000, 30000, 32
    src = rng.integers(0, n, size=e).astype(np.int64)
    dst = rng.integers(0, n, size=e).astype(np.int64)
    feat = oracle.make_features(n, dim, seed=2)
    np.save(tmp_path / "edge_index.npy", np.stack([src, dst], axis=1))
    np.save(tmp_path / "node_feat.npy", feat)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    ds = SharedCSCDataset(str(tmp_path), comm, "cuda:0", shm_name=f"/coala_ds_edges_{os.getpid()}")
    g = ds[0]
    perm = np.argsort(dst, kind="stable")
    assert (g.num_nodes, g.num_edges) == (n, e)
    assert np.array_equal(g.indptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]))
    assert np.array_equal(g.indices.cpu().numpy(), src[perm])
    ds.close()
    # the conversion at size, on the GPU
    N, E = 10_000_000, 60_000_000
    gen = torch.Generator(device="cuda").manual_seed(3)
    s = torch.randint(0, N, (E,), generator=gen, device="cuda")
    d = torch.randint(0, N, (E,), generator=gen, device="cuda")
    indptr, indices, eids = csc_from_edge_index(s, d, N)
    assert int(indptr[0]) == 0 and int(indptr[-1]) == E and bool((indptr[1:] >= indptr[:-1]).all())
    col = torch.repeat_interleave(torch.arange(N, device="cuda"), indptr[1:] - indptr[:-1])
    assert torch.equal(d[eids], col) and torch.equal(s[eids], indices)
    same_col = col[1:] == col[:-1]
    assert bool((eids[1:][same_col] > eids[:-1][same_col]).all())   # inside a column: the order of the edge list


@pytest.mark.parametrize("layout", ["IGB", "OGB"])
def test_shared_csc_dataset_reference_directory_layouts(hiplib, oracle, tmp_path, layout):
    """The two directory trees the reference's loaders read (examples/ssd_gnn_dataloader.py:401-563 IGB, :687-854 OGB): file
    locations, IGB's first-60/20/20 split by node id, OGB's NaN labels (unlabelled) and its split over the labelled nodes."""
    import torch
    from COALA_GNN import MPI_Comm_Manager
    from COALA_GNN.datasets import SharedCSCDataset
    rng = np.random.default_rng(2)
    n, e, dim = 2000, 12000, 16
    src, dst = rng.integers(0, n, size=e).astype(np.int64), rng.integers(0, n, size=e).astype(np.int64)
    perm = np.argsort(dst, kind="stable")
    indptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.int64)
    feat = oracle.make_features(n, dim, seed=3)
    if layout == "IGB":
        paper = tmp_path / "small" / "processed" / "paper"
        cites = tmp_path / "small" / "processed" / "paper__cites__paper"
        paper.mkdir(parents=True); cites.mkdir(parents=True)
        np.save(paper / "node_feat.npy", feat)
        np.save(paper / "node_label_19.npy", (np.arange(n) % 19).astype(np.float32))
        np.save(cites / "csc_indptr.npy", indptr); np.save(cites / "csc_indices.npy", src[perm]); np.save(cites / "csc_edge_ids.npy", perm)
        kw = {"layout": "IGB", "dataset_size": "small"}
    else:
        raw = tmp_path / "raw"
        raw.mkdir()
        labels = (np.arange(n) % 172).astype(np.float32)
        labels[rng.random(n) < 0.7] = np.nan                         # papers100M: most nodes carry no label
        np.save(raw / "node_feat.npy", feat); np.save(raw / "node_label.npy", labels)
        np.save(raw / "edge_index.npy", np.stack([src, dst], axis=0))  # [2, E]; no preprocessed CSC: converted when opened
        kw = {"layout": "OGB", "num_classes": 172}
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    ds = SharedCSCDataset(str(tmp_path), comm, "cuda:0", shm_name=f"/coala_ds_{layout}_{os.getpid()}", **kw)
    g = ds[0]
    assert (g.num_nodes, g.num_edges, ds.dim) == (n, e, dim)
    assert np.array_equal(g.indptr.cpu().numpy(), indptr) and np.array_equal(g.indices.cpu().numpy(), src[perm])
    assert np.array_equal(ds.feat_data[::29].cpu().numpy(), feat[::29])
    tr, va, te = (g.ndata[k].numpy() for k in ("train_mask", "val_mask", "test_mask"))
    assert not (tr & va).any() and not (tr & te).any() and not (va & te).any()
    lab = g.ndata["label"].cpu().numpy()
    if layout == "IGB":
        assert tr[: int(0.6 * n)].all() and tr.sum() == int(0.6 * n) and va.sum() == int(0.2 * n) and (tr | va | te).all()
        assert np.array_equal(lab, np.arange(n) % 19)
    else:
        known = np.where(~np.isnan(labels))[0]
        assert np.array_equal(np.where(tr)[0], known[: int(0.6 * len(known))]) and (tr | va | te).sum() == len(known)
        assert np.array_equal(lab[known], labels[known].astype(np.int64)) and (lab[np.isnan(labels)] == -1).all()
    ds.close()


def test_example_training_script_on_a_dataset_directory(tmp_path):
    """examples/train_synthetic.py --path: the reference's IGB directory tree (features, labels, the raw edge list) -> shared pinned
    cold tier + CSC converted on the GPU -> colouring -> the training loop."""
    rng = np.random.default_rng(4)
    n, e, dim = 30000, 240000, 32
    paper = tmp_path / "small" / "processed" / "paper"
    cites = tmp_path / "small" / "processed" / "paper__cites__paper"
    paper.mkdir(parents=True); cites.mkdir(parents=True)
    np.save(paper / "node_feat.npy", rng.random((n, dim), dtype=np.float32))
    np.save(paper / "node_label_19.npy", (np.arange(n) % 19).astype(np.float32))
    np.save(cites / "edge_index.npy", np.stack([rng.integers(0, n, size=e), rng.integers(0, n, size=e)], axis=1).astype(np.int64))
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_synthetic.py"), "--path", str(tmp_path), "--data", "IGB",
                          "--dataset_size", "small", "--batch_size", "128", "--epochs", "1", "--cache_size", "2",
                          # the rest as in the reference's launch scripts (examples/Distribution_compare_script.sh:27)
                          "--fan_out", "10,5,5", "--num_layers", "2", "--feat_cpu", "--model_type", "sage", "--cache_backend", "isolated",
                          "--distribution", "baseline"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("Epoch Time:") == 1 and "GPU hit ratio:" in out.stdout and "final loss" in out.stdout and "Test Acc" in out.stdout
    assert f"Total number of iterations: {int(0.6 * n) // 128 - 1}" in out.stdout
