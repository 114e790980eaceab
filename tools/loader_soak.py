#!/usr/bin/env python3
"""Long-run check of the loaders' stream hand-offs: every delivered feature tensor is compared with the table's formula ON THE
CONSUMER'S STREAM, behind a deliberately long "training" kernel sequence that keeps reading the previous step's rows while the
next fetch runs on the side stream.  A missing wait / record_stream shows up as a mismatch.  Development tool.

  python tools/loader_soak.py [--steps 3000 --rows 4000000 --dim 1024 --cache-mb 1024]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from COALA_GNN import COALA_GNN_DataLoader, MPI_Comm_Manager, Node_Distributor, SSD_INFO  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402
from COALA_GNN.synthetic import alloc_pinned_table, block_colors, feature_rows_torch, powerlaw_csc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--cache-mb", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    comm = MPI_Comm_Manager(0)
    comm.initialize_nested_process_group("isolated")
    table = alloc_pinned_table(args.rows, args.dim, 0, 0)
    indptr, indices = powerlaw_csc(args.rows, 10.0, seed=0, device="cuda")
    tmp = tempfile.mkdtemp(prefix="coala_soak_")
    color, tk, sc, _ = block_colors(args.rows, nodes_per_color=4096)
    files = [os.path.join(tmp, f) for f in ("color.npy", "topk.npy", "score.npy")]
    np.save(files[0], color); np.save(files[1], tk); np.save(files[2], sc)
    train = torch.randperm(int(0.6 * args.rows), generator=torch.Generator().manual_seed(1))[: (args.steps + 1) * args.batch]
    junk = torch.randn(2048, 2048, device="cuda")
    for name, kw in (("default (one thread, two side streams)", {}), ("prefetch=2 (producer thread)", {"prefetch": 2})):
        sampler = NeighborSampler([5, 5], seed=0)
        g = sampler.make_graph(indptr, indices)
        nd = Node_Distributor(comm, train, args.batch, *files, parsing_method="baseline")
        loader = COALA_GNN_DataLoader(SSD_INFO(1, args.dim * 4, 1024, 0), nd, g, sampler, args.batch, args.dim, [5, 5], args.cache_mb, "cuda:0",
                                      cache_backend="isolated", sim_buf=table, num_rows=args.rows, **kw)
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        prev = None
        t0 = time.time()
        steps = 0
        for input_nodes, seeds, blocks, feat in loader:
            bad += (feat != feature_rows_torch(input_nodes, args.dim, 0)).any(dim=1).sum()
            for _ in range(3):                       # a "training step" that runs long after the host has moved on
                junk = torch.tanh(junk @ junk * 1e-3)
            if prev is not None:                     # ... and still reads the PREVIOUS step's rows at its end
                bad += (prev[1] != feature_rows_torch(prev[0], args.dim, 0)).any(dim=1).sum()
            prev = (input_nodes, feat)
            steps += 1
        torch.cuda.synchronize()
        print(f"{name}: {steps} steps in {time.time() - t0:.1f}s, rows that differ from the table: {int(bad)}", flush=True)
        assert int(bad) == 0
        del loader, nd
        g.close()
    table.close()


if __name__ == "__main__":
    main()
