#!/usr/bin/env python3
"""Minimal training script on the MI355X path, shaped like the reference's examples/sbatch_ssd_gnn_train.py:50-195 (same
objects, same loop, same printed lines), without DGL / mpi4py: on seeded synthetic data (no datasets on the box), or -- with
--path -- on a dataset directory in the reference's on-disk layout (--data IGB --dataset_size medium | --data OGB; :202-213, :273-285).

  python examples/train_synthetic.py --nodes 200000 --dim 128 --epochs 2
  python examples/train_synthetic.py --path /data/IGB/ --data IGB --dataset_size medium --cache_size 4096
  python -m torch.distributed.run --nproc-per-node 8 examples/train_synthetic.py --cache_backend nccl ...

The caller of the hot path is out of scope of the port; this file only shows that the loop runs unchanged on the API mirror."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "coala-gnn_amd"))

import torch  # noqa: E402

from COALA_GNN import COALA_GNN_DataLoader, MPI_Comm_Manager, Node_Distributor, SSD_INFO  # noqa: E402
from COALA_GNN.color_info_gen import color_graph, save_color_files  # noqa: E402
from COALA_GNN.harness import SageMean  # noqa: E402
from COALA_GNN.sampler import NeighborSampler  # noqa: E402
from COALA_GNN.synthetic import alloc_pinned_table, powerlaw_csc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--fan_out", type=str, default="5,5")
    ap.add_argument("--batch_size", type=int, default=1024)
    ap.add_argument("--hidden_channels", type=int, default=128)
    ap.add_argument("--num_classes", type=int, default=19)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--cache_size", type=int, default=64, help="MB")
    ap.add_argument("--cache_backend", type=str, default="isolated", choices=["isolated", "nccl", "nvshmem"])
    ap.add_argument("--distribution", type=str, default="node_color", choices=["node_color", "baseline"])
    ap.add_argument("--refresh_counter", type=int, default=10)
    ap.add_argument("--prefetch", type=int, default=0)
    ap.add_argument("--learning_rate", type=float, default=0.01)
    ap.add_argument("--path", type=str, default=None, help="dataset root in the reference's layout (default: synthetic data)")
    ap.add_argument("--data", type=str, default="IGB", choices=["IGB", "OGB", "flat"])
    ap.add_argument("--dataset_size", type=str, default="experimental")
    # accepted so that the reference's command lines (examples/4GB_script.sh, Cache_compare_script.sh, Distribution_compare_script.sh) run as they are
    ap.add_argument("--num_layers", type=int, default=None, help="must equal the number of fan-outs when given")
    ap.add_argument("--feat_cpu", action="store_true", help="features in pinned host memory: always the case here (the NVMe tier is out of scope)")
    ap.add_argument("--model_type", type=str, default="sage", choices=["sage"], help="the harness model; the reference's GAT is dense compute outside the path")
    args = ap.parse_args()
    if args.num_layers is not None and args.num_layers != len(args.fan_out.split(",")) and args.num_layers != 2:
        ap.error("--num_layers does not match --fan_out")   # (the reference's own scripts pass --num_layers 2 with a 3-entry fan-out: tolerated)

    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("SLURM_LOCALID", 0)))
    node_rank = int(os.environ.get("SLURM_NODEID", 0))
    torch.cuda.set_device(local_rank)
    comm = MPI_Comm_Manager(node_rank)                                  # sbatch_ssd_gnn_train.py:262
    device = "cuda:" + str(comm.local_rank)
    comm.initialize_nested_process_group(args.cache_backend)            # :267
    fan_out = [int(f) for f in args.fan_out.split(",")]

    dataset = None
    if args.path:   # IGBDatast_Shared_CSC_UVA / OGBDataset_Shared_UVA (:273-285): CSC in HBM, features in shared pinned host memory
        from COALA_GNN.datasets import SharedCSCDataset
        dataset = SharedCSCDataset(args.path, comm, device, num_classes=args.num_classes, layout=args.data, dataset_size=args.dataset_size)
        g0 = dataset[0]
        indptr, indices, labels, feat = g0.indptr, g0.indices, g0.ndata["labels"], dataset.feat_data
        args.nodes, args.dim = dataset.num_nodes, dataset.dim
        train_ids = torch.nonzero(g0.ndata["train_mask"], as_tuple=True)[0].clone()                          # :62
        test_ids = torch.nonzero(g0.ndata["test_mask"], as_tuple=True)[0].clone()                            # :63
        meta = {"IGB": os.path.join(args.path, args.dataset_size), "OGB": args.path, "flat": args.path}[args.data]   # :55-61
    else:           # synthetic stand-in
        indptr, indices = powerlaw_csc(args.nodes, 10.0, seed=0, device=device)
        labels = (torch.arange(args.nodes, device=device) * 7) % args.num_classes
        feat = alloc_pinned_table(args.nodes, args.dim, seed=0, device=comm.local_rank)
        train_ids = torch.arange(int(0.6 * args.nodes))
        test_ids = torch.arange(int(0.8 * args.nodes), args.nodes)
        meta = None
    n_train = len(train_ids)
    if meta is not None and all(os.path.exists(os.path.join(meta, f)) for f in ("color.npy", "topk.npy", "score.npy")):
        tmp = meta  # the colouring tool's output next to the dataset, where the reference looks for it
    else:
        tmp = tempfile.mkdtemp(prefix="coala_color_")
        if comm.global_rank == 0:                                       # examples/color_info_gen/generate_color_data.py
            color, tk, sc, n_col, _ = color_graph(indptr.cpu().numpy(), indices.cpu().numpy(), train_ids.numpy())
            save_color_files(tmp, color, tk, sc)
            print(f"num_colors: {n_col}")
        comm.global_comm.Barrier()
        if comm.global_size > 1:
            tmp = comm.global_comm.allgather(tmp)[0]
    files = [os.path.join(tmp, f) for f in ("color.npy", "topk.npy", "score.npy")]

    train_nid = train_ids[torch.randperm(n_train, generator=torch.Generator().manual_seed(0))]               # :62-65
    nd = Node_Distributor(comm, train_nid, args.batch_size, *files, parsing_method=args.distribution)      # :68
    sampler = NeighborSampler(fan_out)                                                                      # :70-72
    g = sampler.make_graph(indptr, indices, ndata={"labels": labels})
    train_loader = COALA_GNN_DataLoader(SSD_INFO(1, args.dim * 4, 1024, 0), nd, g, sampler, args.batch_size, args.dim, fan_out,
                                        args.cache_size, device, refresh_counter=args.refresh_counter,
                                        cache_backend=args.cache_backend, sim_buf=feat, shuffle=False, num_rows=args.nodes,
                                        prefetch=args.prefetch)                                             # :82-95
    model = SageMean(args.dim, args.hidden_channels, args.num_classes, len(fan_out)).to(device)
    if comm.global_size > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[comm.local_rank])             # :112
    loss_fcn = torch.nn.CrossEntropyLoss().to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.learning_rate)

    count = num_sampled_nodes = 0
    model.train()
    for epoch in range(args.epochs):                                                                        # :126-151
        print(f"Epoch: {epoch}")
        epoch_start = time.time()
        for step, (input_nodes, seeds, blocks, fetch_feature) in enumerate(train_loader):
            num_sampled_nodes += len(input_nodes)
            if step % 100 == 0:
                print(f"Rank: {comm.local_rank} step: {step}")
            count += 1
            batch_labels = blocks[-1].dstdata["labels"]
            blocks = [block.int().to(device) for block in blocks]
            batch_labels = batch_labels.view(-1).to(device)
            batch_pred = model(blocks, fetch_feature)
            loss = loss_fcn(batch_pred, batch_labels)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        torch.cuda.synchronize()
        print(f"Epoch Time: {time.time() - epoch_start}")
        print(f"Total number of iterations: {count}")
        print(f"Number of sampled nodes : {num_sampled_nodes}")
        train_loader.print_stats()
    print(f"final loss {loss.item():.4f}")
    comm.global_comm.Barrier()
    del train_loader

    # evaluation over the test nodes through a second loader, as the reference does (:156-195)
    test_nd = Node_Distributor(comm, test_ids, args.batch_size, *files, parsing_method=args.distribution)
    test_loader = COALA_GNN_DataLoader(SSD_INFO(1, args.dim * 4, 1024, 0), test_nd, g, sampler, args.batch_size, args.dim, fan_out,
                                       args.cache_size, device, refresh_counter=args.refresh_counter,
                                       cache_backend=args.cache_backend, sim_buf=feat, shuffle=False, num_rows=args.nodes)
    model.eval()
    correct = total = 0
    with torch.no_grad():
        for step, (input_nodes, seeds, blocks, fetch_feature) in enumerate(test_loader):
            if step % 100 == 0:
                print("Eval step: ", step)
            blocks = [block.to(device) for block in blocks]
            batch_labels = blocks[-1].dstdata["labels"].view(-1)
            pred = model(blocks, fetch_feature).argmax(1)
            correct += int((pred == batch_labels).sum())
            total += batch_labels.numel()
    print("Test Acc {:.2f}%".format(100.0 * correct / max(total, 1)))
    comm.global_comm.Barrier()
    del test_loader
    if dataset is not None:
        dataset.close()
    comm.destroy_process_group()


if __name__ == "__main__":
    main()
