"""Small consumer of the loader's 4-tuple, used by bench.py's epoch leg and the tests: a 2-layer GraphSAGE (mean) in plain
torch on the native Block objects.  Stands where examples/models.py:DistSAGE + dgl.nn.SAGEConv stand in the reference's
training script (examples/sbatch_ssd_gnn_train.py:98-145); the model itself is out of scope (dense compute downstream of
the path), this is harness plumbing."""
import time

import torch

__all__ = ["SageMean", "train_steps", "FlatGradAllReduce"]


class SageMean(torch.nn.Module):
    def __init__(self, in_dim, hidden, n_classes, n_layers=2):
        super().__init__()
        dims = [in_dim] + [hidden] * (n_layers - 1) + [n_classes]
        self.lin_self = torch.nn.ModuleList(torch.nn.Linear(dims[i], dims[i + 1]) for i in range(n_layers))
        self.lin_nbr = torch.nn.ModuleList(torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(n_layers))

    def forward(self, blocks, h):
        for i, b in enumerate(blocks):
            h = self.lin_self[i](b.dst_rows(h)) + self.lin_nbr[i](b.mean_aggregate(h))
            if i + 1 < len(blocks):
                h = torch.relu(h)
        return h


class FlatGradAllReduce(object):
    """Data-parallel gradient averaging for the harness model: every parameter's gradient is a view into ONE flat buffer, averaged across the
    ranks with a single all-reduce per step (RCCL for GPU tensors).  Stands where DistributedDataParallel stands in the reference's script
    (examples/sbatch_ssd_gnn_train.py:112), with the same result for a model without unused parameters; torch's DDP costs this 1.6 ms
    training step another 1.0 ms of host time per iteration (measured at world size 1: 2.53 against 1.53 ms/step), which would bound a
    multi-GPU epoch whose fetch takes a third of that."""

    def __init__(self, model, group=None):
        import torch.distributed as dist
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=self.params[0].dtype, device=self.params[0].device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._dist = dist

    def zero(self):
        self.flat.zero_()         # (optimizer.zero_grad() would drop the views)

    def reduce(self):
        if self._dist.is_initialized():
            self._dist.all_reduce(self.flat, group=self.group)
            if self.world > 1:
                self.flat.div_(self.world)


def train_steps(loader, model, optimizer, max_steps, device, stop_check=None, check_every=64, grad_sync=None):
    """Runs up to max_steps iterations of the reference's training loop body; returns (steps, seconds, sampled_nodes).
    stop_check (optional): called every check_every steps on the host; a true answer ends the loop early (a caller with a time limit; in a
    multi-rank run it must give every rank the same answer at the same step -- the loop is full of collectives)."""
    loss_fn = torch.nn.CrossEntropyLoss()
    steps = nodes = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for input_nodes, seeds, blocks, feat in loader:
        nodes += len(input_nodes)
        labels = blocks[-1].dstdata["labels"].view(-1).to(device)
        blocks = [b.int().to(device) for b in blocks]
        loss = loss_fn(model(blocks, feat), labels)
        if grad_sync is not None:
            grad_sync.zero()
            loss.backward()
            grad_sync.reduce()
        else:
            optimizer.zero_grad()
            loss.backward()
        optimizer.step()
        steps += 1
        if steps >= max_steps:
            break
        if stop_check is not None and steps % check_every == 0 and stop_check():
            break
    torch.cuda.synchronize()
    return steps, time.perf_counter() - t0, nodes
