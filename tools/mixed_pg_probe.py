"""Which backend serves which call on the default "cpu:gloo,cuda:nccl" world?  Two ranks on ONE GPU: everything routed to gloo
works, anything routed to RCCL fails with a duplicate-GPU error -- which is the information wanted (development tool).
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/mixed_pg_probe.py"""
import datetime
import os

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)


def step(name, fn):
    try:
        r = fn()
        print(rank, "OK  ", name, "" if r is None else r, flush=True)
    except Exception as e:  # noqa: BLE001
        print(rank, "FAIL", name, repr(e)[:160].replace("\n", " "), flush=True)


dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
step("barrier()", lambda: dist.barrier())
out = [None] * world
step("all_gather_object", lambda: (dist.all_gather_object(out, rank), out)[1])
g = dist.new_group(ranks=list(range(world)), backend="gloo")
t = torch.ones(3) * rank
step("gloo subgroup all_reduce(cpu)", lambda: (dist.all_reduce(t, group=g), t.tolist())[1])
t2 = torch.ones(3) * rank
step("default group all_reduce(cpu)", lambda: (dist.all_reduce(t2), t2.tolist())[1])
t3 = torch.ones(3) * rank
step("default group broadcast(cpu)", lambda: (dist.broadcast(t3, src=0), t3.tolist())[1])
step("barrier(group=gloo subgroup)", lambda: dist.barrier(group=g))
tc = torch.ones(3, device="cuda") * rank
step("default group all_reduce(cuda) [expected to fail here: two ranks, one GPU]", lambda: (dist.all_reduce(tc), tc.tolist())[1])
