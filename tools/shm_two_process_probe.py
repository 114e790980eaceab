#!/usr/bin/env python3
"""The reference's shared cold tier as a NON-creator rank sees it (COALA_GNN_Modules/shared_UVA.cuh:78-100: shm_open of an existing
object, mmap, cudaHostRegister): this process creates the 41 GB segment through SharedUVAManager as local rank 0, fills it with the
synthetic table through its device alias, and stays alive holding its mapping while a SECOND process (bench.py --cold-tier shm
--shm-attach) maps the same object, registers it and runs the default fetch workload over its own mapping.

    python tools/shm_two_process_probe.py > gpurun_out/r03/bench_shm_second_mapping.json"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "coala-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
import COALA_GNN_Pybind as P  # noqa: E402
from COALA_GNN.Shared_Tensor import tensor_from_pointer  # noqa: E402
from COALA_GNN.synthetic import feature_rows_torch  # noqa: E402

rows, dim = int(os.environ.get("ROWS", 10_000_000)), 1024
name = f"/coala_probe_shared_{os.getpid()}"
torch.cuda.set_device(0)
t0 = time.perf_counter()
seg = P.SharedUVAManager(name, rows * dim * 4, 0, 0, 0, local_rank=0, device=0)
print(f"[creator] {rows * dim * 4 / 1e9:.2f} GB segment created + registered in {time.perf_counter() - t0:.2f}s", file=sys.stderr, flush=True)
alias = tensor_from_pointer(seg.get_device_ptr(), (rows, dim), torch.float32, "cuda:0")
for lo in range(0, rows, 1 << 18):
    hi = min(rows, lo + (1 << 18))
    feature_rows_torch(torch.arange(lo, hi, dtype=torch.int64, device="cuda"), dim, 0, out=alias[lo:hi])
torch.cuda.synchronize()
print(f"[creator] filled at {time.perf_counter() - t0:.2f}s; starting the second process", file=sys.stderr, flush=True)
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cold-tier", "shm", "--shm-attach", name, "--rows", str(rows), "--steps", "200", "--epoch-steps", "0",
       "--no-fanout-leg", "--no-color-affinity-leg", "--no-cpu-baseline", "--no-allhit"]
out = subprocess.run(cmd, capture_output=True, text=True)
sys.stderr.write(out.stderr[-3000:])
sys.stdout.write(out.stdout)
del alias
seg.cleanup()
sys.exit(out.returncode)
