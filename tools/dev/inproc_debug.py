"""debug: native in-process fetch, report which rows are wrong"""
import ctypes as C, os, sys, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "coala-gnn_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import oracle as O
O.build()
import COALA_GNN_Pybind as P
from COALA_GNN_Pybind import _capi
from COALA_GNN.COALA_GNN_Manager import NativeExchange
from _util import PinnedTable
L = _capi.load()
G, dim, cache_mb, rounds = int(sys.argv[1]), int(sys.argv[2]), 2, int(sys.argv[3])
num_rows = 12000
feat = O.make_features(num_rows, dim, seed=6)
ctrl = P.SSD_GNN_SSD_Controllers(1, 4096, 1024, 0, 0, dim, True)
tables = [PinnedTable(P, np.ascontiguousarray(feat[r::G])) for r in range(G)]
caches = [P.Isolated_Cache(ctrl, None, r, G, cache_mb, tables[r].device_ptr, num_rows=num_rows, rank=r, cold_partitioned=True) for r in range(G)]
group = C.c_void_p(); _capi.check(L.coala_comm_group_create(G, C.byref(group)))
exs = [NativeExchange(None, 0, r, G, 0, inproc_group=group, rounds=rounds) for r in range(G)]
rng = np.random.default_rng(77 + G)
steps = 3
plan = [[rng.choice(num_rows // 2, size=1500, replace=False).astype(np.int64) for _ in range(G)] for _ in range(steps)]
got = [[None] * G for _ in range(steps)]
bar = threading.Barrier(G, timeout=120)
def worker(r):
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for step in range(steps):
            ids = plan[step][r]
            idx = torch.from_numpy(ids).cuda()
            out = torch.full((len(ids), dim), -9.0, dtype=torch.float32, device="cuda")
            exs[r].fetch(caches[r], out.data_ptr(), idx.data_ptr(), len(ids))
            stream.synchronize()
            got[step][r] = out.cpu().numpy()
            bar.wait()
ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
[t.start() for t in ts]; [t.join() for t in ts]
for step in range(steps):
    for r in range(G):
        want = feat[plan[step][r]]
        bad = np.where((got[step][r] != want).any(axis=1))[0]
        if len(bad):
            ids = plan[step][r][bad]
            unwritten = (got[step][r][bad] == -9.0).all(axis=1).sum()
            # position of each bad id inside its bucket
            info = []
            for b in bad[:12]:
                o = plan[step][r][b] % G
                bucket = [i for i in range(len(plan[step][r])) if plan[step][r][i] % G == o]
                info.append((int(b), int(o), bucket.index(b), len(bucket)))
            print(f"step {step} rank {r}: {len(bad)} bad rows, {unwritten} unwritten; owners {np.bincount(ids % G, minlength=G).tolist()}; (pos, owner, k-in-bucket, bucket) {info}")
print("done")
