#!/usr/bin/env python3
"""What the host of a GPU box looks like to the cold tier: NUMA nodes, the GPU's node, CPU / memory limits of this job, /dev/shm,
and how long it takes to make host memory GPU-visible both ways the product offers:

  hipHostMalloc (coala_pinned_alloc, PinnedFeatureTable)            -- what bench.py's default cold tier uses
  POSIX shm + hipHostRegister (coala_shm_open, SharedUVAManager)     -- the reference's own kind (shared_UVA.cuh:60-100)

    python tools/host_probe.py [--gb 1,8,41]          -> one JSON object on stdout (profiles/r03_host_probe.json)
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "coala-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _read(path, default=None):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return default


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=str, default="1,8,41")
    args = ap.parse_args()
    res = {"what": "host topology and the cost of making host memory GPU-visible"}
    res["cpus_online"] = os.cpu_count()
    res["cpus_allowed"] = sorted(os.sched_getaffinity(0))
    res["numa_nodes"] = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        mem = _read(os.path.join(d, "meminfo"), "")
        total = [l for l in mem.splitlines() if "MemTotal" in l]
        free = [l for l in mem.splitlines() if "MemFree" in l]
        res["numa_nodes"][os.path.basename(d)] = {"cpulist": _read(os.path.join(d, "cpulist")),
                                                  "MemTotal_kB": int(total[0].split()[-2]) if total else None,
                                                  "MemFree_kB": int(free[0].split()[-2]) if free else None}
    res["meminfo"] = {k: v for k, v in (l.split(":", 1) for l in (_read("/proc/meminfo", "") or "").splitlines()[:6])}
    res["cgroup_memory_max"] = _read("/sys/fs/cgroup/memory.max") or _read("/sys/fs/cgroup/memory/memory.limit_in_bytes")
    res["cgroup_cpuset"] = _read("/sys/fs/cgroup/cpuset.cpus.effective")
    try:
        st = os.statvfs("/dev/shm")
        res["dev_shm_free_GB"] = round(st.f_bavail * st.f_frsize / 1e9, 1)
    except OSError as e:
        res["dev_shm_free_GB"] = repr(e)
    import resource
    res["rlimit_memlock"] = resource.getrlimit(resource.RLIMIT_MEMLOCK)
    res["drm_cards"] = {os.path.basename(os.path.dirname(os.path.dirname(p))): {"numa_node": _read(p), "pci": os.path.basename(os.path.realpath(os.path.dirname(p))),
                                                                                  "local_cpulist": _read(os.path.join(os.path.dirname(p), "local_cpulist"))}
                        for p in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/numa_node"))}
    from COALA_GNN import numa
    res["numa_module"] = numa.describe(0)

    import torch
    from COALA_GNN_Pybind import _capi
    L = _capi.load()
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    res["pci_bus_id_after_init"] = numa.pci_bus_id_of_device(0)
    timings = []
    for gb in [float(x) for x in args.gb.split(",") if x]:
        nbytes = int(gb * 1e9) // 4096 * 4096
        row = {"GB": gb}
        # hipHostMalloc
        hp, dp = C.c_void_p(), C.c_void_p()
        t0 = time.perf_counter()
        rc = L.coala_pinned_alloc(nbytes, 0, C.byref(hp), C.byref(dp))
        row["hipHostMalloc_s"] = round(time.perf_counter() - t0, 3) if rc == 0 else _capi.last_error()
        if rc == 0:
            t0 = time.perf_counter()
            L.coala_pinned_free(hp)
            row["hipHostFree_s"] = round(time.perf_counter() - t0, 3)
        # shm + hipHostRegister: untouched pages (registration faults them in), then a second mapping of the same, populated, object
        if isinstance(res["dev_shm_free_GB"], float) and res["dev_shm_free_GB"] * 1e9 < nbytes * 1.05:
            row["shm"] = "does not fit /dev/shm"
        else:
            name = f"/coala_probe_{os.getpid()}".encode()
            h = C.c_void_p()
            t0 = time.perf_counter()
            rc = L.coala_shm_open(name, nbytes, 1, 0, C.byref(h))
            row["shm_create_register_s"] = round(time.perf_counter() - t0, 3) if rc == 0 else _capi.last_error()
            if rc == 0:
                h2 = C.c_void_p()
                t0 = time.perf_counter()
                rc2 = L.coala_shm_open(name, nbytes, 0, 0, C.byref(h2))
                row["shm_second_mapping_register_s"] = round(time.perf_counter() - t0, 3) if rc2 == 0 else _capi.last_error()
                if rc2 == 0:
                    L.coala_shm_close(h2, 0)
                t0 = time.perf_counter()
                L.coala_shm_close(h, 1)
                row["shm_unregister_unlink_s"] = round(time.perf_counter() - t0, 3)
        timings.append(row)
        print(f"[host_probe] {row}", file=sys.stderr, flush=True)
    res["registration"] = timings
    print(json.dumps(res))


if __name__ == "__main__":
    main()
