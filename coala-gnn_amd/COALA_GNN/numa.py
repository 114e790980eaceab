"""NUMA placement of a rank: run its host threads -- and first-touch whatever host memory it registers for the GPU -- on the NUMA
node its GPU hangs off.

The cold tier is zero-copy pinned host memory read by the GPU over PCIe; a shard on the other socket is read across the
inter-socket fabric before it reaches the GPU's root complex.  What decides where the memory lands depends on its kind
(measured on a 2-socket MI355X host, profiles/r03_cold_tier_kinds.txt):

  hipHostMalloc (coala_pinned_alloc, PinnedFeatureTable)   the runtime allocates from the pool of the GPU's OWN node, whatever CPU
      asks: a rank deliberately bound to the far socket still got its 41 GB on the near node (K2 55.8 vs 55.7 GB/s) -- the
      owner-partitioned tier of bench.py is placed correctly by construction.
  POSIX shm + hipHostRegister (coala_shm_open, SharedUVAManager: the reference's kind, COALA_GNN_Modules/shared_UVA.cuh:60-100)
      pages land where they are first touched / pinned: created by a process on the far socket the segment sits there and K2 reads
      it at 52.4 instead of 53.5 GB/s.  The reference leaves this to whichever CPU its local rank 0 happens to run on.

So the binding matters for the shared segment (local rank 0 creates and pins it: bind THAT rank first) and for the host threads that
enqueue work and poll completion signals; it is harmless otherwise.  No libnuma on the image: sysfs + os.sched_setaffinity only.
The GPU's PCI address is found WITHOUT touching the GPU (the binding has to happen before the HIP runtime starts its helper
threads): KFD's topology lists the GPUs in the order HIP enumerates them; {ROCR,HIP,CUDA}_VISIBLE_DEVICES index lists are applied
on top (cross-checked against hipDeviceGetPCIBusId by tools/host_probe.py).

    COALA_NUMA = auto (default) | off | far | <node number>
       far: the node FURTHEST from the GPU's (tools/numa_probe.sh: the deliberate wrong placement, to measure what it costs)
"""
import glob
import os

__all__ = ["bind_to_device_node", "describe", "device_pci_address", "pci_bus_id_of_device", "current_placement"]

_applied = None   # what bind_to_device_node did in this process (bench.py reports it)
_SYS = "/sys"     # (tests point this at a fake tree)


def _read(path, default=None):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return default


def _parse_cpulist(s):
    cpus = set()
    for part in (s or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def _kfd_gpus():
    """PCI addresses of the GPUs in KFD node order (= HIP's enumeration order before the *_VISIBLE_DEVICES filters)."""
    out = []
    nodes = sorted(glob.glob(_SYS + "/class/kfd/kfd/topology/nodes/[0-9]*"), key=lambda p: int(os.path.basename(p)))
    for n in nodes:
        props = {}
        for line in (_read(os.path.join(n, "properties"), "") or "").splitlines():
            kv = line.split()
            if len(kv) == 2:
                props[kv[0]] = kv[1]
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue   # a CPU node
            loc = int(props["location_id"])
            dom = int(props.get("domain", "0"))
        except (KeyError, ValueError):
            continue
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    return out


def _visible(gpus):
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is None or v.strip() == "":
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None   # UUIDs: cannot be resolved without the runtime
        if any(i < 0 or i >= len(gpus) for i in idx):
            return None
        gpus = [gpus[i] for i in idx]
    return gpus


def device_pci_address(device_index):
    """'dddd:bb:dd.f' of HIP device `device_index`, from sysfs alone (None when it cannot be told)."""
    gpus = _visible(_kfd_gpus())
    if gpus and 0 <= int(device_index) < len(gpus):
        return gpus[int(device_index)]
    cards = sorted(glob.glob(_SYS + "/class/drm/card[0-9]*/device"), key=lambda p: int(os.path.basename(os.path.dirname(p))[4:]))
    cards = [c for c in cards if _read(os.path.join(c, "vendor")) == "0x1002"]
    if len(cards) == 1 and int(device_index) == 0:   # a one-GPU box: nothing to get wrong
        return os.path.basename(os.path.realpath(cards[0]))
    return None


def _node_of_pci(addr):
    if not addr:
        return None
    v = _read(f"{_SYS}/bus/pci/devices/{addr}/numa_node")
    try:
        v = int(v)
    except (TypeError, ValueError):
        return None
    return v if v >= 0 else None


def _nodes():
    res = {}
    for d in glob.glob(_SYS + "/devices/system/node/node[0-9]*"):
        res[int(os.path.basename(d)[4:])] = _parse_cpulist(_read(os.path.join(d, "cpulist"), ""))
    return res


def _far_node(near, nodes):
    dist = (_read(f"{_SYS}/devices/system/node/node{near}/distance", "") or "").split()
    best, best_d = None, -1
    for n in sorted(nodes):
        if n == near or not nodes[n]:
            continue
        d = int(dist[n]) if n < len(dist) and dist[n].isdigit() else 20
        if d > best_d:
            best, best_d = n, d
    return best


def describe(device_index):
    """What is known about the placement of device `device_index` (no side effects)."""
    addr = device_pci_address(device_index)
    nodes = _nodes()
    return {"device": int(device_index), "pci": addr, "gpu_numa_node": _node_of_pci(addr),
            "nodes": {str(n): len(c) for n, c in sorted(nodes.items())}, "allowed_cpus": len(os.sched_getaffinity(0)),
            "kfd_gpus": _kfd_gpus()}


def bind_to_device_node(device_index, mode=None):
    """Restrict this process (the calling thread and every thread it starts later) to the CPUs of the GPU's NUMA node.  Call it
    BEFORE the first GPU call and before the cold tier is allocated.  -> dict describing what was done; never raises for a host
    it cannot read (containers without sysfs, single-node machines): it then does nothing and says so."""
    global _applied
    mode = (mode if mode is not None else os.environ.get("COALA_NUMA", "auto")).strip().lower()
    info = {"mode": mode, "device": int(device_index), "pci": None, "gpu_numa_node": None, "bound_node": None, "cpus": None, "applied": False}
    if mode == "off":
        info["why"] = "COALA_NUMA=off"
        _applied = info
        return info
    nodes = _nodes()
    addr = device_pci_address(device_index)
    gpu_node = _node_of_pci(addr)
    info["pci"], info["gpu_numa_node"] = addr, gpu_node
    target = gpu_node
    if mode == "far":
        target = _far_node(gpu_node, nodes) if gpu_node is not None else None
    elif mode not in ("auto", ""):
        try:
            target = int(mode)
        except ValueError:
            info["why"] = f"COALA_NUMA={mode!r} not understood"
            _applied = info
            return info
    if target is None or target not in nodes:
        info["why"] = "the GPU's NUMA node is not exposed by sysfs (single-node host or restricted container)" if gpu_node is None else "no such node"
        _applied = info
        return info
    if len([n for n in nodes if nodes[n]]) < 2:
        info["why"] = "one NUMA node: nothing to choose"
        info["bound_node"] = target
        _applied = info
        return info
    allowed = os.sched_getaffinity(0)
    cpus = nodes[target] & allowed
    if not cpus:
        info["why"] = f"none of node {target}'s CPUs is in this job's cpuset"
        _applied = info
        return info
    os.sched_setaffinity(0, cpus)
    info.update(bound_node=target, cpus=len(cpus), applied=True)
    _applied = info
    return info


def current_placement():
    """What bind_to_device_node did in this process (None if it was never called)."""
    return _applied


def pci_bus_id_of_device(device_index):
    """The runtime's own answer (hipDeviceGetPCIBusId) -- initialises the GPU; used to cross-check device_pci_address."""
    import ctypes as C
    from COALA_GNN_Pybind import _capi
    buf = C.create_string_buffer(32)
    _capi.check(_capi.load().coala_device_pci_bus_id(int(device_index), buf, 32))
    return buf.value.decode().lower()


def node_of_memory(addr):
    """NUMA node of the page at virtual address `addr` of this process (move_pages(2) with no target: a query), or None."""
    import ctypes as C
    libc = C.CDLL(None, use_errno=True)
    page = C.c_void_p(int(addr) & ~4095)
    status = C.c_int(-1)
    # long move_pages(int pid, unsigned long count, void **pages, const int *nodes, int *status, int flags);  x86-64: 279
    rc = libc.syscall(279, 0, C.c_ulong(1), C.byref(page), None, C.byref(status), 0)
    if rc != 0 or status.value < 0:
        return None
    return int(status.value)
