"""COALA_GNN/numa.py on a fake sysfs tree: the GPU's PCI address from KFD's topology (no GPU call), its NUMA node, and the CPU mask a
rank binds itself to before it pins its shard of the cold tier.  (The real thing is measured on the GPU box: profiles/r03_cold_tier_kinds.txt.)"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def numa(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location("coala_numa_under_test", os.path.join(ROOT, "coala-gnn_amd", "COALA_GNN", "numa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                     # loaded by path, as bench.py does: must not import torch or the package
    sys_root = tmp_path / "sys"

    def put(rel, text):
        p = sys_root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    # two sockets of 4 CPUs; KFD: node 0 and 1 are CPUs, 2 and 3 GPUs at 0000:5a:00.0 (socket 0) and 0000:f1:00.0 (socket 1)
    put("devices/system/node/node0/cpulist", "0-1,4-5\n")
    put("devices/system/node/node1/cpulist", "2-3,6-7\n")
    put("devices/system/node/node0/distance", "10 32\n")
    put("devices/system/node/node1/distance", "32 10\n")
    for n, simd, loc in ((0, 0, 0), (1, 0, 0), (2, 1024, (0x5a << 8)), (3, 1024, (0xf1 << 8))):
        put(f"class/kfd/kfd/topology/nodes/{n}/properties", f"cpu_cores_count 4\nsimd_count {simd}\nlocation_id {loc}\ndomain 0\n")
    put("bus/pci/devices/0000:5a:00.0/numa_node", "0\n")
    put("bus/pci/devices/0000:f1:00.0/numa_node", "1\n")
    monkeypatch.setattr(mod, "_SYS", str(sys_root))
    state = {"mask": set(range(8))}
    monkeypatch.setattr(mod.os, "sched_getaffinity", lambda pid: set(state["mask"]))
    monkeypatch.setattr(mod.os, "sched_setaffinity", lambda pid, cpus: state.update(mask=set(cpus)))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "COALA_NUMA"):
        monkeypatch.delenv(v, raising=False)
    mod._state = state
    return mod


def test_pci_address_and_node_from_kfd_topology(numa, monkeypatch):
    assert numa.device_pci_address(0) == "0000:5a:00.0" and numa.device_pci_address(1) == "0000:f1:00.0"
    assert numa.device_pci_address(2) is None
    assert numa.describe(1)["gpu_numa_node"] == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")            # the launcher gave this rank the second GPU as device 0
    assert numa.device_pci_address(0) == "0000:f1:00.0"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")  # a UUID cannot be resolved without the runtime: no guess
    assert numa.device_pci_address(0) is None


def test_bind_modes(numa, monkeypatch):
    info = numa.bind_to_device_node(1)                         # auto: the GPU's own node
    assert info["applied"] and info["bound_node"] == 1 and numa._state["mask"] == {2, 3, 6, 7} and numa.current_placement() is info
    numa._state["mask"] = set(range(8))
    info = numa.bind_to_device_node(0, mode="far")             # the deliberate wrong placement of tools/numa_probe.sh
    assert info["bound_node"] == 1 and info["gpu_numa_node"] == 0 and numa._state["mask"] == {2, 3, 6, 7}
    numa._state["mask"] = set(range(8))
    monkeypatch.setenv("COALA_NUMA", "off")
    info = numa.bind_to_device_node(0)
    assert not info["applied"] and numa._state["mask"] == set(range(8))
    monkeypatch.setenv("COALA_NUMA", "0")                      # an explicit node
    assert numa.bind_to_device_node(1)["bound_node"] == 0 and numa._state["mask"] == {0, 1, 4, 5}
    numa._state["mask"] = {2, 3}                               # a cpuset without any CPU of the target node: leave it alone
    monkeypatch.setenv("COALA_NUMA", "auto")
    info = numa.bind_to_device_node(0)
    assert not info["applied"] and "cpuset" in info["why"] and numa._state["mask"] == {2, 3}


def test_hosts_it_cannot_read_are_left_alone(numa, tmp_path, monkeypatch):
    monkeypatch.setattr(numa, "_SYS", str(tmp_path / "nothing_here"))
    info = numa.bind_to_device_node(0)
    assert not info["applied"] and numa._state["mask"] == set(range(8))
