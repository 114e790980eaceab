"""world_size > 1 on CPU with gloo: the host logic of the owner-partitioned path (topology, count exchange, all-to-all-v
splits, un-permute) and of the seed scheduler, with the oracle standing in for the device primitives (tests/_gloo_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, world, tmp_path):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   COALA_TEST_TMP=str(tmp_path), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert f"rank {r} ok" in out


@pytest.mark.parametrize("world", [2, 3])
def test_all_to_all_exchange_gloo(world, tmp_path):
    _launch("exchange", world, tmp_path)


def test_scheduler_one_domain_two_ranks(tmp_path):
    _launch("sched1", 2, tmp_path)


def test_scheduler_two_domains(tmp_path):
    _launch("sched2", 2, tmp_path)


def test_scheduler_two_domains_of_two_ranks(tmp_path):
    """2 domains x 2 ranks (world 4), the shape of bench.py's colour-affinity leg at N = 4: every domain master parses its domain's share, the
    broadcast to the domain's second rank runs in the distribution helper of both, every global batch is partitioned exactly in both modes."""
    _launch("sched2x2", 4, tmp_path)
