"""GPU: the one-shot peer-write all-reduce (msweep_amd/csrc/peer_comm.hpp, MSWEEP_ALLREDUCE=peer) with thread-ranks on
ONE GPU: raw messages against numpy, the EC-sharded solve against the host-staged transport bit for bit, and the
bounded wait.  Each case is one child process (tests/peer_allreduce_worker.py): the variable is read when the
communicators are created, and the ranks' kernels need hardware queues of their own."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(mode, n_ranks, **env):
    e = dict(os.environ, GPU_MAX_HW_QUEUES="8", **env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "peer_allreduce_worker.py"), mode, str(n_ranks)],
                       capture_output=True, text=True, timeout=300, env=e)
    assert p.returncode == 0, p.stdout + p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_peer_messages_equal_numpy_sums(n_ranks):
    """24 messages of 1 ... 15 004 words (integers exact, doubles summed in rank order), the inbox outgrown twice."""
    r = _run("messages", n_ranks, MSWEEP_ALLREDUCE="peer")
    assert r["ok"] and not r["err"], r
    print(f"\npeer all-reduce, {n_ranks} thread-ranks on one GPU, 15 004 words: {r['ms_per_call_15004_words'] * 1e3:.1f} us "
          "per call (host barrier per launch included)")


def test_peer_sharded_solve_is_the_host_staged_one_bit_for_bit():
    peer = _run("solve", 3, MSWEEP_ALLREDUCE="peer")
    host = _run("solve", 3, MSWEEP_ALLREDUCE="rccl")
    assert not peer["err"] and not host["err"], (peer["err"], host["err"])
    for r in range(3):
        assert peer["ranks"][r]["iters"] == host["ranks"][0]["iters"]
        assert peer["ranks"][r]["bound"] == host["ranks"][0]["bound"]
        np.testing.assert_array_equal(peer["ranks"][r]["theta"], host["ranks"][0]["theta"])
        assert peer["ranks"][r]["again_same"]
        assert peer["ranks"][r]["collectives"] >= 2 * peer["ranks"][r]["iters"]
    print(f"\n3 thread-ranks: collective_ms per solve  peer {peer['ranks'][0]['collective_ms']:.2f}  "
          f"host-staged {host['ranks'][0]['collective_ms']:.2f}  ({peer['ranks'][0]['collectives']} collectives)")


def test_peer_wait_is_bounded():
    r = _run("timeout", 2, MSWEEP_ALLREDUCE="peer", MSWEEP_PEER_TEST_SKIP_RANK="1", MSWEEP_PEER_TIMEOUT_MS="300")
    assert not r["err"], r
    assert "did not deliver its message" in r["said"][0], r


def _run_processes(mode, n_ranks, **env):
    """n_ranks worker PROCESSES, one rank each (Comm.shm), all on cuda:0."""
    e = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    name = f"/msweep_test_{os.getpid()}_{mode.replace('-', '_')}_{env.get('MSWEEP_ALLREDUCE', 'host')}"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_allreduce_worker.py"), mode, str(n_ranks),
                               str(r), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
             for r in range(n_ranks)]
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=240)
            assert p.returncode == 0, so + se
            outs.append(json.loads(so.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


def test_peer_inboxes_over_hipipc_between_processes():
    """One rank per PROCESS (three processes on this GPU, meeting in a shared-memory segment -- RCCL refuses two ranks on
    one device): the inboxes are exchanged as hipIpc handles and written by the peers' kernels, as between the GPUs of
    a node.  Messages against numpy; the sharded solve against the host-staged transport of the same processes and
    against the thread-ranks, bit for bit."""
    n = 3
    for r in _run_processes("proc-messages", n, MSWEEP_ALLREDUCE="peer"):
        assert r["ok"], r
        print(f"\npeer all-reduce between {n} processes on one GPU (hipIpc inboxes), 15 004 words: "
              f"{r['ms_per_call_15004_words'] * 1e3:.1f} us per call, rank {r['rank']}")
    peer = _run_processes("proc-solve", n, MSWEEP_ALLREDUCE="peer")
    host = _run_processes("proc-solve", n, MSWEEP_ALLREDUCE="rccl")
    threads = _run("solve", n, MSWEEP_ALLREDUCE="rccl")
    for r in range(n):
        for other in (host[r], threads["ranks"][0]):
            assert peer[r]["iters"] == other["iters"] and peer[r]["bound"] == other["bound"]
            np.testing.assert_array_equal(peer[r]["theta"], other["theta"])
    print(f"\n{n} processes: collective_ms per solve  peer {peer[0]['collective_ms']:.2f}  host-staged {host[0]['collective_ms']:.2f}  "
          f"({peer[0]['collectives']} collectives)")


def test_unknown_transport_is_refused():
    r = subprocess.run([sys.executable, "-c", "from msweep_amd.core import Comm; Comm.local(2)"], capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, MSWEEP_ALLREDUCE="ring"), cwd=ROOT)
    assert r.returncode != 0 and "expected `rccl` or `peer`" in r.stderr
