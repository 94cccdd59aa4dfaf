"""N > 1 ranks, one process per rank, torch.distributed/gloo rendezvous on 127.0.0.1.

CPU (not gpu): the product's host logic -- decomposition, allreduce-based initialisation, the three-phase halo exchange
driver, the transport vtable -- on 2 real processes, bit-exact against the oracle's virtual ranks.
GPU (-m gpu): the HIP path on 2 and 4 ranks that share the test box's single GPU (host-staged gloo transport; the
production transport is RCCL over xGMI, which needs one GPU per rank and is exercised by bench.py --gpus N).
"""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, grid, eam, n, extra=(), timeout=600, env_extra=None):
    world = grid[0] * grid[1] * grid[2]
    port = str(_free_port())
    env = dict(os.environ, OMP_NUM_THREADS="2", **(env_extra or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), mode, str(r), str(world), port,
                               *map(str, grid), str(eam), str(n), *map(str, extra)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
    return outs


@pytest.mark.parametrize("grid,eam,n", [((2, 1, 1), 1, 8), ((1, 1, 2), 1, 8), ((2, 1, 1), 0, 14)])
def test_host_logic_two_processes_gloo(grid, eam, n):
    outs = _launch("host", grid, eam, n)
    assert all("host-mode OK" in o for o in outs)


@pytest.mark.parametrize("eam,n", [(1, 12), (0, 22)])
def test_host_logic_2x2x2_eight_processes_gloo(eam, n):
    """BASELINE configs 4/5 run on a 2x2x2 rank grid: on every axis the minus and the plus neighbour are the SAME peer
    (decomposition.c:57-66), so both messages of an axis phase travel between one pair of ranks and must not be swapped.
    Eight real processes, bit-exact against the oracle's eight virtual ranks (EAM 12^3 and LJ 22^3)."""
    outs = _launch("host", (2, 2, 2), eam, n, timeout=900)
    assert all("host-mode OK" in o for o in outs)


@pytest.mark.gpu
@pytest.mark.parametrize("grid,eam,n,method,use_async", [((2, 1, 1), 1, 10, "cta_cell", 0), ((2, 2, 1), 1, 12, "thread_atom", 1),
                                                           ((1, 2, 1), 0, 14, "thread_atom", 0), ((2, 1, 2), 0, 20, "cta_cell", 1),
                                                           ((2, 1, 1), 1, 12, "thread_atom_nl", 0), ((1, 2, 1), 0, 22, "thread_atom_nl", 1),
                                                           ((1, 1, 2), 0, 22, "cta_cell_pairlist", 1),
                                                           # three ranks on an axis: minus and plus neighbours are different ranks
                                                           ((3, 1, 1), 1, 12, "cta_cell", 1), ((1, 1, 3), 0, 21, "thread_atom", 0), ((1, 3, 1), 1, 14, "thread_atom_nl", 1),
                                                           ((2, 1, 2), 1, 12, "cta_cell+H", 1), ((2, 1, 1), 0, 22, "thread_atom_nl+H", 0)])
def test_gpu_path_multi_rank_shared_device(grid, eam, n, method, use_async):
    outs = _launch("gpu", grid, eam, n, extra=(method, use_async))
    assert "gpu-mode OK" in outs[0]
    assert "sized exchanges" in outs[0] and " 0 sized exchanges" not in outs[0]      # the no-handshake protocol carried the run


@pytest.mark.gpu
@pytest.mark.parametrize("grid,eam,n,method,use_async", [((2, 1, 1), 1, 10, "cta_cell", 1), ((1, 2, 1), 0, 14, "thread_atom", 0)])
def test_gpu_path_multi_rank_exact_size_handshake(grid, eam, n, method, use_async):
    """COMD_HALO_HANDSHAKE=1: the exact-size handshake of round 1 (what the first exchange and the post-rebuild exchanges still use)."""
    outs = _launch("gpu", grid, eam, n, extra=(method, use_async), env_extra={"COMD_HALO_HANDSHAKE": "1"})
    assert "gpu-mode OK" in outs[0] and " 0 sized exchanges" in outs[0]


@pytest.mark.gpu
@pytest.mark.parametrize("eam,n,method,use_async", [(0, 14, "thread_atom", 0), (1, 10, "cta_cell", 1), (1, 10, "thread_atom_nl", 1), (0, 22, "thread_atom_nl", 0)])
def test_rccl_transport_loopback(eam, n, method, use_async):
    """comm_rccl.hip on real hardware: a one-rank RCCL communicator carries all six halo messages per exchange (size handshake +
    grouped ncclSend/ncclRecv to itself), the EAM dF/drho exchange and the energy / atom-count reductions."""
    outs = _launch("rccl", (1, 1, 1), eam, n, extra=(method, use_async), env_extra={"COMD_LOOPBACK_TRANSPORT": "1"})
    assert "rccl-loopback OK" in outs[0]
