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


@pytest.mark.gpu
def test_message_that_outgrows_its_agreed_size_stops_the_run():
    """The sized protocol gives a message last step's count + 12.5 % + 64 atoms.  A hot (3000 K), strongly displaced EAM lattice moves atoms
    across cell faces every step: with the default slack 80 steps run through; with the slack set to nothing (COMD_HALO_SLACK=0,0) the first
    message that grows must be refused by the pack kernel, raise the device status flag, and the run must stop at the next status read with a
    message that names the rule -- never carry on with a truncated halo."""
    def run(extra_env):
        env = dict(os.environ, COMD_LOOPBACK_TRANSPORT="1", OMP_NUM_THREADS="2", **extra_env)
        proc = subprocess.run([sys.executable, os.path.join(HERE, "multirank_worker.py"), "rccl_hot", "0", "1", str(_free_port()), "1", "1", "1", "1", "10", "cta_cell", "0"],
                              capture_output=True, text=True, env=env, timeout=600)
        return proc.returncode, proc.stdout + proc.stderr
    rc, out = run({})
    assert rc == 0 and "rccl-hot run finished" in out, out[-2000:]
    rc, out = run({"COMD_HALO_SLACK": "0,0"})
    assert rc != 0 and "rccl-hot run finished" not in out, out[-2000:]
    assert "halo message overflowed its buffer, or grew by more than" in out, out[-2000:]


@pytest.mark.gpu
def test_bench_reports_what_rccl_saw_and_maps_one_copy_of_it():
    """bench.py forms the communicator the way comd-hip does (comdCommInitFromEnv; no torch in the rank process since round 3), so exactly ONE librccl and
    ONE libamdhip64 -- the copies under /opt/rocm that libcomd_hip.so was linked against -- may be mapped; the communicator must report the launched
    rank count, and the line must carry these facts so that a SCALE record shows what carried the halo messages."""
    import json
    env = dict(os.environ, COMD_LOOPBACK_TRANSPORT="1", MASTER_PORT=str(_free_port()))
    proc = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--nx", "20", "--steps", "3", "--warmup", "1",
                           "--no-variants", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert cfg["transport"] == "rccl-loopback" and cfg["rccl_ranks"] == 1 and cfg["rank_devices"] == [0]
    assert len(cfg["librccl"]) == 1 and len(cfg["libamdhip64"]) == 1, (cfg["librccl"], cfg["libamdhip64"])
    assert all(os.path.realpath(p).startswith("/opt/rocm") for p in cfg["librccl"] + cfg["libamdhip64"]) and cfg["one_rocm_runtime_on_every_rank"]
    assert "measured" not in line
    # [round 4] the self-check of the sized halo protocol: the same steps again with the size handshake, bit for bit (CoMD.c:413-440 beside it)
    assert line["sized_matches_handshake"] is True
    assert line["handshake_run"]["dE_pot_eV"] == 0.0 and line["handshake_run"]["dE_kin_eV"] == 0.0 and line["handshake_run"]["d_atoms"] == 0
    assert abs(line["eFinal_over_eInitial"] - 1.0) < 1e-4


@pytest.mark.gpu
def test_bench_self_check_catches_a_run_that_differs(tmp_path):
    """The comparison is live: when the second run is made to differ (COMD_BENCH_SELFTEST_PERTURB=1 gives it one time step more), bench.py says
    sized_matches_handshake: false and exits non-zero."""
    import json
    env = dict(os.environ, COMD_LOOPBACK_TRANSPORT="1", MASTER_PORT=str(_free_port()), COMD_BENCH_SELFTEST_PERTURB="1")
    proc = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--nx", "20", "--steps", "3", "--warmup", "1",
                           "--no-variants", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 4, (proc.returncode, proc.stderr[-2000:])
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["sized_matches_handshake"] is False


@pytest.mark.gpu
@pytest.mark.parametrize("pot,ranks", [("lj", 2), ("eam", 4)])
def test_bench_through_the_launcher_with_several_ranks(pot, ranks):
    """The driver's command line for N > 1 -- python -m torch.distributed.run --nproc-per-node N bench.py --gpus N -- rehearsed on ONE device: the ranks share the GPU
    (COMD_FORCE_DEVICE=0), RCCL refuses a communicator with one device twice, and --allow-host-staged sends the halo messages over gloo instead (the line says
    "measured": false).  Everything else is the N > 1 path of the day the ranks have a GPU each: decomposition 2x1x1 / 2x2x1 (one / two message axes, the rest mirrored),
    overlap mode, the sized halo protocol checked against the handshake run, one JSON line from rank 0, exit code 0 on every rank."""
    import json
    env = dict(os.environ, COMD_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", str(ranks), "--pot", pot, "--nx", "16", "--steps", "4", "--warmup", "2", "--allow-host-staged"]
    proc = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["config"]["decomposition"] == {2: "2x1x1", 4: "2x2x1"}[ranks] and line["config"]["halo_overlap"] is True
    assert line["measured"] is False and "gloo-host-staged" in line["config"]["transport"]
    assert line["sized_matches_handshake"] is True and line["handshake_run"]["d_atoms"] == 0
    assert abs(line["eFinal_over_eInitial"] - 1.0) < 1e-4


def test_bench_refuses_to_run_many_ranks_without_rccl():
    """Two ranks on a machine without a GPU (no device, hence no RCCL communicator): bench.py must exit non-zero on every rank instead of
    printing a host-staged number -- a SCALE record produced that way would look measured.  Runs here, on the CPU."""
    # asked in a child: importing torch here would map its bundled HIP runtime next to the one libcomd_hip.so (loaded by other test files of
    # this process) links, and the two tear each other down at exit
    gpu = subprocess.run([sys.executable, "-c", "import sys, torch; sys.exit(0 if torch.cuda.is_available() else 1)"], capture_output=True, timeout=300)
    if gpu.returncode == 0:
        pytest.skip("a GPU is visible: the refusal path needs a machine where RCCL cannot form")
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--nx", "10", "--steps", "2", "--warmup", "1"],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode != 0 for p in procs), [o[1][-500:] for o in outs]
    assert all('"value"' not in o[0] for o in outs)
