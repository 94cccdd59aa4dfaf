"""Worker for the multi-process tests.  Launched by tests/test_multirank.py through torch.multiprocessing / subprocess:

    python multirank_worker.py <mode> <rank> <world> <port> <px> <py> <pz> <eam> <n> [method] [async]

mode "host": CPU only.  Product host logic on `world` real processes over torch.distributed/gloo: decomposition, lattice,
             momenta (allreduce), the atom halo-exchange driver with a numpy pack/unpack -- compared with the oracle's
             virtual ranks, bit for bit.
mode "gpu" : every rank drives the HIP path on cuda:0 (ranks share the one GPU of the test box); halo messages are device
             buffers moved by the host-staged gloo transport.  Energies and per-atom forces are compared with the oracle.
"""
import ctypes
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

SINGLE = os.environ.get("COMD_PRECISION", "double") == "single"
REC = struct.Struct("<ii6f" if SINGLE else "<ii6d")      # the reference's AtomMsg (haloExchange.h:32-38), real_t fields
import json  # noqa: E402
TOL = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_values.json")))["tolerances_single" if SINGLE else "tolerances"]
# after a trajectory of tens of steps round-off differences have grown: two decades on the one-evaluation bounds (both precisions)
TRAJ_F = 100 * TOL["force_rel_to_max"]
TRAJ_R = 1e-4 if SINGLE else 1e-10


def host_mode(pkg, orc, transport, rank, world, grid, eam, n):
    args = ["-x", n, "-y", n, "-z", n, "-i", grid[0], "-j", grid[1], "-k", grid[2], "-r", 0.2] + (["-e"] if eam else [])
    sim = pkg.Simulation(args, host_only=True)
    o = orc.Oracle(n, grid, eam=eam, delta=0.2, cap=sim.max_atoms)
    lib = sim.lib
    nbr, coord = sim.neighbor_ranks()
    # 1. initial state of my brick == the oracle's virtual rank (lattice partition, allreduced COM removal + rescale)
    c = sim.cells()
    nl = sim.n_local_boxes
    mine = {}
    for b in range(nl):
        for i in range(c["nAtoms"][b]):
            mine[int(c["gid"][b, i])] = tuple(c[k][b, i] for k in ("px", "py", "pz"))
    # the oracle has already redistributed; compare momenta by gid over the atoms I own now
    allp = o.gather(orc.P)
    for g, p in mine.items():
        assert tuple(allp[g]) == p, (rank, g)

    # 2. product's exchange driver with a numpy pack/unpack over gloo
    cells_of = {f: sim.face_cells(0, f) for f in range(6)}
    shift = {}
    for f in range(6):
        v = (ctypes.c_double * 3)()
        lib.comdFaceShift(sim.ptr, f, v)
        shift[f] = tuple(v)
    h = lib.comdHostAtoms(sim.ptr).contents
    cap, ntot = sim.max_atoms, sim.n_total_boxes

    def view(ptr, dt):
        return np.ctypeslib.as_array(ptr, shape=(ntot * cap,))

    gid, spec = view(h.gid, np.int32), view(h.iSpecies, np.int32)
    arr = {k: view(getattr(h, k), np.float32 if SINGLE else np.float64) for k in ("rx", "ry", "rz", "px", "py", "pz")}
    nat = np.ctypeslib.as_array(h.nAtoms, shape=(ntot,))

    # local rebinning on the host mirror (what updateLinkCellsGpu does on the device): re-insert every local atom by coordinate
    recs = []
    for b in range(nl):
        for s in range(b * cap, b * cap + nat[b]):
            recs.append((int(gid[s]), int(spec[s]), arr["rx"][s], arr["ry"][s], arr["rz"][s], arr["px"][s], arr["py"][s], arr["pz"][s]))
    nat[:] = 0
    for g_, t_, x, y, z, px, py, pz in recs:
        lib.comdPutAtomInBox(sim.ptr, g_, t_, (ctypes.c_double * 3)(x, y, z), (ctypes.c_double * 3)(px, py, pz))

    def load(parms, data, face, buf):
        out = bytearray()
        sx, sy, sz = shift[face]
        for b in cells_of[face]:
            for s in range(b * cap, b * cap + nat[b]):
                out += REC.pack(int(gid[s]), int(spec[s]), arr["rx"][s] + sx, arr["ry"][s] + sy, arr["rz"][s] + sz,
                                arr["px"][s], arr["py"][s], arr["pz"][s])
        ctypes.memmove(buf, bytes(out), len(out))
        return len(out)

    def unload(parms, data, face, nbytes, buf):
        raw = ctypes.string_at(buf, nbytes)
        for k in range(nbytes // REC.size):
            g_, t_, x, y, z, px, py, pz = REC.unpack_from(raw, k * REC.size)
            lib.comdPutAtomInBox(sim.ptr, g_, t_, (ctypes.c_double * 3)(x, y, z), (ctypes.c_double * 3)(px, py, pz))

    keep = (pkg.LOAD_FN(load), pkg.UNLOAD_FN(unload))
    lib.comdHaloExchangeHost(sim.ptr, *keep)

    oc = o.rank_cells(rank)
    assert np.array_equal(nat, oc["nAtoms"]), (rank, np.nonzero(nat != oc["nAtoms"])[0][:10])
    for b in range(ntot):
        k = nat[b]
        order = np.argsort(gid[b * cap:b * cap + k])
        assert np.array_equal(gid[b * cap:b * cap + k][order], oc["gid"][b, :k]), (rank, b)
        for name in ("rx", "ry", "rz", "px", "py", "pz"):
            assert np.array_equal(arr[name][b * cap:b * cap + k][order], oc[name][b, :k]), (rank, b, name)
    sim.close()
    print(f"rank {rank}: host-mode OK ({sum(nat[:nl])} local atoms, {sum(nat[nl:])} halo atoms)")


def gpu_mode(pkg, orc, dist, rank, world, grid, eam, n, method, use_async, transport=None):
    import torch
    pkg.setup_gpu(0, rank)
    hilbert = method.endswith("+H")                          # "+H": Hilbert numbering of the link cells on every rank
    if hilbert:
        method = method[:-2]
    pairlist = method == "cta_cell_pairlist"
    if pairlist:
        method = "cta_cell"
    args = ["-x", n, "-y", n, "-z", n, "-i", grid[0], "-j", grid[1], "-k", grid[2], "-r", 0.1, "-m", method, "-a", use_async] + (["-e"] if eam else [])
    if hilbert:
        args += ["-H"]
    if pairlist:
        args += ["-L", "-S", 0.03]
        method = "cta_cell_nl"                               # below: run long, expect rebuilds
    if method.endswith("_nl") and not eam and not pairlist:
        args += ["-S", 0.03]                                 # LJ: the default skin (1.16 A) outlasts the test; 0.35 A does not
    sim = pkg.Simulation(args)
    steps = 40 if method.endswith("_nl") else 12          # list mode: long enough for several list builds and for atoms to change owner

    def gather_all(which):
        a = torch.from_numpy(sim.gather(which))
        dist.all_reduce(a)
        return a.numpy()

    f0 = gather_all(2)
    e0 = sim.energy()
    sim.step(steps)
    r1, f1, e1 = gather_all(0), gather_all(2), sim.energy()
    sim.sum_atoms()
    assert sim.energy()[2] == 4 * n ** 3
    if rank == 0:
        o = orc.Oracle(n, grid, eam=eam, delta=0.1)
        fo, eo = o.gather(orc.F), o.energy()
        err = np.abs(f0 - fo).max(axis=1)
        if err.max() >= TOL["force_rel_to_max"] * np.abs(fo).max():      # say where before failing: the worst atoms and where they sit
            worst = np.argsort(err)[-8:][::-1]
            r0 = o.gather(orc.R)
            print("force mismatch at step 0: %d atoms off by more than the tolerance; worst:" % int((err >= TOL["force_rel_to_max"] * np.abs(fo).max()).sum()))
            for g in worst:
                print("  gid %6d  |df| %.3e  r = (%.3f, %.3f, %.3f)  f = %s  oracle %s" % (g, err[g], *r0[g], f0[g], fo[g]))
        assert np.abs(f0 - fo).max() < TOL["force_rel_to_max"] * np.abs(fo).max()
        assert abs(e0[0] - eo[0]) / e0[2] < TOL["energy_per_atom_step0"] and abs(e0[1] - eo[1]) / e0[2] < 10 * TOL["kinetic_per_atom"]
        o.step(steps)
        fo, eo = o.gather(orc.F), o.energy()
        ro = o.gather(orc.R)
        ext = n * 3.615
        d = r1 - ro
        d -= np.rint(d / ext) * ext                       # an atom may sit on either side of a periodic face
        assert np.abs(d).max() < TRAJ_R
        assert np.abs(f1 - fo).max() < TRAJ_F * np.abs(fo).max()
        assert abs((e1[0] + e1[1]) - (eo[0] + eo[1])) / e1[2] < TOL["energy_per_atom_trace"]
        if method.endswith("_nl"):
            assert 1 < sim.nl_builds < steps, sim.nl_builds
        print(f"gpu-mode OK: {world} ranks {grid}, {'EAM' if eam else 'LJ'} {n}^3 {method} async={use_async}: E/atom {(e1[0]+e1[1])/e1[2]:.12f}, "
              f"{transport.n_sized if transport else 0} sized exchanges")
    sim.close()


def rccl_loopback_mode(pkg, orc, eam, n, method, use_async):
    """One rank whose halo messages, reductions and broadcasts all go through the RCCL transport (ncclSend/ncclRecv to itself):
    the only way to run comm_rccl.hip on a one-GPU box.  COMD_LOOPBACK_TRANSPORT=1 is set by the launcher."""
    pkg.setup_gpu(0, 0)
    t = pkg.rccl_transport(0, 1, pkg.rccl_unique_id())
    pkg.init_parallel(0, 1, t)
    assert pkg.lib_host().loopbackParallel() == 1
    args = ["-x", n, "-y", n, "-z", n, "-r", 0.1, "-m", method, "-a", use_async] + (["-e"] if eam else [])
    steps = 12
    if method.endswith("_nl") and not eam:
        args += ["-S", 0.03]                                 # LJ lists: a skin short enough for rebuilds (and size re-handshakes) inside the run
        steps = 40
    sim = pkg.Simulation(args)
    o = orc.Oracle(n, eam=eam, delta=0.1)
    fo = o.gather(orc.F)
    assert np.abs(sim.gather(2) - fo).max() < TOL["force_rel_to_max"] * np.abs(fo).max()
    sim.step(steps)
    o.step(steps)
    e1, eo, fo = sim.energy(), o.energy(), o.gather(orc.F)
    assert np.abs(sim.gather(2) - fo).max() < TRAJ_F * np.abs(fo).max()
    assert abs((e1[0] + e1[1]) - (eo[0] + eo[1])) / e1[2] < TOL["energy_per_atom_trace"]
    sim.sum_atoms()
    assert sim.energy()[2] == 4 * n ** 3
    sim.close()
    pkg.lib_hip().comdCommFinalize()
    print(f"rccl-loopback OK: {'EAM' if eam else 'LJ'} {n}^3 {method} async={use_async}: E/atom {(e1[0]+e1[1])/e1[2]:.12f}")


def main():
    mode, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    if mode == "rccl_hot":
        # a hot, strongly displaced lattice on the RCCL loopback: atoms cross cell faces every step, so the halo message counts move.
        # No checker here: the caller looks at how the run ends (tests/test_multirank.py, the outgrown-bound test).
        pkg = ge.load_package()
        pkg.setup_gpu(0, 0)
        pkg.init_parallel(0, 1, pkg.rccl_transport(0, 1, pkg.rccl_unique_id()))
        n = int(sys.argv[9])
        sim = pkg.Simulation(["-x", n, "-y", n, "-z", n, "-r", 0.4, "-T", 3000, "-m", sys.argv[10]] + (["-e"] if int(sys.argv[8]) else []))
        for _ in range(8):
            sim.step(10)
        sim.sum_atoms()
        assert sim.energy()[2] == 4 * n ** 3
        sim.close()
        print("rccl-hot run finished")
        return
    if mode == "rccl":
        pkg, orc = ge.load_package(), ge.load_oracle()
        rccl_loopback_mode(pkg, orc, int(sys.argv[8]), int(sys.argv[9]), sys.argv[10], int(sys.argv[11]))
        return
    grid = tuple(int(v) for v in sys.argv[5:8])
    eam, n = int(sys.argv[8]), int(sys.argv[9])
    method = sys.argv[10] if len(sys.argv) > 10 else "thread_atom"
    use_async = int(sys.argv[11]) if len(sys.argv) > 11 else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, orc = ge.load_package(), ge.load_oracle()
    transport = pkg.GlooTransport(dist)
    pkg.init_parallel(rank, world, transport.struct)
    if mode == "host":
        host_mode(pkg, orc, transport, rank, world, grid, eam, n)
    else:
        gpu_mode(pkg, orc, dist, rank, world, grid, eam, n, method, use_async, transport)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
