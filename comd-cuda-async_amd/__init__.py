"""ctypes binding of the MI355X CoMD hot path (libcomd_host.so + libcomd_hip.so).

This is plumbing for tests and bench.py: every call goes through the C ABI declared in
include/comd_hip.h and comd-cuda-async_amd/csrc/host/comd_host.h.  There is no Python or CPU
fallback: if the HIP library is missing, loading fails with an ImportError that says how to build it.

The directory name contains '-', so import it with `__graft_entry__.load_package()`.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
REPO_ROOT = os.path.dirname(_HERE)
POT_DIR = os.path.join(REPO_ROOT, "pots")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)

# One build per precision (make PRECISION=single -> lib*_sp.so, real_t = float: the reference's DOUBLE_PRECISION = OFF, mytype.h:8-21).
# A process binds ONE of them: COMD_PRECISION=single selects the float build before the first library is loaded.
PRECISION = os.environ.get("COMD_PRECISION", "double")
if PRECISION not in ("double", "single"):
    raise ImportError(f"COMD_PRECISION={PRECISION!r}: expected 'double' or 'single'")
_SFX = "_sp" if PRECISION == "single" else ""
c_real = ctypes.c_float if PRECISION == "single" else ctypes.c_double
c_real_p = ctypes.POINTER(c_real)


class HostAtoms(ctypes.Structure):
    """include/comd_hip.h HostAtoms"""
    _fields_ = [("nAtoms", c_int_p), ("gid", c_int_p), ("iSpecies", c_int_p)] + \
               [(n, c_real_p) for n in ("rx", "ry", "rz", "px", "py", "pz", "fx", "fy", "fz", "e")]


SENDRECV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)
SENDRECV2_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                ctypes.POINTER(ctypes.c_int))
ALLREDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
BCAST_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
BARRIER_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
SENDRECV2SIZED_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)


LOAD_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
UNLOAD_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)


class CommTransport(ctypes.Structure):
    """include/comd_hip.h CommTransport"""
    _fields_ = [("ctx", ctypes.c_void_p), ("sendrecv", SENDRECV_FN), ("sendrecv2", SENDRECV2_FN), ("allreduce", ALLREDUCE_FN),
                ("bcast", BCAST_FN), ("barrier", BARRIER_FN), ("sendrecv2sized", SENDRECV2SIZED_FN)]


_libs = {}


def _load(name):
    if name in _libs:
        return _libs[name]
    path = os.path.join(_CSRC, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C {_CSRC}{' PRECISION=single' if _SFX else ''}` (hipcc --offload-arch=gfx950). "
                          "There is no CPU fallback for the product path.")
    _libs[name] = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    return _libs[name]


def lib_hip():
    """libcomd_hip.so: HIP kernels + C-ABI launch wrappers."""
    lib = _load(f"libcomd_hip{_SFX}.so")
    if not getattr(lib, "_typed", False):
        lib.SetupGpu.argtypes = [ctypes.c_int] * 3
        lib.SetupGpu.restype = ctypes.c_int
        lib.comdDeviceCount.restype = ctypes.c_int
        lib.comdCommGetUniqueId.argtypes = [ctypes.c_char_p]
        lib.comdCommInitRank.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(CommTransport)]
        lib.comdForceTimingEnable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.comdForceTimingReset.argtypes = [ctypes.c_void_p]
        lib.comdForceTimingTotalMs.argtypes = [ctypes.c_void_p, c_int_p]
        lib.comdForceTimingTotalMs.restype = ctypes.c_double
        lib.comdForceTimingAuxMs.argtypes = [ctypes.c_void_p, c_int_p]
        lib.comdForceTimingAuxMs.restype = ctypes.c_double
        lib.comdDeviceMemInfo.argtypes = [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
        lib.comdCommInitFromEnv.argtypes = [ctypes.POINTER(CommTransport), c_int_p, c_int_p, c_int_p]
        lib.comdCommInitFromEnv.restype = ctypes.c_int
        lib.comdCommInfo.argtypes = [c_int_p, c_int_p, c_int_p]
        lib.comdCommInfo.restype = ctypes.c_int
        lib.comdEventCreate.restype = ctypes.c_void_p
        lib.comdEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.comdEventElapsedMs.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.comdEventElapsedMs.restype = ctypes.c_float
        lib.comdEventDestroy.argtypes = [ctypes.c_void_p]
        lib.comdStreamSynchronize.argtypes = [ctypes.c_void_p]
        lib.comdMemcpyDtoH.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        lib.comdMemcpyHtoD.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        lib._typed = True
    return lib


def lib_host():
    """libcomd_host.so: the C host (CLI, decomposition, link cells, potentials, halo exchange, time step)."""
    lib_hip()
    lib = _load(f"libcomd_host{_SFX}.so")
    if not getattr(lib, "_typed", False):
        vp = ctypes.c_void_p
        lib.comdCreate.restype = vp
        lib.comdCreate.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
        lib.comdCreateHostOnly.restype = vp
        lib.comdCreateHostOnly.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
        lib.comdMain.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
        lib.comdDestroy.argtypes = [vp]
        lib.comdSimGpu.argtypes = [vp]
        lib.comdSimGpu.restype = vp
        lib.comdGetEnergy.argtypes = [vp, c_double_p]
        lib.comdNumGlobal.argtypes = [vp]
        lib.comdNumLocalSlots.argtypes = [vp]
        lib.comdFetchAtoms.argtypes = [vp]
        lib.comdFetchAtoms.restype = ctypes.POINTER(HostAtoms)
        lib.comdHostAtoms.argtypes = [vp]
        lib.comdHostAtoms.restype = ctypes.POINTER(HostAtoms)
        lib.comdGridInfo.argtypes = [vp, c_int_p]
        lib.comdEamTable.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
        lib.comdEamTable.restype = ctypes.c_int
        lib.comdNeighborListBuilds.argtypes = [vp]
        lib.comdNeighborListBuilds.restype = ctypes.c_int
        lib.comdSimBoxFromTuple.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.comdSimBoxFromCoord.argtypes = [vp, c_double_p]
        lib.comdFaceCells.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.comdNeighborRanks.argtypes = [vp, c_int_p, c_int_p]
        lib.comdHaloExchangeHost.argtypes = [vp, LOAD_FN, UNLOAD_FN]
        lib.comdFaceShift.argtypes = [vp, ctypes.c_int, c_double_p]
        lib.comdPutAtomInBox.argtypes = [vp, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p]
        lib.comdGatherByGid.argtypes = [vp, ctypes.c_int, c_double_p]
        lib.comdScatterByGid.argtypes = [vp, ctypes.c_int, c_double_p]
        lib.timestep.argtypes = [vp, ctypes.c_int, c_real]
        lib.timestep.restype = ctypes.c_double
        for fn in ("redistributeAtoms", "computeForce", "kineticEnergyGpu", "sumAtoms"):
            getattr(lib, fn).argtypes = [vp]
        lib.initParallel.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(CommTransport)]
        lib.getMyRank.restype = ctypes.c_int
        lib.getNRanks.restype = ctypes.c_int
        lib._typed = True
    return lib


def _argv(args):
    argv = [b"comd"] + [str(a).encode() for a in args]
    arr = (ctypes.c_char_p * (len(argv) + 1))(*argv, None)
    return len(argv), arr


def setup_gpu(device=0, rank=0, verbose=False):
    """SetupGpu (gpu_utility.c:32-71).  Raises if no HIP device is visible."""
    hip = lib_hip()
    if hip.comdDeviceCount() < 1:
        raise RuntimeError("no HIP device visible: the product path has no CPU fallback")
    return hip.SetupGpu(device, rank, 1 if verbose else 0)


def init_parallel(rank=0, n_ranks=1, transport=None):
    lib_host().initParallel(rank, n_ranks, ctypes.byref(transport) if transport is not None else None)


def rccl_transport(rank, n_ranks, unique_id):
    """Join the RCCL communicator (after setup_gpu) and return the transport to hand to init_parallel."""
    t = CommTransport()
    rc = lib_hip().comdCommInitRank(unique_id, rank, n_ranks, ctypes.byref(t))
    if rc != 0:
        raise RuntimeError("comdCommInitRank failed")
    return t


def rccl_transport_from_env():
    """Join the RCCL communicator the way comd-hip does (comdCommInitFromEnv: RANK / WORLD_SIZE / LOCAL_RANK, rendezvous through a file keyed by
    the launcher's pid).  No torch, no second HIP runtime in the process.  Returns the transport, or None when the communicator cannot be formed."""
    t = CommTransport()
    r, n, l = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    if lib_hip().comdCommInitFromEnv(ctypes.byref(t), ctypes.byref(r), ctypes.byref(n), ctypes.byref(l)) != 0:
        return None
    return t


def device_mem_info():
    f, t = ctypes.c_long(0), ctypes.c_long(0)
    lib_hip().comdDeviceMemInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value, t.value


def rccl_comm_info():
    """(ranks, my rank, device) as the RCCL communicator itself reports them (ncclCommCount / UserRank / CuDevice)."""
    n, r, d = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    if lib_hip().comdCommInfo(ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)) != 0:
        raise RuntimeError("no RCCL communicator")
    return n.value, r.value, d.value


def mapped_libraries(pattern):
    """Paths of the shared objects mapped into this process whose name contains `pattern` (from /proc/self/maps)."""
    out = []
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.split()[-1]
            if pattern in os.path.basename(path) and path not in out:
                out.append(path)
    return out


def rccl_unique_id():
    buf = ctypes.create_string_buffer(128)
    if lib_hip().comdCommGetUniqueId(buf) != 0:
        raise RuntimeError("comdCommGetUniqueId failed")
    return buf.raw


class Simulation:
    """One rank's SimFlat (CoMDTypes.h:75-135) driven through the C ABI.

    args are the reference's CoMD command-line flags, e.g. ["-x", 20, "-y", 20, "-z", 20, "-m", "thread_atom"].
    """

    def __init__(self, args, host_only=False):
        import numpy as np
        self._np = np
        self.lib = lib_host()
        args = list(args)
        if "-d" not in args and "--potDir" not in args:
            args += ["-d", POT_DIR]
        if "--quiet" not in args:
            args += ["--quiet"]
        argc, argv = _argv(args)
        self.host_only = host_only
        self.ptr = (self.lib.comdCreateHostOnly if host_only else self.lib.comdCreate)(argc, argv)
        if not self.ptr:
            raise RuntimeError("simulation creation failed")
        info = (ctypes.c_int * 6)()
        self.lib.comdGridInfo(self.ptr, info)
        self.grid = tuple(info[0:3])
        self.n_local_boxes, self.n_total_boxes, self.max_atoms = info[3], info[4], info[5]

    # --- time stepping (timestep.c) ---
    def step(self, n, dt=1.0):
        return self.lib.timestep(self.ptr, n, dt)

    def redistribute(self):
        self.lib.redistributeAtoms(self.ptr)

    def compute_force(self):
        self.lib.computeForce(self.ptr)

    def eam_table(self, which):
        """(x0, invDx, padded samples) of the EAM table 0 phi / 1 rho / 2 F as the host hands it to AllocateGpu."""
        import numpy as np
        x0, inv = ctypes.c_double(), ctypes.c_double()
        n = self.lib.comdEamTable(self.ptr, which, ctypes.byref(x0), ctypes.byref(inv), None)
        v = np.empty(n + 3)
        self.lib.comdEamTable(self.ptr, which, ctypes.byref(x0), ctypes.byref(inv), v.ctypes.data_as(ctypes.c_void_p))
        return x0.value, inv.value, v

    @property
    def nl_builds(self):
        """Verlet-list builds so far (thread_atom_nl)."""
        return self.lib.comdNeighborListBuilds(self.ptr)

    def kinetic_energy(self):
        self.lib.kineticEnergyGpu(self.ptr)

    def sum_atoms(self):
        self.lib.sumAtoms(self.ptr)

    # --- device-side timing of the force launches (comd_hip.h comdForceTiming*, per simulation) ---
    def force_timing(self, on):
        hip = lib_hip()
        gpu = ctypes.c_void_p(self.lib.comdSimGpu(self.ptr))
        hip.comdForceTimingEnable(gpu, 1 if on else 0)
        if on:
            hip.comdForceTimingReset(gpu)

    def force_timing_total(self):
        """(milliseconds, launches) of the force kernels since force_timing(True)."""
        n = ctypes.c_int(0)
        ms = lib_hip().comdForceTimingTotalMs(ctypes.c_void_p(self.lib.comdSimGpu(self.ptr)), ctypes.byref(n))
        return ms, int(n.value)

    def force_timing_aux(self):
        """(milliseconds, launches) of what the force evaluations launched beside the force kernels (list builds, cell marks)."""
        n = ctypes.c_int(0)
        ms = lib_hip().comdForceTimingAuxMs(ctypes.c_void_p(self.lib.comdSimGpu(self.ptr)), ctypes.byref(n))
        return ms, int(n.value)

    def force_path_info(self):
        """What the force wrappers decided: LJ candidate lists in use, records of the EAM brick image, Verlet-list format, bricks in the thread-per-atom fall-back."""
        hip = lib_hip()
        gpu = ctypes.c_void_p(self.lib.comdSimGpu(self.ptr))
        a, b = (ctypes.c_int * 4)(), (ctypes.c_int * 3)()
        hip.comdForcePathInfo(gpu, a)
        hip.comdEamBrickStats(gpu, b)
        return {"lj_candidate_lists_active": bool(a[0]), "eam_brick_image_records": a[1], "neighbor_list_format": a[2], "eam_brick_cells": a[3] & 255, "eam_brick_lists_made": a[3] >> 8,
                "eam_bricks_in_thread_per_atom_fallback": b[0], "eam_bricks_per_launch": b[1]}

    # --- results ---
    def energy(self):
        """(ePotential, eKinetic, nGlobal) totals in eV."""
        out = (ctypes.c_double * 3)()
        self.lib.comdGetEnergy(self.ptr, out)
        return out[0], out[1], int(out[2])

    @property
    def n_global(self):
        return self.lib.comdNumGlobal(self.ptr)

    def gather(self, which, out=None):
        """Per-atom array of this rank's local atoms keyed by gid: 0 r, 1 p, 2 f -> (n,3); 3 e, 4 rhobar, 5 dfEmbed -> (n,)."""
        np = self._np
        n = self.n_global
        if out is None:
            out = np.zeros((n, 3) if which < 3 else (n,), dtype=np.float64)
        self.lib.comdGatherByGid(self.ptr, which, out.ctypes.data_as(c_double_p))
        return out

    def scatter(self, which, arr):
        np = self._np
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self.lib.comdScatterByGid(self.ptr, which, arr.ctypes.data_as(c_double_p))

    def cells(self):
        """Slot arrays (device state copied to the host mirror; host mirror only when host_only):
        dict of nAtoms [nTotalBoxes], gid [nTotalBoxes, maxAtoms] and r/p/f components."""
        np = self._np
        h = (self.lib.comdHostAtoms if self.host_only else self.lib.comdFetchAtoms)(self.ptr).contents
        nb, cap = self.n_total_boxes, self.max_atoms
        out = {"nAtoms": np.ctypeslib.as_array(h.nAtoms, shape=(nb,)).copy(),
               "gid": np.ctypeslib.as_array(h.gid, shape=(nb, cap)).copy()}
        for k in ("rx", "ry", "rz", "px", "py", "pz", "fx", "fy", "fz", "e"):
            out[k] = np.ctypeslib.as_array(getattr(h, k), shape=(nb, cap)).copy()
        return out

    def box_from_tuple(self, ix, iy, iz):
        return self.lib.comdSimBoxFromTuple(self.ptr, ix, iy, iz)

    def box_from_coord(self, r):
        v = (ctypes.c_double * 3)(*r)
        return self.lib.comdSimBoxFromCoord(self.ptr, v)

    def face_cells(self, kind, face):
        np = self._np
        n = self.lib.comdFaceCells(self.ptr, kind, face, None)
        a = np.zeros(n, dtype=np.int32)
        self.lib.comdFaceCells(self.ptr, kind, face, a.ctypes.data)
        return a

    def neighbor_ranks(self):
        nbr = (ctypes.c_int * 6)()
        coord = (ctypes.c_int * 3)()
        self.lib.comdNeighborRanks(self.ptr, nbr, coord)
        return list(nbr), list(coord)

    def close(self):
        if self.ptr:
            self.lib.comdDestroy(self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def run_main(args):
    """The reference's main() (CoMD.c:86-187): full stdout report + YAML file."""
    argc, argv = _argv(list(args))
    return lib_host().comdMain(argc, argv)


class GlooTransport:
    """CommTransport backed by torch.distributed (gloo): host staging, any number of processes per GPU.

    The production transport is RCCL over xGMI (rccl_transport).  This one exists so that the multi-rank host logic
    can be exercised where RCCL cannot run: on CPU-only machines (host buffers, the CPU test-suite) and with several
    ranks sharing ONE GPU (device buffers are staged through pinned-less host copies).  It plays the role the
    reference's plain-MPI path does (haloExchange.c:1493-1522 + parallel.c:100-118).
    """

    def __init__(self, dist):
        import numpy as np
        import torch
        self.dist, self.torch, self.np = dist, torch, np
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.hip = None
        self.n_sized = 0          # exchanges that went through the pre-agreed-size path (tests look at it)
        self._keep = (SENDRECV_FN(self._guard(self._sendrecv)), SENDRECV2_FN(self._guard(self._sendrecv2)), ALLREDUCE_FN(self._guard(self._allreduce)),
                      BCAST_FN(self._guard(self._bcast)), BARRIER_FN(self._guard(self._barrier)), SENDRECV2SIZED_FN(self._guard(self._sendrecv2sized)))
        self.struct = CommTransport(None, *self._keep)

    @staticmethod
    def _guard(fn):
        """ctypes swallows exceptions raised inside callbacks (the C caller would carry on with garbage): make them fatal."""
        def wrapped(*args):
            try:
                return fn(*args)
            except BaseException:
                import traceback
                traceback.print_exc()
                sys.stderr.flush()
                os._exit(70)
        return wrapped

    def _exchange(self, send_bytes, dest, source, recv_cap):
        torch, dist = self.torch, self.dist
        n_send = torch.tensor([len(send_bytes)], dtype=torch.int64)
        n_recv = torch.zeros(1, dtype=torch.int64)
        if dest == self.rank and source == self.rank:
            return bytes(send_bytes)
        reqs = [dist.isend(n_send, dest, tag=1), dist.irecv(n_recv, source, tag=1)]
        for r in reqs:
            r.wait()
        n = int(n_recv[0])
        if n > recv_cap:
            raise RuntimeError(f"incoming message of {n} bytes exceeds the {recv_cap}-byte buffer")
        out = torch.empty(max(n, 1), dtype=torch.uint8)
        payload = torch.frombuffer(bytearray(send_bytes), dtype=torch.uint8) if len(send_bytes) else torch.zeros(1, dtype=torch.uint8)
        reqs = []
        if len(send_bytes):
            reqs.append(dist.isend(payload, dest, tag=2))
        if n:
            reqs.append(dist.irecv(out, source, tag=2))
        for r in reqs:
            r.wait()
        return out.numpy()[:n].tobytes()

    def _sendrecv(self, ctx, send_buf, send_len, dest, recv_buf, recv_cap, source, device, stream):
        if device:
            hip = self.hip or lib_hip()
            hip.comdStreamSynchronize(ctypes.c_void_p(stream))
            staging = ctypes.create_string_buffer(max(send_len, 1))
            if send_len:
                hip.comdMemcpyDtoH(staging, ctypes.c_void_p(send_buf), ctypes.c_long(send_len))
            data = self._exchange(staging.raw[:send_len], dest, source, recv_cap)
            if data:
                up = ctypes.create_string_buffer(data, len(data))
                hip.comdMemcpyHtoD(ctypes.c_void_p(recv_buf), up, ctypes.c_long(len(data)))
        else:
            data = self._exchange(ctypes.string_at(send_buf, send_len), dest, source, recv_cap)
            ctypes.memmove(recv_buf, data, len(data))
        return len(data)

    def _sendrecv2(self, ctx, send_m, n_m, dst_m, recv_p, send_p, n_p, dst_p, recv_m, recv_cap, device, stream, n_recv):
        # same pairing as the RCCL transport: my minus-face message lands in dst_m's "from plus" buffer and vice versa
        n_recv[0] = self._sendrecv(ctx, send_m, n_m, dst_m, recv_p, recv_cap, dst_p, device, stream)
        n_recv[1] = self._sendrecv(ctx, send_p, n_p, dst_p, recv_m, recv_cap, dst_m, device, stream)

    def _sendrecv2sized(self, ctx, send_m, n_m, dst_m, recv_p, n_rp, send_p, n_p, dst_p, recv_m, n_rm, device, stream):
        """Both ends agree on every size: payloads only, no size messages (the RCCL transport's rcclSendrecv2Sized)."""
        torch, dist = self.torch, self.dist
        if not device:
            raise RuntimeError("sized exchange is a device-buffer path")
        hip = self.hip or lib_hip()
        hip.comdStreamSynchronize(ctypes.c_void_p(stream))
        self.n_sized += 1

        def down(ptr, n):
            buf = ctypes.create_string_buffer(max(n, 1))
            if n:
                hip.comdMemcpyDtoH(buf, ctypes.c_void_p(ptr), ctypes.c_long(n))
            return torch.frombuffer(bytearray(buf.raw[:max(n, 1)]), dtype=torch.uint8)

        out_m, out_p = down(send_m, n_m), down(send_p, n_p)
        in_p, in_m = torch.empty(max(n_rp, 1), dtype=torch.uint8), torch.empty(max(n_rm, 1), dtype=torch.uint8)
        if dst_m == self.rank and dst_p == self.rank:          # loopback
            assert n_m == n_rp and n_p == n_rm
            in_p[:n_rp] = out_m[:n_m]
            in_m[:n_rm] = out_p[:n_p]
        else:
            # posting order as in RCCL: minus-face message first -- with dst_m == dst_p the peer's first send meets my first receive
            reqs = [dist.isend(out_m, dst_m, tag=3), dist.isend(out_p, dst_p, tag=4),
                    dist.irecv(in_p, dst_p, tag=3), dist.irecv(in_m, dst_m, tag=4)]
            for r in reqs:
                r.wait()
        for ptr, t, n in ((recv_p, in_p, n_rp), (recv_m, in_m, n_rm)):
            if n:
                raw = t.numpy()[:n].tobytes()
                hip.comdMemcpyHtoD(ctypes.c_void_p(ptr), ctypes.create_string_buffer(raw, n), ctypes.c_long(n))

    def _allreduce(self, ctx, buf, count, dtype):
        np, torch, dist = self.np, self.torch, self.dist
        ctype = ctypes.c_double if dtype == 1 else ctypes.c_float if dtype == 3 else ctypes.c_int
        arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctype)), shape=(count,))
        t = torch.from_numpy(arr.copy())
        if dtype in (1, 3) and self.world > 2:
            # floating-point sums in RANK ORDER (gather, then add 0, 1, 2, ...): the result does not depend on gloo's reduction tree, so
            # runs are reproducible and comparable bit for bit with a serial sum over the ranks (what the tests' checker does)
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t)
            acc = parts[0].clone()
            for p in parts[1:]:
                acc += p
            arr[:] = acc.numpy()
            return
        dist.all_reduce(t, op=dist.ReduceOp.MAX if dtype == 2 else dist.ReduceOp.SUM)
        arr[:] = t.numpy()

    def _bcast(self, ctx, buf, length, root):
        np, torch = self.np, self.torch
        arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_uint8)), shape=(length,))
        t = torch.from_numpy(arr.copy())
        self.dist.broadcast(t, src=root)
        arr[:] = t.numpy()

    def _barrier(self, ctx):
        self.dist.barrier()
