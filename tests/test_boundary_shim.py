"""The drop-in boundary, put through a compiler and a linker (no GPU needed).

1. include/comd_hip_shim.h + tests/shim/reference_callsites.c: the reference's device-library call expressions, as the reference
   writes them (ljForce.c:141, eam.c:203-259, timestep.c:137-160, :187, :224-275, haloExchange.c:1617-1633, :1686, :1872, :1885),
   compile with gcc -Wall -Wextra -Werror against the shim and link against libcomd_hip.so with no undefined symbol.
2. Every symbol of the reference's link surface (what its 19 host objects leave undefined without gpu_kernels.cu and
   comm.cc, + the gpu_utility.c entry points) is defined by libcomd_hip.so; the table below records what each one is.
"""
import ctypes
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "comd-cuda-async_amd", "csrc")

IMPLEMENTED, NOOP, ANSWER0, TRAP = "implemented", "no-op kept for the call site", "answers 0: 'not in use'", "defined, exits if called (libmp layer)"
SURFACE = {
    # device compute (gpu_kernels.h:13-24, 79-83)
    **{n: IMPLEMENTED for n in ("ljForceGpu", "eamForce1Gpu", "eamForce2Gpu", "eamForce3Gpu", "eamForce1GpuAsync", "eamForce2GpuAsync",
                                "eamForce3GpuAsync", "advanceVelocityGpu", "advancePositionGpu", "computeEnergy")},
    "updateNeighborsGpu": NOOP, "updateNeighborsGpuAsync": NOOP,        # the kernels gather from the cell table: no 27*MAXATOMS offset table to refresh
    # redistribute (gpu_kernels.h:84-86)
    "updateLinkCellsGpu": IMPLEMENTED, "sortAtomsGpu": IMPLEMENTED, "buildAtomListGpu": NOOP,      # no a_list / i_list / b_list
    # halo pack / unpack (gpu_kernels.h:28-72)
    **{n: IMPLEMENTED for n in ("compactCellsGpu", "getAtomMsgSoAPtr", "unloadAtomsBufferToGpu", "loadForceBufferFromGpu", "unloadForceBufferToGpu")},
    **{n: TRAP for n in ("loadAtomsBufferFromGpu_Async", "loadAtomsBufferFromGpu_Comm", "unloadAtomsBufferToGpu_Async", "unloadAtomsBufferToGpu_Comm",
                         "loadForceBufferFromGpu_Async", "loadForceBufferFromGpu_Comm", "unloadForceBufferToGpu_Async", "unloadForceBufferToGpu_Comm",
                         "unloadForceScanCells", "exchangeDataForceGpu_KI")},
    # neighbour lists / pairlists (gpu_kernels.h:25, 73-78, 87-92)
    **{n: IMPLEMENTED for n in ("emptyNeighborListGpu", "neighborListUpdateRequiredGpu", "pairlistUpdateRequiredGpu", "buildNeighborListGpu")},
    "initHashTableGpu": NOOP, "emptyHashTableGpu": NOOP,                # counters only: atoms keep their slots between list builds
    # comm.h:40-74
    **{n: ANSWER0 for n in ("comm_use_comm", "comm_use_async", "comm_use_gpu_comm", "comm_select_device", "comm_init")},
    "comm_finalize": NOOP,
    **{n: TRAP for n in ("comm_irecv", "comm_isend", "comm_isend_on_stream", "comm_send_ready", "comm_send_ready_on_stream", "comm_wait_ready_on_stream",
                         "comm_wait_all", "comm_wait_all_on_stream", "comm_flush", "comm_progress")},
    # gpu_utility.h:55-69
    **{n: IMPLEMENTED for n in ("SetupGpu", "AllocateGpu", "CopyDataToGpu", "SetBoundaryCells", "GetDataFromGpu", "DestroyGpu", "emptyHaloCellsGpu",
                                "initLinkCellsGpu")},
    # [round 4] the rest of gpu_utility.h:60-69 (staging of the cpu_nl path and the DEBUG block of timestep.c:309): all real
    **{n: IMPLEMENTED for n in ("GetLocalAtomsFromGpu", "updateGpuHalo", "updateNAtomsGpu", "updateNAtomsCpu", "cudaCopyDtH", "compactHaloCells")},
}


def test_link_surface_is_fully_defined(pkg):
    assert len(SURFACE) == 52 + 8 + 6                    # 49 undefined symbols (+3 with -DDO_MPI) + the gpu_utility.c entry points + the staging names of gpu_utility.h:60-69
    lib = pkg.lib_hip()
    missing = [n for n in SURFACE if not hasattr(lib, n)]
    assert not missing, missing
    # the dynamic symbol table agrees (hasattr resolves through dlsym; nm shows they are this library's own definitions)
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(CSRC, "libcomd_hip.so")], capture_output=True, text=True).stdout
    defined = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert set(SURFACE) <= defined, sorted(set(SURFACE) - defined)
    # "not in use" answers: the reference's host objects then take their plain send/receive path (haloExchange.c:726-730)
    for n in ("comm_use_comm", "comm_use_async", "comm_use_gpu_comm", "comm_use_gdrdma"):
        fn = getattr(lib, n)
        fn.restype = ctypes.c_int
        assert fn() == 0, n


def test_trap_symbols_say_what_they_are():
    """A comm-layer entry point reached by mistake stops the program with its own name (checked in a child process)."""
    code = ("import ctypes, sys; lib = ctypes.CDLL(%r); lib.comm_isend()" % os.path.join(CSRC, "libcomd_hip.so"))
    proc = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True)
    assert proc.returncode != 0 and "comm_isend" in proc.stderr and "not part of this build" in proc.stderr


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not on PATH")
def test_reference_call_expressions_compile_and_link(tmp_path):
    src = os.path.join(ROOT, "tests", "shim", "reference_callsites.c")
    obj, so = str(tmp_path / "callsites.o"), str(tmp_path / "libcallsites.so")
    cc = subprocess.run(["gcc", "-std=gnu11", "-O1", "-fPIC", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-4000:]
    ld = subprocess.run(["gcc", "-shared", "-o", so, obj, "-L" + CSRC, "-lcomd_hip", "-Wl,--no-undefined", "-Wl,-rpath," + CSRC], capture_output=True, text=True)
    assert ld.returncode == 0, ld.stderr[-4000:]
    # every shim_* scaffold made it into the object, and what it calls resolves into libcomd_hip.so
    syms = subprocess.run(["nm", obj], capture_output=True, text=True).stdout
    for fn in ("shim_ljForce", "shim_eamForceGpu", "shim_advance", "shim_kineticEnergyGpu", "shim_redistributeAtomsGpu", "shim_loadAtomsBuffer",
               "shim_unloadAtomsBuffer", "shim_loadForceBuffer", "shim_unloadForceBuffer"):
        assert f" T {fn}" in syms, fn
    undefined = {line.split()[-1] for line in syms.splitlines() if " U " in line}
    assert {"ljForceGpu", "eamForce1GpuAsync", "eamForce3Gpu", "advanceVelocityGpu", "advancePositionGpu", "computeEnergy", "updateLinkCellsGpu", "sortAtomsGpu",
            "compactCellsGpu", "unloadAtomsBufferToGpu", "loadForceBufferFromGpu", "unloadForceBufferToGpu", "pairlistUpdateRequiredGpu"} <= undefined


# ---------------------------------------------------------------------------------------------------------------------
# The reference's REAL host files through the shim (build container only: /root/reference does not exist on the GPU box, and nothing of it is copied).
REFERENCE = "/root/reference/src-mpi"
HOT_PATH_FILES = ("ljForce", "timestep", "eam", "haloExchange")                   # the four host files of the hot path
OTHER_HOST_FILES = ("neighborList", "hashTable", "linkCells", "initAtoms", "mytype")   # reference host code the four call into (its CPU list path); no device-library call in them


@pytest.mark.skipif(not os.path.isdir(REFERENCE) or shutil.which("gcc") is None, reason="needs the reference checkout (build container) and gcc")
def test_reference_host_files_compile_and_link_against_the_shim(tmp_path):
    """COMPILE AND LINK ONLY -- nothing built here is ever run, and it is no oracle.  The reference's ljForce.c, timestep.c, eam.c and haloExchange.c, as
    they lie (symlinked into a scratch directory so that `#include "gpu_types.h"` etc. resolve to five one-line headers that include
    include/comd_hip_shim.h in the place of cuda_runtime.h / nvToolsExt.h / gpu_kernels.h / gpu_utility.h / gpu_types.h), must compile without an error,
    every device-library symbol they leave undefined must be one libcomd_hip.so defines, and together with the reference's own host files they call into
    they must link against libcomd_hip.so + libcomd_host.so with --no-undefined (libcomd_host.so stands in for the reference's parallel.c /
    performanceTimers.c / decomposition.c, same names and signatures)."""
    src = tmp_path / "src"
    inc = tmp_path / "inc"
    src.mkdir(); inc.mkdir()
    redirected = ("gpu_kernels.h", "gpu_utility.h", "gpu_types.h")
    for f in os.listdir(REFERENCE):
        if f.endswith((".c", ".h")) and f not in redirected:
            os.symlink(os.path.join(REFERENCE, f), src / f)
    for h in redirected:
        (src / h).write_text('#include "comd_hip_shim.h"\n')
    for h in ("cuda_runtime.h", "nvToolsExt.h"):
        (inc / h).write_text('#include "comd_hip_shim.h"\n')
    mpi_inc = "/opt/conda/include"                                                 # comm.h:3 includes <mpi.h> for its prototypes' MPI_Comm (MPICH ships in the image)
    if not os.path.exists(os.path.join(mpi_inc, "mpi.h")):
        pytest.skip("no mpi.h in this image (comm.h needs MPI_Comm)")
    flags = ["gcc", "-std=gnu11", "-O1", "-fPIC", "-w", "-DCOMD_DOUBLE", "-DNDEBUG", "-DMAXATOMS=64", "-I" + str(inc), "-I" + os.path.join(ROOT, "include"), "-I" + mpi_inc]
    objs = []
    for f in HOT_PATH_FILES + OTHER_HOST_FILES:
        obj = str(tmp_path / (f + ".o"))
        cc = subprocess.run(flags + ["-c", str(src / (f + ".c")), "-o", obj], capture_output=True, text=True)
        assert cc.returncode == 0, f"{f}.c: " + cc.stderr[-3000:]
        objs.append(obj)
    hip_defined = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", os.path.join(CSRC, "libcomd_hip.so")], capture_output=True, text=True).stdout.splitlines() if l.strip()}
    wanted = set()
    for obj in objs[:len(HOT_PATH_FILES)]:
        wanted |= {l.split()[-1] for l in subprocess.run(["nm", "-u", obj], capture_output=True, text=True).stdout.splitlines() if l.strip()}
    device_calls = wanted & (set(SURFACE) | {n for n in hip_defined if n.startswith("comd")})
    assert device_calls <= hip_defined, sorted(device_calls - hip_defined)
    assert {"ljForceGpu", "eamForce1Gpu", "eamForce3GpuAsync", "advanceVelocityGpu", "advancePositionGpu", "computeEnergy", "updateLinkCellsGpu", "sortAtomsGpu",
            "compactCellsGpu", "unloadAtomsBufferToGpu", "loadForceBufferFromGpu", "unloadForceBufferToGpu", "emptyHaloCellsGpu", "pairlistUpdateRequiredGpu"} <= device_calls
    assert "neighborListForceRebuildGpu" in wanted and "neighborListForceRebuildGpu" in hip_defined          # (gpu_neighborList.h:55: host code in the reference, same signature here)
    ld = subprocess.run(["gcc", "-shared", "-o", str(tmp_path / "libreference_host.so")] + objs + ["-L" + CSRC, "-lcomd_host", "-lcomd_hip", "-Wl,--no-undefined", "-Wl,-rpath," + CSRC, "-lm"],
                        capture_output=True, text=True)
    assert ld.returncode == 0, ld.stderr[-4000:]


# ---------------------------------------------------------------------------------------------------------------------
# The adapter EXECUTED: whole time steps of a live simulation through the reference's call expressions (tests/shim/shim_driver.c).
class _ShimWorld(ctypes.Structure):
    _fields_ = [("nAtomCells", ctypes.c_int * 6), ("atomCells", ctypes.c_void_p * 6),
                ("nForceCells", ctypes.c_int * 6), ("forceSend", ctypes.c_void_p * 6), ("forceRecv", ctypes.c_void_p * 6),
                ("shift", (ctypes.c_double * 3) * 6),
                ("nLocalBoxes", ctypes.c_int), ("nTotalBoxes", ctypes.c_int), ("method", ctypes.c_int), ("eam", ctypes.c_int), ("gpuAsync", ctypes.c_int),
                ("cutoff", ctypes.c_double),
                ("_driver_private", ctypes.c_byte * 1024)]      # AtomExchangeParms, ForceExchangeParms, host buffers: filled by the driver


def _build_driver(tmp_path):
    so = str(tmp_path / "libshim_driver.so")
    cmd = ["gcc", "-std=gnu11", "-O1", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "shim"),
           os.path.join(ROOT, "tests", "shim", "shim_driver.c"), "-o", so, "-L" + CSRC, "-lcomd_hip", "-Wl,--no-undefined", "-Wl,-rpath," + CSRC]
    cc = subprocess.run(cmd, capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-4000:]
    return ctypes.CDLL(so)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not on PATH")
@pytest.mark.parametrize("eam,method,use_async,n", [(0, "thread_atom", 0, 10), (0, "cta_cell", 0, 10), (1, "cta_cell", 0, 8), (1, "cta_cell", 1, 10), (1, "thread_atom", 1, 10)])
def test_reference_side_adapter_drives_whole_time_steps(gpu, orc, tmp_path, eam, method, use_async, n):
    """Twelve velocity-Verlet steps of a hot, displaced lattice driven ENTIRELY through include/comd_hip_shim.h: shim_advanceVelocity/Position,
    shim_redistributeAtomsGpu, shim_ljForce / shim_eamForceGpu (both gpuAsync branches), shim_kineticEnergyGpu, and the atom and dF/drho halo
    exchanges through the four host-staged pack/unpack adapters (self-exchange through host buffers, parallel.c:112-117).  Energies, forces and
    positions against the oracle at the usual tolerances; atoms must migrate between cells and through the periodic faces on the way."""
    import json
    import numpy as np
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_values.json")))
    single = os.environ.get("COMD_PRECISION", "double") == "single"
    tol = G["tolerances_single" if single else "tolerances"]
    drv = _build_driver(tmp_path)
    methods = {"thread_atom": 0, "cta_cell": 4}                                   # comd_hip.h: THREAD_ATOM, CTA_CELL
    args = ["-x", n, "-y", n, "-z", n, "-r", 0.3, "-T", 4000, "-m", method, "-a", use_async] + (["-e"] if eam else [])
    with gpu.Simulation(args) as sim:
        o = orc.Oracle(n, eam=eam, delta=0.3, temperature=4000.0, cap=max(sim.max_atoms, 64))
        w = _ShimWorld()
        keep = []
        for f in range(6):
            a = np.ascontiguousarray(sim.face_cells(0, f), dtype=np.int32)
            s_ = np.ascontiguousarray(sim.face_cells(1, f), dtype=np.int32)
            r_ = np.ascontiguousarray(sim.face_cells(2, f), dtype=np.int32)
            keep += [a, s_, r_]
            w.nAtomCells[f], w.atomCells[f] = len(a), a.ctypes.data
            w.nForceCells[f], w.forceSend[f], w.forceRecv[f] = len(s_), s_.ctypes.data, r_.ctypes.data
            assert len(s_) == len(r_)
            v = (ctypes.c_double * 3)()
            sim.lib.comdFaceShift(sim.ptr, f, v)
            for c in range(3):
                w.shift[f][c] = v[c]
        w.nLocalBoxes, w.nTotalBoxes, w.method, w.eam, w.gpuAsync = sim.n_local_boxes, sim.n_total_boxes, methods[method], eam, use_async
        w.cutoff = o.L.oracle_cutoff(o.ptr)
        e = (ctypes.c_double * 2)()
        drv.shim_world_run.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ShimWorld), ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        oc0 = o.rank_cells(0)
        drv.shim_world_run(ctypes.c_void_p(sim.lib.comdSimGpu(sim.ptr)), ctypes.byref(w), 12, 1.0, e)
        o.step(12)
        ep, ek = o.energy()
        ng = sim.n_global
        assert abs(e[0] - ep) / ng < 10 * tol["energy_per_atom_trace"] and abs(e[1] - ek) / ng < 10 * tol["energy_per_atom_trace"]
        fo, ro = o.gather(orc.F), o.gather(orc.R)
        assert np.abs(sim.gather(2) - fo).max() <= 100 * tol["force_rel_to_max"] * np.abs(fo).max()      # (twelve steps of round-off growth, as in the trajectory tests)
        ext = n * 3.615
        d = sim.gather(0) - ro
        d -= np.rint(d / ext) * ext
        assert np.abs(d).max() < (1e-4 if single else 1e-10)
        # cell membership, local AND halo, is the oracle's -- in gid order, as the dF/drho exchange needs it -- and it is not the initial one: the
        # re-binning, the migration through the periodic faces and the exchange had work to do
        c, oc = sim.cells(), o.rank_cells(0)
        assert np.array_equal(c["nAtoms"], oc["nAtoms"])
        for b in range(sim.n_total_boxes):
            k = c["nAtoms"][b]
            assert np.array_equal(c["gid"][b, :k], oc["gid"][b, :k]), b
        moved = sum(1 for b in range(sim.n_local_boxes) if oc["nAtoms"][b] != oc0["nAtoms"][b] or not np.array_equal(oc["gid"][b, :oc["nAtoms"][b]], oc0["gid"][b, :oc0["nAtoms"][b]]))
        assert moved > 0
