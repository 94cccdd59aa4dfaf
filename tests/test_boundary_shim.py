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
}


def test_link_surface_is_fully_defined(pkg):
    assert len(SURFACE) == 52 + 8                        # 49 undefined symbols (+3 with -DDO_MPI) + the gpu_utility.c entry points
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
