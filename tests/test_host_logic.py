"""Host-side logic of the product (no GPU): lattice, momenta, link-cell numbering, halo cell lists, CLI, C-ABI exports.
Everything is compared bit-for-bit with the oracle, which is itself pinned to the reference (test_oracle_golden.py)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def single_rank(pkg):
    pkg.init_parallel(0, 1, None)


def _by_gid(cells, n_local):
    out = {}
    for b in range(n_local):
        for i in range(cells["nAtoms"][b]):
            out[int(cells["gid"][b, i])] = tuple(cells[k][b, i] for k in ("rx", "ry", "rz", "px", "py", "pz"))
    return out


@pytest.mark.parametrize("eam,n", [(0, 10), (1, 6), (1, 9)])
def test_initial_state_matches_oracle_bit_for_bit(pkg, orc, eam, n):
    """FCC sites, gids, cell assignment, Maxwell-Boltzmann momenta (lcg61/gasdev streams, COM removal, rescale)."""
    with pkg.Simulation(["-x", n, "-y", n, "-z", n] + (["-e"] if eam else []), host_only=True) as s:
        o = orc.Oracle(n, eam=eam, cap=s.max_atoms)
        assert s.grid == o.rank_grid(0)[0]
        c, oc = s.cells(), o.rank_cells(0)
        nl = s.n_local_boxes
        assert np.array_equal(c["nAtoms"][:nl], oc["nAtoms"][:nl])
        for b in range(nl):
            k = c["nAtoms"][b]
            assert np.array_equal(c["gid"][b, :k], oc["gid"][b, :k])
            for name in ("rx", "ry", "rz", "px", "py", "pz"):
                assert np.array_equal(c[name][b, :k], oc[name][b, :k]), name


def test_displaced_state_matches_oracle_by_gid(pkg, orc):
    with pkg.Simulation(["-x", 8, "-y", 8, "-z", 8, "-r", 0.3, "-e"], host_only=True) as s:
        o = orc.Oracle(8, eam=1, delta=0.3, cap=s.max_atoms)
        assert _by_gid(s.cells(), s.n_local_boxes) == _by_gid(o.rank_cells(0), s.n_local_boxes)


def test_cell_numbering_and_tie_rules(pkg, orc):
    """getBoxFromTuple for every tuple incl. halo, getBoxFromCoord incl. positions exactly on faces and just outside."""
    with pkg.Simulation(["-x", 9, "-y", 7, "-z", 8, "-e"], host_only=True) as s:
        o = orc.Oracle((9, 7, 8), eam=1, cap=s.max_atoms)
        gx, gy, gz = s.grid
        seen = set()
        for ix in range(-1, gx + 1):
            for iy in range(-1, gy + 1):
                for iz in range(-1, gz + 1):
                    b = s.box_from_tuple(ix, iy, iz)
                    assert b == o.L.oracle_box_from_tuple(o.ptr, 0, ix, iy, iz)
                    seen.add(b)
        assert seen == set(range(s.n_total_boxes)), "numbering is a bijection onto [0, nTotalBoxes)"
        lat = 3.615
        ext = np.array([9, 7, 8]) * lat
        rng = np.random.default_rng(7)
        box = ext / np.array(s.grid)
        pts = list(rng.uniform(-0.99, 1.0, (300, 3)) * 0 + (-0.99 * box + rng.uniform(0.0, 1.0, (300, 3)) * (ext + 1.98 * box)))
        eps = np.finfo(float).eps
        for a in range(3):
            for v in (0.0, ext[a], np.nextafter(ext[a], 0), np.nextafter(0.0, -1), ext[a] * (1 - eps), -1e-9, ext[a] + 1e-9):
                p = ext * 0.5
                p[a] = v
                pts.append(p.copy())
        for p in pts:
            arr = np.ascontiguousarray(p, dtype=np.float64)
            assert s.box_from_coord(p) == o.L.oracle_box_from_coord(o.ptr, 0, arr.ctypes.data), p


@pytest.mark.parametrize("eam,n", [(0, 14), (1, 7)])
def test_halo_cell_lists(pkg, orc, eam, n):
    with pkg.Simulation(["-x", n, "-y", n, "-z", n] + (["-e"] if eam else []), host_only=True) as s:
        o = orc.Oracle(n, eam=eam, cap=s.max_atoms)
        for face in range(6):
            for kind in range(3):
                assert np.array_equal(s.face_cells(kind, face), o.face_cells(0, kind, face))


def test_capacity_rule(pkg):
    """LJ capacity is a multiple of 64 (waves never straddle cells); EAM capacity is a multiple of 4 with >= 10 % + 8 head-room."""
    with pkg.Simulation(["-x", 20, "-y", 20, "-z", 20], host_only=True) as s:
        assert s.max_atoms % 64 == 0 and s.max_atoms >= s.cells()["nAtoms"].max() * 1.1
    with pkg.Simulation(["-x", 20, "-y", 20, "-z", 20, "-e"], host_only=True) as s:
        assert s.max_atoms % 4 == 0 and s.max_atoms >= s.cells()["nAtoms"].max() * 1.1 + 8 - 4
    with pkg.Simulation(["-x", 10, "-y", 10, "-z", 10, "--maxAtoms", 192], host_only=True) as s:
        assert s.max_atoms == 192


def test_abi_exports_every_declared_symbol(pkg):
    """libcomd_hip.so exports every function include/comd_hip.h declares (load only; no device calls)."""
    header = open(os.path.join(ROOT, "include", "comd_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", header, flags=re.M))
    names -= {"sendrecv", "sendrecv2", "sendrecv2sized", "allreduce", "bcast", "barrier"}
    assert {"ljForceGpu", "eamForce1Gpu", "eamForce2Gpu", "eamForce3Gpu", "advanceVelocityGpu", "advancePositionGpu",
            "computeEnergy", "updateLinkCellsGpu", "buildAtomListGpu", "sortAtomsGpu", "compactCellsGpu",
            "unloadAtomsBufferToGpu", "loadForceBufferFromGpu", "unloadForceBufferToGpu", "getAtomMsgSoAPtr",
            "AllocateGpu", "CopyDataToGpu", "GetDataFromGpu", "DestroyGpu", "SetBoundaryCells"} <= names
    lib = pkg.lib_hip()
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_means_loud_failure(pkg):
    """Without a GPU the product refuses to run: no CPU fallback."""
    if pkg.lib_hip().comdDeviceCount() > 0:
        pytest.skip("a device is visible")
    with pytest.raises(RuntimeError):
        pkg.setup_gpu(0, 0)
    exe = os.path.join(ROOT, "comd-cuda-async_amd", "csrc", "comd-hip")
    proc = subprocess.run([exe, "-x", "10", "-y", "10", "-z", "10"], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert proc.returncode != 0 and "no HIP device" in proc.stderr


def test_product_never_links_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may mention it."""
    pkg_dir = os.path.join(ROOT, "comd-cuda-async_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "comd_oracle" not in text and "oracle_binding" not in text, f
    out = subprocess.run(["ldd", os.path.join(pkg_dir, "csrc", "libcomd_host.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_lj_thread_atom_keeps_its_scalar_load_stream(tmp_path):
    """The LJ thread_atom kernel is fast because the neighbour positions arrive through s_load_dwordx16 (wave-uniform j).
    A store ahead of those loads makes hipcc fall back to per-lane global_load without a word (4.7 -> 6.2 ms on MI355X);
    pin the ISA: the full-wave path must keep its scalar loads."""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    src = os.path.join(ROOT, "comd-cuda-async_amd", "csrc", "hip", "comd_device.hip")
    out = tmp_path / "dev.s"
    proc = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-comment", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", str(out), src], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    text = out.read_text()
    kernels = re.findall(r"^(_Z20LJ_Force_thread_atomILb[01]ELb([01])EEv6LjArgsi11LjWaveLists):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 4
    for name, listed, body in kernels:
        assert body.count("s_load_dwordx16") >= 3, name             # the stencil walk (the only path when LISTED = 0, the fallback otherwise)
        if listed == "1":
            # candidate lists: eight byte offsets per s_load_dwordx8, then ONE s_load_dwordx8 per candidate record with the offset in an
            # SGPR (no 64-bit address arithmetic, no x4 + x2 split) -- two unrolled loops (own cell, other cells) of 8 candidates
            assert len(re.findall(r"s_load_dwordx8 s\[\d+:\d+\], s\[\d+:\d+\], s\d+", body)) >= 16, name


@pytest.mark.parametrize("extra,pot_name", [((), "Cu_u6.eam"), (("-t", "setfl", "-p", "Cu01.eam.alloy"), "Cu01.eam.alloy")])
def test_eam_tables_of_host_and_oracle_agree(pkg, orc, extra, pot_name):
    """funcfl (eam.c:802-872) and setfl (eam.c:680-757) readers: grid, padding and every sample of phi, rho and F, bit for bit."""
    import ctypes
    sim = pkg.Simulation(["-x", 6, "-y", 6, "-z", 6, "-e"] + list(extra), host_only=True)
    o = orc.Oracle(6, eam=1, pot_name=pot_name)
    L = orc.lib()
    for which in range(3):
        x0, inv, v = sim.eam_table(which)
        n, ox0, oinv = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        assert L.oracle_eam_table(o.ptr, which, ctypes.byref(n), ctypes.byref(ox0), ctypes.byref(oinv), None) == 0
        ov = np.empty(n.value + 3)
        L.oracle_eam_table(o.ptr, which, ctypes.byref(n), ctypes.byref(ox0), ctypes.byref(oinv), ov.ctypes.data_as(ctypes.c_void_p))
        assert (n.value + 3, x0, inv) == (len(v), ox0.value, oinv.value)
        assert np.array_equal(v, ov)
        assert v[0] == v[1] and v[-1] == v[-2] == v[-3]      # eam.c:510-513 padding
    sim.close()


@pytest.mark.parametrize("n,eam", [(8, 1), ((12, 9, 14), 0), ((7, 6, 9), 1)])
def test_hilbert_numbering_is_a_bijection_and_local(pkg, n, eam):
    """-H: local cells are renumbered along a Hilbert curve (any grid, the reference: power-of-two grids only, linkCells.c:151-178).
    tuple -> id -> tuple is the identity, ids are dense, consecutive ids are face neighbours wherever the enclosing power-of-two cube
    is full (always for a power-of-two grid), and the halo cells keep their numbers."""
    nx, ny, nz = (n, n, n) if isinstance(n, int) else n
    args = ["-x", nx, "-y", ny, "-z", nz] + (["-e"] if eam else [])
    plain = pkg.Simulation(args, host_only=True)
    hil = pkg.Simulation(args + ["-H"], host_only=True)
    gx, gy, gz = hil.grid
    ids = np.array([[[hil.box_from_tuple(ix, iy, iz) for iz in range(gz)] for iy in range(gy)] for ix in range(gx)])
    assert sorted(ids.ravel()) == list(range(gx * gy * gz))
    where = {int(ids[ix, iy, iz]): (ix, iy, iz) for ix in range(gx) for iy in range(gy) for iz in range(gz)}
    steps = [sum(abs(a - b) for a, b in zip(where[i], where[i + 1])) for i in range(gx * gy * gz - 1)]
    if all(v & (v - 1) == 0 for v in (gx, gy, gz)) and gx == gy == gz:
        assert max(steps) == 1
    assert np.mean(np.array(steps) == 1) > 0.6          # a Hilbert walk with the cells outside the grid skipped
    for t in [(-1, 0, 0), (gx, 1, 1), (0, -1, gz - 1), (1, gy, 0), (0, 0, -1), (gx, gy, gz), (-1, -1, -1)]:
        assert hil.box_from_tuple(*t) == plain.box_from_tuple(*t)
    # every atom sits in the cell its coordinates name, under either numbering
    for sim in (plain, hil):
        c = sim.cells()
        for b in range(0, sim.n_local_boxes, 7):
            for i in range(c["nAtoms"][b]):
                assert sim.box_from_coord((c["rx"][b, i], c["ry"][b, i], c["rz"][b, i])) == b
    assert plain.cells()["nAtoms"][:plain.n_local_boxes].sum() == hil.cells()["nAtoms"][:hil.n_local_boxes].sum()
    plain.close(); hil.close()


def test_bench_never_divides_a_stale_pmc_record_by_a_live_time(tmp_path, monkeypatch):
    """bench.py's `traffic` and `valu_issue_frac` are stored rocprofv3 PMC counts divided by this run's kernel time.  A record is valid for ONE version of the
    kernel source: profiles/rNN_summarize.py stores a hash of the kernel's header with it, bench.profiled() refuses a record whose hash is not the tree's
    (and says so in the provenance string), and `profiles/r04_summarize.py --check` lists stale records."""
    import importlib
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    good = bench.kernel_source_hash("eam", "cta_cell")
    assert len(good) == 16 and good == bench.kernel_source_hash("eam", "thread_atom_nl") != bench.kernel_source_hash("lj", "thread_atom")
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda pot, method: good)
    (prof / "r04_traffic.json").write_text(json.dumps({"eam/cta_cell/80": {"fetch_KiB": 1.0, "write_KiB": 2.0, "kernel_source_sha16": good},
                                                       "lj/thread_atom/80": {"fetch_KiB": 1.0, "write_KiB": 2.0, "kernel_source_sha16": "0" * 16}}))
    (prof / "r03_traffic.json").write_text(json.dumps({"lj/thread_atom_nl/80": {"fetch_KiB": 1.0, "write_KiB": 2.0}}))
    rec, prov = bench.profiled("eam", "cta_cell", 80)
    assert rec and prov == "profiles/r04_traffic.json"
    rec, prov = bench.profiled("lj", "thread_atom", 80)
    assert rec is None and "STALE" in prov
    rec, prov = bench.profiled("lj", "thread_atom_nl", 80)                     # a record without a hash (rounds 1-3) is stale by definition
    assert rec is None and "STALE" in prov
    assert bench.profiled("eam", "thread_atom", 80) == (None, None)
