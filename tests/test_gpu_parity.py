"""Parity of the HIP path (through the C ABI) with the oracle on identical inputs -- run with `-m gpu` on an MI355X.

Tolerances (fp64): per-atom force |df| <= 1e-11 * max|f|, per-atom energy 1e-11 eV, energy per atom along a trace
2e-12 eV/atom against the reference's recorded values (tests/golden/reference_values.json, "tolerances").
Index work (cell membership, gid order, halo images, positions after the periodic shift) must be bit-exact.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))
SINGLE = os.environ.get("COMD_PRECISION", "double") == "single"      # the float build: its own checker build, its own (stated) tolerances
TOL = G["tolerances_single" if SINGLE else "tolerances"]
S = G["survey_recorded"]
METHODS = ["thread_atom", "cta_cell"]


def _args(n, eam=0, delta=0.0, method="thread_atom", extra=()):
    nx, ny, nz = (n, n, n) if isinstance(n, int) else n
    return ["-x", nx, "-y", ny, "-z", nz, "-r", delta, "-m", method] + (["-e"] if eam else []) + list(extra)


def _per_atom(sim):
    ep, ek, n = sim.energy()
    return (ep + ek) / n, ep / n, ek / n


# ---------------------------------------------------------------- forces and energies of one evaluation
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("eam,n,delta", [(0, 10, 0.0), (0, 10, 0.1), (0, (10, 14, 11), 0.2), (1, 8, 0.0), (1, 8, 0.1), (1, (7, 9, 12), 0.3)])
def test_forces_match_oracle(gpu, orc, method, eam, n, delta):
    with gpu.Simulation(_args(n, eam, delta, method)) as sim:
        o = orc.Oracle(n, eam=eam, delta=delta, cap=max(sim.max_atoms, 64))
        f, e = sim.gather(2), sim.gather(3)
        fo, eo = o.gather(orc.F), o.gather(orc.U)
        assert np.abs(f - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0)
        assert np.abs(e - eo).max() <= TOL["per_atom_energy_abs"]
        ep, ek, ng = sim.energy()
        op, ok = o.energy()
        assert abs(ep - op) / ng < TOL["energy_per_atom_step0"] and abs(ek - ok) / ng < TOL["kinetic_per_atom"]
        if eam:
            assert np.abs(sim.gather(4) - o.gather(orc.RHOBAR)).max() < TOL["eam_density_abs"]
            assert np.abs(sim.gather(5) - o.gather(orc.DFEMBED)).max() < TOL["eam_dfembed_abs"]


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("cap", [44, 100, 128])
def test_eam_any_cell_capacity(gpu, orc, method, cap):
    """--maxAtoms that is neither a power of two nor below a wave (round 1's cta_cell kernel addressed slots as lane & (cap - 1) and
    hung or dropped slots for such values): forces, energies, rhobar and F' against the oracle, then a short trace."""
    with gpu.Simulation(_args(8, 1, 0.1, method, ["--maxAtoms", cap])) as sim:
        assert sim.max_atoms == cap
        o = orc.Oracle(8, eam=1, delta=0.1, cap=max(cap, 64))
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]
        assert np.abs(sim.gather(4) - o.gather(orc.RHOBAR)).max() < TOL["eam_density_abs"]
        assert np.abs(sim.gather(5) - o.gather(orc.DFEMBED)).max() < TOL["eam_dfembed_abs"]
        sim.step(10)
        o.step(10)
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs((ep + ek) - (op + ok)) / ng < TOL["energy_per_atom_trace"]


@pytest.mark.parametrize("eam,n,method", [(1, 10, "cta_cell"), (1, 10, "thread_atom"), (0, 14, "cta_cell"), (0, 14, "thread_atom")])
def test_overlap_mode_small_interior(gpu, orc, eam, n, method):
    """-a 1 on one rank: interior cells (27 for EAM 10^3, 8 for LJ 14^3) on one stream, the two boundary rings on the other.
    Small interior lists are launches of fewer than 8 workgroups -- the XCD-dealt cell walk must still cover every cell."""
    with gpu.Simulation(_args(n, eam, 0.1, method, ["-a", 1])) as sim:
        o = orc.Oracle(n, eam=eam, delta=0.1, cap=max(sim.max_atoms, 64))
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0)
        sim.step(10)
        o.step(10)
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs((ep + ek) - (op + ok)) / ng < TOL["energy_per_atom_trace"]


def test_lj_cells_larger_than_the_launch_estimate(gpu, orc, monkeypatch):
    """thread_atom sizes its workgroups from the host's last occupancy reading; a cell that outgrew it must still be complete.
    Forcing one wave per cell makes every cell take the extra-chunk path (chunks 1 and 2 through the generic per-lane code)."""
    monkeypatch.setenv("COMD_LJ_WAVES", "1")
    with gpu.Simulation(_args(10, 0, 0.1, "thread_atom")) as sim:
        o = orc.Oracle(10, eam=0, delta=0.1)
        sim.step(3)
        o.step(3)
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= 1e-10 * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]


@pytest.mark.parametrize("env", [{}, {"COMD_LJ_CTA_SLABS": "1"}])
def test_lj_cta_cell_both_forms(gpu, orc, monkeypatch, env):
    """LJ cta_cell: the default form (every wave stages its own box-pruned candidates in a private LDS region, flushing it whenever it is
    full -- a box this small has cells of ~200 atoms, so regions of 440 records are flushed several times per cell) and the slab form
    that -L builds on (COMD_LJ_CTA_SLABS=1), against the oracle after a few steps."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with gpu.Simulation(_args((12, 10, 11), 0, 0.15, "cta_cell")) as sim:
        o = orc.Oracle((12, 10, 11), eam=0, delta=0.15, cap=max(sim.max_atoms, 64))
        sim.step(2)
        o.step(2)
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]


@pytest.mark.parametrize("env", [{"COMD_EAM_IMAGE": "128"}, {"COMD_EAM_IMAGE": "1000"}, {"COMD_EAM_BRICK": "2,2"}, {"COMD_EAM_BRICK": "3,5"},
                                 {"COMD_EAM_CTA": "cell", "COMD_EAM_STENCIL": "128"}, {"COMD_EAM_CTA": "cell", "COMD_EAM_STENCIL": "368"}, {"COMD_EAM_CTA": "cell"}])
def test_eam_cells_whose_stencil_outgrows_the_lds_slice(gpu, orc, monkeypatch, env):
    """cta_cell stages the cells around a brick in an LDS image sized from the fullest block the first launch finds; a brick whose block holds more
    atoms than the image is walked thread-per-atom by the same workgroup, in pass 1 AND pass 3 (no rows are handed over for it).  COMD_EAM_IMAGE
    forces an image that every brick (128) or part of the bricks (1000; the blocks of this box hold 860-1170 atoms) outgrow; COMD_EAM_BRICK
    other brick shapes, one that does not divide the grid; COMD_EAM_CTA=cell round 2's kernel (a wave stages every cell's stencil for itself), with
    its own slice-overflow legs."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with gpu.Simulation(_args(12, 1, 0.1, "cta_cell")) as sim:
        o = orc.Oracle(12, eam=1, delta=0.1, cap=max(sim.max_atoms, 64))
        sim.step(3)
        o.step(3)
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]
        assert np.abs(sim.gather(4) - o.gather(orc.RHOBAR)).max() < TOL["eam_density_abs"]
        assert np.abs(sim.gather(5) - o.gather(orc.DFEMBED)).max() < TOL["eam_dfembed_abs"]


@pytest.mark.parametrize("env", [{"COMD_LJ_PRUNE": "0"}, {"COMD_LJ_LIST_CAP": "64"}, {"COMD_LJ_LIST_CAP": "2200"}, {"COMD_LJ_LIST_BUDGET_MB": "1"}, {}])
def test_lj_wave_candidate_lists_and_their_fallbacks(gpu, orc, monkeypatch, env):
    """thread_atom tests only the stencil atoms within the cutoff of each wave's bounding box (LJ_WaveCandidates).  The four legs: lists
    off (the plain 27-cell walk), rows too short for any wave (every wave falls back to the walk), rows that fit the tail waves' lists but
    not the full waves' (both paths inside one launch), and the default.  All four must give the oracle's forces."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with gpu.Simulation(_args((12, 10, 11), 0, 0.15, "thread_atom")) as sim:
        o = orc.Oracle((12, 10, 11), eam=0, delta=0.15, cap=max(sim.max_atoms, 64))
        sim.step(2)
        o.step(2)
        f, fo = sim.gather(2), o.gather(orc.F)
        assert np.abs(f - fo).max() <= TOL["force_rel_to_max"] * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]


@pytest.mark.parametrize("env", [{}, {"COMD_EAM_GROUPS": "0"}, {"COMD_EAM_BRICK": "2,3"}])
def test_eam_overlap_mode_takes_whole_bricks(gpu, orc, monkeypatch, env):
    """-a 1 launches every EAM pass once over the boundary cells and once over the interior cells.  cta_cell makes that split at brick
    granularity (every cell of a brick that holds a boundary cell goes with the boundary launch: no brick is staged twice per pass;
    comd_device.hip eamBrickGroupOf).  Forces, energies, densities and dF/drho after three steps must be the oracle's -- with the groups,
    with the lists taken cell by cell (COMD_EAM_GROUPS=0), and with a brick shape that does not divide the grid."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with gpu.Simulation(_args((14, 12, 13), 1, 0.2, "cta_cell", extra=("-a", 1))) as sim:
        o = orc.Oracle((14, 12, 13), eam=1, delta=0.2, cap=max(sim.max_atoms, 64))
        sim.step(3)
        o.step(3)
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]
        assert np.abs(sim.gather(4) - o.gather(orc.RHOBAR)).max() < TOL["eam_density_abs"]
        assert np.abs(sim.gather(5) - o.gather(orc.DFEMBED)).max() < TOL["eam_dfembed_abs"]


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("env", [{}, {"COMD_EAM_ATOM_HANDOVER": "0"}, {"COMD_EAM_IMAGE": "128"}, {"COMD_EAM_IMAGE": "1500"}, {"COMD_EAM_ATOM_BRICK": "2,3"},
                                 {"COMD_EAM_ATOM_BRICK": "4,5"}, {"COMD_EAM_ATOM_ROWS": "16"}, {"COMD_EAM_ATOM_ROWS": "48"}, {"COMD_EAM_THREAD_ATOM": "cell"}, {"COMD_EAM_GROUPS": "0"}, {"COMD_EAM_ABLATE": "16"}])
def test_eam_thread_atom_on_the_brick_image(gpu, orc, monkeypatch, env, overlap):
    """-m thread_atom -e: a thread per atom inside a brick workgroup (eam_atom_brick_kernels.h).  Legs: the default; pass 3 testing again instead of reading
    the rows of pass 1; an image every brick (128 records) or part of the bricks (1500) outgrow -- those take the streaming form, in both passes; brick shapes
    that do not divide the grid / fill all four waves; rows shorter than the neighbour count of every atom (16: each walks its stencil a second time) or of
    some atoms (48); stencil runs too long for byte offsets (COMD_EAM_ABLATE=16 lowers the limit from 256 records to 64: every atom walks without a row, in both
    passes); round 2's kernel; the lists of -a 1 taken cell by cell (COMD_EAM_GROUPS=0: bricks that hold cells of both lists are staged under two
    selections, the hand-over must notice) instead of as whole bricks.  Each without and with -a 1 (every pass once over the boundary and once over the
    interior cells).  Forces, energies, densities, dF/drho against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with gpu.Simulation(_args((14, 12, 13), 1, 0.2, "thread_atom", extra=("-a", overlap))) as sim:
        o = orc.Oracle((14, 12, 13), eam=1, delta=0.2, cap=max(sim.max_atoms, 64))
        sim.step(3)
        o.step(3)
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]
        assert np.abs(sim.gather(4) - o.gather(orc.RHOBAR)).max() < TOL["eam_density_abs"]
        assert np.abs(sim.gather(5) - o.gather(orc.DFEMBED)).max() < TOL["eam_dfembed_abs"]


def test_eam_thread_atom_hand_over_changes_no_bit(gpu, monkeypatch):
    """The rows pass 1 leaves for pass 3 hold what pass 3's own test would find, in the same order: forces and energies with and without the hand-over are the
    same to the last bit -- on one launch per pass and with -a 1 (bricks staged under two selections; the two modes order the atoms of a cell differently and
    are not compared with each other)."""
    got = []
    monkeypatch.setenv("COMD_EAM_GROUPS", "0")      # (-a 1 with the lists taken cell by cell: the case in which pass 3 finds cells staged under another selection)
    for handover, overlap in (("1", 0), ("0", 0), ("1", 1), ("0", 1)):
        monkeypatch.setenv("COMD_EAM_ATOM_HANDOVER", handover)
        with gpu.Simulation(_args((12, 14, 11), 1, 0.15, "thread_atom", extra=("-a", overlap))) as sim:
            sim.step(4)
            got.append((sim.gather(2).copy(), sim.gather(3).copy()))
    for k in (0, 2):
        assert np.array_equal(got[k][0], got[k + 1][0]) and np.array_equal(got[k][1], got[k + 1][1])


def test_eam_pass_3_over_another_partition_than_pass_1_stops_the_run():
    """The contract of the cta_cell passes (include/comd_hip.h): eamForce3Gpu[Async] must cover the cells with the partition of the eamForce1Gpu[Async] calls of
    the same force evaluation -- the 16-bit numbers pass 1 leaves index the LDS image of a brick, and the image holds the stencils of the SELECTED cells only.
    Pass 1 over all cells followed by pass 3 over a cell list would read wrong neighbours silently; pass 1 records the selection it staged every cell's brick
    for, pass 3 compares, and the mismatch stops the run with a message (in a child process: the stop is exit(-1))."""
    import subprocess
    import sys
    code = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package()
pkg.setup_gpu(0, 0, verbose=False); pkg.init_parallel(0, 1, None)
hip = pkg.lib_hip()
with pkg.Simulation(["-x", 12, "-y", 12, "-z", 12, "-e", "-m", "cta_cell", "-r", 0.1]) as sim:
    gpu = ctypes.c_void_p(sim.lib.comdSimGpu(sim.ptr))
    CTA_CELL = 4
    hip.eamForce1Gpu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    hip.eamForce3GpuAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    hip.comdDeviceMalloc.restype = ctypes.c_void_p; hip.comdDeviceMalloc.argtypes = [ctypes.c_long]
    hip.comdCheckStatus.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    cells = np.arange(0, sim.n_local_boxes, 3, dtype=np.int32)                  # every third cell: bricks with some cells selected, some not
    d = ctypes.c_void_p(hip.comdDeviceMalloc(cells.nbytes))
    hip.comdMemcpyHtoD(d, cells.ctypes.data_as(ctypes.c_void_p), cells.nbytes)
    hip.eamForce1Gpu(gpu, CTA_CELL, 0)                                          # pass 1: all cells in one launch
    hip.comdCheckStatus(gpu, b"after pass 1"); print("pass 1 fine", flush=True)
    hip.eamForce3GpuAsync(gpu, len(cells), d, CTA_CELL, None, 0)                # pass 3: a cell list
    hip.comdCheckStatus(gpu, b"after pass 3"); print("NOT STOPPED", flush=True)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, COMD_EAM_GROUPS="0"))
    assert "pass 1 fine" in proc.stdout and "NOT STOPPED" not in proc.stdout, proc.stdout[-500:] + proc.stderr[-1500:]
    assert proc.returncode != 0 and "another partition" in proc.stderr, proc.stderr[-1500:]


def test_lj_wave_candidate_lists_do_not_lose_a_pair(gpu, monkeypatch):
    """The list build prunes in single precision on positions relative to the corner of the local domain, with a margin for that rounding
    (comd_device.hip ljBoxMarginsF); the force kernel decides every pair on the fp64 records.  With the lists and with the plain 27-cell
    walk the forces may differ by the order of a tail wave's partial sums (1e-14 eV/A here); a pair lost at the cutoff would be a jump
    of 2e-5 eV/A (the LJ force is not shifted) -- on a long box (coordinates up to ~250 A, where a float ulp is 1.5e-5 A) with a strongly
    disturbed lattice."""
    out = []
    for prune in ("1", "0"):
        monkeypatch.setenv("COMD_LJ_PRUNE", prune)
        with gpu.Simulation(_args((70, 7, 7), 0, 0.2, "thread_atom")) as sim:
            sim.step(2)
            out.append((sim.gather(2).copy(), sim.gather(3).copy()))
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-10 and np.abs(out[0][1] - out[1][1]).max() < 1e-12
    assert (np.abs(out[0][0] - out[1][0]).max(axis=1) == 0).mean() > 0.8        # the full waves add in the walk's order


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("case", ["eam_6_delta", "lj_8_delta"])
def test_reference_recorded_forces(gpu, case, method):
    """SECONDARY cross-check: the -r 0.1 force statistics of reference_values.json "survey_recorded" were written down by the survey stage from a
    stub-header build that nothing here regenerates; they pin nothing by themselves.  What ties the forces to the reference is
    tests/test_finite_difference.py: F = -dU/dr of the energies pinned to CoMD.c:897-899."""
    ref = S[case]
    with gpu.Simulation(_args(ref["nx"], ref["eam"], ref["delta"], method)) as sim:
        _, u, _ = _per_atom(sim)
        assert abs(u - ref["U_per_atom"]) < 1e-12
        f = sim.gather(2)
        norms = np.linalg.norm(f, axis=1)
        assert abs(norms.sum() - ref["sum_norm_f"]) < 1e-9 * ref["sum_norm_f"]
        assert abs(norms.max() - ref["max_norm_f"]) < 1e-11
        assert np.abs(f[0] - np.array(ref["f_gid0"])).max() < TOL["force_rel_to_max"] * ref["max_norm_f"]


# ---------------------------------------------------------------- energy traces
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("case", ["lj_20", "eam_20"])
def test_energy_trace_matches_reference(gpu, orc, case, method):
    """BASELINE configs[0] (20^3, 100 steps).  Two checkers side by side: the ORACLE stepped through the same 100 steps (the pinned one: oracle/comd_oracle.c
    is held to the reference-held energies and to the reference's own TUs, tests/test_oracle_golden.py) -- potential and kinetic energy per atom at steps
    10, 50 and 100 -- and the `survey_recorded` trace (recorded by the survey stage from a stand-in-header build: a secondary cross-check that pins nothing)."""
    ref = S[case]
    with gpu.Simulation(_args(ref["nx"], ref["eam"], 0.0, method)) as sim:
        o = orc.Oracle(ref["nx"], eam=ref["eam"], cap=max(sim.max_atoms, 64))
        n = sim.n_global
        e, u, _ = _per_atom(sim)
        assert abs(u - o.energy()[0] / n) < TOL["energy_per_atom_step0"]
        assert abs(u - ref["step0"]["U"]) < TOL["energy_per_atom_step0"]
        assert abs(e - ref["step0"]["E"]) < TOL["energy_per_atom_step0"]
        done = 0
        for step in (10, 50, 100):
            sim.step(step - done)
            o.step(step - done)
            done = step
            et, ut, kt = _per_atom(sim)
            op, ok = o.energy()
            assert abs(et - (op + ok) / n) < TOL["energy_per_atom_trace"], (step, et - (op + ok) / n)
            # (U and K exchange energy: a round-off difference in a force moves both, opposite ways, ten times further than their sum)
            assert abs(ut - op / n) < 10 * TOL["energy_per_atom_trace"] and abs(kt - ok / n) < 10 * TOL["energy_per_atom_trace"], (step, ut - op / n, kt - ok / n)
            assert abs(et - ref["E_at"][str(step)]) < TOL["energy_per_atom_trace"], step
        e, u, k = _per_atom(sim)
        assert abs(u - ref["step100"]["U"]) < 5e-12
        sim.sum_atoms()
        assert sim.energy()[2] == 4 * ref["nx"] ** 3


def test_eam_cohesive_energy(gpu):
    """CoMD.c:898: perfect Adams-EAM lattice, -3.538079224691 eV/atom."""
    with gpu.Simulation(_args(10, 1, 0.0, "cta_cell", ["-T", 0])) as sim:
        assert abs(_per_atom(sim)[1] - G["repo_native"]["eam_adams_cohesive_energy"]["value"]) < TOL["energy_per_atom_step0"]


@pytest.mark.parametrize("method,extra", [("thread_atom", ()), ("cta_cell", ()), ("thread_atom_nl", ()), ("cta_cell", ("-L",))])
def test_lj_cohesive_energy(gpu, orc, method, extra):
    """CoMD.c:897: the reference's documented LJ cohesive energy, -1.243619295058 eV/atom (cutoff 2.5 sigma, see
    tests/test_oracle_golden.py::test_lj_cohesive_energy_repo_native).  Every LJ kernel at --ljCutoffSigmas 2.5: the fixture to 1e-10,
    the oracle at the same cutoff to the usual tolerances, with and without displacements (forces are zero on the perfect lattice)."""
    ref = G["repo_native"]["lj_cohesive_energy_2p5_sigma"]
    with gpu.Simulation(_args(10, 0, 0.0, method, ["-T", 0, "--ljCutoffSigmas", 2.5, *extra])) as sim:
        _, u, k = _per_atom(sim)
        assert k == 0.0 and abs(u - ref["value"]) < ref["tolerance"]
        assert np.abs(sim.gather(2)).max() < 1e-11
    with gpu.Simulation(_args(10, 0, 0.15, method, ["--ljCutoffSigmas", 2.5, *extra])) as sim:
        o = orc.Oracle(10, eam=0, delta=0.15, lj_cutoff_sigmas=2.5)
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * np.abs(fo).max()
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]
        sim.step(10)
        o.step(10)
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs((ep + ek) - (op + ok)) / ng < TOL["energy_per_atom_trace"]


# ---------------------------------------------------------------- Verlet neighbour lists (-m thread_atom_nl)
@pytest.mark.parametrize("eam,n,delta", [(0, 12, 0.0), (0, (11, 13, 12), 0.2), (1, 8, 0.1), (1, (7, 9, 12), 0.3)])
def test_neighbor_list_forces_match_oracle(gpu, orc, eam, n, delta):
    """Lists hold everything within cutoff + skin; the force pass keeps r <= cutoff only, so forces do not depend on the skin."""
    with gpu.Simulation(_args(n, eam, delta, "thread_atom_nl")) as sim:
        o = orc.Oracle(n, eam=eam, delta=delta)
        fo, eo = o.gather(orc.F), o.gather(orc.U)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0)
        assert np.abs(sim.gather(3) - eo).max() <= TOL["per_atom_energy_abs"]
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs(ep - op) / ng < TOL["energy_per_atom_step0"]
        assert sim.nl_builds == 1


@pytest.mark.parametrize("key,skin", [("lj_20", 0.1), ("eam_20", 0.1), ("eam_20", 0.02), ("eam_20", 0.3)])
def test_neighbor_list_trace_matches_reference(gpu, key, skin):
    """100 steps with lists reused between builds reproduce the reference's energy trace (its own CPU runs with skins
    0.02 / 0.1 / 0.3 agree to 12 digits); the lists must be rebuilt now and then, not every step."""
    ref = S[key]
    with gpu.Simulation(_args(ref["nx"], ref["eam"], 0.0, "thread_atom_nl", ["-S", skin])) as sim:
        done = 0
        for step in (10, 50, 100):
            sim.step(step - done)
            done = step
            assert abs(_per_atom(sim)[0] - ref["E_at"][str(step)]) < TOL["energy_per_atom_trace"], step
        # LJ: skin/2 = 0.58 A is more than an atom travels in 100 fs -> the first list lasts; EAM skin 0.02: 0.05 A -> frequent rebuilds
        assert 1 <= sim.nl_builds < (102 if skin < 0.05 else 40), sim.nl_builds
        if skin < 0.05:
            assert sim.nl_builds > 5
        sim.sum_atoms()
        assert sim.energy()[2] == 4 * ref["nx"] ** 3


@pytest.mark.parametrize("eam,skin", [(1, 0.02), (1, 0.1), (0, 0.02)])
def test_deferred_list_check_is_the_blocking_check_made_two_drifts_early(eam, skin):
    """COMD_NL_DEFERRED=1: the host asks "must the lists be rebuilt?" without draining the stream -- answered from what the drift kernels up to the one two before the last
    found, against a threshold below skin/2 by what two steps can add (comd_hip.h comdNeighborListUpdateDeferredGpu).  Against the blocking form of the reference (the
    default: the exact rule tested after every drift): the same physics -- energies per atom agree to the trace tolerance, no atom lost -- and rebuilds that come a little
    earlier, never later: at least as many builds, not many more.  (A child process per mode: the host reads the switch once.)"""
    import subprocess
    import sys
    key = "eam_20" if eam else "lj_20"
    ref = S[key]
    code = r'''
import json, os, sys
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package()
pkg.setup_gpu(0, 0, verbose=False); pkg.init_parallel(0, 1, None)
with pkg.Simulation(%r) as sim:
    sim.step(100)
    ep, ek, n = sim.energy()
    sim.sum_atoms()
    print(json.dumps({"e": (ep + ek) / n, "builds": sim.nl_builds, "atoms": sim.energy()[2]}))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), _args(ref["nx"], ref["eam"], 0.0, "thread_atom_nl", ["-S", skin]))
    out = {}
    for mode in ("0", "1"):
        proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, COMD_NL_DEFERRED=mode))
        assert proc.returncode == 0, proc.stderr[-1500:]
        out[mode] = json.loads(proc.stdout.strip().splitlines()[-1])
        assert out[mode]["atoms"] == 4 * ref["nx"] ** 3
        assert abs(out[mode]["e"] - ref["E_at"]["100"]) < TOL["energy_per_atom_trace"], (mode, out[mode])
    b_sync, b_def = out["0"]["builds"], out["1"]["builds"]
    assert b_sync <= b_def <= int(1.7 * b_sync) + 2, (b_sync, b_def)
    if skin < 0.05 and eam:
        assert b_sync > 5


@pytest.mark.parametrize("eam,n,extra", [(0, 12, ()), (1, 8, ()), (1, 8, ("-t", "setfl", "-p", "Cu01.eam.alloy"))])
def test_neighbor_list_global_slot_format(gpu, orc, monkeypatch, eam, n, extra):
    """The plain list format (32-bit global slots, neighbours gathered from global memory) is what cells too large for the LDS
    kernels and EAM tables too large for the LDS (setfl) fall back to; COMD_NL_GLOBAL=1 selects it for any run."""
    if not extra:
        monkeypatch.setenv("COMD_NL_GLOBAL", "1")
    with gpu.Simulation(_args(n, eam, 0.1, "thread_atom_nl", list(extra))) as sim:
        o = orc.Oracle(n, eam=eam, delta=0.1, **({"pot_name": "Cu01.eam.alloy"} if extra else {}))
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0)
        sim.step(25)
        o.step(25)
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs((ep + ek) - (op + ok)) / ng < TOL["energy_per_atom_trace"]


# ---------------------------------------------------------------- pairlists for LJ cta_cell (-L)
@pytest.mark.parametrize("n,delta", [(12, 0.0), ((11, 13, 12), 0.2)])
def test_pairlist_forces_match_oracle(gpu, orc, n, delta):
    """-L: the first force call generates one bit per (wave, 8-neighbour trip), the second one uses them; both must give the oracle's forces."""
    with gpu.Simulation(_args(n, 0, delta, "cta_cell", ["-L"])) as sim:
        o = orc.Oracle(n, eam=0, delta=delta)
        fo, eo = o.gather(orc.F), o.gather(orc.U)
        for call in range(2):                                   # 0: generated during initialisation; 1: bits in use
            if call:
                sim.compute_force()
            assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0), call
            assert np.abs(sim.gather(3) - eo).max() <= TOL["per_atom_energy_abs"], call


@pytest.mark.parametrize("skin", [0.1, 0.02])
def test_pairlist_trace_matches_reference(gpu, skin):
    ref = S["lj_20"]
    with gpu.Simulation(_args(20, 0, 0.0, "cta_cell", ["-L", "-S", skin])) as sim:
        done = 0
        for step in (10, 50, 100):
            sim.step(step - done)
            done = step
            assert abs(_per_atom(sim)[0] - ref["E_at"][str(step)]) < TOL["energy_per_atom_trace"], step
        assert (sim.nl_builds > 5) if skin < 0.05 else (sim.nl_builds >= 1)
        sim.sum_atoms()
        assert sim.energy()[2] == 4 * 20 ** 3


def test_neighbor_list_80_cubed_properties(gpu):
    """Full size (BASELINE configs[1] workload) with lists: recorded step-0 energy, conservation, nothing lost, lists reused."""
    ref = S["lj_80_8ranks"]
    with gpu.Simulation(_args(80, 0, 0.0, "thread_atom_nl")) as sim:
        e0, u0, _ = _per_atom(sim)
        assert abs(u0 - ref["step0"]["U"]) < 1e-10
        sim.step(20)
        e1 = _per_atom(sim)[0]
        assert abs(e1 - ref["E_at"]["20"]) < TOL["energy_per_atom_trace"] * 2      # the reference's 8-rank CPU run
        assert abs(e1 - e0) < 2e-5
        assert sim.nl_builds == 1
        sim.sum_atoms()
        assert sim.energy()[2] == 4 * 80 ** 3


SETFL = ["-t", "setfl", "-p", "Cu01.eam.alloy"]


@pytest.mark.parametrize("method", METHODS)
def test_setfl_cohesive_energy(gpu, method):
    """CoMD.c:899: perfect lattice under Mishin's Cu01.eam.alloy (setfl, 10000-sample tables, cutoff 5.507 A): -3.539999969176."""
    ref = G["repo_native"]["eam_mishin_cohesive_energy"]
    with gpu.Simulation(_args(10, 1, 0.0, method, ["-T", 0] + SETFL)) as sim:
        assert abs(_per_atom(sim)[1] - ref["value"]) < ref["tolerance"]


@pytest.mark.parametrize("method", METHODS)
def test_setfl_forces_match_oracle(gpu, orc, method):
    """setfl tables are too large for the LDS: cta_cell reads them through L2 and stages up to 640 stencil atoms per wave."""
    n = (8, 9, 10)
    with gpu.Simulation(_args(n, 1, 0.12, method, SETFL)) as sim:
        o = orc.Oracle(n, eam=1, delta=0.12, cap=max(sim.max_atoms, 64), pot_name="Cu01.eam.alloy")
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0)
        assert np.abs(sim.gather(3) - o.gather(orc.U)).max() <= TOL["per_atom_energy_abs"]
        assert np.abs(sim.gather(4) - o.gather(orc.RHOBAR)).max() < TOL["eam_density_abs"]
        sim.step(20)
        o.step(20)
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs((ep + ek) - (op + ok)) / ng < TOL["energy_per_atom_trace"]


@pytest.mark.parametrize("eam,n", [(0, 12), (1, 9)])
def test_trajectory_tracks_oracle(gpu, orc, eam, n):
    """25 steps from a displaced lattice: positions, momenta and forces stay together atom by atom."""
    with gpu.Simulation(_args(n, eam, 0.15)) as sim:
        o = orc.Oracle(n, eam=eam, delta=0.15, cap=max(sim.max_atoms, 64))
        sim.step(25)
        o.step(25)
        assert np.abs(sim.gather(0) - o.gather(orc.R)).max() < 1e-11
        assert np.abs(sim.gather(1) - o.gather(orc.P)).max() < 1e-10 * np.abs(o.gather(orc.P)).max()
        assert np.abs(sim.gather(2) - o.gather(orc.F)).max() < 1e-9 * np.abs(o.gather(orc.F)).max()


# ---------------------------------------------------------------- index work: bit-exact
def _assert_cells_equal(c, oc, n_total):
    assert np.array_equal(c["nAtoms"], oc["nAtoms"])
    for b in range(n_total):
        k = c["nAtoms"][b]
        assert np.array_equal(c["gid"][b, :k], oc["gid"][b, :k]), b
        for name in ("rx", "ry", "rz", "px", "py", "pz"):
            assert np.array_equal(c[name][b, :k], oc[name][b, :k]), (b, name)


@pytest.mark.parametrize("eam,n,shift", [(1, 8, (2.1, -1.9, 2.3)), (1, (7, 9, 8), (-2.4, 2.2, -0.3)), (0, 10, (5.0, -4.5, 3.9))])
def test_redistribution_is_bit_exact(gpu, orc, eam, n, shift):
    """Rigid translation by a fraction of a cell: ~half the atoms change cell, many wrap periodically.  Afterwards every
    cell -- local and halo -- must hold exactly the oracle's atoms, in gid order, with bit-identical r (incl. the
    periodic shift added by the halo pack) and p; forces and energy are translation invariant."""
    # the device appends movers before it squeezes the holes they leave, so a cell transiently holds old + incoming
    # atoms; with 3/4 of all atoms moving at once (never the case in MD) that needs 2x the usual head-room
    extra = ["--maxAtoms", 64 if eam else 384]
    with gpu.Simulation(_args(n, eam, 0.05, extra=extra)) as sim:
        o = orc.Oracle(n, eam=eam, delta=0.05, cap=sim.max_atoms)
        _assert_cells_equal(sim.cells(), o.rank_cells(0), sim.n_total_boxes)      # initial redistribute
        f0 = sim.gather(2)
        u0 = sim.energy()[0]
        r = sim.gather(0)
        assert np.array_equal(r, o.gather(orc.R))
        sim.scatter(0, r + np.array(shift))
        o.scatter(orc.R, r + np.array(shift))
        sim.redistribute(); sim.compute_force(); sim.kinetic_energy()
        o.redistribute()
        c, oc = sim.cells(), o.rank_cells(0)
        _assert_cells_equal(c, oc, sim.n_total_boxes)
        moved = sum(int(c["nAtoms"][b]) for b in range(sim.n_local_boxes))
        assert moved == sim.n_global
        assert abs(sim.energy()[0] - u0) < 0.1 * TOL["translation_invariance_rel"] * abs(u0)
        assert np.abs(sim.gather(2) - f0).max() < TOL["translation_invariance_rel"] * np.abs(f0).max()


def test_halo_cells_are_periodic_images(gpu):
    """Every halo atom is a copy of a local atom displaced by a lattice vector of the box; gid order inside cells."""
    n = 8
    with gpu.Simulation(_args(n, 1, 0.1)) as sim:
        c = sim.cells()
        ext = n * 3.615
        local = {}
        for b in range(sim.n_local_boxes):
            for i in range(c["nAtoms"][b]):
                local[int(c["gid"][b, i])] = (c["rx"][b, i], c["ry"][b, i], c["rz"][b, i])
        assert len(local) == sim.n_global
        n_halo = 0
        for b in range(sim.n_local_boxes, sim.n_total_boxes):
            g = c["gid"][b, :c["nAtoms"][b]]
            assert np.all(np.diff(g) > 0)
            for i, gid in enumerate(g):
                d = np.array([c["rx"][b, i], c["ry"][b, i], c["rz"][b, i]]) - np.array(local[int(gid)])
                k = np.rint(d / ext)
                assert np.abs(d - k * ext).max() < TOL["position_abs"] and np.abs(k).max() == 1
                n_halo += 1
        assert n_halo > sim.n_global          # 8^3 EAM: the halo shell holds more images than there are atoms


# ---------------------------------------------------------------- properties at BASELINE size
@pytest.mark.parametrize("eam,method,steps", [(0, "thread_atom", 10), (1, "cta_cell", 20), (1, "thread_atom", 20)])
def test_full_size_properties(gpu, eam, method, steps):
    """80^3 (2,048,000 atoms), BASELINE configs 2 and 3: the oracle is too slow here, so check what the physics guarantees:
    step-0 energy equals the recorded reference value, total energy is conserved, total momentum stays zero,
    forces sum to zero, no atom is lost, cells stay sorted."""
    ref = S["lj_80_8ranks" if not eam else "eam_80_8ranks"]
    with gpu.Simulation(_args(80, eam, 0.0, method)) as sim:
        e0, u0, k0 = _per_atom(sim)
        if eam:
            # the reference's own captured GPU log of this exact problem (step-0 row, repo-native pin) ...
            log = G["repo_native"]["eam_80_step0_gpu_log"]
            assert abs(u0 - log["U"]) < 2e-12 and abs(k0 - log["K"]) < 2e-12 and abs(e0 - log["E"]) < 2e-12
            # ... and its CPU path, whose single running sum over 2 M atoms carries ~1e-10 of round-off
            assert abs(u0 - ref["step0"]["U"]) < 1e-10
        else:
            assert abs(u0 - ref["step0"]["U"]) < TOL["energy_per_atom_step0"]
        sim.step(steps)
        e1 = _per_atom(sim)[0]
        key = str(steps)
        if key in ref["E_at"]:
            assert abs(e1 - ref["E_at"][key]) < TOL["energy_per_atom_trace"] * 2
        assert abs(e1 - e0) < 5e-6 * abs(e0)
        sim.sum_atoms()
        assert sim.energy()[2] == 4 * 80 ** 3
        c = sim.cells()
        nl = sim.n_local_boxes
        mask = np.arange(sim.max_atoms)[None, :] < c["nAtoms"][:nl, None]
        for name in ("px", "py", "pz", "fx", "fy", "fz"):
            tot = c[name][:nl][mask].sum()
            scale = np.abs(c[name][:nl][mask]).sum()
            assert abs(tot) < 1e-11 * scale, name
        g = np.where(mask, c["gid"][:nl], np.iinfo(np.int32).max)
        assert np.all(np.diff(g.astype(np.int64), axis=1) >= 0)
        assert np.array_equal(np.sort(c["gid"][:nl][mask]), np.arange(4 * 80 ** 3))


# ---------------------------------------------------------------- the executable: CLI, stdout table, validation block, YAML
@pytest.mark.parametrize("launcher_env", [False, True])
def test_comd_hip_executable_report(gpu, tmp_path, launcher_env):
    """`comd-hip` is the reference's CoMD binary for this path: same flags, same table (CoMD.c:478-493), same validation block
    (CoMD.c:421-438), a YAML side file (yamlOutput.c:45-67).  LJ 20^3, 20 steps: the rows must carry the reference's energies.
    launcher_env: started the way a multi-process launcher would (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT), one rank, loopback:
    the RCCL id travels through the rendezvous file and every halo message through ncclSend/ncclRecv."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "comd-cuda-async_amd", "csrc", "comd-hip")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if launcher_env:
        env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29611", COMD_LOOPBACK_TRANSPORT="1", COMD_RDZV_DIR=str(tmp_path))
    proc = subprocess.run([exe, "-x", "20", "-y", "20", "-z", "20", "-N", "20", "-n", "10", "-d", os.path.join(root, "pots")],
                          capture_output=True, text=True, cwd=tmp_path, env=env, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = proc.stdout
    assert "#  Loop   Time(fs)       Total Energy   Potential Energy     Kinetic Energy  Temperature   (us/atom)     # Atoms" in out
    rows = {int(m.group(1)): [float(v) for v in m.group(2).split()]
            for m in re.finditer(r"^\s+(\d+)\s+(\d+\.\d+\s+-\d+\.\d+\s+-\d+\.\d+\s+\d+\.\d+\s+\d+\.\d+\s+\d+\.\d+\s+\d+)\s*$", out, flags=re.M)}
    assert set(rows) == {0, 10, 20}
    ref = S["lj_20"]
    assert abs(rows[0][1] - ref["step0"]["E"]) < 2e-12 and abs(rows[0][2] - ref["step0"]["U"]) < 2e-12
    assert abs(rows[10][1] - ref["E_at"]["10"]) < 2e-12
    assert rows[0][4] == 600.0 and rows[20][6] == 32000
    assert "Simulation Validation:" in out and "no atoms lost" in out
    ratio = float(re.search(r"eFinal/eInitial : (\S+)", out).group(1))
    assert abs(ratio - 1.0) < 1e-4            # the reference's energy-conservation check (CoMD.c:413-440)
    assert "Average all atom update rate:" in out and "Timings for Rank 0" in out
    yamls = [f for f in os.listdir(tmp_path) if f.endswith(".yaml")]
    assert len(yamls) == 1
    y = open(os.path.join(tmp_path, yamls[0])).read()
    for key in ("Mini-Application Name", "Command Line Parameters:", "nx: 20", "Simulation data:", "Potential data:", "Performance Results:", "AtomUpdateRate:"):
        assert key in y, key


# ---------------------------------------------------------------- seeded sweep over box shapes, displacements, methods and overlap modes
def _sweep_cases():
    import random
    rnd = random.Random(7)
    cases = []
    for eam, methods, lo, hi in ((0, ["thread_atom", "cta_cell", "thread_atom_nl", "cta_cell -L"], 8, 15), (1, ["thread_atom", "cta_cell", "thread_atom_nl"], 5, 13)):
        for m in methods:
            for _ in range(4):
                n = (rnd.randint(lo, hi), rnd.randint(lo, hi), rnd.randint(lo, hi))
                cases.append((eam, m, n, rnd.choice([0.0, 0.05, 0.15, 0.3]), rnd.choice([0, 1]), rnd.choice([3, 7])))
    return cases


@pytest.mark.parametrize("eam,method,n,delta,overlap,steps", _sweep_cases())
def test_seeded_sweep_matches_oracle(gpu, orc, eam, method, n, delta, overlap, steps):
    """Non-cubic boxes (2 to 4 cells per axis, uneven cell occupancies), displaced lattices, every force method, both overlap modes:
    forces at step 0 and after a few steps, and the total energy, against the oracle."""
    m = method.split()
    with gpu.Simulation(_args(n, eam, delta, m[0], ["-a", overlap] + m[1:])) as sim:
        o = orc.Oracle(n, eam=eam, delta=delta, cap=max(sim.max_atoms, 64))
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0)
        sim.step(steps)
        o.step(steps)
        fo = o.gather(orc.F)
        assert np.abs(sim.gather(2) - fo).max() <= 1e-9 * max(np.abs(fo).max(), 1.0)
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs((ep + ek) - (op + ok)) / ng < TOL["energy_per_atom_trace"]


# ---------------------------------------------------------------- determinism
@pytest.mark.parametrize("eam,n,method", [(0, 14, "thread_atom"), (0, 14, "cta_cell"), (0, 14, "thread_atom_nl"), (1, 12, "cta_cell"), (1, 12, "thread_atom_nl")])
def test_runs_are_bit_reproducible(gpu, eam, n, method):
    """No floating-point atomics anywhere on the path and every cell in gid order: two runs of the same input agree to the last bit
    (the reference's energy reduction and its unsorted halo cells make its runs differ, gpu_reduce.h:32-98)."""
    out = []
    for _ in range(2):
        with gpu.Simulation(_args(n, eam, 0.1, method, ["-a", 1])) as sim:
            sim.step(25)
            out.append((sim.gather(0), sim.gather(1), sim.gather(2), sim.energy()))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a, b)
    assert out[0][3] == out[1][3]


# ---------------------------------------------------------------- Hilbert cell numbering (-H)
@pytest.mark.parametrize("eam,n,method", [(0, (12, 10, 14), "thread_atom"), (0, 12, "cta_cell"), (0, 14, "thread_atom_nl"), (1, (9, 12, 7), "cta_cell"),
                                          (1, 10, "thread_atom"), (1, 12, "thread_atom_nl")])
def test_hilbert_numbering_changes_nothing_but_the_cell_ids(gpu, orc, eam, n, method):
    """-H renumbers the local link cells along a Hilbert curve: which atoms share a cell, their order inside it (gid) and the order in
    which a cell's 27 neighbours are visited are untouched.  The sums taken over ALL atoms (centre-of-mass velocity and temperature at
    start-up on the host, the energy reduction on the device) run in cell order, so they move in their last bits and with them the
    momenta; everything must still agree with the natural numbering and with the oracle to round-off after 25 steps (overlap mode on,
    lists rebuilt on the way)."""
    out = []
    for extra in ([], ["-H"]):
        with gpu.Simulation(_args(n, eam, 0.1, method, ["-a", 1] + extra)) as sim:
            sim.step(25)
            out.append((sim.gather(0), sim.gather(1), sim.gather(2), sim.gather(3), sim.energy()))
    r0, p0, f0, e0, en0 = out[0]
    r1, p1, f1, e1, en1 = out[1]
    assert np.abs(r0 - r1).max() < 1e-12 and np.abs(p0 - p1).max() < 1e-12 * np.abs(p0).max()
    assert np.abs(f0 - f1).max() < 1e-11 * np.abs(f0).max() and np.abs(e0 - e1).max() < 1e-11
    assert abs(en0[0] - en1[0]) / en0[2] < 1e-13 and abs(en0[1] - en1[1]) / en0[2] < 1e-13 and en0[2] == en1[2]
    o = orc.Oracle(n, eam=eam, delta=0.1)
    o.step(25)
    fo = o.gather(orc.F)
    assert np.abs(f1 - fo).max() <= 1e-9 * np.abs(fo).max()
    assert abs((en1[0] + en1[1]) - sum(o.energy())) / en1[2] < TOL["energy_per_atom_trace"]


# ---------------------------------------------------------------- cubic-spline EAM tables (-P): PARITY-UNPINNED
@pytest.mark.parametrize("method", METHODS + ["thread_atom_nl"])
@pytest.mark.parametrize("n,delta", [(8, 0.0), ((7, 9, 12), 0.2)])
def test_spline_tables_match_the_restatement(gpu, orc, method, n, delta):
    """-P evaluates phi and rho as cubic splines in r^2 (gpu_utility.c:377-430, gpu_common.h:95-129).  The reference implements this
    on the GPU only and holds no output of it, so the check is against the oracle's restatement of that device code (parity
    unpinned) -- plus one sanity bound from the physics: a spline through the same samples stays close to the quadratic tables."""
    with gpu.Simulation(_args(n, 1, delta, method, ["-P"])) as sim:
        o = orc.Oracle(n, eam=1, delta=delta, spline=True)
        fo, eo = o.gather(orc.F), o.gather(orc.U)
        assert np.abs(sim.gather(2) - fo).max() <= TOL["force_rel_to_max"] * max(np.abs(fo).max(), 1.0)
        assert np.abs(sim.gather(3) - eo).max() <= TOL["per_atom_energy_abs"]
        assert np.abs(sim.gather(4) - o.gather(orc.RHOBAR)).max() < TOL["eam_density_abs"]
        quad = orc.Oracle(n, eam=1, delta=delta)
        assert 0 < abs(o.energy()[0] - quad.energy()[0]) / o.n_global < 1e-4          # different interpolant, same potential
        sim.step(20)
        o.step(20)
        (ep, ek, ng), (op, ok) = sim.energy(), o.energy()
        assert abs((ep + ek) - (op + ok)) / ng < TOL["energy_per_atom_trace"]
