"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POT_DIR = os.path.join(ROOT, "pots")
R, P, F, U, RHOBAR, DFEMBED = range(6)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # COMD_PRECISION=single: the restatement of the reference's single-precision build (real_t = float), same C interface
    path = os.path.join(ROOT, "oracle", "liboracle_sp.so" if os.environ.get("COMD_PRECISION", "double") == "single" else "liboracle.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} missing: run `make -C {os.path.join(ROOT, 'oracle')}`")
    L = ctypes.CDLL(path)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.oracle_create.restype = vp
    L.oracle_create.argtypes = [ci] * 6 + [cd, ci, ctypes.c_char_p, ctypes.c_char_p, cd, cd, cd, ci]
    L.oracle_destroy.argtypes = [vp]
    L.oracle_step.argtypes = [vp, ci]
    for fn in ("oracle_redistribute", "oracle_compute_force", "oracle_kinetic_energy"):
        getattr(L, fn).argtypes = [vp]
    L.oracle_advance_velocity.argtypes = [vp, cd]
    L.oracle_advance_position.argtypes = [vp, cd]
    for fn in ("oracle_n_global", "oracle_n_ranks", "oracle_rank_cell_cap"):
        getattr(L, fn).argtypes = [vp]
        getattr(L, fn).restype = ci
    for fn in ("oracle_e_potential", "oracle_e_kinetic", "oracle_cutoff", "oracle_mass", "oracle_lattice", "oracle_loop_seconds"):
        getattr(L, fn).argtypes = [vp]
        getattr(L, fn).restype = cd
    L.oracle_gather.argtypes = [vp, ci, vp]
    L.oracle_scatter.argtypes = [vp, ci, vp]
    L.oracle_rank_grid.argtypes = [vp, ci, vp, vp, vp]
    L.oracle_rank_natoms.argtypes = [vp, ci, vp]
    L.oracle_rank_gid.argtypes = [vp, ci, vp]
    L.oracle_rank_array.argtypes = [vp, ci, ci, ci, vp]
    L.oracle_box_from_tuple.argtypes = [vp, ci, ci, ci, ci]
    L.oracle_box_from_coord.argtypes = [vp, ci, vp]
    L.oracle_face_cells.argtypes = [vp, ci, ci, ci, vp]
    L.oracle_lcg61.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.oracle_lcg61.restype = cd
    L.oracle_mkSeed.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_mkSeed.restype = ctypes.c_uint64
    L.oracle_gasdev.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.oracle_gasdev.restype = cd
    L.oracle_eam_interpolate.argtypes = [vp, ci, cd, ctypes.POINTER(cd), ctypes.POINTER(cd)]
    L.oracle_eam_table.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(cd), ctypes.POINTER(cd), vp]
    L.oracle_use_splines.argtypes = [vp]
    L.oracle_threads.restype = ci
    L.oracle_set_threads.argtypes = [ci]
    L.oracle_set_lj_cutoff_sigmas.argtypes = [cd]
    _lib = L
    return L


class Oracle:
    """CPU restatement of the reference path; all ranks of the decomposition live in this process."""

    def __init__(self, n, procs=(1, 1, 1), eam=0, temperature=600.0, delta=0.0, dt=1.0, cap=0, lat=-1.0, pot_name="Cu_u6.eam", spline=False,
                 lj_cutoff_sigmas=5.0):
        self.L = lib()
        nx, ny, nz = (n, n, n) if isinstance(n, int) else n
        self.L.oracle_set_lj_cutoff_sigmas(lj_cutoff_sigmas)       # 5 = the reference (ljForce.c:114); 2.5 meets the fixture of CoMD.c:897
        self.ptr = self.L.oracle_create(nx, ny, nz, procs[0], procs[1], procs[2], lat, eam,
                                        POT_DIR.encode(), pot_name.encode(), temperature, delta, dt, cap)
        self.L.oracle_set_lj_cutoff_sigmas(5.0)
        if not self.ptr:
            raise RuntimeError("oracle_create failed")
        if spline:
            self.L.oracle_use_splines(self.ptr)          # -P: parity-unpinned restatement of the reference's device code
        self.n_global = self.L.oracle_n_global(self.ptr)
        self.n_ranks = self.L.oracle_n_ranks(self.ptr)
        self.cap = self.L.oracle_rank_cell_cap(self.ptr)

    def step(self, n):
        self.L.oracle_step(self.ptr, n)

    def redistribute(self):
        self.L.oracle_redistribute(self.ptr)

    def compute_force(self):
        self.L.oracle_compute_force(self.ptr)

    def kinetic_energy(self):
        self.L.oracle_kinetic_energy(self.ptr)

    def energy(self):
        return self.L.oracle_e_potential(self.ptr), self.L.oracle_e_kinetic(self.ptr)

    def gather(self, which):
        out = np.zeros((self.n_global, 3) if which <= F else (self.n_global,), dtype=np.float64)
        self.L.oracle_gather(self.ptr, which, out.ctypes.data)
        return out

    def scatter(self, which, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self.L.oracle_scatter(self.ptr, which, arr.ctypes.data)

    def rank_grid(self, rank=0):
        g = (ctypes.c_int * 3)()
        nl, nt = ctypes.c_int(), ctypes.c_int()
        self.L.oracle_rank_grid(self.ptr, rank, g, ctypes.byref(nl), ctypes.byref(nt))
        return tuple(g), nl.value, nt.value

    def rank_cells(self, rank=0):
        _, _, nt = self.rank_grid(rank)
        out = {"nAtoms": np.zeros(nt, dtype=np.int32), "gid": np.zeros((nt, self.cap), dtype=np.int32)}
        self.L.oracle_rank_natoms(self.ptr, rank, out["nAtoms"].ctypes.data)
        self.L.oracle_rank_gid(self.ptr, rank, out["gid"].ctypes.data)
        for which, names in ((R, ("rx", "ry", "rz")), (P, ("px", "py", "pz")), (F, ("fx", "fy", "fz"))):
            for comp, name in enumerate(names):
                a = np.zeros((nt, self.cap), dtype=np.float64)
                self.L.oracle_rank_array(self.ptr, rank, which, comp, a.ctypes.data)
                out[name] = a
        for which, name in ((U, "e"), (RHOBAR, "rhobar"), (DFEMBED, "dfEmbed")):
            a = np.zeros((nt, self.cap), dtype=np.float64)
            self.L.oracle_rank_array(self.ptr, rank, which, 0, a.ctypes.data)
            out[name] = a
        return out

    def face_cells(self, rank, kind, face):
        n = self.L.oracle_face_cells(self.ptr, rank, kind, face, None)
        a = np.zeros(n, dtype=np.int32)
        self.L.oracle_face_cells(self.ptr, rank, kind, face, a.ctypes.data)
        return a

    def close(self):
        if self.ptr:
            self.L.oracle_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
