"""The single-precision build (make PRECISION=single: real_t = float, the reference's DOUBLE_PRECISION = OFF, mytype.h:8-21, Makefile:12).

A process binds one precision (the two builds export the same symbols), so every leg runs the ordinary test files in a child process
with COMD_PRECISION=single: the product loads lib*_sp.so, the checker loads oracle/liboracle_sp.so (the same restatement compiled with
real_t = float) and the tolerances are tests/golden/reference_values.json "tolerances_single" (their derivation is written there).
Index work -- lattice, momenta, cell membership, gid order, halo images -- stays bit-exact in either precision.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "comd-cuda-async_amd", "csrc")


def _run(files, k, marker, timeout):
    env = dict(os.environ, COMD_PRECISION="single")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", marker, "-p", "no:cacheprovider", *files] + (["-k", k] if k else [])
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = proc.stdout[-3000:] + proc.stderr[-2000:]
    assert proc.returncode == 0, tail
    return proc.stdout


def test_single_precision_libraries_exist_and_export_the_abi():
    """libcomd_hip_sp.so exports what include/comd_hip.h declares, same as the double build (checked in a child: one precision per process)."""
    for lib in ("libcomd_hip_sp.so", "libcomd_host_sp.so"):
        assert os.path.exists(os.path.join(CSRC, lib)), f"{lib} missing: make -C {CSRC} PRECISION=single"
    assert os.path.exists(os.path.join(ROOT, "oracle", "liboracle_sp.so"))
    out = _run(["tests/test_host_logic.py"], "test_abi_exports_every_declared_symbol or test_product_never_links_the_oracle", "not gpu", 300)
    assert "passed" in out


def test_single_precision_host_logic_is_bit_exact_against_the_checker():
    """Lattice, Maxwell-Boltzmann momenta (mixed float / double expressions evaluated as the reference's C does), cell numbering incl. tie
    rules, halo cell lists, and the multi-process exchange driver (2 and 8 ranks): bit for bit against the float build of the checker."""
    out = _run(["tests/test_host_logic.py", "tests/test_multirank.py"],
               "test_initial_state or test_displaced_state or test_cell_numbering or test_halo_cell_lists or test_host_logic", "not gpu", 900)
    assert "passed" in out


def test_single_precision_checker_meets_the_reference_held_energies():
    """liboracle_sp.so itself against the values the reference holds (CoMD.c:897-899, the step-0 row of the K20 log), at the float tolerances:
    the float checker is pinned by more than being the same source as the double one."""
    out = _run(["tests/test_oracle_golden.py"], "repo_native or step0_row_of_k20_log or setfl_mishin_cohesive_energy", "not gpu", 600)
    assert "4 passed" in out, out[-800:]


@pytest.mark.gpu
def test_single_precision_gpu_parity():
    """Forces, energies, densities, redistribution, lists, pairlists, setfl tables and reproducibility of the float kernels on the GPU."""
    k = ("test_forces_match_oracle or test_eam_any_cell_capacity or test_overlap_mode_small_interior or test_neighbor_list_forces_match_oracle "
         "or test_redistribution_is_bit_exact or test_pairlist_forces_match_oracle or test_setfl_forces_match_oracle or test_runs_are_bit_reproducible "
         "or test_halo_cells_are_periodic_images or test_neighbor_list_global_slot_format")
    out = _run(["tests/test_gpu_parity.py"], k, "gpu", 1500)
    assert "passed" in out and "failed" not in out


@pytest.mark.gpu
def test_single_precision_multi_rank():
    out = _run(["tests/test_multirank.py"], "test_gpu_path_multi_rank_shared_device and (cta_cell-0 or thread_atom-1 or thread_atom_nl-0) or test_rccl_transport_loopback and cta_cell", "gpu", 1500)
    assert "passed" in out and "failed" not in out
