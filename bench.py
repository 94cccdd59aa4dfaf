#!/usr/bin/env python3
"""bench.py -- atom-updates/s of the CoMD hot path (force + velocity-Verlet + redistribute + halo) on MI355X.

    python bench.py                       # 1 GPU, LJ Cu 80^3 (BASELINE.json configs[1]), thread_atom, fp64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # weak scaling: 80^3 atoms per GPU

A "step" is one full time step of timestep() (timestep.c:48-100).  The timed region is exactly K steps bracketed by a
barrier + device synchronisation on both sides; the maximum over ranks is reported.  `value` is the whole-job rate:
nGlobal * K / seconds.  Atoms are resident in HBM when the timed region starts (they are generated on the host once,
uploaded by CopyDataToGpu, and never leave the device).

Extra objects on the JSON line:
  roofline     : the dominant kernel (the LJ or EAM force kernel) against the 8 TB/s HBM roof, from HIP events recorded
                 on the launch stream around every force launch of the timed region (comdForceTiming*).
                 Algorithmic bytes per atom: LJ force 56 B, EAM force (3 passes) 176 B.
  cpu_baseline : the CPU restatement (oracle/, kind "port") timed on this host's cores on a bounded sample of the same
                 workload (same potential, same density, fewer atoms and steps), rank 0 at N = 1 only.
  variants     : N = 1 only: the same K timed steps with the other force methods (cta_cell, the Verlet-list method
                 thread_atom_nl) and the other potential, for comparison; `value` is always the named configuration.
                 The BASELINE.json configurations among them (EAM 80^3 cta_cell = configs[2]) carry a `roofline` object of their own.
  target_line  : N = 1 only: BASELINE.json's north_star line, LJ 256^3 (67 M atoms) thread_atom, 10 timed steps after 2 -- value, ms/step,
                 kernel and whole-evaluation ms, HBM and fp64 fractions, memory of the candidate lists; skipped with a stated reason when
                 the device has less than 60 GB free.

N > 1: the ranks form their RCCL communicator the way comd-hip does (comdCommInitFromEnv: RANK / WORLD_SIZE / LOCAL_RANK from the launcher, rendezvous
through a file keyed by the launcher's pid and restart count); the two collectives the harness needs (barrier, max of the elapsed times) go through
CommTransport.  No torch in the rank processes: libcomd_hip.so runs on the one HIP runtime and the one librccl it was linked against (/opt/rocm).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0
FORCE_BYTES = {"lj": 56.0, "eam": 176.0}        # algorithmic bytes per atom per force evaluation
STEP_BYTES = {"lj": 276.0, "eam": 396.0}        # ... per full time step (force + 2 half kicks + drift)
GRIDS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}
# fp64 vector ceiling (the bound that actually binds these kernels): useful FLOP per atom per force evaluation.
# LJ 5 sigma: ~4000 candidates x (3 sub + mul + 2 fma = 8 FLOP) + ~550 pairs x ~25 FLOP; EAM: 2 passes x (~283 x 8 + ~42 x ~70).
FP64_VECTOR_PEAK_TFLOPS = 78.6
FORCE_FLOP = {"lj": 4000 * 8 + 550 * 25, "eam": 2 * (283 * 8 + 42 * 70)}
# LJ thread_atom tests only the candidates its per-wave lists keep (within the cutoff of the wave's bounding box): ~2350 of the 4000
# (SQ_INSTS_SMEM of the force kernel, profiles/r02_pmc_summary.json) -- the FLOP it executes usefully, not the stencil's
FORCE_FLOP_LISTED = {"lj": 2350 * 8 + 550 * 25}
# Verlet lists (skin 10 %): ~732 (LJ) / ~57 (EAM) listed neighbours take the place of the stencil candidates
FORCE_FLOP_NL = {"lj": 732 * 8 + 550 * 25, "eam": 2 * (57 * 8 + 42 * 70)}
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0                 # wave-instructions/s the chip can issue: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per fp64 wave-instruction
KERNEL_NAME = {("lj", "thread_atom"): "LJ_Force_thread_atom", ("lj", "cta_cell"): "LJ_Force_cta_cell_boxes", ("lj", "thread_atom_nl"): "LJ_Force_nl_slabs", ("lj", "cta_cell_pairlist"): "LJ_Force_cta_cell",
               ("eam", "thread_atom"): "EAM_Force_atom_brick<1> + <3>", ("eam", "cta_cell"): "EAM_Force_cta_brick<1> + <3>",
               ("eam", "thread_atom_nl"): "EAM_Force_cta_brick<1, listed> + <3, listed> (Verlet rows)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pot", choices=["lj", "eam"], default="lj")
    ap.add_argument("--method", choices=["thread_atom", "thread_atom_nl", "cta_cell", "cta_cell_pairlist"], default=None)
    ap.add_argument("--nx", type=int, default=80, help="unit cells per GPU along each axis")
    ap.add_argument("--async-halo", type=int, default=None, help="-a flag: overlap interior force with the halo exchange")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the other force methods (reported as `variants` at N = 1)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-target-line", action="store_true", help="skip the LJ 256^3 leg (N = 1 default run)")
    ap.add_argument("--precision", choices=["double", "single"], default="double",
                    help="single: the PRECISION=single build (real_t = float); never the headline -- the default run reports it as a variant")
    ap.add_argument("--allow-host-staged", action="store_true",
                    help="N > 1 only: if the RCCL communicator cannot be formed, run over host-staged gloo messages instead of failing "
                         "(the line then carries \"measured\": false -- it is a functional rehearsal, not an xGMI number)")
    return ap.parse_args()


class stdout_to_stderr:
    """RCCL prints a version banner to stdout when a communicator is created; stdout must carry the JSON line only."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def cpu_baseline(pot, seconds):
    """Time the oracle on this host: 20^3 cells of the same lattice/potential (BASELINE configs[0] for LJ).
    Two legs: this GPU's share of the host cores (OpenMP over cells), and ONE core -- the reference is single-threaded
    per rank ("Threading: none", yamlOutput.c:87), so the 1-core figure is the like-for-like one."""
    orc = ge.load_oracle()
    lib = orc.lib()
    n = 20
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1

    def leg(threads, budget):
        lib.oracle_set_threads(threads)
        t0 = time.time()
        o = orc.Oracle(n, eam=1 if pot == "eam" else 0)
        o.step(2)
        per_step = max((time.time() - t0) / 3.0, 1e-4)
        steps = max(3, min(2000, int(budget / per_step)))
        before = lib.oracle_loop_seconds(o.ptr)
        o.step(steps)
        loop = lib.oracle_loop_seconds(o.ptr) - before
        return o.n_global * steps / loop, int(lib.oracle_threads()), steps, loop, o.n_global

    rate, used, steps, loop, n_atoms = leg(min(cores, 16), seconds * 0.5)      # the GPU box gives one GPU a 16-core share
    rate1, _, steps1, loop1, _ = leg(1, seconds * 0.5)
    # the headline figure is ONE core: the reference runs one thread per rank ("Threading: none", yamlOutput.c:87; SURVEY.md 8d "1 core for C1"); the share of the
    # host cores this GPU gets (OpenMP over cells) is reported beside it
    return {"value": rate1, "unit": "atom-updates/s", "cores": 1, "kind": "port",
            "sample": f"{pot.upper()} Cu {n}^3 FCC ({n_atoms} atoms), {steps1} steps, oracle/comd_oracle.c (27-cell stencil form), 1 thread, {loop1:.1f} s",
            "all_cores_of_this_gpu": {"value": rate, "unit": "atom-updates/s", "cores": used,
                                      "sample": f"same workload, {steps} steps, OpenMP over cells on {used} threads, {loop:.1f} s"}}


def profiled(pot, method, nx):
    """The committed rocprofv3 PMC passes for this workload (profiles/rNN_traffic.json, newest round first): HBM-side bytes (FETCH_SIZE + WRITE_SIZE) and VALU
    wave-instructions per force evaluation.  PMC counters cannot be read from inside the timed process, so these are the last profiled values, and the
    record says which file they come from."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name))).get(f"{pot}/{method}/{nx}")
        except (OSError, ValueError):
            rec = None
        if rec:
            # a record made for another version of the kernel must not be divided by this run's time: profiles/rNN_summarize.py stores a hash of the kernel's
            # source header with every record; a record whose hash is not the current one is reported as stale, not used
            if rec.get("kernel_source_sha16") != kernel_source_hash(pot, method):      # (records of rounds 1-3 carry no hash: stale by definition)
                return None, f"profiles/{name} (STALE: collected for another version of the kernel source; re-run profiles/r04_collect.sh)"
            return rec, f"profiles/{name}"
    return None, None


KERNEL_SOURCES = {("lj", "thread_atom"): ["lj_kernels.h"], ("lj", "cta_cell"): ["lj_kernels.h"], ("lj", "cta_cell_pairlist"): ["lj_kernels.h"], ("lj", "thread_atom_nl"): ["nl_kernels.h"],
                  ("eam", "cta_cell"): ["eam_brick_kernels.h"], ("eam", "thread_atom_nl"): ["eam_brick_kernels.h"], ("eam", "thread_atom"): ["eam_atom_brick_kernels.h"]}
COMMON_SOURCES = ["device_common.h"]              # (interpolate(), the reciprocal square root, the lane reductions: part of every force kernel)


def kernel_source_hash(pot, method):
    """sha256 (first 16 hex digits) of the header(s) that hold the force kernel of this path: what a stored PMC record is valid for."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[(pot, method)] + COMMON_SOURCES:
        h.update(open(os.path.join(ROOT, "comd-cuda-async_amd", "csrc", "hip", name), "rb").read())
    return h.hexdigest()[:16]


def roofline_object(pot, method, nx, n_local, force_ms, aux_ms, precision, one_gpu=True):
    """`roofline` of one force path: algorithmic bytes / live kernel time against the HBM roof, the fp64 (fp32) vector ceiling beside it, the
    cost of a whole force evaluation (kernel + list build), and the counter-derived figures of the committed PMC passes with their provenance."""
    flop = (FORCE_FLOP_NL if method.endswith("_nl") else FORCE_FLOP)[pot]
    if pot == "lj" and method == "thread_atom" and os.environ.get("COMD_LJ_PRUNE", "1") != "0":
        flop = FORCE_FLOP_LISTED["lj"]
    achieved = FORCE_BYTES[pot] * n_local / (force_ms * 1e-3) / 1e9 if force_ms > 0 else None
    rec, prov = profiled(pot, method, nx) if one_gpu else (None, None)
    vec = "fp64_vector" if precision == "double" else "fp32_vector"
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
           "traffic": (rec["fetch_KiB"] + rec["write_KiB"]) * 1024.0 if rec and precision == "double" else None,
           "traffic_provenance": (f"{prov}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, raw counter values summed over the launches of one force "
                                  "evaluation -- a stored value, not this run's") if rec and precision == "double" else prov,
           "kernel": KERNEL_NAME[(pot, method)], "kernel_ms_per_step": force_ms,
           "force_evaluation_ms": force_ms + aux_ms,      # every launch of one force call: the kernel(s) + list build / cell marks
           "algorithmic_bytes_per_atom": FORCE_BYTES[pot],
           vec: {"achieved_TFLOPs": flop * n_local / (force_ms * 1e-3) / 1e12 if force_ms > 0 else None, "peak_TFLOPs": FP64_VECTOR_PEAK_TFLOPS, "flop_per_atom": flop,
                 "frac": flop * n_local / (force_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS if force_ms > 0 else None,
                 "note": "a FLOP MODEL (constants at the top of bench.py), not a counter"}}
    if rec and precision == "double" and rec.get("insts_valu") and force_ms > 0:
        rate = rec["insts_valu"] / (force_ms * 1e-3)
        out["valu_issue_frac"] = rate / VALU_ISSUE_PEAK
        out["valu_issue_provenance"] = (f"{prov}: SQ_INSTS_VALU of the force kernel(s) per evaluation ({rec['insts_valu']:.4g}) / this run's kernel time, against "
                                        f"{VALU_ISSUE_PEAK:.3g} wave-instructions/s (256 CUs x 4 SIMDs x 2.4 GHz / 4)")
    return out


def main():
    a = parse()
    os.environ["COMD_PRECISION"] = a.precision               # read by the binding when it picks lib*.so / lib*_sp.so: one precision per process
    global FP64_VECTOR_PEAK_TFLOPS
    if a.precision == "single":                                # half the bytes per real_t field (species stays an int), the fp32 vector peak
        FORCE_BYTES.update(lj=28.0, eam=88.0)
        STEP_BYTES.update(lj=28.0 + 112.0, eam=88.0 + 112.0)
        FP64_VECTOR_PEAK_TFLOPS = 157.3
    method = a.method or ("thread_atom" if a.pot == "lj" else "cta_cell")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world
    if a.gpus not in GRIDS:
        sys.exit("supported GPU counts: 1, 2, 4, 8")
    px, py, pz = GRIDS[a.gpus]
    use_async = a.async_halo if a.async_halo is not None else (1 if a.gpus > 1 else 0)

    import ctypes
    # One GPU can rehearse the multi-GPU code path: COMD_LOOPBACK_TRANSPORT=1 sends every halo message and reduction of the
    # single rank through RCCL (to itself), with the same bootstrap as a torch.distributed.run launch.
    loopback = world == 1 and os.environ.get("COMD_LOOPBACK_TRANSPORT", "0") not in ("", "0")
    transport_name = None
    rccl_info = None
    transport = None
    dist = None                                              # torch.distributed: only on the explicit --allow-host-staged path
    pkg = ge.load_package()
    hip = pkg.lib_hip()
    if "COMD_FORCE_DEVICE" in os.environ:               # debugging aid: several ranks on one GPU
        local_rank = int(os.environ["COMD_FORCE_DEVICE"])

    def refuse():
        # a SCALE record must never be produced over host-staged messages by accident
        if rank == 0:
            sys.stderr.write("bench.py: the RCCL communicator could not be formed on every rank; refusing to fall back to host-staged "
                             "gloo messages (pass --allow-host-staged for a functional rehearsal).  The ranks that did reach ncclCommInitRank wait for this one: "
                             "the launcher (torch.distributed.run) ends them when this rank exits non-zero.\n")
        sys.exit(3)

    if world > 1 or loopback:
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        os.environ.setdefault("MASTER_PORT", "29533")
        have_gpu = hip.comdDeviceCount() > local_rank
        if have_gpu:
            pkg.setup_gpu(local_rank, rank, verbose=False)     # stdout carries exactly one line: the JSON
            with stdout_to_stderr():                           # RCCL announces itself on stdout
                transport = pkg.rccl_transport_from_env()
        if transport is not None:
            pkg.init_parallel(rank, world, transport)
            # what the communicator itself says: N ranks, one device each -- and which librccl / libamdhip64 this process runs (ONE of each: no torch
            # in the process, so the copies libcomd_hip.so was linked against, /opt/rocm's)
            n_comm, r_comm, dev = pkg.rccl_comm_info()
            if n_comm != world:
                sys.exit(f"bench.py: the RCCL communicator reports {n_comm} ranks, launched with {world}")
            devs = (ctypes.c_int * world)()
            devs[rank] = dev + 1
            transport.allreduce(transport.ctx, ctypes.cast(devs, ctypes.c_void_p), world, 0)
            libs = {"librccl": pkg.mapped_libraries("librccl"), "libamdhip64": pkg.mapped_libraries("libamdhip64")}
            foreign = ctypes.c_int(0 if all(len(v) == 1 and os.path.realpath(v[0]).startswith("/opt/rocm") for v in libs.values()) else 1)
            transport.allreduce(transport.ctx, ctypes.cast(ctypes.byref(foreign), ctypes.c_void_p), 1, 2)      # max over the ranks
            rccl_info = {"rccl_ranks": n_comm, "rank_devices": [d - 1 for d in devs], "librccl": libs["librccl"], "libamdhip64": libs["libamdhip64"],
                         "one_rocm_runtime_on_every_rank": foreign.value == 0}
            assert foreign.value == 0 or "COMD_ALLOW_FOREIGN_RUNTIME" in os.environ, f"a HIP runtime or RCCL outside /opt/rocm is mapped: {libs}"
        elif not a.allow_host_staged:
            refuse()
        else:                                                # explicit opt-in: host-staged messages over gloo, marked as not measured
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with stdout_to_stderr():
                dist.init_process_group("gloo", rank=rank, world_size=world)
            if not have_gpu:
                sys.exit("bench.py: no HIP device visible (the product path has no CPU fallback)")
            transport_name = "gloo-host-staged (RCCL communicator could not be formed)"
            gloo = pkg.GlooTransport(dist)
            pkg.init_parallel(rank, world, gloo.struct)
    else:
        pkg.setup_gpu(local_rank, rank, verbose=False)
        pkg.init_parallel(0, 1, None)

    def barrier():
        if transport is not None:
            transport.barrier(transport.ctx)
        elif dist is not None:
            dist.barrier()

    def max_over_ranks(seconds):
        if transport is not None:
            # MAX of a non-negative double through the integer MAX reduction: the bit pattern of an IEEE double >= 0 orders as its value does
            # (the transport's dtype 1 is a SUM; two 32-bit halves, high word decided first, would need two rounds)
            hi = ctypes.c_int(int(seconds))                                      # whole seconds
            transport.allreduce(transport.ctx, ctypes.cast(ctypes.byref(hi), ctypes.c_void_p), 1, 2)
            lo = ctypes.c_int(int((seconds - int(seconds)) * 1e9) if int(seconds) == hi.value else -1)      # nanoseconds of the ranks that hold the largest second
            transport.allreduce(transport.ctx, ctypes.cast(ctypes.byref(lo), ctypes.c_void_p), 1, 2)
            return hi.value + lo.value * 1e-9
        if dist is not None:
            import torch
            t = torch.tensor([seconds], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0])
        return seconds

    def measure(pot, meth, steps, warmup, nx=None, mem=False, handshake=False):
        """One Simulation of `pot`/`meth`: W untimed steps, then exactly K steps between barrier + device syncs.
        handshake: exact message sizes are swapped before every exchange (COMD_HALO_HANDSHAKE=1) -- the cross-check of the sized protocol."""
        nx = nx or a.nx
        pairlist = meth == "cta_cell_pairlist"                # the reference's -L: pairlist bits for the CTA-per-cell LJ kernel
        args = ["-x", nx * px, "-y", nx * py, "-z", nx * pz, "-i", px, "-j", py, "-k", pz,
                "-m", "cta_cell" if pairlist else meth, "-a", use_async] + (["-e"] if pot == "eam" else []) + (["-L"] if pairlist else [])
        free0 = pkg.device_mem_info()[0] if mem else 0
        pkg.lib_host().comdSetHaloHandshake(1 if handshake else -1)
        sim = pkg.Simulation(args)
        e0 = sim.energy()
        free1 = pkg.device_mem_info()[0] if mem else 0

        def sync_all():
            hip.comdDeviceSynchronize()
            barrier()
            hip.comdDeviceSynchronize()

        sim.step(warmup)
        sync_all()
        free2 = pkg.device_mem_info()[0] if mem else 0
        sim.force_timing(True)
        builds0 = sim.nl_builds
        t0 = time.perf_counter()
        sim.step(steps)
        sync_all()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        force_ms, n_launch = sim.force_timing_total()
        aux_ms, _ = sim.force_timing_aux()
        sim.force_timing(False)
        ep, ek, n_global = sim.energy()
        sim.sum_atoms()
        assert sim.energy()[2] == n_global, "atoms were lost"
        res = {"elapsed": elapsed, "force_ms": force_ms, "aux_ms": aux_ms, "launches": n_launch, "ep": ep, "ek": ek, "n_global": n_global,
               "cap": sim.max_atoms, "nl_builds": sim.nl_builds - builds0, "e_initial": e0[0] + e0[1], "path": sim.force_path_info(),
               "mem_GB": {"simulation": (free0 - free1) / 1e9, "allocated_by_the_first_steps": (free1 - free2) / 1e9} if mem else None}
        sim.close()
        return res

    m = measure(a.pot, method, a.steps, a.warmup)
    elapsed, force_ms, n_global, ep, ek = m["elapsed"], m["force_ms"], m["n_global"], m["ep"], m["ek"]

    if rank == 0:
        n_local = n_global / a.gpus
        value = n_global * a.steps / elapsed
        roof = roofline_object(a.pot, method, a.nx, n_local, force_ms / a.steps, m["aux_ms"] / a.steps, a.precision, one_gpu=a.gpus == 1)
        roof.update({"launches_timed": m["launches"], "whole_step_achieved_GBs": STEP_BYTES[a.pot] * value / a.gpus / 1e9,
                     "note": "fp64 ALU-bound stencil: ~4000 (LJ; ~2350 after the per-wave box pruning of thread_atom) / ~283 (EAM) candidate pairs per atom against 56 / 176 algorithmic bytes (SURVEY.md 8d)"})
        out = {
            "metric": "atom_updates_per_sec", "value": value, "unit": "atom-updates/s",
            "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if a.precision == "double" else "f32", "data": "synthetic",
            "config": {"workload": f"{a.pot.upper()} Cu FCC {a.nx}^3 unit cells per GPU ({int(n_local)} atoms/GPU, {n_global} total), "
                                   f"{method} kernel, {'fp64' if a.precision == 'double' else 'fp32 (PRECISION=single build)'}, T=600 K, dt=1 fs",
                       "decomposition": f"{px}x{py}x{pz}", "halo_overlap": bool(use_async), "cell_capacity": m["cap"],
                       **({"transport": transport_name or ("rccl-loopback" if loopback else "rccl")} if a.gpus > 1 or loopback else {})},
            "per_gpu_value": value / a.gpus,
            "energy_per_atom_eV": (ep + ek) / n_global,
            "eFinal_over_eInitial": (ep + ek) / m["e_initial"],      # CoMD.c:413-440 "Simulation Validation": energy conservation over warm-up + timed steps
            "force_path": m["path"],                                  # what the device library decided (candidate lists in use, brick image, list format, fall-backs)
            "roofline": roof,
        }
        out["config"]["energy_reductions_timed"] = 1                  # the K steps are ONE timestep() call: one energy reduction (27 us) where the reference's loop does one per printRate = 10 steps
        if method == "thread_atom_nl":
            out["config"]["neighbor_list_builds_timed"] = m["nl_builds"]
        if rccl_info:
            out["config"].update(rccl_info)
        if transport_name:                                  # host-staged rehearsal: not an xGMI measurement
            out["measured"] = False
    # N > 1 (and the one-GPU loopback rehearsal): the halo messages of the timed run were sized WITHOUT a handshake, from the counts both ends hold of the same
    # message one step earlier.  Run the same W + K steps again from the same initial state with exact sizes swapped before every exchange: runs are
    # bit-reproducible, so the two must agree to the last bit -- a mis-paired same-peer message (decomposition.c:57-66: two ranks on an axis make both
    # neighbours of a phase ONE peer) or a truncated one cannot.
    mismatch = False
    if (a.gpus > 1 or loopback) and (transport is not None or dist is not None):
        # (COMD_BENCH_SELFTEST_PERTURB=1, tests only: the second run takes one step more, so that the comparison can be seen to fail)
        c = measure(a.pot, method, a.steps + (1 if os.environ.get("COMD_BENCH_SELFTEST_PERTURB") == "1" else 0), a.warmup, handshake=True)
        pkg.lib_host().comdSetHaloHandshake(-1)
        differs = 0 if (c["ep"], c["ek"], c["n_global"]) == (ep, ek, n_global) else 1
        if transport is not None:
            same = ctypes.c_int(differs)
            transport.allreduce(transport.ctx, ctypes.cast(ctypes.byref(same), ctypes.c_void_p), 1, 2)      # (the energies are global sums: every rank holds the same pair)
            mismatch = same.value != 0
        else:                                                    # the host-staged rehearsal
            import torch
            t = torch.tensor([differs])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mismatch = int(t[0]) != 0
        if rank == 0:
            out["sized_matches_handshake"] = not mismatch
            out["handshake_run"] = {"ms_per_step": 1e3 * c["elapsed"] / a.steps, "energy_per_atom_eV": (c["ep"] + c["ek"]) / c["n_global"],
                                    "dE_pot_eV": c["ep"] - ep, "dE_kin_eV": c["ek"] - ek, "d_atoms": c["n_global"] - n_global}
    # the other force methods on the same workload (one GPU only): not the headline, reported beside it
    variants = []
    if a.gpus == 1 and not a.no_variants:
        for pot, meth in (("lj", "thread_atom"), ("lj", "thread_atom_nl"), ("lj", "cta_cell"), ("lj", "cta_cell_pairlist"), ("eam", "cta_cell"), ("eam", "thread_atom_nl"), ("eam", "thread_atom")):
            if (pot, meth) == (a.pot, method):
                continue
            v = measure(pot, meth, a.steps, a.warmup)
            entry = {"workload": f"{pot.upper()} Cu FCC {a.nx}^3, {meth}", "value": v["n_global"] * a.steps / v["elapsed"],
                     "ms_per_step": 1e3 * v["elapsed"] / a.steps, "force_ms_per_step": v["force_ms"] / a.steps,
                     "force_evaluation_ms": (v["force_ms"] + v["aux_ms"]) / a.steps,
                     "energy_per_atom_eV": (v["ep"] + v["ek"]) / v["n_global"], "cell_capacity": v["cap"],
                     **({"neighbor_list_builds_timed": v["nl_builds"]} if meth.endswith(("_nl", "_pairlist")) else {})}
            if (pot, meth) in (("eam", "cta_cell"), ("lj", "thread_atom")):      # BASELINE.json configs[2] / configs[1]: a roofline object of their own
                entry["baseline_config"] = "configs[2]" if pot == "eam" else "configs[1]"
                entry["roofline"] = roofline_object(pot, meth, a.nx, v["n_global"], v["force_ms"] / a.steps, v["aux_ms"] / a.steps, a.precision)
            elif (pot, meth) == ("eam", "thread_atom"):      # the second EAM kernel shape north_star names: its roofline object too
                entry["roofline"] = roofline_object(pot, meth, a.nx, v["n_global"], v["force_ms"] / a.steps, v["aux_ms"] / a.steps, a.precision)
            variants.append(entry)
    # the single-precision build on the two BASELINE workloads: a child process each (the two builds export the same symbols)
    if a.gpus == 1 and not a.no_variants and a.precision == "double":
        import subprocess
        for pot in ("lj", "eam"):
            cmd = [sys.executable, os.path.abspath(__file__), "--precision", "single", "--pot", pot, "--steps", str(a.steps), "--warmup", str(a.warmup),
                   "--nx", str(a.nx), "--no-variants", "--no-cpu-baseline", "--no-target-line"]
            try:
                child = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                v = json.loads(child.stdout.strip().splitlines()[-1])
                variants.append({"workload": v["config"]["workload"], "dtype": "f32", "value": v["value"], "ms_per_step": v["ms_per_step"],
                                 "force_ms_per_step": v["roofline"]["kernel_ms_per_step"], "energy_per_atom_eV": v["energy_per_atom_eV"],
                                 "cell_capacity": v["config"]["cell_capacity"]})
            except Exception as exc:                          # a variant line must never take the headline down with it
                variants.append({"workload": f"{pot.upper()} Cu FCC {a.nx}^3, PRECISION=single build", "dtype": "f32", "error": repr(exc)[:200]})
    # BASELINE.json's target line: LJ 256^3 on one GPU (north_star: ">= 50 % of the HBM roofline on LJ force at 256^3")
    target = None
    if a.gpus == 1 and not a.no_variants and not a.no_target_line and a.precision == "double" and not loopback:
        free, total = pkg.device_mem_info()
        if free < 60e9:
            target = {"workload": "LJ Cu FCC 256^3, thread_atom", "skipped": f"{free / 1e9:.0f} GB of device memory free, the leg needs ~45 GB (atoms 16 GB, candidate lists 17 GB, packed positions 6 GB) and asks for 60"}
        else:
            try:
                TS = 10
                t = measure("lj", "thread_atom", TS, 2, nx=256, mem=True)
                ms = 1e3 * t["elapsed"] / TS
                roof = roofline_object("lj", "thread_atom", 256, t["n_global"], t["force_ms"] / TS, t["aux_ms"] / TS, "double")
                target = {"workload": f"LJ Cu FCC 256^3 ({t['n_global']} atoms), thread_atom, fp64, {TS} timed steps after 2", "value": t["n_global"] * TS / t["elapsed"],
                          "unit": "atom-updates/s", "ms_per_step": ms, "kernel_ms_per_step": t["force_ms"] / TS, "force_evaluation_ms": (t["force_ms"] + t["aux_ms"]) / TS,
                          "lj_candidate_lists_active": t["path"]["lj_candidate_lists_active"],
                          "hbm_frac": roof["frac"], "fp64_vector_frac": roof["fp64_vector"]["frac"], "energy_per_atom_eV": (t["ep"] + t["ek"]) / t["n_global"],
                          "device_memory_GB": t["mem_GB"], "note": "north_star asks for >= 50 % of the HBM roof on this line; a 5-sigma fp64 stencil is VALU-bound near 7 % (SURVEY.md 0.10, DESIGN.md 3)"}
            except Exception as exc:
                target = {"workload": "LJ Cu FCC 256^3, thread_atom", "error": repr(exc)[:300]}
    if rank == 0:
        if variants:
            out["variants"] = variants
        if target:
            out["target_line"] = target
        if a.gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.pot, a.cpu_seconds)
        print(json.dumps(out))
    if transport is not None:
        barrier()
        hip.comdCommFinalize()
    elif dist is not None:
        dist.barrier()
        hip.comdCommFinalize()
        dist.destroy_process_group()
        # the rehearsal path has torch in the process, i.e. a second HIP runtime beside the one libcomd_hip.so links: their exit handlers free each other's state
        # ("double free or corruption" after the line has been printed).  Leave without running them.
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(4 if mismatch else 0)
    if mismatch:
        sys.stderr.write("bench.py: the run with sized halo messages and the run with the size handshake DISAGREE (sized_matches_handshake: false): "
                         "the line above is not a valid measurement\n")
        sys.exit(4)


if __name__ == "__main__":
    main()
