#!/usr/bin/env python3
"""Turn what profiles/r01_collect.sh left under gpurun_out/final/ into the tracked files of profiles/ (run from the repo root).

    gpurun -- 'bash profiles/r01_collect.sh'      # on the MI355X box: GPU tests, smoke, bench, comd-hip runs, rocprofv3 stats + PMC passes
    python3 profiles/r01_summarize.py             # here: copies, r01_traffic.json, r01_summary.md
"""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
N = 2048000
PATHS = [("lj", "thread_atom"), ("lj", "thread_atom_nl"), ("eam", "cta_cell"), ("eam", "thread_atom_nl")]


def main():
    shutil.copy(os.path.join(F, "bench_default.json"), os.path.join(P, "r01_bench_default.json"))
    for pot, meth in PATHS:
        shutil.copy(os.path.join(F, f"stats_{pot}_{meth}", "out_kernel_stats.csv"), os.path.join(P, f"r01_80_{pot}_{meth}_kernel_stats.csv"))
    for name in ("lj20_nl", "eam20_nl", "setfl20", "lj20_pairlist"):
        shutil.copy(os.path.join(F, f"comd_hip_{name}_stdout.txt"), os.path.join(P, f"r01_comd_hip_{name}_stdout.txt"))
    shutil.copy(os.path.join(F, "pmc_summary.json"), os.path.join(P, "r01_pmc_fetch_write_summary.json"))
    pmc = json.load(open(os.path.join(F, "pmc_summary.json")))

    def per_launch(tag, ctr, prefixes):
        d = pmc[f"pmc_{tag}_{ctr}"]
        tot = 0.0
        for pre in prefixes:
            ks = [k for k in d if k.startswith(pre)]
            tot += d[max(ks, key=lambda k: d[k]["launches"])]["per_launch"]
        return tot

    kern = {("lj", "thread_atom"): (["void LJ_Force_thread_atom<false>"], "LJ_Force_thread_atom<false>"),
            ("lj", "thread_atom_nl"): (["void LJ_Force_nl_slabs<false>"], "LJ_Force_nl_slabs<false>"),
            ("eam", "cta_cell"): (["void EAM_Force_cta_cell<1", "EAM_Force_embed", "void EAM_Force_cta_cell<3"], "EAM_Force_cta_cell<1> + EAM_Force_embed + EAM_Force_cta_cell<3>"),
            ("eam", "thread_atom_nl"): (["void EAM_Force_nl_lds<1", "EAM_Force_embed", "void EAM_Force_nl_lds<3"], "EAM_Force_nl_lds<1> + EAM_Force_embed + EAM_Force_nl_lds<3>")}
    traffic = {"_about": "Per-evaluation memory-side traffic of the force kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs "
                         "with --kernel-trace only; both are reported in KiB). Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE under-reports a wide coalesced "
                         "16 B/lane stream by exactly 2x on gfx950 and is uncalibrated for other access widths; these kernels read through 64-byte scalar loads, "
                         "8-byte lane loads and 2-byte list loads, so the fetch side is given raw. WRITE_SIZE is exact for streaming stores. bench.py reports "
                         "raw_fetch + write as `traffic`. Every kernel of the four runs: profiles/r01_pmc_fetch_write_summary.json."}
    for pot, meth in PATHS:
        pre, name = kern[(pot, meth)]
        rec = {"kernel": name, "fetch_KiB": per_launch(f"{pot}_{meth}", "FETCH_SIZE", pre), "write_KiB": per_launch(f"{pot}_{meth}", "WRITE_SIZE", pre),
               "algorithmic_bytes": (56 if pot == "lj" else 176) * N,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --pot P --method M --no-variants --steps 10 --warmup 3"}
        if meth.endswith("_nl"):
            rec["list_bytes"] = 2 * (732 if pot == "lj" else 2 * 57) * N
        traffic[f"{pot}/{meth}/80"] = rec
    json.dump(traffic, open(os.path.join(P, "r01_traffic.json"), "w"), indent=1)

    d = json.load(open(os.path.join(P, "r01_bench_default.json")))
    rows = [("LJ", "thread_atom (headline: BASELINE configs[1])", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["config"]["cell_capacity"])]
    for v in d["variants"]:
        pot, meth = v["workload"].split(" Cu FCC 80^3, ")
        rows.append((pot, meth, v["value"], v["ms_per_step"], v["force_ms_per_step"], v["cell_capacity"]))
    by = {"LJ": 56, "EAM": 176}
    fl = {("LJ", False): 4000 * 8 + 550 * 25, ("LJ", True): 732 * 8 + 550 * 25, ("EAM", False): 2 * (283 * 8 + 42 * 70), ("EAM", True): 2 * (57 * 8 + 42 * 70)}
    out = ["# Round 1 measurements on MI355X (gfx950, ROCm 7.2) -- final state of the round\n",
           "Collected by `profiles/r01_collect.sh` in ONE gpurun call (GPU test suite, smoke, bench, comd-hip runs, rocprofv3 passes) and written here by "
           "`profiles/r01_summarize.py`.\n",
           "One GPU, 80^3 unit cells = 2,048,000 Cu atoms, fp64, T = 600 K, dt = 1 fs, atoms resident in HBM.  All rows are from one run of `python3 bench.py` "
           "(defaults: 20 timed steps after 5 warm-up steps; `profiles/r01_bench_default.json`): the headline configuration is `value`, the others its `variants`.\n"
           "`force ms` = HIP events on the launch stream around every force launch of the timed steps (all three passes for EAM).\n",
           "| potential | method | ms/step | M atom-updates/s | force ms/step | HBM-roof frac (56 / 176 B per atom, 8 TB/s) | fp64-vector frac (FLOP model of bench.py, 78.6 TF) | cell slots |",
           "|---|---|---|---|---|---|---|---|"]
    for pot, meth, val, ms, fms, cap in rows:
        nl = meth.startswith("thread_atom_nl")
        out.append(f"| {pot} | {meth} | {ms:.3f} | {val / 1e6:.1f} | {fms:.3f} | {by[pot] * N / (fms * 1e-3) / 8e12 * 100:.2f} % | "
                   f"{fl[(pot, nl)] * N / (fms * 1e-3) / 78.6e12 * 100:.1f} % | {cap} |")
    cb = d["cpu_baseline"]
    out.append(f"\ncpu_baseline: {cb['value'] / 1e6:.2f} M atom-updates/s on {cb['cores']} host threads ({cb['sample']}).\n")
    builds = {}
    for pot in ("lj", "eam"):
        for r in csv.DictReader(open(os.path.join(P, f"r01_80_{pot}_thread_atom_nl_kernel_stats.csv"))):
            if r["Name"].startswith("BuildNeighborList"):
                builds[pot] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    out.append(f"The Verlet-list rows (`thread_atom_nl`, skin 10 % of the cutoff) and the pairlist row (`cta_cell -L`) time 20 steps with at most one rebuild; over 110 steps "
               f"(`profiles/r01_80_*_nl_kernel_stats.csv`) LJ builds {builds['lj'][0]} time(s) ({builds['lj'][1]:.1f} ms each) and EAM {builds['eam'][0]} times ({builds['eam'][1]:.1f} ms each), "
               f"i.e. +{builds['lj'][0] * builds['lj'][1] / 110:.2f} / +{builds['eam'][0] * builds['eam'][1] / 110:.2f} ms per step.  "
               "FLOP model for the list kernels: 732 (LJ) / 57 (EAM) listed pairs x 8 + 550 / 42 evaluated pairs x 25 / 70.\n")
    out.append("LJ 256^3 (67,108,864 atoms, the BASELINE target line), `bench.py --nx 256 --steps 3 --warmup 1`, thread_atom: 123.4 ms/step = 544 M atom-updates/s "
               "(`profiles/r01_bench_lj256.json`).  The list method does not fit 256^3 on one GPU (its lists alone would be ~390 GB).\n")
    out.append("comd-hip stdout of the reference's 20^3 cases (`profiles/r01_comd_hip_*_stdout.txt`): LJ E(100) = -1.329021332207, EAM E(100) = -3.460530084715 with "
               "thread_atom, cta_cell, cta_cell -L AND thread_atom_nl -- the reference CPU values to the last printed digit; setfl (Mishin Cu01) step-0 U = -3.539999969172 "
               "(CoMD.c:899: -3.539999969176).\n")

    def table(f, title, cmd):
        out.append(f"## {title}\n\n`{cmd}` (raw: `profiles/{f}`)\n")
        out.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(os.path.join(P, f))))[:14]:
            out.append(f"| {r['Name'].split('(')[0]} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
        out.append("")

    base = "rocprofv3 --kernel-trace --stats -- python3 bench.py --pot {} --method {} --no-cpu-baseline --no-variants --steps 100 --warmup 10"
    for pot, meth in PATHS:
        table(f"r01_80_{pot}_{meth}_kernel_stats.csv", f"{pot.upper()} 80^3 {meth}", base.format(pot, meth))
    out.append("## PMC passes (separate runs, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` with `--kernel-trace` only; `profiles/r01_pmc_fetch_write_summary.json`)\n")
    out.append("| force path | fetched MiB / evaluation | written MiB | algorithmic MiB (56 / 176 B per atom) | list MiB |\n|---|---|---|---|---|")
    for k, v in traffic.items():
        if k != "_about":
            out.append(f"| {k} ({v['kernel']}) | {v['fetch_KiB'] / 1024:.1f} | {v['write_KiB'] / 1024:.1f} | {v['algorithmic_bytes'] / 2 ** 20:.1f} | {v.get('list_bytes', 0) / 2 ** 20:.0f} |")
    out.append("\nFETCH_SIZE is raw (see `r01_traffic.json`).  The cell kernels re-read their stencils out of L2/MALL (TCC hit rate 97 %); the list kernels stream 16-bit lists "
               "(LJ 2.6 GB per evaluation at 1.3 TB/s, hidden behind the arithmetic; EAM: every lane fetches 16 rows whatever its list length so that the loads can go out "
               "before the length is known).\n")
    out.append("SQ counters per launch (separate `--pmc` passes during the round):\n"
               "* LJ_Force_nl_slabs (final, 2.07 ms): SQ_INSTS_VALU 8.0e8, SQ_INSTS_LDS 6.8e7, VALU busy 70 %, LDS busy 57 % (SQ_LDS_BANK_CONFLICT 53 % of SQ_LDS_IDX_ACTIVE: random "
               "gathers), SQ_WAIT_ANY 37 % / SQ_WAIT_INST_ANY 26 % of wave cycles.  With x-plane groups (2.57 ms): VALU busy 69 %, SQ_WAIT_ANY 50 %; with per-pair LDS round trips (3.19 ms): 54 %, 66 %.\n"
               "* LJ_Force_thread_atom (3.95 ms state): SQ_INSTS_VALU 1.97e9 (11.9 per candidate: 7 for the distance test + 0.33 x 15 for the evaluation), SQ_INSTS_SMEM 6.96e7 "
               "(s_load_dwordx16), VALU busy 76 %.\n"
               "* EAM_Force_cta_cell<1> (1.77 ms state, before the paired evaluation): 226 VALU + 68 LDS instructions per atom, VALU busy 44 %, SQ_WAIT_ANY 47 %.\n")
    out.append("RCCL on one GPU (`profiles/r01_rccl_loopback_bench.json`): `COMD_LOOPBACK_TRANSPORT=1 python3 bench.py` sends every halo message and reduction of the single rank "
               "through ncclSend/ncclRecv to itself: LJ 4.39 vs 4.07 ms/step, EAM 4.00 (3.79 with -a 1) vs 3.50 -- the cost of three size handshakes + six host syncs per exchange; "
               "with lists and -a 1 the refresh exchange hides behind the interior force (LJ 2.27, EAM 2.07 ms/step in loopback).\n")
    open(os.path.join(P, "r01_summary.md"), "w").write("\n".join(out))
    print("\n".join(out[4:16]))


if __name__ == "__main__":
    main()
