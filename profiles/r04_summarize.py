#!/usr/bin/env python3
"""Turn what profiles/r04_collect.sh left under gpurun_out/r04final/ into the tracked files of profiles/ (run from the repo root).

    gpurun --timeout 1190 -- 'bash profiles/r04_collect.sh'     # on the MI355X box
    python3 profiles/r04_summarize.py                            # here: copies, r04_traffic.json, r04_summary.md
"""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "r04final")
P = os.path.join(ROOT, "profiles")
N = 2048000
PATHS = [("lj", "thread_atom"), ("lj", "thread_atom_nl"), ("lj", "cta_cell"), ("eam", "cta_cell"), ("eam", "thread_atom_nl"), ("eam", "thread_atom")]
KERN = {("lj", "thread_atom"): ["LJ_Force_thread_atom<false, true>"], ("lj", "thread_atom_nl"): ["LJ_Force_nl_slabs<false>"], ("lj", "cta_cell"): ["LJ_Force_cta_cell_boxes<false>"],
        ("eam", "cta_cell"): ["EAM_Force_cta_brick<1", "EAM_Force_cta_brick<3"], ("eam", "thread_atom_nl"): ["EAM_Force_cta_brick<1", "EAM_Force_cta_brick<3"],
        ("eam", "thread_atom"): ["EAM_Force_atom_brick<1", "EAM_Force_atom_brick<3"]}


def stats_file(pot, meth):
    g = glob.glob(os.path.join(F, f"stats_{pot}_{meth}", "**", "out_kernel_stats.csv"), recursive=True)
    return g[0] if g else None


def main():
    shutil.copy(os.path.join(F, "bench_default.json"), os.path.join(P, "r04_bench_default.json"))
    for pot, meth in PATHS:
        f = stats_file(pot, meth)
        if f:
            shutil.copy(f, os.path.join(P, f"r04_80_{pot}_{meth}_kernel_stats.csv"))
    g = glob.glob(os.path.join(F, "stats_loopback_lj", "**", "out_kernel_stats.csv"), recursive=True)
    if g:
        shutil.copy(g[0], os.path.join(P, "r04_80_lj_thread_atom_rccl_loopback_kernel_stats.csv"))
    for name in ("lj20", "lj20_device_timers", "lj40_device_timers", "eam20", "eam40_device_timers", "eam20_thread_atom", "lj20_nl", "eam20_nl", "setfl20", "lj20_pairlist", "sp_lj20", "sp_eam20"):
        shutil.copy(os.path.join(F, f"comd_hip_{name}_stdout.txt"), os.path.join(P, f"r04_comd_hip_{name}_stdout.txt"))
    shutil.copy(os.path.join(F, "pmc_summary.json"), os.path.join(P, "r04_pmc_summary.json"))
    loop = {}
    for tag in ("plain_lj", "loopback_lj", "loopback_lj_handshake", "loopback_lj_overlap", "plain_eam", "loopback_eam", "loopback_eam_handshake", "loopback_eam_overlap",
                "plain_lj_messages", "plain_eam_messages", "plain_eam_nl", "plain_eam_nl_messages", "plain_eam_nl_round3_kernel", "loopback_eam_nl_overlap",
                "plain_eam_thread_atom", "plain_eam_thread_atom_round2_kernel", "plain_eam_thread_atom_no_handover", "loopback_eam_thread_atom_overlap"):
        try:
            d = json.loads(open(os.path.join(F, tag + ".json")).read().strip().splitlines()[-1])
        except (OSError, IndexError, ValueError):
            continue
        loop[tag] = {"ms_per_step": d["ms_per_step"], "force_ms_per_step": d["roofline"]["kernel_ms_per_step"], "force_evaluation_ms": d["roofline"]["force_evaluation_ms"],
                     "value": d["value"], "config": d["config"],
                     **{k: d[k] for k in ("sized_matches_handshake", "handshake_run", "eFinal_over_eInitial", "force_path") if k in d}}
    loop["_about"] = ("One GPU, 80^3.  plain: no transport (a rank that is its own neighbour unpacks straight from its send buffers).  loopback: COMD_LOOPBACK_TRANSPORT=1, every halo "
                      "message and reduction goes through a one-rank RCCL communicator (ncclSend/ncclRecv to itself) with the sized protocol (no handshake, no host sync); "
                      "loopback_*_handshake: COMD_HALO_HANDSHAKE=1, the exact-size handshake of round 1 (three host syncs per exchange); loopback_*_overlap: the sized protocol with -a 1 "
                      "(interior cells on a second stream during the exchange: the mode bench.py runs for N > 1).  [round 4] every loopback line carries the self-check of the sized "
                      "protocol (sized_matches_handshake: the same steps again with the handshake, bit for bit); plain_*_messages: COMD_HALO_MIRROR=0, the pack / unpack path of round 3 on the "
                      "self-neighbour axes instead of the direct mirror; plain_eam_nl_round3_kernel: COMD_EAM_NL=lds, round 3's list kernel; plain_eam_thread_atom*: -m thread_atom -e on the brick image "
                      "(eam_atom_brick_kernels.h), with COMD_EAM_THREAD_ATOM=cell round 2's kernel, with COMD_EAM_ATOM_HANDOVER=0 pass 3 testing again.")
    json.dump(loop, open(os.path.join(P, "r04_rccl_loopback_bench.json"), "w"), indent=1)

    pmc = json.load(open(os.path.join(F, "pmc_summary.json")))
    hashes = json.load(open(os.path.join(F, "kernel_source_hashes.json")))      # of the sources that were on the GPU box when the counters were read

    def per_launch(tag, ctr, prefixes):
        d = pmc.get(f"pmc_{tag}_{ctr}", {})
        tot = 0.0
        for pre in prefixes:
            ks = [k for k in d if k.startswith(pre) and ctr in d[k]]
            if not ks:
                return None
            tot += d[max(ks, key=lambda k: d[k][ctr]["launches"])][ctr]["per_launch"]
        return tot

    traffic = {"_about": "Per-evaluation memory-side traffic of the force kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs with "
                         "--kernel-trace only; both in KiB).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE under-reports a wide coalesced 16 B/lane stream by exactly 2x on gfx950 "
                         "and is uncalibrated for other widths; these kernels read through 64-byte scalar loads, 8-byte lane loads and 2-byte list loads, so the fetch side is given raw. "
                         "WRITE_SIZE is exact for streaming stores.  bench.py reports raw_fetch + write as `traffic`.  Every force kernel of the runs: profiles/r04_pmc_summary.json."}
    for pot, meth in [("lj", "thread_atom"), ("lj", "thread_atom_nl"), ("eam", "cta_cell"), ("eam", "thread_atom_nl"), ("eam", "thread_atom")]:
        pre = KERN[(pot, meth)]
        fe, wr = per_launch(f"{pot}_{meth}", "FETCH_SIZE", pre), per_launch(f"{pot}_{meth}", "WRITE_SIZE", pre)
        if fe is None or wr is None:
            continue
        iv = None
        if f"pmc_{pot}_{meth}_SQ1" in pmc:
            dd = pmc[f"pmc_{pot}_{meth}_SQ1"]
            iv = 0.0
            for pr in pre:
                ks = [k for k in dd if k.startswith(pr) and "SQ_INSTS_VALU" in dd[k]]
                iv = iv + dd[max(ks, key=lambda k: dd[k]["SQ_INSTS_VALU"]["launches"])]["SQ_INSTS_VALU"]["per_launch"] if ks and iv is not None else None
        traffic[f"{pot}/{meth}/80"] = {"kernel": " + ".join(pre), "fetch_KiB": fe, "write_KiB": wr, "algorithmic_bytes": (56 if pot == "lj" else 176) * N,
                                       "kernel_source_sha16": hashes.get(f"{pot}/{meth}"),
                                       **({"insts_valu": iv} if iv else {}),
                                       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --pot P --method M --no-variants --steps 10 --warmup 3"}
    json.dump(traffic, open(os.path.join(P, "r04_traffic.json"), "w"), indent=1)

    d = json.load(open(os.path.join(P, "r04_bench_default.json")))
    out = ["# Round 4 measurements on MI355X (gfx950, ROCm 7.2) -- final state of the round\n",
           "Collected by `profiles/r04_collect.sh` in ONE gpurun call (GPU test suite, smoke, bench, comd-hip runs, rocprofv3 passes) and written here by `profiles/r04_summarize.py`.\n",
           "One GPU, 80^3 unit cells = 2,048,000 Cu atoms, T = 600 K, dt = 1 fs, atoms resident in HBM.  All rows are from one run of `python3 bench.py` "
           "(20 timed steps after 5 warm-up steps; `profiles/r04_bench_default.json`): the headline configuration is `value`, the others its `variants`.  "
           "`force ms` = HIP events on the launch stream around every force launch of the timed steps.\n",
           "| workload | dtype | ms/step | M atom-updates/s | force ms/step | cell slots | round 3 ms/step |", "|---|---|---|---|---|---|---|"]
    r1 = {"LJ thread_atom": 3.11, "LJ thread_atom_nl": 2.04, "LJ cta_cell": 3.14, "LJ cta_cell_pairlist": 4.72, "EAM cta_cell": 1.45, "EAM thread_atom_nl": 1.96, "EAM thread_atom": 3.69}      # round 3 (profiles/r03_summary.md)
    rows = [(f"LJ Cu FCC 80^3, thread_atom (headline: BASELINE configs[1])", "f64", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_per_step"], d["config"]["cell_capacity"], r1["LJ thread_atom"])]
    for v in d.get("variants", []):
        if "error" in v:
            continue
        key = v["workload"].replace(" Cu FCC 80^3,", "").strip()
        rows.append((v["workload"], v.get("dtype", "f64"), v["ms_per_step"], v["value"], v["force_ms_per_step"], v.get("cell_capacity", ""), r1.get(key, "")))
    for w, dt, ms, val, fms, cap, old in rows:
        out.append(f"| {w} | {dt} | {ms:.3f} | {val / 1e6:.1f} | {fms:.3f} | {cap} | {old} |")
    rf = d["roofline"]
    out.append(f"\nHeadline roofline object: kernel `{rf['kernel']}`, {rf['kernel_ms_per_step']:.3f} ms per launch, achieved {rf['achieved']:.1f} GB/s of {rf['peak']:.0f} "
               f"(frac {rf['frac']:.4f}); fp64 vector model {rf['fp64_vector']['achieved_TFLOPs']:.1f} of {rf['fp64_vector']['peak_TFLOPs']} TFLOP/s (frac {rf['fp64_vector']['frac']:.2f}); "
               f"traffic {rf['traffic'] and rf['traffic'] / 2 ** 20 or 0:.1f} MiB per launch (raw FETCH_SIZE + WRITE_SIZE) vs {56 * N / 2 ** 20:.1f} MiB algorithmic.\n")
    cb = d["cpu_baseline"]
    out.append(f"cpu_baseline (oracle, kind port): {cb['value'] / 1e6:.3f} M atom-updates/s on {cb['cores']} core ({cb['sample']}); "
               f"the GPU's share of the host cores: {cb['all_cores_of_this_gpu']['value'] / 1e6:.2f} M atom-updates/s on {cb['all_cores_of_this_gpu']['cores']} threads ({cb['all_cores_of_this_gpu']['sample']}).\n")
    lp = loop
    out.append("RCCL on one GPU (`profiles/r04_rccl_loopback_bench.json`): ms/step plain / loopback with the sized protocol / loopback with round 1's handshake: "
               f"LJ {lp['plain_lj']['ms_per_step']:.3f} / {lp['loopback_lj']['ms_per_step']:.3f} / {lp['loopback_lj_handshake']['ms_per_step']:.3f}; "
               f"EAM {lp['plain_eam']['ms_per_step']:.3f} / {lp['loopback_eam']['ms_per_step']:.3f} / {lp['loopback_eam_handshake']['ms_per_step']:.3f}."
               + (f"  With the overlap mode (`-a 1`: what `bench.py --gpus N` runs for N > 1) through the loopback: LJ {lp['loopback_lj_overlap']['ms_per_step']:.3f}, EAM {lp['loopback_eam_overlap']['ms_per_step']:.3f}."
                  if "loopback_lj_overlap" in lp and "loopback_eam_overlap" in lp else "") + "\n")
    tl = d.get("target_line")
    if tl and "value" in tl:
        out.append(f"LJ 256^3 (the BASELINE target line; `target_line` of the same bench run): {tl['ms_per_step']:.1f} ms/step = {tl['value'] / 1e6:.0f} M atom-updates/s, force kernel "
                   f"{tl['kernel_ms_per_step']:.1f} ms, whole force evaluation {tl['force_evaluation_ms']:.1f} ms, HBM frac {tl['hbm_frac']:.4f}, fp64 model frac {tl['fp64_vector_frac']:.2f}, "
                   f"device memory {tl['device_memory_GB']}.\n")
    for v in d.get("variants", []):
        if "roofline" in v:
            rr = v["roofline"]
            out.append(f"{v['workload']} ({v.get('baseline_config', '')}) roofline object: {rr['kernel_ms_per_step']:.3f} ms per force evaluation, achieved {rr['achieved']:.0f} GB/s "
                       f"(frac {rr['frac']:.4f}), fp64 model frac {rr['fp64_vector']['frac']:.2f}" + (f", VALU issue frac {rr['valu_issue_frac']:.2f}" if 'valu_issue_frac' in rr else "") + ".\n")

    def table(fn, title, cmd):
        out.append(f"## {title}\n\n`{cmd}` (raw: `profiles/{fn}`)\n")
        out.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(os.path.join(P, fn))))[:12]:
            out.append(f"| {r['Name'].split('(')[0]} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
        out.append("")

    base = "rocprofv3 --kernel-trace --stats -- python3 bench.py --pot {} --method {} --no-cpu-baseline --no-variants --steps 100 --warmup 10"
    for pot, meth in PATHS:
        fn = f"r04_80_{pot}_{meth}_kernel_stats.csv"
        if os.path.exists(os.path.join(P, fn)):
            table(fn, f"{pot.upper()} 80^3 {meth}", base.format(pot, meth))
    fn = "r04_80_lj_thread_atom_rccl_loopback_kernel_stats.csv"
    if os.path.exists(os.path.join(P, fn)):
        table(fn, "LJ 80^3 thread_atom through the RCCL loopback (ncclDevKernel rows = the three grouped send/recv per exchange)",
              "COMD_LOOPBACK_TRANSPORT=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants --steps 50 --warmup 5")
    out.append("## PMC passes (separate runs, `--pmc ...` with `--kernel-trace` only; per-launch averages in `profiles/r04_pmc_summary.json`)\n")
    out.append("| force path | fetched MiB / evaluation | written MiB | algorithmic MiB |\n|---|---|---|---|")
    for k, v in traffic.items():
        if k != "_about":
            out.append(f"| {k} ({v['kernel']}) | {v['fetch_KiB'] / 1024:.1f} | {v['write_KiB'] / 1024:.1f} | {v['algorithmic_bytes'] / 2 ** 20:.1f} |")
    out.append("")
    for tag, names in (("lj_thread_atom", ["LJ_Force_thread_atom<false, true>"]), ("lj_thread_atom_nl", ["LJ_Force_nl_slabs<false>"]),
                       ("eam_cta_cell", ["EAM_Force_cta_brick<1, true, false, false>", "EAM_Force_cta_brick<3, true, false, false>"]),
                       ("eam_thread_atom_nl", ["EAM_Force_cta_brick<1, true, false, true>", "EAM_Force_cta_brick<3, true, false, true>", "EAM_Force_cta_brick<0"]),
                       ("eam_thread_atom", ["EAM_Force_atom_brick<1", "EAM_Force_atom_brick<3"])):
        for nm in names:
            sq = {}
            for grp in ("SQ1", "SQ2", "TCC"):
                for k, v in pmc.get(f"pmc_{tag}_{grp}", {}).items():
                    if k.startswith(nm):
                        sq.update({c: x["per_launch"] for c, x in v.items()})
            if sq:
                wc = sq.get("SQ_WAVE_CYCLES", 0)
                line = f"* `{nm}` per launch: " + ", ".join(f"{c} {x:.3g}" for c, x in sorted(sq.items()))
                if wc:
                    line += f" -- VALU instructions per atom {sq.get('SQ_INSTS_VALU', 0) / N:.0f}; wave cycles spent waiting (SQ_WAIT_ANY) {100 * sq.get('SQ_WAIT_ANY', 0) / wc:.0f} %"
                if sq.get("TCC_HIT_sum"):
                    line += f"; L2 hit rate {100 * sq['TCC_HIT_sum'] / (sq['TCC_HIT_sum'] + sq['TCC_MISS_sum']):.1f} %"
                out.append(line)
    out.append("")
    # non-force time per step and what the direct mirror of the self-neighbour axes bought (one rank: all three axes)
    out.append("## Step time outside the force evaluation (ms/step - force_evaluation_ms), one rank\n")
    out.append("| run | ms/step | force evaluation ms | non-force ms |\n|---|---|---|---|")
    for tag in ("plain_lj", "plain_lj_messages", "plain_eam", "plain_eam_messages", "plain_eam_nl", "plain_eam_nl_messages", "plain_eam_nl_round3_kernel",
                "plain_eam_thread_atom", "plain_eam_thread_atom_round2_kernel", "plain_eam_thread_atom_no_handover"):
        if tag in loop:
            v = loop[tag]
            out.append(f"| {tag} | {v['ms_per_step']:.3f} | {v['force_evaluation_ms']:.3f} | {v['ms_per_step'] - v['force_evaluation_ms']:.3f} |")
    out.append("")
    for tag in ("loopback_lj", "loopback_eam", "loopback_lj_overlap", "loopback_eam_overlap", "loopback_eam_nl_overlap", "loopback_eam_thread_atom_overlap"):
        if tag in loop and "sized_matches_handshake" in loop[tag]:
            out.append(f"* `{tag}`: sized_matches_handshake = {loop[tag]['sized_matches_handshake']}, eFinal/eInitial = {loop[tag].get('eFinal_over_eInitial')}")
    out.append("")
    open(os.path.join(P, "r04_summary.md"), "w").write("\n".join(out))
    print("\n".join(out[4:20]))


def current_hashes():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    return {f"{pot}/{meth}": bench.kernel_source_hash(pot, meth) for pot, meth in bench.KERNEL_SOURCES}


def check():
    """Every force path's NEWEST stored PMC record (profiles/rNN_traffic.json) must have been collected for the kernel source that is in the tree now: bench.py divides
    these stored counts by the live kernel time (`traffic`, `valu_issue_frac`), and a record of an older kernel would be a wrong number that looks measured."""
    cur = current_hashes()
    stale = []
    for key, h in cur.items():
        for name in sorted(glob.glob(os.path.join(P, "r*_traffic.json")), reverse=True):
            rec = json.load(open(name)).get(key + "/80")
            if rec:
                if rec.get("kernel_source_sha16") != h:
                    stale.append(f"{key}: {os.path.basename(name)} was collected for source {rec.get('kernel_source_sha16')}, the tree holds {h}")
                break
    return stale


if __name__ == "__main__":
    import sys
    if "--hashes" in sys.argv:
        print(json.dumps(current_hashes()))
    elif "--check" in sys.argv:
        bad = check()
        print("\n".join(bad) if bad else "every stored PMC record matches the kernel source in the tree")
        sys.exit(1 if bad else 0)
    else:
        main()
        bad = check()
        if bad:
            sys.exit("STALE PMC RECORDS (re-run profiles/r04_collect.sh 2):\n" + "\n".join(bad))
