#!/usr/bin/env python3
"""Model: an LJ thread_atom wave whose 64 atoms are split into two (or four) spatially separate lane groups, each with a bounding box of its own.
A candidate that is within the cutoff of only ONE group's box can share an iteration with a candidate that concerns only the OTHER group
(lanes 0-31 take candidate a, lanes 32-63 candidate b: the three subtractions are issued once per group under an exec mask, the other ~13 instructions once).
Measured on the same lattice sample as lj_8x8_model.py: how many iterations that saves, and what the list build would have to test."""
import numpy as np
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location("m", os.path.join(os.path.dirname(__file__), "lj_8x8_model.py"))
# reuse the lattice of the 8x8 model without running its report
src = open(spec.origin).read().split("res = {")[0]
ns = {}
exec(src, ns)
cell_atoms, S, RC = ns["cell_atoms"], ns["S"], ns["RC"]

def near(box_atoms, stencil):
    lo, hi = box_atoms.min(0), box_atoms.max(0)
    d = np.maximum(0, np.maximum(lo - stencil, stencil - hi))
    return (d * d).sum(1) <= RC * RC

T, E, EXTRA = 7, 12, 3          # test, evaluation, extra instructions of a shared iteration (three subtractions issued per group instead of once)
tot = dict(atoms=0, today=0, today_eval=0, both=0, both_eval=0, zipped=0, zipped_eval=0, only=0, tests=0)
tot4 = dict(all4=0, rest=0)
for cx in range(1, S - 1):
    for cy in range(1, S - 1):
        for cz in range(1, S - 1):
            own = cell_atoms(cx, cy, cz)
            stencil = np.concatenate([cell_atoms(cx + a, cy + b, cz + c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)])
            for w0 in range(0, len(own), 64):
                g = own[w0:w0 + 64]
                if len(g) < 64:
                    continue                      # the tail wave keeps its replicas
                tot["atoms"] += 64
                k = near(g, stencil)
                cand = stencil[k]
                d2 = ((g[:, None, :] - cand[None, :, :]) ** 2).sum(-1)
                hit = (d2 <= RC * RC) & (d2 > 0)
                tot["today"] += len(cand); tot["today_eval"] += int(hit.any(0).sum())
                # halves by the median of the widest axis of the wave's box
                ax = int(np.argmax(g.max(0) - g.min(0)))
                order = np.argsort(g[:, ax]) if os.environ.get("SPLIT", "axis") == "axis" else np.arange(64); A, B = order[:32], order[32:]
                kA, kB = near(g[A], cand), near(g[B], cand)
                both = kA & kB; oa = kA & ~kB; ob = kB & ~kA
                nz = min(oa.sum(), ob.sum()); rest = max(oa.sum(), ob.sum()) - nz
                tot["both"] += int(both.sum()); tot["both_eval"] += int(hit[:, both].any(0).sum())
                # shared iterations: pair the i-th A-only with the i-th B-only candidate
                ia, ib = np.flatnonzero(oa), np.flatnonzero(ob)
                ev = 0
                for t in range(nz):
                    ev += bool(hit[A, ia[t]].any() or hit[B, ib[t]].any())
                tot["zipped"] += int(nz); tot["zipped_eval"] += ev
                left = ia[nz:] if len(ia) > nz else ib[nz:]
                tot["only"] += int(rest); tot["zipped_eval"] += 0
                tot["both_eval"] += int(hit[:, left].any(0).sum())      # unpaired one-sided candidates run as ordinary iterations
                tot["tests"] += 3 * len(stencil)
n = tot["atoms"]
today = (tot["today"] * T + tot["today_eval"] * E) / n
new = ((tot["both"] + tot["only"]) * T + tot["both_eval"] * E + tot["zipped"] * (T + EXTRA) + tot["zipped_eval"] * E) / n
print(f"full waves sampled: {n // 64}")
print(f"today : {tot['today'] / (n / 64):.0f} candidates per wave, {tot['today_eval'] / (n / 64):.0f} evaluated: {today:.0f} wave-instructions per atom")
print(f"halves: {tot['both'] / (n / 64):.0f} concern both halves, {tot['zipped'] / (n / 64):.0f} shared iterations (two one-sided candidates each), {tot['only'] / (n / 64):.0f} unpaired;")
print(f"        {new:.0f} wave-instructions per atom ({100 * (1 - new / today):.1f} % fewer); the list build tests every stencil atom against 3 boxes per wave instead of 1")
