#!/usr/bin/env python3
"""Model of the 'eight atoms x eight candidate streams per wave' LJ kernel on the 80^3 lattice, before writing it.

The shipped LJ_Force_thread_atom maps a wave to 64 gid-consecutive atoms of a cell (a slab 0.43 cells thick); the neighbour j is wave-uniform
(scalar loads) and the pair evaluation runs whenever ANY of the 64 lanes is inside the cutoff -- 22 % useful lanes.
The proposal: lanes = pairs.  A wave takes 8 gid-consecutive atoms; lane = 8 * atom + stream; the candidates of the group's bounding box are
dealt to the 8 streams, one candidate per lane per trip; a DPP row reduction adds the 8 streams of an atom at the end.

What this script measures on a periodic sample of the real lattice (T = 600 K thermal displacements do not change the picture; perfect lattice +
small random displacement), with the kernel's own orders (atoms of a cell in gid order):
  * candidates per 64-atom wave box (today) and per 8-atom group box, and the in-cutoff pairs per atom;
  * wave-instructions per atom of both mappings from the same per-instruction costs (test 7, evaluation 12 VALU, measured on the shipped kernel);
  * the box tests the LIST BUILD has to make (today: every stencil atom against the <= 3 wave boxes of a cell).
"""
import numpy as np

LAT, NX, RC = 3.615, 80, 5 * 2.315
G = int(NX * LAT / RC)                      # 24 cells per axis at 80^3
H = NX * LAT / G                            # 12.05 A
S = 5                                       # sample: the cells of a 5^3 block and everything around them
rng = np.random.default_rng(1)

# atoms of a (S+2)^3 block of cells, by cell, in gid order (gid = ib + 4 (iz + nz (iy + ny ix)))
basis = np.array([[.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
nmax = int(np.ceil((S + 2) * H / LAT)) + 1
ii = np.arange(-1, nmax)
ix, iy, iz, ib = np.meshgrid(ii, ii, ii, np.arange(4), indexing="ij")
pos = (np.stack([ix, iy, iz], -1) + basis[ib]) * LAT - H          # shift so that cell (0,0,0) is the halo layer
gid = ib + 4 * (iz + 1000 * (iy + 1000 * ix))
pos = pos.reshape(-1, 3) + rng.uniform(-0.05, 0.05, (pos.size // 3, 3))
gid = gid.reshape(-1)
cell = np.floor(pos / H).astype(int)
ok = np.all((cell >= -1) & (cell <= S), axis=1)
pos, gid, cell = pos[ok], gid[ok], cell[ok]
key = (cell[:, 0] + 1) * (S + 2) ** 2 + (cell[:, 1] + 1) * (S + 2) + (cell[:, 2] + 1)
order = np.lexsort((gid, key))
pos, key = pos[order], key[order]
starts = np.searchsorted(key, np.arange((S + 2) ** 3 + 1))


def cell_atoms(cx, cy, cz):
    k = (cx + 1) * (S + 2) ** 2 + (cy + 1) * (S + 2) + (cz + 1)
    return pos[starts[k]:starts[k + 1]]


def box_candidates(group, stencil):
    lo, hi = group.min(0), group.max(0)
    d = np.maximum(0, np.maximum(lo - stencil, stencil - hi))
    return int(((d * d).sum(1) <= RC * RC).sum())


def any_hit_trips(group, cand, width):
    """trips of `width` candidates per trip in which at least one (atom, candidate) pair is inside the cutoff, and the useful pair slots"""
    d2 = ((group[:, None, :] - cand[None, :, :]) ** 2).sum(-1)
    hit = (d2 <= RC * RC) & (d2 > 0)
    return hit


res = {"cand64": [], "cand8": [], "pairs": [], "eval64": [], "useful64": [], "eval8": [], "useful8": [], "atoms": 0, "boxtests64": 0, "boxtests8": 0}
for cx in range(1, S - 1):
    for cy in range(1, S - 1):
        for cz in range(1, S - 1):
            own = cell_atoms(cx, cy, cz)
            stencil = np.concatenate([cell_atoms(cx + a, cy + b, cz + c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)])
            n = len(own)
            res["atoms"] += n
            # today: waves of 64 slots
            for w0 in range(0, n, 64):
                g = own[w0:w0 + 64]
                lo, hi = g.min(0), g.max(0)
                d = np.maximum(0, np.maximum(lo - stencil, stencil - hi))
                cand = stencil[(d * d).sum(1) <= RC * RC]
                hit = any_hit_trips(g, cand, 1)
                res["cand64"].append((len(cand), len(g)))
                res["eval64"].append((int(hit.any(0).sum()), len(g)))          # evaluation slots: candidates some lane accepts
                res["useful64"].append(hit.sum())
                res["pairs"].append(hit.sum() / len(g))
                res["boxtests64"] += len(stencil)
            # proposal: groups of 8
            for g0 in range(0, n, 8):
                g = own[g0:g0 + 8]
                lo, hi = g.min(0), g.max(0)
                d = np.maximum(0, np.maximum(lo - stencil, stencil - hi))
                cand = stencil[(d * d).sum(1) <= RC * RC]
                hit = any_hit_trips(g, cand, 8)
                trips = -(-len(cand) // 8)
                anyhit = sum(hit[:, t * 8:(t + 1) * 8].any() for t in range(trips))
                res["cand8"].append((len(cand), len(g), trips, anyhit))
                res["useful8"].append(hit.sum())
                res["boxtests8"] += len(stencil)

atoms = res["atoms"]
c64 = sum(c * n for c, n in res["cand64"]) / atoms
e64 = sum(e * n for e, n in res["eval64"]) / atoms
c8 = sum(c * n for c, n, _, _ in res["cand8"]) / atoms
trips8 = sum(t for _, _, t, _ in res["cand8"])
any8 = sum(a for _, _, _, a in res["cand8"])
pairs = sum(res["useful64"]) / atoms
T, E = 7, 12
print(f"cells sampled {(S-2)**3}, atoms {atoms}, in-cutoff pairs per atom {pairs:.0f}")
print(f"today      : candidates per atom (64-atom wave boxes) {c64:.0f}, evaluation slots per atom {e64:.0f} ({pairs / e64:.0%} useful lanes)")
print(f"             a wave issues {c64:.0f} x {T} + {e64:.0f} x {E} = {c64 * T + e64 * E:.0f} VALU instructions for its 64 atoms: {(c64 * T + e64 * E) / 64:.0f} wave-instructions per atom")
print(f"             (measured, SQ_INSTS_VALU r02: 589 -- the model leaves out the x3 replicas of the under-filled tail wave and the loop overhead)")
EXTRA = 3                                   # per trip: list entry and record addressing, loop bookkeeping (the scalar path of today pays none of it in VALU)
w8 = (trips8 * (T + EXTRA) + any8 * E) / atoms
print(f"proposal   : candidates per atom (8-atom group boxes) {c8:.0f} -> {trips8 / atoms * 8:.0f} pair slots per atom, {sum(res['useful8']) / (any8 * 64.0):.0%} useful lanes in the evaluation")
print(f"             trips per atom {trips8 / atoms:.1f}, of which with a hit {any8 / atoms:.1f}: wave-instructions per atom = trips x ({T} + {EXTRA}) + hit trips x {E} = {w8:.0f}  (+ ~5 for the row reductions)")
print(f"list build : box tests per atom today {res['boxtests64'] / atoms:.0f}, with 8-atom groups {res['boxtests8'] / atoms:.0f} ({res['boxtests8'] / res['boxtests64']:.1f} x); list entries per atom {c64 / 64 * 1:.0f} -> {c8 / 8:.0f} ({c8 / 8 / (c64 / 64):.1f} x the list bytes)")
