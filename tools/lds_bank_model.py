#!/usr/bin/env python3
"""LDS bank model of the mix kernels' sample reads (no GPU needed).

A lane of spatial_mix_pair / spatial_mix renders 16 consecutive frames, so at any step the 64 lanes of a wave read
`clip[w + ds * (16 * lane + j)]` (and the next sample): a stride of about 16 * ds words between neighbouring lanes, ds = the
source's resample ratio (1 +- 0.1 with the bench's velocities).  ds_read2_b32 is served as two ds_read_b32: two groups of
32 lanes over 32 banks, one LDS cycle per group when no two lanes of the group want different words of one bank
(/opt/skills/guides/MI355X_MICROARCH.md, LDS).  This script counts the cycles per group for the window layouts that were
considered: none, one pad word per 16 / 8 / 32 samples (what SFLAG_PAD does for |ds - 1| < ODDIO_PAD_EPS), and
permutations of the lane -> block assignment.  Result: exactly 16 (the unpadded layout at ds = 1: 11 cycles) and the
other resonances aside, a fractional stride leaves about 2 cycles per group whatever the layout -- 32 addresses that
fall quasi-randomly on 32 banks -- against the 2.36 measured (SQ_LDS_IDX_ACTIVE / (SQ_LDS_IDX_ACTIVE -
SQ_LDS_BANK_CONFLICT), profiles/r06_final_pmc_summary.json).  One cycle per group needs |frac(stride)| < 1/32, i.e. a
layout chosen per source to 0.2 % of ds; the repack pass that would make one costs more LDS cycles than it saves.
"""
import numpy as np


def group_cycles(x):
    """LDS cycles of one ds_read_b32 over 64 lanes with word addresses x: per 32-lane group, the fullest bank."""
    tot = 0
    for g in range(2):
        ua = np.unique(x[g * 32:(g + 1) * 32])
        tot += np.bincount(ua % 32, minlength=32).max()
    return tot / 2.0


def cycles(ds, w, block_of_lane, layout):
    return np.mean([group_cycles(layout(np.floor(w + ds * (16 * block_of_lane + j)).astype(np.int64))) for j in range(16)])


LAYOUTS = {
    "none": lambda x: x,
    "pad/16": lambda x: x + (x >> 4),
    "pad/8": lambda x: x + (x >> 3),
    "pad/32": lambda x: x + (x >> 5),
    "3pad/16": lambda x: x + 3 * (x >> 4),
}


def main():
    rng = np.random.default_rng(1)
    ident = np.arange(64)
    print("cycles per 32-lane group (1 = conflict-free), ds +- 0.005 around the column's value")
    print("ds     " + "  ".join("%7s" % k for k in LAYOUTS))
    for ds in np.arange(0.90, 1.101, 0.01):
        row = [np.mean([cycles(ds + d, rng.uniform(0, 64), ident, f) for d in rng.uniform(-0.005, 0.005, 20)]) for f in LAYOUTS.values()]
        print("%.2f   " % ds + "  ".join("%7.2f" % v for v in row))
    # the bench's scene: positions uniform in [-50, 50]^3, velocities in [-20, 20]^3 -> ds = 1 - v_radial / 343
    n = 1500
    p = rng.uniform(-50, 50, (n, 3)); v = rng.uniform(-20, 20, (n, 3))
    ds_all = 1.0 - (p * v).sum(1) / np.linalg.norm(p, axis=1) / 343.0
    perms = {"lane l -> block l": ident,
             "even blocks in lanes 0-31": np.concatenate([np.arange(0, 64, 2), np.arange(1, 64, 2)]),
             "bit-reversed": np.array([int(format(i, "06b")[::-1], 2) for i in range(64)])}
    w = rng.uniform(0, 64, n)
    print("\nthe bench's distribution of ds (%d sources):" % n)
    base = {}
    for name, perm in perms.items():
        base[name] = np.array([cycles(d, ww, perm, LAYOUTS["none"]) for d, ww in zip(ds_all, w)])
        print("  %-28s unpadded %.2f" % (name, base[name].mean()))
    padded = np.array([cycles(d, ww, ident, LAYOUTS["pad/16"]) for d, ww in zip(ds_all, w)])
    print("  pad/16 for every source      %.2f" % padded.mean())
    print("  the better of the two, per source  %.2f" % np.minimum(base["lane l -> block l"], padded).mean())
    shipped = np.where(np.abs(ds_all - 1.0) < 0.004, padded, base["lane l -> block l"])
    print("  shipped rule (pad/16 when |ds - 1| < 0.004)  %.2f" % shipped.mean())


if __name__ == "__main__":
    main()
