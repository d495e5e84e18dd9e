#!/usr/bin/env python
"""profiles/pmc_latest.json from a tools/profile_pmc.sh summary.

HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB and come
from separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced stream, which is what the mix kernel's window loads are, so the read side is
doubled; WRITE_SIZE is taken as is (uncalibrated, < 1 % of the traffic here)."""
import json
import sys

def _targs(k):     # spatial_mix<FULL, STORE, FUSED, RING, DMX> -> its template arguments (older trees: fewer)
    return [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []


import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oddio_amd import _lib  # noqa: E402  (the source hash only; nothing is loaded)

summary, sources, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = json.load(open(summary))
if len(sys.argv) > 4 and sys.argv[4] == "buffered":
    # the buffered set's path: buffered_walk + buffered_write + spatial_mix<.., RING> (+ the reduce of its partial tiles)
    ks = {"walk": [k for k in d if k.startswith("buffered_walk")], "write": [k for k in d if k.startswith("buffered_write")],
          "reads": [k for k in d if k.startswith("spatial_mix<") and _targs(k)[3:4] == ["true"]]}
    res = {"sources": sources, "kernels": {}, "correction": "read side x2 (gfx950 FETCH_SIZE counts 64 B per 128 B request on wide coalesced loads)"}
    total = 0.0
    for name, cand in ks.items():
        if not cand:
            continue
        k = max(cand, key=lambda k: d[k].get("_dispatches") or 0)
        c = d[k]
        b = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        total += b
        res["kernels"][name] = {"kernel": k, "dispatches": c.get("_dispatches"), "hbm_bytes_per_launch": b, "counters": {n: v for n, v in c.items() if not n.startswith("_")}}
    res["hbm_bytes_per_callback"] = total
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_callback": total}))
    sys.exit(0)
# the FAST instantiation the timed region runs -- spatial_mix<FULL, false, true> (fused arithmetic) -- not the row render of the
# ORDERED-mode callbacks (<.., true, ..>) nor the unfused repeat of the timed region (<.., false, false>); older trees: <FULL, false>
def _ring(k):      # spatial_mix<FULL, STORE, FUSED, RING, DMX>: the buffered set's ring reads
    return _targs(k)[3:4] == ["true"]
d_all = d
d = {k: v for k, v in d.items() if not (k.startswith("spatial_mix<") and _ring(k))}
cands = [k for k in d if k.startswith("spatial_mix<true, false, true") or k.startswith("spatial_mix<false, false, true")]
# round 5: large FAST-mode scenes run spatial_mix_pair<FULL, FUSED> (pair_kernels.h) instead
pair = [k for k in d if k.startswith("spatial_mix_pair<true, true") or k.startswith("spatial_mix_pair<false, true")]
if pair and (not cands or max(d[k].get("_dispatches") or 0 for k in pair) >= max(d[k].get("_dispatches") or 0 for k in cands)):
    cands = pair
if not cands:
    cands = [k for k in d if k.startswith("spatial_mix<true, false") or k.startswith("spatial_mix<false, false")]
k = max(cands, key=lambda k: d[k].get("_dispatches") or 0)
c = d[k]
res = {
    "kernel": k, "sources": sources, "dispatches": c.get("_dispatches"),
    "FETCH_SIZE_KiB": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB": c.get("WRITE_SIZE"),
    "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
    "correction": "read side x2 (gfx950 FETCH_SIZE counts 64 B per 128 B request on wide coalesced loads)",
    "counters": {n: v for n, v in c.items() if not n.startswith("_")},
    # the kernels these counters were measured on (bench.py prints roofline.traffic_stale when the tree's differ)
    "mix_kernel_source_hash": _lib.mix_kernel_source_hash(),
}
# ORDERED mode: the row render (spatial_mix<.., true, ..>) + ordered_sum of one callback
rows = [k for k in d if k.startswith("spatial_mix<true, true") or k.startswith("spatial_mix<false, true")]
sums = [k for k in d if k.startswith("ordered_sum")]
if rows and sums and "FETCH_SIZE" in d[rows[0]] and "FETCH_SIZE" in d[sums[0]]:
    r_, s_ = d[rows[0]], d[sums[0]]
    res["ordered_hbm_bytes_per_callback"] = (2.0 * r_["FETCH_SIZE"] + r_.get("WRITE_SIZE", 0.0) + 2.0 * s_["FETCH_SIZE"] + s_.get("WRITE_SIZE", 0.0)) * 1024.0
    res["ordered_kernels"] = {rows[0]: {"FETCH_SIZE_KiB": r_["FETCH_SIZE"], "WRITE_SIZE_KiB": r_.get("WRITE_SIZE")},
                              sums[0]: {"FETCH_SIZE_KiB": s_["FETCH_SIZE"], "WRITE_SIZE_KiB": s_.get("WRITE_SIZE")}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({"hbm_bytes_per_launch": res["hbm_bytes_per_launch"]}))
