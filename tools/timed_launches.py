#!/usr/bin/env python
"""The spatial_mix launches of bench.py's TIMED region out of a rocprofv3 kernel trace of the whole command.
`--stats` averages every launch of the command -- hundreds of preconditioning callbacks on a drifting workload, the
stage-timing and host-output callbacks after the timed region -- while `roofline.avg_kernel_ms` is the timed region's.
After the timed region bench.py launches the accumulate instantiation 8 (stage timing) + 2 + 6 (host output) more times.
usage: timed_launches.py <kernel_trace.csv> [steps=20] [launches_after=16]   -> csv on stdout, summary on stderr"""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
after = int(sys.argv[3]) if len(sys.argv) > 3 else 16
def _timed_kernel(name):      # the FAST instantiation of the Seek set: spatial_mix<true, false, true[, false[, false]]> (older trees: <true, false>)
    if name.startswith("void oddio_hip::spatial_mix_pair<"):      # round 5: large FAST-mode scenes (pair_kernels.h): <FULL, FUSED>
        a = [x.strip() for x in name[name.index("<") + 1:name.index(">")].split(",")]
        return a[:2] == ["true", "true"] and a[4:5] != ["true"]      # (not <.., LANE16>: callbacks of 16 k < 1024 frames)
    if not name.startswith("void oddio_hip::spatial_mix<"):
        return False
    a = [x.strip() for x in name[name.index("<") + 1:name.index(">")].split(",")]
    return a[:2] == ["true", "false"] and (len(a) == 2 or a[2] == "true") and a[3:4] != ["true"]


rows = [r for r in csv.DictReader(open(path)) if _timed_kernel(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
timed = rows[len(rows) - after - steps: len(rows) - after]
print("launch,start_ns,duration_us,gap_before_us")
prev_end = None
durs = []
for k, r in enumerate(timed):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    durs.append((b - a) / 1e3)
    print(f"{k},{a},{(b - a) / 1e3:.2f},{'' if prev_end is None else f'{(a - prev_end) / 1e3:.2f}'}")
    prev_end = b
allv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(f"timed region: {len(durs)} launches, mean {sum(durs) / len(durs):.2f} us (min {min(durs):.2f}, max {max(durs):.2f}); "
      f"all {len(allv)} launches of the command: mean {sum(allv) / len(allv):.2f} us", file=sys.stderr)
