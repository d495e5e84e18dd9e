#!/usr/bin/env python
"""Repeats the scene fuzz of a few seeds many times (run two of these at once to shake the timing) and prints every failure in full."""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import test_hip_fuzz as t
seeds = [int(x) for x in sys.argv[2:]]
bad = 0
for rep in range(int(sys.argv[1])):
    for sd in seeds:
        try:
            t.test_random_operations_bit_exact(sd)
        except AssertionError as e:
            bad += 1
            print("rep", rep, "seed", sd, "\n".join(str(e).splitlines()[:12]), flush=True)
print("done, failures:", bad)
