"""Print the few numbers of a bench.py JSON line that kernel A/B runs compare (stdin)."""
import json
import sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln); r = d["roofline"]
        print(" ".join(sys.argv[1:]), "step_ms=%.4f mix_ms=%.4f frac=%.3f fcb=%.3f pre=%.3f red=%.3f" % (
            d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r["frac_callback"], r["prepass_ms"], r.get("reduce_ms", 0)))
