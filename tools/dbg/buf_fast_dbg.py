import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_hip_buffered_fast as T
import oddio_amd as oa
from oracle import oracle_c as oc
INTERVAL = T.INTERVAL
def run(mode, fast, n=400, ncb=6, which=2):
    oa_, control, scene, ref, ctl, rng = T.populate(n, 21, max_distance=60.0, buffer_duration=0.05, cube=25.0, vmax=20.0, mode=mode)
    scene.set_buffered_fast(fast)
    (gh, go), (sph, spo) = ctl[which]
    gh.set_gain(-6.0); go.set_gain(-6.0)
    sph.set_speed(1.05); spo.set_speed(1.05)
    for cb in range(ncb):
        a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
        err = np.abs(b-a)
        bad = err.max(axis=1)>1e-6
        print(f"mode={mode} fast={fast} cb={cb} slow={scene.debug_buffered_slow()} maxerr={err.max():.3e} rel={err.max()/np.abs(a).max():.3e} first bad frame={np.argmax(bad) if bad.any() else -1} nbad={bad.sum()}")
    scene.close()
run(oa.MODE_ORDERED, True)
run(oa.MODE_FAST, True)
run(oa.MODE_FAST, False)
run(oa.MODE_ORDERED, True, n=3)
