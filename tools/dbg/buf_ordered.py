import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
import oddio_amd as oa
from oddio_amd import synth
S=int(sys.argv[1]) if len(sys.argv)>1 else 65536
dev=torch.device("cuda",0)
sc=synth.make_scene(1,S)
control, scene = oa.SpatialScene(max_sources=S, max_frames=1024)
scene.reserve_buffered(S)
clips=torch.rand((4096,65536),device=dev)*2-1
frames=[oa.Frames.from_device_ptr(48000, clips.data_ptr()+4*65536*i, 65536, device=0, copy=False) for i in range(4096)]
ids=control.play_buffered_frames_batch([frames[i%4096] for i in range(S)], np.full(S,0.5), [oa.FILTER_SPEED, oa.FILTER_GAIN], np.stack([np.full(S,1.05,np.float32), np.ones(S,np.float32)],axis=1), sc["position"], sc["velocity"], sc["radius"], 100.0, 48000, 0.1)
out=torch.zeros((1024,2),device=dev)
iv=np.float32(1)/np.float32(48000)
host=np.zeros((1024,2),np.float32)
for _ in range(3): scene.sample(iv, host)
scene.set_buffered_fast(False)
for _ in range(3): scene.sample_device(iv,out.data_ptr(),1024)
scene.synchronize()
scene.set_buffered_fast(True)
for mode in (oa.MODE_FAST, oa.MODE_ORDERED, oa.MODE_FAST):
    scene.set_mode(mode)
    for _ in range(3): scene.sample_device(iv,out.data_ptr(),1024)
    scene.synchronize()
    scene.set_profiling(1)
    t0=time.perf_counter()
    for _ in range(6): scene.sample_device(iv,out.data_ptr(),1024)
    scene.synchronize()
    print("mode",mode,"ms/cb",(time.perf_counter()-t0)/6*1e3, "stages", scene.buffered_ms_history(6).mean(axis=0), "slow", scene.debug_buffered_slow())
    scene.set_profiling(0)
