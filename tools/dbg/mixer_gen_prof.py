import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oddio_amd as oa
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
n_clips, length = 4096, 48000 * 3
clips = (torch.rand((n_clips, length), device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
frames = [oa.Frames.from_device_ptr(48000, clips.data_ptr() + 4 * length * i, length, device=0, copy=False) for i in range(n_clips)]
control, mixer = oa.Mixer(max_sources=S, max_frames=1024)
for i in range(S):
    gc, g = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(frames[(i * 2654435761) % n_clips], 0.25)))
    control.play(g)
iv = np.float32(1.0) / np.float32(48000)
for mode in (0, 1):
    mixer.set_mode(mode)
    for _ in range(12):
        mixer.sample_n(iv, 1024)
mixer.close()
