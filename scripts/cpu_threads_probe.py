import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import synth, HHConfig
from oracle import encoder as OE
cfg = HHConfig(num_frames=16, depth=2)
sd = synth.encoder_state(cfg, 0, with_text=False)
v = synth.make_batch(cfg, 1, 0)["video"]
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    with torch.no_grad():
        OE.vision_forward(v, sd, cfg)
        t = time.time(); OE.vision_forward(v, sd, cfg); print(th, "threads: 2-block enc fwd", round(time.time() - t, 2), "s", flush=True)
