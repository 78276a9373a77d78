"""Create / init / process / destroy in a loop and watch free device memory (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reevr_amd
from reevr_amd import synth
from tests import impulse_cases as IC

ir = synth.synth_ir(200000, 2, 3)
x = np.stack([synth.synth_input(70000, c) for c in range(2)])
raw = IC.raw_channels(100000, 4, 5)
free0 = None
for i in range(120):
    s = reevr_amd.ConvolverSet(2, bg_stream=bool(i & 1))
    assert s.init(512, 8192, list(ir), max_len=70000)
    s.process(x[:, :512]); s.process(x)
    if i % 3 == 0:
        assert s.init(256, 8192, list(ir[:, :50000]), max_len=4096)     # re-init with another geometry
        s.process(x[:, :4096])
    imp = reevr_amd.Impulse(); imp.prepare(48000.0); imp.setRaw(*raw); imp.decayMagnitude = IC.MAGS["tilt"]; imp.recalcImpulse()
    sc = reevr_amd.StereoConvolver(); sc.prepare(512); sc.loadImpulse(imp); sc.process(x[0, :512], x[1, :512], 512)
    s.close(); imp.close(); del sc
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if i == 10:
        free0 = free
    if i % 20 == 0:
        print(i, "free MiB", free // (1 << 20))
print("drift since iteration 10: %.1f MiB" % ((free0 - free) / (1 << 20)))
assert free0 - free < 64 << 20, "device memory keeps shrinking"
print("leak check ok")
