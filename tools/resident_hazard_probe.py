import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import reevr_amd
from reevr_amd import synth
irs = list(synth.synth_ir(20000, 2, 1))
x = np.stack([synth.synth_input(512 * 50, c) for c in range(2)])
a = reevr_amd.ConvolverSet(2, persistent=True)
assert a.init(512, 8192, irs, max_len=512)
for i in range(10):
    a.process(x[:, i * 512:(i + 1) * 512])
t0 = time.time()
b = reevr_amd.ConvolverSet(2)
assert b.init(512, 8192, irs, max_len=512)
t1 = time.time()
b.process(x[:, :512])
t2 = time.time()
b.close()
t3 = time.time()
print("while another set's kernel is resident: init %.3f s, process %.3f s, destroy %.3f s" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
for i in range(10, 20):
    a.process(x[:, i * 512:(i + 1) * 512])
a.close()
print("done")
