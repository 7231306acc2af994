import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krepp_amd import capi, synth
g = synth.evolve_genomes("(a:0.1,b:0.1);", 5_000_000, seed=3)["a"]
offs = np.array([0, len(g)], np.uint64)
pp = [28, 26, 24, 23, 19, 16, 13, 10, 8, 7, 5, 3, 2]
for dev in (None, 0, 0):
    t = time.time(); k, n1, n2 = capi.minimizers(g, offs, 29, 35, 13, pp, device=dev); dt = time.time() - t
    print("device" if dev is not None else "cpu", f"{dt*1e3:.1f} ms wall (incl. copies, sort)", len(k), n2 / n1)
