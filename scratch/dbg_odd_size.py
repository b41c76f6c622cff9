import sys, numpy as np
sys.path.insert(0, '/root/repo')
import imagemosaicing_amd as im
from tests import oracle_lib
from tests.synth_frames import terrain
o = oracle_lib.load_oracle()
ctx = im.Context(0)
for (w, h) in [(2001, 1501), (2050, 1538), (4100, 260)]:
    img = terrain(w, h, seed=w)
    kp, desc = ctx.SiftExtract(1, img)
    okp, odesc = o.sift(img)
    print(w, h, len(kp), len(okp), np.array_equal(kp.view(np.uint8), okp.view(np.uint8)), np.array_equal(desc.astype(np.uint8), odesc))
