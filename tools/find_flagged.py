import sys; sys.path.insert(0,'.')
import numpy as np
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer
t = DeviceTracer(0)
pr, lt = scenes.default_scene()
t.upload_scene(pr, lt)
p = abi.make_params(640,640,5)
r = t.flag_reasons(p)
for code in np.unique(r):
    if code == 0: continue
    ys, xs = np.nonzero(r == code)
    print(f"L{code>>4}c{code&15}: n={len(ys)} first: ", [(int(x),int(y)) for x,y in zip(xs[:6], ys[:6])])
