import sys; sys.path.insert(0,'.')
import numpy as np
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer
t = DeviceTracer(0)
def show(name, prims, lights, p):
    t.upload_scene(prims, lights)
    r = t.flag_reasons(p)
    vals, cnts = np.unique(r, return_counts=True)
    print(name, "flagged %.3f%%" % (100*(r>0).mean()), {f"L{v>>4}c{v&15}": int(c) for v,c in zip(vals,cnts) if v})
pr, lt = scenes.default_scene()
for g in (1.0, 0.0):
    t.set_guard(g); print("guard", g)
    show("default 640 D5", pr, lt, abi.make_params(640,640,5))
    show("default 4K D5", pr, lt, abi.make_params(3840,2160,5))
    gp, gl = scenes.glass_scene()
    show("glass 256x144 D8", gp, gl, abi.make_params(256,144,8))
    sp, sl = scenes.random_spheres_scene()
    show("spheres 192x108 D5", sp, sl, abi.make_params(192,108,5))
