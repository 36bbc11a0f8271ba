import sys, time; sys.path.insert(0,'.')
import numpy as np
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer
from tests import oracle_lib
t = DeviceTracer(0)
def cmp(name, prims, lights, p):
    t.upload_scene(prims, lights)
    s0 = t.stats()
    dev = t.render(p)
    s1 = t.stats()
    ex = t.render(p, exact=True)
    ora = oracle_lib.render(prims, lights, p)
    px = dev['bgr'].shape[0]*dev['bgr'].shape[1]
    bad_id = int((dev['hit_id'] != ora['hit_id']).sum())
    bad_hm = int((dev['hit_mask'] != ora['hit_mask']).any(-1).sum())
    bad_sm = int((dev['spawn_mask'] != ora['spawn_mask']).any(-1).sum())
    diff = np.abs(dev['bgr'].astype(int) - ora['bgr'].astype(int))
    dex = np.abs(ex['bgr'].astype(int) - ora['bgr'].astype(int))
    print(f"{name}: px {px} requeued {s1['requeued_pixels']-s0['requeued_pixels']} ({(s1['requeued_pixels']-s0['requeued_pixels'])/px:.4%}) "
          f"bad id/hm/sm {bad_id}/{bad_hm}/{bad_sm} colour maxdiff {diff.max()} px!=0 {(diff.max(-1)>0).mean():.4%} px>1 {(diff.max(-1)>1).sum()} | exact maxdiff {dex.max()}")
pr, lt = scenes.default_scene()
for (w,h,d) in [(64,64,4),(160,90,5),(640,640,4),(640,640,8),(3840,2160,5)]:
    cmp(f"default {w}x{h} D{d}", pr, lt, abi.make_params(w,h,d))
gp, gl = scenes.glass_scene()
cmp("glass 256x144 D8", gp, gl, abi.make_params(256,144,8))
sp, sl = scenes.random_spheres_scene()
cmp("spheres 192x108 D5", sp, sl, abi.make_params(192,108,5))
for g in (0.0, 0.01):
    t.set_guard(g)
    print("guard", g)
    cmp("default 640x640 D5", pr, lt, abi.make_params(640,640,5))
    cmp("glass 256x144 D8", gp, gl, abi.make_params(256,144,8))
t.set_guard(1.0)
