"""Frame time of the id/mask-emitting kernel variants against the image-only one (4K, device buffers)."""
import sys; sys.path.insert(0, '.')
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.device import DeviceTracer
t = DeviceTracer(0)
pr, lt = scenes.default_scene(); t.upload_scene(pr, lt)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream()
for d in (5, 8):
    p = abi.make_params(3840, 2160, d); w = abi.mask_words(d)
    bgr = torch.zeros((2160, 3840, 3), dtype=torch.uint8, device=dev)
    hid = torch.zeros((2160, 3840), dtype=torch.int32, device=dev)
    hm = torch.zeros((2160, 3840, w), dtype=torch.int32, device=dev); sm = torch.zeros_like(hm)
    for name, args in (("image only", (bgr.data_ptr(),)), ("+ hit_id + masks", (bgr.data_ptr(), hid.data_ptr(), hm.data_ptr(), sm.data_ptr()))):
        for _ in range(3): t.render_device(p, st.cuda_stream, *args)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): t.render_device(p, st.cuda_stream, *args)
        e1.record(); torch.cuda.synchronize()
        print(f"default 4K D{d} {name}: {e0.elapsed_time(e1)/30:.4f} ms/frame", flush=True)
