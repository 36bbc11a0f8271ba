"""Host-side pieces that need no GPU: the frame writers, and the C++ scene model, whose
flattened records must be the very bytes the Python scene builder produces."""
import os
import subprocess

import numpy as np

from raytracer_amd import abi, frame, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DUMP_SRC = r"""
#include <cstdio>
#include "scene_model.hpp"
// links scene_model.cpp only for the Screen; Tracer() is never called here
int main(int argc, char** argv) {
    Color bgr{0.58, 0.76, 0.92};
    const Light ambient{bgr, 0.3f};
    const Color grey = Color{1.0, 1.0, 1.0} * 0.5;
    const Color violet{0.5, 0.313, 0.64};
    const Color orange{0.9, 0.4, 0.298};
    std::vector<std::unique_ptr<Object>> objs;
    std::vector<std::unique_ptr<PointLight>> lights;
    objs.push_back(std::make_unique<Plane>(gmtl::Vec3d{0, -3, 0}, gmtl::Vec3d{0, 1, 0},
                                           DiffuseMaterial{grey, grey, Color{0, 0, 0} * 0.5, 0, 0}, ambient));
    objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{2, 0.5, 10}, 1,
                                            DiffuseMaterial{violet, violet, violet, .7, .9f, 1.1}, ambient));
    objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{1.5, 0, 13}, 1,
                                            DiffuseMaterial{orange, orange, orange, 0.2f, 0.f}, ambient));
    objs.push_back(std::make_unique<Plane>(gmtl::Vec3d{1, 2, 3}, gmtl::Vec3d{1, -2, 0.5},
                                           DiffuseMaterial{grey, grey, grey}, ambient));
    lights.push_back(std::make_unique<PointLight>(Color{1, 1, 1}, 1.f, gmtl::Vec3d{0, 4, 7}));
    lights.push_back(std::make_unique<PointLight>(Color{1, 1, 1}, .3f, gmtl::Vec3d{10, 8, 3}));
    FILE* f = fopen(argv[1], "wb");
    for (auto& o : objs) { rt_prim p = o->Flatten(); fwrite(&p, sizeof p, 1, f); }
    for (auto& l : lights) { rt_light p = l->Flatten(); fwrite(&p, sizeof p, 1, f); }
    fclose(f);
    // Screen: store, swap, write
    Screen scr(5, 3, argv[2]);
    scr.ClearScreenColor(bgr);
    scr.StorePixel(1, 2, 1, Color{1.0, 0.5, 0.25});
    scr.StorePixel(1, 2, 1, Color{0.1, 0.9, 0.1});   // equal depth: per-channel max
    scr.StorePixel(4, 0, 1, Color{2.0, -1.0, 0.999});
    scr.DrawScreen();
    ViewFrustum v(640, 640, 60, 1000.0);
    printf("%.17g %.17g\n", v.PixelWidth(), v.PixelHeight());
    // Two frames through the whole-frame store Tracer() uses.  As in the reference the depth
    // buffer is never reset (src/screen.h:81-99): the second frame arrives at the depth the
    // first one left, i.e. through the equal-depth merge, over a zeroed back buffer.
    Screen two(4, 2, argv[3]);
    two.ClearScreenColor(bgr);
    std::vector<uint8_t> f1(4 * 2 * 3, 200), f2(4 * 2 * 3);
    for (size_t i = 0; i < f2.size(); ++i) f2[i] = static_cast<uint8_t>(10 + i);
    uint8_t* dst = two.BeginFrameStore(1);
    if (!dst) return 3;                          // depth 0 everywhere: frame 1 overwrites
    std::copy(f1.begin(), f1.end(), dst);
    two.DrawScreen();
    two.ClearScreenColor(bgr);                   // depth is 1 by now: a clear at depth 0 shows nothing
    dst = two.BeginFrameStore(1);
    if (!dst) return 4;                          // equal depth over a zeroed buffer: still a plain store
    std::copy(f2.begin(), f2.end(), dst);
    if (two.BeginFrameStore(1)) return 5;        // a second store at that depth must merge
    std::vector<uint8_t> f3(4 * 2 * 3, 30);
    two.StoreRows(0, 2, 1, f3.data());           // per-channel max with frame 2
    two.StorePixel(0, 0, 2, Color{0.0, 0.0, 0.0});  // nearer: replaces
    two.DrawScreen();
    return 0;
}
"""


def test_cpp_scene_model_flattens_to_python_scene_bytes(tmp_path):
    host = os.path.join(ROOT, "raytracer_amd", "host")
    src = tmp_path / "dump.cpp"
    src.write_text(DUMP_SRC)
    exe = tmp_path / "dump"
    # scene_model.cpp references the rt_* entry points; link the built library
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", host, "-o", str(exe), str(src),
         os.path.join(host, "scene_model.cpp"), "-L", os.path.join(ROOT, "raytracer_amd"), "-lrt_hip",
         "-Wl,-rpath," + os.path.join(ROOT, "raytracer_amd")],
        check=True,
    )
    blob, bmp, bmp2 = tmp_path / "scene.bin", tmp_path / "s.bmp", tmp_path / "two.bmp"
    out = subprocess.run([str(exe), str(blob), str(bmp), str(bmp2)], check=True, capture_output=True, text=True).stdout
    raw = blob.read_bytes()
    prims = np.frombuffer(raw[: 4 * abi.PRIM_DTYPE.itemsize], dtype=abi.PRIM_DTYPE)
    lights = np.frombuffer(raw[4 * abi.PRIM_DTYPE.itemsize :], dtype=abi.LIGHT_DTYPE)
    pp, pl = scenes.default_scene()
    assert prims[:3].tobytes() == pp.tobytes()
    assert lights.tobytes() == pl.tobytes()
    extra = scenes.plane((1, 2, 3), (1, -2, 0.5), (0.5, 0.5, 0.5))
    assert prims[3].tobytes() == extra.tobytes()  # constructor normalised (1,-2,.5) like make_normal
    # Screen semantics, through the BMP the C++ writer produced and the Python reader
    img = frame.read_bmp(str(bmp))
    assert img.shape == (3, 5, 3)
    back = np.array([int(0.92 * 255), int(0.76 * 255), int(0.58 * 255)], dtype=np.uint8)
    assert (img[0, 0] == back).all()
    assert (img[2, 1] == [63, 229, 255]).all()  # B=max(.25,.1)*255, G=max(.5,.9)*255, R=max(1,.1)*255
    assert (img[0, 4] == [254, 0, 255]).all()  # clamped to [0,1], truncated
    # the second of two frames: max(frame 2, 30) per channel, pixel (0,0) replaced at depth 2
    two = frame.read_bmp(str(bmp2))
    want = np.maximum(np.arange(10, 34, dtype=np.uint8).reshape(2, 4, 3), 30)
    want[0, 0] = 0
    assert np.array_equal(two, want)
    import math

    pw, ph = (float(v) for v in out.split())
    assert pw == 2 * 1 * math.tan(0.5 * 60 * (3.141592653589793 / 180.0)) / 640 and ph == pw


def test_bmp_ppm_writers(tmp_path):
    rng = np.random.default_rng(5)
    for (h, w) in ((4, 4), (3, 5), (2, 7)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"a{w}.bmp")
        frame.write_bmp(p, img)
        assert os.path.getsize(p) == 54 + h * (3 * w + (-3 * w) % 4)
        assert np.array_equal(frame.read_bmp(p), img)
        q = str(tmp_path / f"a{w}.ppm")
        frame.write_ppm(q, img)
        data = open(q, "rb").read()
        head = b"P6\n%d %d\n255\n" % (w, h)
        assert data.startswith(head)
        rgb = np.frombuffer(data[len(head) :], dtype=np.uint8).reshape(h, w, 3)
        assert np.array_equal(rgb, img[::-1, :, ::-1])


BUILDERS_SRC = r"""
#include <cstdio>
#include "scene_builders.hpp"
static void dump(FILE* f, const ObjectList& objs, const LightList& lights) {
    for (auto& o : objs) { rt_prim p = o->Flatten(); fwrite(&p, sizeof p, 1, f); }
    for (auto& l : lights) { rt_light p = l->Flatten(); fwrite(&p, sizeof p, 1, f); }
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "wb");
    Color bgr{0.58, 0.76, 0.92};
    { ObjectList o; LightList l; SetupScene(o, l, bgr); dump(f, o, l); }
    { ObjectList o; LightList l; SetupRandomSpheresScene(o, l, bgr); dump(f, o, l); }
    { ObjectList o; LightList l; SetupGlassScene(o, l, bgr); dump(f, o, l); }
    { ObjectList o; LightList l; SetupScene(o, l, bgr); OrbitScene(o, l, 0.7); dump(f, o, l); }
    { ObjectList o; LightList l; SetupGlassScene(o, l, bgr, 5, 3, 77); OrbitScene(o, l, -2.1); dump(f, o, l); }
    fclose(f);
    return 0;
}
"""


def test_cpp_scene_builders_match_python_builders(tmp_path):
    """The C++ host's scenes for every benchmark configuration (default, 1024 spheres, glass,
    camera orbit) flatten to the bytes of the Python builders: one scene definition, two hosts."""
    host = os.path.join(ROOT, "raytracer_amd", "host")
    src = tmp_path / "dump2.cpp"
    src.write_text(BUILDERS_SRC)
    exe = tmp_path / "dump2"
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", host, "-o", str(exe), str(src),
         os.path.join(host, "scene_model.cpp"), os.path.join(host, "scene_builders.cpp"),
         "-L", os.path.join(ROOT, "raytracer_amd"), "-lrt_hip", "-Wl,-rpath," + os.path.join(ROOT, "raytracer_amd")],
        check=True,
    )
    blob = tmp_path / "scenes.bin"
    subprocess.run([str(exe), str(blob)], check=True)
    raw = blob.read_bytes()
    d = scenes.default_scene()
    g5 = scenes.glass_scene(5, 3, 77)
    want = [d, scenes.random_spheres_scene(), scenes.glass_scene(), scenes.orbit_scene(*d, 0.7),
            scenes.orbit_scene(*g5, -2.1)]
    off = 0
    for k, (pr, lt) in enumerate(want):
        n = pr.nbytes + lt.nbytes
        assert raw[off : off + pr.nbytes] == pr.tobytes(), f"scene {k}: objects differ"
        assert raw[off + pr.nbytes : off + n] == lt.tobytes(), f"scene {k}: lights differ"
        off += n
    assert off == len(raw)
