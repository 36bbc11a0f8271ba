"""GPU parity: the HIP library (through its C ABI) against the CPU checker on the same
scene bytes.  Run on the GPU box: python -m pytest tests -m gpu."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes
from tests import oracle_lib
from raytracer_amd.strips import rank_params as strips_params
from tests.parity import assert_parity

pytestmark = pytest.mark.gpu


def _shares_rays(prims, params):
    """Whether the image-only launch of this render hands rays from lane to lane inside a wave
    (queue launches that use the anchor grids): a pixel's colour is then the same terms added
    up in another order, and a channel may differ by one from the launch that also writes ids."""
    flags = int(params["flags"])
    spheres = len(prims) - int((prims["type"] == abi.RT_PRIM_PLANE).sum())
    grid = (spheres >= 8 or flags & abi.RT_FLAG_GRID) and not flags & abi.RT_FLAG_NOGRID and spheres > 0
    tree = (spheres >= 40 or flags & abi.RT_FLAG_TREE) and not flags & abi.RT_FLAG_LINEAR
    return bool(grid and (tree or int(params["max_depth"]) > 5))


def _assert_same_image(got, want, shares, what):
    if shares:
        assert_parity({"bgr": got}, {"bgr": want}, what)
    else:
        assert np.array_equal(got, want), what


def _both(tracer, prims, lights, params, counters=True):
    tracer.upload_scene(prims, lights)
    dev = tracer.render(params, want_counters=counters)
    if counters:
        # the counting kernels are separate instantiations: the plain ones (with and without
        # the id/mask outputs) must give the same frame
        plain = tracer.render(params)
        for k in ("bgr", "hit_id", "hit_mask", "spawn_mask"):
            assert np.array_equal(plain[k], dev[k]), f"counting and plain kernels differ in {k}"
        walked = params.copy()
        walked["flags"] = int(params["flags"]) | abi.RT_FLAG_NOFRONT
        image_only = tracer.render(walked, want_ids=False, want_masks=False)
        _assert_same_image(image_only["bgr"], dev["bgr"], _shares_rays(prims, params), "image-only kernel differs")
    ora = oracle_lib.render(prims, lights, params, want_counters=counters)
    if counters:
        # the image-only render as callers get it: pixels that end at the primary ray are
        # finished in single precision (same hits and occluders, colours to the tolerance)
        image_only = tracer.render(params, want_ids=False, want_masks=False)
        assert_parity({"bgr": image_only["bgr"]}, {"bgr": ora["bgr"]}, "image-only")
    return dev, ora


def test_collision_and_normalize_bits(tracer):
    """TestCollision (sphere, plane) and gmtl normalize on random rays: every double the
    device produces equals the CPU's bit for bit (division and sqrt are correctly rounded
    on both)."""
    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    rng = np.random.default_rng(1234)
    n = 200_000
    rays = np.empty((n, 6))
    rays[:, 0:3] = rng.uniform(-3, 3, (n, 3))
    target = np.array([2.0, 0.5, 10.0]) + rng.normal(0, 0.8, (n, 3))
    d = target - rays[:, 0:3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3:6] = d
    rays[: n // 10, 3:6] *= rng.uniform(1e-3, 1e3, (n // 10, 1))  # un-normalised directions too
    # directions with zero, signed-zero, tiny, huge and denormal components: the device's
    # normalisation leaves its short sequence for these and must still match bit for bit
    special = np.array([
        [0.0, 1.0, 0.0], [-0.0, 3.0, 4.0], [0.0, -0.0, -2.0], [1e-200, 1.0, 0.0], [1e-300, 1e-300, 0.0],
        [0.0, 0.0, 0.0], [1e300, 1e300, 1.0], [1e-160, 1e-160, 1e-160], [5e-324, 1.0, 1.0],
        [1e-120, 2e-120, -3e-120], [1e200, -1e200, 1e-200], [3e-310, 0.0, 0.0],
    ])
    rays[n // 10 : n // 10 + len(special), 3:6] = special
    for prim in range(len(prims)):
        t_d, h_d, u_d = tracer.debug_intersect(rays, prim)
        t_o, h_o, u_o = oracle_lib.intersect(prims, rays, prim)
        assert np.array_equal(h_d, h_o), f"prim {prim}: hit flags differ"
        assert np.array_equal(t_d.view(np.uint64), t_o.view(np.uint64)), f"prim {prim}: t bits differ"
        assert np.array_equal(u_d.view(np.uint64), u_o.view(np.uint64)), f"prim {prim}: normalize bits differ"
        assert 0 < h_o.sum() < n or prim == 0


@pytest.mark.parametrize("size", [(64, 64), (160, 90), (37, 23)])
@pytest.mark.parametrize("depth", [2, 3, 4, 5, 8])
def test_default_scene(tracer, size, depth):
    prims, lights = scenes.default_scene()
    p = abi.make_params(size[0], size[1], depth)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, f"default {size} D{depth}")


def test_default_scene_reference_size(tracer):
    """The reference's own configuration: 640x640, MAX_DEPTH 4."""
    prims, lights = scenes.default_scene()
    p = abi.make_params(640, 640, 4)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, "default 640x640 D4")
    assert int(dev["counters"]["rays"]) == 893544


def test_default_scene_4k_depth5(tracer):
    """The benchmark configuration, compared pixel by pixel with the checker."""
    prims, lights = scenes.default_scene()
    p = abi.make_params(3840, 2160, 5)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, "default 4K D5")
    assert int(dev["counters"]["rays"]) == 20139306


def test_random_spheres(tracer):
    prims, lights = scenes.random_spheres_scene()
    assert len(prims) == 1025 and len(lights) == 8
    p = abi.make_params(96, 54, 5)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, "1024 spheres 96x54 D5")


def test_random_spheres_4k_rows(tracer):
    """4K frame of the 1024-sphere scene checked on a strided subset of bands (pixels are
    independent, so a strip is computed exactly as inside the full frame)."""
    prims, lights = scenes.random_spheres_scene()
    p = abi.make_params(3840, 2160, 5, band_rows=1, band_first=7, band_step=270, band_count=8)
    assert abi.strip_rows(p) == 8
    tracer.upload_scene(prims, lights)
    dev = tracer.render(p)
    full_cols = slice(0, 3840, 1)
    ora = oracle_lib.render(prims, lights, p)
    assert_parity(dev, ora, "1024 spheres 4K rows")


def test_glass_4k_depth8_rows(tracer):
    """The glass benchmark frame at its full size (the fixed grid of 1280 blocks with the
    deep-stack workspace) checked against the checker on strided rows."""
    prims, lights = scenes.glass_scene()
    p = abi.make_params(3840, 2160, 8, band_rows=1, band_first=5, band_step=135, band_count=16)
    assert abi.strip_rows(p) == 16
    tracer.upload_scene(prims, lights)
    dev = tracer.render(p, want_counters=True)
    ora = oracle_lib.render(prims, lights, p, want_counters=True)
    assert_parity(dev, ora, "glass 4K D8 rows")
    # the whole frame through the image-only kernel: its rows equal the strip's
    import torch

    full = abi.make_params(3840, 2160, 8)
    out = torch.zeros((2160, 3840, 3), dtype=torch.uint8, device="cuda")
    tracer.render_device(full, torch.cuda.current_stream().cuda_stream, out.data_ptr())
    torch.cuda.synchronize()
    rows = abi.strip_image_rows(p)
    whole = out.cpu().numpy()
    _assert_same_image(whole[rows], dev["bgr"], True, "rows of the whole image-only frame")
    # the same call gives the same bytes (which pixels share a wave does not depend on timing)
    tracer.render_device(full, torch.cuda.current_stream().cuda_stream, out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), whole)


def test_many_lights_saturated(tracer):
    """256 bright lights (the library's limit): the diffuse sum runs far past 1.0, so the
    single-precision accumulation, the clamp and the 8-bit conversion are exercised at the top
    of the range; and dim ones, where 256 small terms add up."""
    for inten, what in ((0.9, "bright"), (0.004, "dim")):
        prims, lights = scenes.glass_scene(n_spheres=10, n_lights=256, seed=0x256)
        lights = lights.copy()
        lights["intensity"] = scenes.f32(inten)
        p = abi.make_params(96, 64, 5)
        dev, ora = _both(tracer, prims, lights, p)
        assert_parity(dev, ora, f"256 lights {what}")
        if what == "bright":
            assert (ora["bgr"] == 255).mean() > 0.2


@pytest.mark.parametrize("name", ["default", "orbit", "inside", "tilted_planes", "glass", "no_objects_in_view"])
def test_screen_bounds(tracer, name):
    """Image-only renders of small scenes skip the tiles no primary ray can hit anything in
    (screen bounds worked out on the host): the frame equals the one traced pixel by pixel
    (RT_FLAG_NOSKY), whole and as strips of every shape, at widths with and without dword rows."""
    if name == "default":
        prims, lights = scenes.default_scene()
    elif name == "orbit":
        prims, lights = scenes.orbit_scene(*scenes.default_scene(), 2.2)
    elif name == "inside":
        prims, lights = scenes.default_scene()
        prims = prims.copy()
        prims[1]["p"] = (0.0, 0.0, 0.5)   # the camera inside a sphere
    elif name == "tilted_planes":
        k = (0.6, 0.6, 0.6)
        prims = scenes._stack([
            scenes.plane((0, -3, 0), (0.1, 1, 0.05), k, reflectivity=0.3),
            scenes.plane((-8, 0, 0), (1, 0.05, 0.1), (0.8, 0.3, 0.3), reflectivity=0.5),
            scenes.plane((0, 0, -1), (0, 0, 1), k, reflectivity=0.0),      # behind the camera, facing it
            scenes.sphere((3, 4, 16), 1.5, (0.9, 0.8, 0.2), reflectivity=1.0),
            scenes.sphere((-30, 20, 40), 2.0, (0.2, 0.8, 0.2)),             # far off axis
            scenes.sphere((0, 0, -5), 1.0, (0.2, 0.2, 0.9)),                # behind the camera
        ], abi.PRIM_DTYPE)
        lights = scenes._stack([scenes.point_light((1, 1, 1), 0.6, (0, 8, 10))], abi.LIGHT_DTYPE)
    elif name == "glass":
        prims, lights = scenes.glass_scene(n_spheres=7, n_lights=2)
    else:
        prims, lights = scenes.default_scene()
        prims = prims.copy()
        prims["p"] = prims["p"] * np.array([1.0, 1.0, -1.0])  # everything behind the camera
    tracer.upload_scene(prims, lights)
    shapes = [
        abi.make_params(200, 117, 5),
        abi.make_params(203, 64, 4),                                             # rows that do not start on a dword
        abi.make_params(640, 360, 5, band_rows=8, band_first=1, band_step=3, band_count=15),
        abi.make_params(640, 360, 3, band_rows=1, band_first=0, band_step=7, band_count=52),
        abi.make_params(640, 360, 5, band_rows=5, band_first=2, band_step=4, band_count=18),
        abi.make_params(1920, 1080, 2),
        abi.make_params(64, 9000, 2),                                            # more tile rows than the table holds
    ]
    skipped = 0
    for p in shapes:
        fast = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
        slow = tracer.render(_with_flags(p, abi.RT_FLAG_NOSKY), want_ids=False, want_masks=False)["bgr"]
        assert np.array_equal(fast, slow), (name, int(p["width"]), int(p["band_rows"]))
        skipped += int((fast == fast[-1, 0]).all(axis=-1).sum())
    assert skipped > 0 or name == "inside"


@pytest.mark.parametrize("name", ["default", "room", "floor_contacts", "grazing_floor", "crossing_floors", "six_lights",
                                  "light_on_floor", "far_scene", "no_lights", "tilted_floor"])
def test_pixels_ending_at_the_primary_ray(tracer, name):
    """Image-only renders of small scenes finish the pixels whose primary ray ends the ray tree
    (nothing hit, or one plane with reflectivity 0 and transparency 0) in single precision where
    every test on the way is certain at that precision.  Against the checker's frame: colours
    to the tolerance; against the frame walked in double precision (RT_FLAG_NOFRONT): never more
    than one off; a strip's rows are byte for byte the whole frame's rows."""
    matte = dict(reflectivity=0.0, transparency=0.0)
    k = (0.6, 0.6, 0.6)
    white = (1, 1, 1)
    lights = [scenes.point_light(white, 0.6, (0, 8, 10)), scenes.point_light((1, 0.9, 0.8), 0.3, (-6, 2, 2))]
    depth = 5
    if name == "default":
        prims, lt = scenes.default_scene()
        prims, lights = list(prims), None
    elif name == "room":  # six planes, four of them matte; spheres inside
        prims = [scenes.plane((0, -3, 0), (0, 1, 0), k, **matte), scenes.plane((0, 9, 0), (0, -1, 0), k, **matte),
                 scenes.plane((-8, 0, 0), (1, 0.05, 0.1), (0.8, 0.3, 0.3), reflectivity=0.5),
                 scenes.plane((8, 0, 0), (-1, 0, 0), (0.3, 0.8, 0.3), **matte),
                 scenes.plane((0, 0, 30), (0.02, 0, -1), (0.3, 0.3, 0.8), reflectivity=0.9),
                 scenes.plane((0, 0, -1), (0, 0, 1), k, **matte),
                 scenes.sphere((-2, -1, 12), 2.0, white, reflectivity=0.9, transparency=0.9, refraction_index=1.5),
                 scenes.sphere((3, 0, 16), 3.0, (0.9, 0.8, 0.2), reflectivity=1.0)]
    elif name == "floor_contacts":  # spheres resting on, cutting and hovering just above the floor
        prims = [scenes.plane((0, -3, 0), (0, 1, 0), k, **matte),
                 scenes.sphere((-4, -2, 12), 1.0, white, reflectivity=0.2, transparency=0.8, refraction_index=1.4),
                 scenes.sphere((0, -3, 14), 1.5, (0.9, 0.2, 0.2), **matte),
                 scenes.sphere((4, -1.999999, 10), 1.0, (0.2, 0.9, 0.2), reflectivity=0.6),
                 scenes.sphere((1, 2, 9), 0.7, white, transparency=1.0, refraction_index=1.1)]
    elif name == "grazing_floor":  # a floor seen almost edge-on: the pass leaves the flat band to the walk
        prims = [scenes.plane((0, -0.02, 0), (0, 1, 0.003), k, **matte),
                 scenes.sphere((0, 1.5, 20), 1.5, (0.9, 0.8, 0.2), reflectivity=0.7)]
    elif name == "crossing_floors":  # two matte planes in view: a pixel may hit both
        prims = [scenes.plane((0, -3, 0), (0, 1, 0), k, **matte), scenes.plane((0, 0, 40), (0.3, 0.2, -1), (0.8, 0.4, 0.3), **matte),
                 scenes.sphere((2, 0, 15), 2.0, white, reflectivity=0.5, transparency=0.5, refraction_index=1.3)]
    elif name == "six_lights":
        prims, lt = scenes.glass_scene(n_spheres=5, n_lights=6)
        prims, lights = list(prims), None
    elif name == "light_on_floor":  # lights within and under the floor's surface, one next to a sphere
        prims = [scenes.plane((0, -3, 0), (0, 1, 0), k, **matte), scenes.sphere((0, -1, 12), 2.0, white, reflectivity=0.5)]
        lights = [scenes.point_light(white, 0.5, (1, -3.0, 8)), scenes.point_light(white, 0.5, (-2, -3.000001, 10)),
                  scenes.point_light(white, 0.4, (0, -5, 12)), scenes.point_light(white, 0.4, (2.0001, -1, 12))]
    elif name == "far_scene":  # everything far from the origin: the error bounds scale with the coordinates
        prims = [scenes.plane((0, -3000, 0), (0, 1, 0), k, **matte), scenes.sphere((500, -1000, 9000), 2000.0, white, reflectivity=0.4),
                 scenes.sphere((-2500, -2500, 7000), 500.0, (0.3, 0.5, 0.9), transparency=0.7, refraction_index=1.2)]
        lights = [scenes.point_light(white, 0.7, (0, 8000, 5000)), scenes.point_light(white, 0.3, (-6000, 200, 2000))]
    elif name == "no_lights":
        prims = [scenes.plane((0, -3, 0), (0, 1, 0), k, **matte), scenes.sphere((0, 0, 12), 2.0, white, reflectivity=0.5)]
        lights = []
    else:  # tilted_floor
        prims = [scenes.plane((1, -3, 2), (0.3, 1, -0.2), k, **matte), scenes.plane((0, 30, 0), (0, -1, 0.1), (0.2, 0.3, 0.9), **matte),
                 scenes.sphere((-3, 0, 14), 2.5, white, reflectivity=0.3, transparency=0.6, refraction_index=1.5),
                 scenes.sphere((4, -1, 11), 1.5, (0.9, 0.5, 0.1), reflectivity=0.2)]
    if lights is None:
        lights = lt
    else:
        lights = scenes._stack(lights, abi.LIGHT_DTYPE)
        prims = scenes._stack(prims, abi.PRIM_DTYPE)
    if not isinstance(prims, np.ndarray):
        prims = scenes._stack(prims, abi.PRIM_DTYPE)
    tracer.upload_scene(prims, lights)
    moved = pixels = 0
    for p in (abi.make_params(640, 360, depth), abi.make_params(203, 117, 3), abi.make_params(1280, 720, 2)):
        ora = oracle_lib.render(prims, lights, p, want_ids=False, want_masks=False)
        fast = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
        slow = tracer.render(_with_flags(p, abi.RT_FLAG_NOFRONT), want_ids=False, want_masks=False)["bgr"]
        assert_parity({"bgr": fast}, {"bgr": ora["bgr"]}, f"primary-only pass, {name}")
        assert_parity({"bgr": slow}, {"bgr": ora["bgr"]}, f"walked, {name}")
        delta = np.abs(fast.astype(int) - slow.astype(int)).max(axis=-1)
        assert delta.max() <= 1
        moved += int(np.count_nonzero(delta))
        pixels += delta.size
    assert moved <= pixels // 2000, (name, moved, pixels)
    # strips: a pixel does not depend on how the frame is cut
    base = abi.make_params(640, 360, depth)
    whole = tracer.render(base, want_ids=False, want_masks=False)["bgr"]
    for band_rows, first, step in ((8, 1, 3), (1, 0, 7), (5, 2, 4)):
        p = abi.make_params(640, 360, depth, band_rows=band_rows, band_first=first, band_step=step,
                            band_count=-(-360 // (band_rows * step)))
        part = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
        rows = abi.strip_image_rows(p)
        assert np.array_equal(part, whole[rows]), (name, band_rows, first, step)


@pytest.mark.parametrize("name,size,depth", [("glass", (333, 207), 8), ("glass", (64, 25), 7), ("spheres1024", (203, 117), 5),
                                             ("spheres1024", (150, 75), 8), ("spheres64", (101, 67), 6)])
def test_lanes_sharing_rays(tracer, name, size, depth):
    """Image-only queue launches that use the anchor grids hand pending rays from lane to lane
    inside a wave: ragged frames (tiles with lanes outside the frame), whole and as strips,
    against the checker; the same bytes on every call; within the colour tolerance of the launch
    that also writes ids (which does not share)."""
    prims, lights = {"glass": scenes.glass_scene, "spheres1024": scenes.random_spheres_scene,
                     "spheres64": lambda: scenes.random_spheres_scene(64, 5, seed=77)}[name]()
    w, h = size
    p = abi.make_params(w, h, depth)
    assert _shares_rays(prims, p)
    tracer.upload_scene(prims, lights)
    ora = oracle_lib.render(prims, lights, p, want_ids=False, want_masks=False)
    img = tracer.render(p, want_ids=False, want_masks=False)["bgr"]
    assert_parity({"bgr": img}, {"bgr": ora["bgr"]}, f"{name} {w}x{h} D{depth} image only")
    for _ in range(3):
        assert np.array_equal(tracer.render(p, want_ids=False, want_masks=False)["bgr"], img)
    assert_parity({"bgr": img}, {"bgr": tracer.render(p)["bgr"]}, "against the launch with ids")
    # three interleaved strips of 8-row bands put together (tiles of a strip are cut differently)
    got = np.zeros_like(img)
    for r in range(3):
        sp = strips_params(p, r, 3)
        got[abi.strip_image_rows(sp)] = tracer.render(sp, want_ids=False, want_masks=False)["bgr"]
    assert_parity({"bgr": got}, {"bgr": ora["bgr"]}, f"{name} strips image only")


def test_glass_depth8(tracer):
    prims, lights = scenes.glass_scene()
    p = abi.make_params(128, 72, 8)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, "glass 128x72 D8")
    # the scene must really exercise the deep tree
    assert int(np.bitwise_count(ora["spawn_mask"]).sum(axis=-1).max()) > 64


@pytest.mark.parametrize("n_spheres", [0, 1, 3, 4, 6, 7, 8, 11])
def test_object_counts(tracer, n_spheres):
    """Scenes of 1..12 objects: the kernel reads the scene in groups of four records plus a
    guarded remainder, so every count modulo 4 (with and without whole groups) is rendered."""
    prims, lights = scenes.glass_scene(n_spheres=n_spheres, n_lights=3, seed=0xC0FFEE + n_spheres)
    assert len(prims) == n_spheres + 1
    p = abi.make_params(96, 64, 6)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, f"{n_spheres} spheres + plane")


def test_plane_box(tracer):
    """A room of six planes (floor, ceiling, four walls, two of them tilted) with spheres
    inside: every ray and every shadow feeler ends on a plane, grazing hits included."""
    k = (0.6, 0.6, 0.6)
    prims = [
        scenes.plane((0, -3, 0), (0, 1, 0), k, reflectivity=0.3),
        scenes.plane((0, 9, 0), (0, -1, 0), k, reflectivity=0.0),
        scenes.plane((-8, 0, 0), (1, 0.05, 0.1), (0.8, 0.3, 0.3), reflectivity=0.5),
        scenes.plane((8, 0, 0), (-1, 0, 0), (0.3, 0.8, 0.3), reflectivity=0.5),
        scenes.plane((0, 0, 30), (0.02, 0, -1), (0.3, 0.3, 0.8), reflectivity=0.9),
        scenes.plane((0, 0, -1), (0, 0, 1), k, reflectivity=0.0),
        scenes.sphere((-2, -1, 12), 2.0, (1, 1, 1), reflectivity=0.9, transparency=0.9, refraction_index=1.5),
        scenes.sphere((3, 0, 16), 3.0, (0.9, 0.8, 0.2), reflectivity=1.0),
    ]
    lights = [scenes.point_light((1, 1, 1), 0.6, (0, 8, 10)), scenes.point_light((1, 0.9, 0.8), 0.3, (-6, 2, 2))]
    prims = scenes._stack(prims, abi.PRIM_DTYPE)
    lights = scenes._stack(lights, abi.LIGHT_DTYPE)
    p = abi.make_params(160, 120, 7)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, "plane box")
    assert int(ora["counters"]["feelers_occluded"]) > 0


# Every way the kernels search the objects: container-order loop or sphere tree, each with and
# without the anchor grids (per light and for the camera: the spheres of one cell of lines).
OBJECT_LOOP_MODES = (
    abi.RT_FLAG_LINEAR | abi.RT_FLAG_NOGRID,
    abi.RT_FLAG_TREE | abi.RT_FLAG_NOGRID,
    abi.RT_FLAG_LINEAR | abi.RT_FLAG_GRID,
    abi.RT_FLAG_TREE | abi.RT_FLAG_GRID,
)


def _with_flags(p, flags):
    q = p.copy()
    q["flags"] = flags
    return q


@pytest.mark.parametrize("name", ["default", "glass", "spheres64", "plane_only", "one_sphere"])
def test_sphere_tree_equals_container_order(tracer, name):
    """The sphere tree (large scenes) against the CPU checker and against the kernel's own
    container-order loop, on scenes of every size, forced with RT_FLAG_TREE."""
    if name == "default":
        prims, lights = scenes.default_scene()
    elif name == "glass":
        prims, lights = scenes.glass_scene()
    elif name == "spheres64":
        prims, lights = scenes.random_spheres_scene(64, 4, seed=0xBEEF)
    elif name == "plane_only":
        prims, lights = scenes.glass_scene(n_spheres=0, n_lights=2)
    else:
        prims, lights = scenes.glass_scene(n_spheres=1, n_lights=2)
    p = abi.make_params(128, 96, 6)
    tracer.upload_scene(prims, lights)
    tree = tracer.render(_with_flags(p, abi.RT_FLAG_TREE | abi.RT_FLAG_NOGRID), want_counters=True)
    ora = oracle_lib.render(prims, lights, p, want_counters=True)
    assert_parity(tree, ora, f"tree {name}")
    for mode in OBJECT_LOOP_MODES:
        other = tracer.render(_with_flags(p, mode), want_counters=True)
        for k in ("hit_id", "hit_mask", "spawn_mask"):
            assert np.array_equal(tree[k], other[k]), (k, mode)
        assert tree["counters"] == other["counters"], mode
        image = tracer.render(_with_flags(p, mode), want_ids=False, want_masks=False)
        assert np.array_equal(image["bgr"], other["bgr"]), mode
        if mode & abi.RT_FLAG_NOGRID:
            assert np.array_equal(tree["bgr"], other["bgr"]), mode
        else:
            # launches that use the anchor grids answer most shadow feelers in single precision
            # (an occluder's transmission factor included): colours to the usual tolerance
            assert_parity(other, ora, f"grid {name} mode {mode}")


def test_sphere_tree_ties_and_first_occluder(tracer):
    """Coincident and nested spheres: equal t at different indices (the smaller index must
    win, src/main.cc:42 is a strict <) and several occluders on one feeler (the one with the
    smallest index decides the shadow, src/objects.h:31-47), met by the tree in another
    order than the container's."""
    rng = scenes.SplitMix64(0x71E5)
    prims = [scenes.plane((0, -3, 0), (0, 1, 0), (0.5, 0.5, 0.5), reflectivity=0.2)]
    base = []
    for _ in range(20):
        c = (rng.uniform(-6, 6), rng.uniform(-2, 4), rng.uniform(8, 24))
        base.append((c, rng.uniform(0.5, 1.6)))
    order = list(range(20)) * 3          # every sphere three times, different materials
    for rep, i in enumerate(order):
        c, r = base[i]
        tr = 0.0 if (rep // 20) == 1 else rng.uniform(0.3, 0.9)
        prims.append(scenes.sphere(c, r, (rng.uniform(0.1, 1), rng.uniform(0.1, 1), rng.uniform(0.1, 1)),
                                   reflectivity=rng.uniform(0, 1), transparency=tr, refraction_index=rng.uniform(1.1, 1.5)))
    for i in range(0, 20, 3):            # and a smaller sphere inside some of them
        c, r = base[i]
        prims.insert(1 + i, scenes.sphere(c, 0.5 * r, (0.9, 0.2, 0.2), transparency=0.5, refraction_index=1.3))
    lights = [scenes.point_light((1, 1, 1), 0.4, (0, 10, 6)), scenes.point_light((1, 1, 1), 0.3, (8, 6, 30)),
              scenes.point_light((1, 1, 1), 0.3, (-9, 3, 16))]
    prims = scenes._stack(prims, abi.PRIM_DTYPE)
    lights = scenes._stack(lights, abi.LIGHT_DTYPE)
    assert len(prims) >= 64
    p = abi.make_params(160, 120, 6)
    dev, ora = _both(tracer, prims, lights, p)            # the tree is the default here
    assert_parity(dev, ora, "ties")
    assert int(ora["counters"]["feelers_occluded"]) > 1000
    tracer.upload_scene(prims, lights)
    lin = tracer.render(_with_flags(p, abi.RT_FLAG_LINEAR), want_counters=True)
    assert lin["counters"] == dev["counters"] and np.array_equal(lin["hit_mask"], dev["hit_mask"])


def _scaled(prims, lights, k):
    pr, lt = prims.copy(), lights.copy()
    pr["p"] = pr["p"] * k
    sph = pr["type"] == abi.RT_PRIM_SPHERE
    q = pr["q"].copy()
    q[sph, 0] = np.float32(q[sph, 0] * k).astype(np.float64)  # radii are floats in the reference
    pr["q"] = q
    lt["pos"] = lt["pos"] * k
    return pr, lt


@pytest.mark.parametrize("k", [1e-3, 37.0, 1e4, 1e7])
@pytest.mark.parametrize("name", ["default", "spheres96"])
def test_scene_scales(tracer, name, k):
    """The same scenes at other length scales (millimetres to 10^7 units): the float
    filters' error bounds scale with the coordinates, the 1e-4 ray offset does not, so
    grazing and self-intersection cases change character; results must stay the checker's."""
    prims, lights = scenes.default_scene() if name == "default" else scenes.random_spheres_scene(96, 3, seed=0x5CA1E)
    prims, lights = _scaled(prims, lights, k)
    p = abi.make_params(112, 80, 5)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, f"{name} x{k}")


def test_many_spheres_and_lights(tracer):
    """4000 spheres, 12 lights: tree depth 12, 48 k (light, object) pairs."""
    prims, lights = scenes.random_spheres_scene(4000, 12, seed=0xB16)
    p = abi.make_params(56, 32, 4)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, "4000 spheres")


def _grazing_points(prims, lights, light, rng, per_sphere=400):
    """Surface points whose feeler towards `light` passes a sphere's silhouette within a
    relative 1e-16 .. 1e-3 (inside and outside), from in front of, behind and far beyond it."""
    P = np.asarray(lights[light]["pos"], dtype=np.float64)
    pts = []
    for pr in prims:
        if pr["type"] != abi.RT_PRIM_SPHERE:
            continue
        c, r = np.asarray(pr["p"], dtype=np.float64), float(pr["q"][0])
        w = c - P
        dist = np.linalg.norm(w)
        if dist <= r * 1.001:
            continue
        wh = w / dist
        v = rng.normal(size=(per_sphere, 3))
        v -= (v @ wh)[:, None] * wh
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        q = c + r * (-(r / dist) * wh + np.sqrt(1 - (r / dist) ** 2) * v)      # silhouette seen from the light
        eps = rng.choice([-1.0, 1.0], per_sphere) * 10.0 ** rng.uniform(-16, -3, per_sphere)
        eps[: per_sphere // 10] = 0.0
        qe = c + (q - c) * (1 + eps)[:, None]
        s_ = rng.choice([0.3, 0.97, 1.5, 4.0, 60.0, 3e3, 1e5], per_sphere)
        pts.append(P + s_[:, None] * (qe - P))
    return np.concatenate(pts)


@pytest.mark.parametrize("name", ["default", "glass", "spheres200"])
def test_shadow_feeler_grazing(tracer, name):
    """The feeler search (light-anchored line test, general filter, box test of the tree,
    double tests) on feelers that graze spheres: the first occluder equals the checker's with
    the container-order loop and with the tree; its transmission factor (colour arithmetic,
    single precision on the device) agrees to 2e-5."""
    if name == "default":
        prims, lights = scenes.default_scene()
    elif name == "glass":
        prims, lights = scenes.glass_scene()
    else:
        prims, lights = scenes.random_spheres_scene(200, 3, seed=0x6A2E)
    tracer.upload_scene(prims, lights)
    rng = np.random.default_rng(99)
    for light in range(min(len(lights), 3)):
        pts = _grazing_points(prims, lights, light, rng, per_sphere=400 if name != "spheres200" else 40)
        pts = np.concatenate([pts, rng.uniform(-8, 8, (2000, 3)) + np.array([0.0, 2.0, 14.0])])
        for self_index in (-1, 0, 1):
            f_o, k_o = oracle_lib.occluder(prims, lights, light, self_index, pts)
            for flags in OBJECT_LOOP_MODES:
                f_d, k_d = tracer.debug_occluder(pts, light, self_index, flags)
                bad = np.nonzero(f_d != f_o)[0]
                assert bad.size == 0, f"{name} light {light} self {self_index} flags {flags}: {bad[:5]} {f_d[bad[:5]]} {f_o[bad[:5]]}"
                # the factor only scales a colour: the device evaluates it in single precision,
                # through the anchor grids from the float distance of the line to the centre
                tol = 2e-6 if flags & abi.RT_FLAG_NOGRID else 1.5e-3
                assert np.allclose(k_d, k_o, rtol=2e-5, atol=tol), float(np.abs(k_d - k_o).max())
        assert (f_o >= 0).any() and (f_o < 0).any()


@pytest.mark.parametrize("name", ["default", "glass6", "contacts"])
def test_single_precision_record_searches(tracer, name):
    """The searches of the pass that finishes pixels ending at the primary ray, on their own
    (RT_DEBUG_RECORDS): wherever they say they are certain their answer is the checker's - the
    first occluder of a feeler (points whose feelers graze spheres within a relative 1e-16..1e-3,
    points on and next to the floor, random points), and what a ray from the camera hits first
    (rays grazing the spheres, rays along and across the planes) - and they are certain for all
    but a small share of ordinary rays."""
    matte = dict(reflectivity=0.0, transparency=0.0)
    if name == "default":
        prims, lights = scenes.default_scene()
    elif name == "glass6":
        prims, lights = scenes.glass_scene(n_spheres=6, n_lights=3)
    else:
        prims = scenes._stack([
            scenes.plane((0, -3, 0), (0, 1, 0), (0.6, 0.6, 0.6), **matte),
            scenes.plane((0, 0, 40), (0.3, 0.2, -1), (0.8, 0.4, 0.3), **matte),
            scenes.sphere((-4, -2, 12), 1.0, (1, 1, 1), reflectivity=0.2, transparency=0.8, refraction_index=1.4),
            scenes.sphere((0, -3, 14), 1.5, (0.9, 0.2, 0.2), **matte),
            scenes.sphere((4, -1.999999, 10), 1.0, (0.2, 0.9, 0.2), reflectivity=0.6),
        ], abi.PRIM_DTYPE)
        lights = scenes._stack([scenes.point_light((1, 1, 1), 0.6, (0, 8, 10)), scenes.point_light((1, 0.9, 0.8), 0.3, (-6, 2, 2)),
                                scenes.point_light((1, 1, 1), 0.4, (2.0001, -1, 12))], abi.LIGHT_DTYPE)
    tracer.upload_scene(prims, lights)
    rng = np.random.default_rng(4242)
    planes = [i for i, pr in enumerate(prims) if pr["type"] == abi.RT_PRIM_PLANE]
    sure = total = 0
    for light in range(min(len(lights), 3)):
        pts = _grazing_points(prims, lights, light, rng, per_sphere=400)
        floor = np.stack([rng.uniform(-20, 20, 4000), np.full(4000, -3.0) + rng.choice([0.0, 1e-12, -1e-12, 1e-6], 4000),
                          rng.uniform(2, 60, 4000)], 1)
        pts = np.concatenate([pts, floor, rng.uniform(-8, 8, (2000, 3)) + np.array([0.0, 2.0, 14.0])])
        for self_index in [-1] + planes[:1]:
            f_o, k_o = oracle_lib.occluder(prims, lights, light, self_index, pts)
            f_d, k_d = tracer.debug_occluder(pts, light, self_index, abi.RT_DEBUG_RECORDS)
            certain = f_d != -2
            bad = np.nonzero(certain & (f_d != f_o))[0]
            assert bad.size == 0, f"{name} light {light} self {self_index}: {bad[:5]} {f_d[bad[:5]]} {f_o[bad[:5]]}"
            assert np.allclose(k_d[certain], k_o[certain], rtol=2e-5, atol=1.5e-3), float(np.abs(k_d - k_o)[certain].max())
            if self_index >= 0:  # feelers that start on the floor with the floor skipped, and random points
                sure += int(certain[-6000:].sum())
                total += 6000
    assert sure > 0.97 * total, (sure, total)
    # primary rays: all from the camera
    rays = _grazing_rays(prims, rng, per_sphere=600, origin=(0.0, 0.0, 0.0))
    m = 6000
    d = np.stack([rng.uniform(-0.7, 0.7, m), rng.uniform(-0.4, 0.4, m) * rng.choice([1.0, 1e-3, 1e-6], m), np.ones(m)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([rays, np.concatenate([np.zeros((m, 3)), d], axis=1)])
    i_o, t_o = oracle_lib.nearest(prims, rays)
    i_d, t_d = tracer.debug_nearest(rays, abi.RT_DEBUG_RECORDS)
    certain = i_d != -2
    bad = np.nonzero(certain & (i_d != i_o))[0]
    assert bad.size == 0, f"{name}: {bad[:5]} {i_d[bad[:5]]} {i_o[bad[:5]]}"
    hit = certain & (i_d >= 0)
    assert np.all(np.isin(i_d[hit], planes))  # it only ever names a plane
    assert np.allclose(t_d[hit], t_o[hit], rtol=1e-9)
    # (with two planes in view most rays hit both: which is nearer is left to the walk)
    assert certain[-m:].mean() > (0.2 if name == "contacts" else 0.5) and (i_d == -1).any()


def _grazing_rays(prims, rng, per_sphere=300, origin=None):
    """Unit rays from random origins (or all from `origin`) that pass a sphere's silhouette
    within a relative 1e-16 .. 1e-3, half of them pointing away from it, plus rays grazing the
    planes."""
    rays = []
    for pr in prims:
        if pr["type"] != abi.RT_PRIM_SPHERE:
            continue
        c, r = np.asarray(pr["p"], dtype=np.float64), float(pr["q"][0])
        o = c + rng.normal(size=(per_sphere, 3)) * rng.choice([2.0, 8.0, 300.0, 2e4], (per_sphere, 1)) * max(r, 1.0)
        if origin is not None:
            o = np.tile(np.asarray(origin, dtype=np.float64), (per_sphere, 1))
        w = c - o
        dist = np.linalg.norm(w, axis=1, keepdims=True)
        ok = (dist[:, 0] > 1.01 * r)
        o, w, dist = o[ok], w[ok], dist[ok]
        wh = w / dist
        v = rng.normal(size=w.shape)
        v -= np.sum(v * wh, axis=1, keepdims=True) * wh
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        q = c + r * (-(r / dist) * wh + np.sqrt(1 - (r / dist) ** 2) * v)
        eps = rng.choice([-1.0, 1.0], len(o)) * 10.0 ** rng.uniform(-16, -3, len(o))
        qe = c + (q - c) * (1 + eps)[:, None]
        d = qe - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[::2] *= 1.0  # towards the sphere
        d[1::4] *= -1.0  # a quarter away from it: only what lies behind may be hit
        rays.append(np.concatenate([o, d], axis=1))
    out = np.concatenate(rays)
    if origin is not None:
        return out
    # rays almost parallel to the floor plane y = -3, from just above and just below it
    m = 2000
    o = np.stack([rng.uniform(-20, 20, m), -3 + rng.choice([-1, 1], m) * 10.0 ** rng.uniform(-9, 0, m), rng.uniform(0, 40, m)], 1)
    d = np.stack([rng.normal(size=m), rng.choice([-1, 1], m) * 10.0 ** rng.uniform(-8, -1, m), rng.normal(size=m)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([out, np.concatenate([o, d], axis=1)])


@pytest.mark.parametrize("name", ["default", "glass", "spheres200", "ties"])
def test_nearest_hit_grazing(tracer, name):
    """The nearest-hit search on rays that graze sphere silhouettes and the floor plane:
    index and t equal the checker's bit for bit, with the container-order loop and with the
    tree (whose box test must never drop an object the reference hits)."""
    if name == "default":
        prims, lights = scenes.default_scene()
    elif name == "glass":
        prims, lights = scenes.glass_scene()
    elif name == "spheres200":
        prims, lights = scenes.random_spheres_scene(200, 3, seed=0x6A2E)
    else:  # every sphere twice: equal t at two indices
        prims, lights = scenes.glass_scene(12, 2, seed=5)
        prims = np.concatenate([prims, prims[1:]])
    tracer.upload_scene(prims, lights)
    rays = _grazing_rays(prims, np.random.default_rng(7), per_sphere=300 if name != "spheres200" else 60)
    # the same kind of rays from the camera (origin exactly zero): these take the camera's anchor grid
    cam = _grazing_rays(prims, np.random.default_rng(8), per_sphere=300 if name != "spheres200" else 60, origin=(0.0, 0.0, 0.0))
    cam = cam[: len(cam) // 64 * 64]  # whole waves of camera rays
    rays = np.concatenate([cam, rays])
    i_o, t_o = oracle_lib.nearest(prims, rays)
    for flags in OBJECT_LOOP_MODES:
        i_d, t_d = tracer.debug_nearest(rays, flags)
        bad = np.nonzero(i_d != i_o)[0]
        assert bad.size == 0, f"{name} flags {flags}: rays {bad[:5]} device {i_d[bad[:5]]} checker {i_o[bad[:5]]}"
        assert np.array_equal(t_d.view(np.uint64), t_o.view(np.uint64))
    assert (i_o >= 0).any() and (i_o < 0).any()


@pytest.mark.parametrize("band_rows,step", [(1, 2), (8, 2), (8, 8), (16, 3), (5, 4)])
def test_strips_reassemble(tracer, band_rows, step):
    """Rendering the frame as `step` interleaved strips gives the same bytes as the
    whole frame (what the multi-GPU partition relies on)."""
    prims, lights = scenes.default_scene()
    w, h, d = 200, 117, 5
    tracer.upload_scene(prims, lights)
    full = tracer.render(abi.make_params(w, h, d))
    n_bands = -(-h // band_rows)
    got = {k: np.zeros_like(v) for k, v in full.items() if v is not None and k != "counters"}
    for r in range(step):
        cnt = len(range(r, n_bands, step))
        p = abi.make_params(w, h, d, band_rows=band_rows, band_first=r, band_step=step, band_count=cnt)
        part = tracer.render(p)
        rows = abi.strip_image_rows(p)
        assert len(rows) == part["bgr"].shape[0]
        for k in got:
            got[k][rows] = part[k]
    for k in got:
        assert np.array_equal(got[k], full[k]), k


def test_edge_cases(tracer):
    prims, lights = scenes.default_scene()
    # no lights, a single object, 1x1 frame
    tracer.upload_scene(prims[:1], lights[:0])
    p = abi.make_params(1, 1, 2)
    dev = tracer.render(p, want_counters=True)
    ora = oracle_lib.render(prims[:1], lights[:0], p, want_counters=True)
    assert_parity(dev, ora, "1x1 no lights")
    # empty partition
    tracer.upload_scene(prims, lights)
    p = abi.make_params(32, 32, 4, band_rows=8, band_first=9, band_step=1, band_count=2)
    dev = tracer.render(p)
    assert dev["bgr"].shape == (0, 32, 3)
    # camera inside a sphere, ray origin on a plane
    inside = prims.copy()
    inside[1]["p"] = (0.0, 0.0, 0.5)
    inside[0]["p"] = (0.0, 0.0, 0.0)
    p = abi.make_params(48, 48, 6)
    dev, ora = _both(tracer, inside, lights, p)
    assert_parity(dev, ora, "camera inside sphere")


def test_errors(tracer):
    from raytracer_amd.device import RtError

    prims, lights = scenes.default_scene()
    tracer.upload_scene(prims, lights)
    with pytest.raises(RtError):
        tracer.render(abi.make_params(16, 16, 9))
    with pytest.raises(RtError):
        tracer.render(abi.make_params(0, 16, 4))
    bad = prims.copy()
    bad[0]["type"] = 7
    with pytest.raises(RtError):
        tracer.upload_scene(bad, lights)
    for field, value in (("p", np.nan), ("q", np.inf)):
        bad = prims.copy()
        bad[1][field][0] = value
        with pytest.raises(RtError):
            tracer.upload_scene(bad, lights)
    badl = lights.copy()
    badl[0]["pos"][2] = -np.inf
    with pytest.raises(RtError):
        tracer.upload_scene(prims, badl)


@pytest.mark.parametrize("n_spheres", [6, 60])
def test_zero_and_negative_radii(tracer, n_spheres):
    """The reference only ever squares a radius: a negative one behaves like its absolute
    value, a zero one is a point that is practically never hit.  Both object-loop modes."""
    prims, lights = scenes.random_spheres_scene(n_spheres, 3, seed=0x2E60)
    prims = prims.copy()
    q = prims["q"].copy()
    q[1::3, 0] *= -4.0      # every third sphere negative (and bigger, so that it shows)
    q[2::7, 0] = 0.0
    prims["q"] = q
    p = abi.make_params(120, 80, 5)
    dev, ora = _both(tracer, prims, lights, p)
    assert_parity(dev, ora, f"radii {n_spheres}")
    assert int(ora["counters"]["hits"]) > int(ora["counters"]["primary_hits"]) > 0
    tracer.upload_scene(prims, lights)


@pytest.mark.parametrize("name", ["default", "random", "glass"])
def test_pruned_ray_tree(tracer, name):
    """RT_FLAG_PRUNE: ids, masks and counters of the walk without the children under a
    material coefficient of zero equal the checker's in the same mode; the image is the one of
    the full ray tree (those children are multiplied by zero), and image-only renders - which
    always leave them out - give it too."""
    if name == "default":
        prims, lights = scenes.default_scene()
    elif name == "random":
        prims, lights = scenes.random_spheres_scene(30, 3, seed=0x9A1)
        prims = prims.copy()
        r, t = prims["reflectivity"].copy(), prims["transparency"].copy()
        r[::3] = 0.0
        t[1::4] = 0.0
        prims["reflectivity"], prims["transparency"] = r, t
    else:
        prims, lights = scenes.glass_scene(8, 3)
    p = abi.make_params(160, 96, 6)
    full_dev, full_ora = _both(tracer, prims, lights, p)
    assert_parity(full_dev, full_ora, f"full tree {name}")
    pp = p.copy()
    pp["flags"] = abi.RT_FLAG_PRUNE
    dev = tracer.render(pp, want_counters=True)
    ora = oracle_lib.render(prims, lights, pp, want_counters=True)
    assert_parity(dev, ora, f"pruned tree {name}")
    assert np.array_equal(dev["bgr"], full_dev["bgr"])
    assert np.array_equal(ora["bgr"], full_ora["bgr"])
    assert int(ora["counters"]["rays"]) <= int(full_ora["counters"]["rays"])
    if name != "glass":
        assert int(ora["counters"]["rays"]) < int(full_ora["counters"]["rays"])


@pytest.mark.parametrize("seed", range(8))
def test_random_materials_depths_and_modes(tracer, seed):
    """Random small scenes whose materials mix zero and non-zero reflectivity / transparency,
    at a random depth, in both object-loop modes: full ray tree and pruned ray tree against the
    checker in the same mode, and the image-only render (which prunes) against the full image."""
    rng = np.random.default_rng(0xF0221 + seed)
    n = int(rng.integers(2, 14))
    prims, lights = scenes.random_spheres_scene(n, int(rng.integers(0, 5)), seed=0x1000 + seed)
    prims = prims.copy()
    r, t = prims["reflectivity"].copy(), prims["transparency"].copy()
    r[rng.random(len(r)) < 0.4] = 0.0
    t[rng.random(len(t)) < 0.4] = 0.0
    t[rng.random(len(t)) < 0.1] = 1.0
    prims["reflectivity"], prims["transparency"] = r, t
    depth = int(rng.integers(2, 9))
    w, h = int(rng.integers(20, 90)), int(rng.integers(20, 70))
    for mode in OBJECT_LOOP_MODES:
        p = abi.make_params(w, h, depth)
        p["flags"] = mode
        dev, ora = _both(tracer, prims, lights, p)
        assert_parity(dev, ora, f"fuzz {seed} depth {depth} mode {mode} full")
        pp = p.copy()
        pp["flags"] = mode | abi.RT_FLAG_PRUNE
        cut = tracer.render(pp, want_counters=True)
        cut_ora = oracle_lib.render(prims, lights, pp, want_counters=True)
        assert_parity(cut, cut_ora, f"fuzz {seed} depth {depth} mode {mode} pruned")
        assert np.array_equal(cut["bgr"], dev["bgr"])
