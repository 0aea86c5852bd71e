"""The BASELINE.json scenes, built through the API mirror exactly as the .pine scripts build them.

cbox: /root/reference/scenes/cbox.pine:4-21 (SURVEY.md 8(d) "Concrete inputs").
"""
import numpy as np

from .api import (Scene, Diffuse, Emissive, Uber, Subsurface, Rect, Box, AABB, Sphere, Disk, Cone, Mesh,
                  Film, Uncharted2, ThinLenCamera, translate, rotate_y, scale)

CAMERAS = {
    # name: (from, to, fov)
    "committed": ([0, 0, 0], [0, 0, 1], 0.4),   # scenes/cbox.pine:21
    "readme": ([0, 1, -4], [0, 1, 0], 0.25),    # README.md:32 / cbox.pine:20 (commented)
}


def cbox(size=(640, 640), camera="committed", boxes=True, lamp=True):
    """scenes/cbox.pine; boxes=False gives the Rect-only variant (no order-dependent OBBs)."""
    scene = Scene()
    scene.add("floor", Diffuse([0.9, 0.9, 0.9]))
    scene.add("blue", Diffuse([0.2, 0.5, 0.9]))
    scene.add("red", Diffuse([0.9, 0.1, 0.05]))
    scene.add("green", Diffuse([0.2, 0.9, 0.05]))

    scene.add(Rect([0, 0, 1], [2, 0, 0], [0, 0, 2], True), "floor")
    scene.add(Rect([0, 2, 1], [2, 0, 0], [0, 0, 2]), "floor")
    scene.add(Rect([-1, 1, 1], [0, 0, 2], [0, 2, 0], True), "red")
    scene.add(Rect([1, 1, 1], [0, 0, 2], [0, 2, 0]), "green")
    scene.add(Rect([0, 1, 2], [2, 0, 0], [0, 2, 0], True), "blue")
    if boxes:
        scene.add(Box(AABB([0, 0, 0], [1, 1, 1]),
                      translate([0.0, 0.0, 0.6]) * rotate_y(0.4) * scale([0.6, 0.6, 0.6])), "floor")
        scene.add(Box(AABB([0, 0, 0], [1, 1, 1]),
                      translate([-0.6, 0.0, 1.0]) * rotate_y(-0.4) * scale([0.6, 1.3, 0.6])), "floor")
    # 600 * [1.0, 0.64, 0.185] in PRL is float32 arithmetic
    le = (np.float32(600) * np.array([1.0, 0.64, 0.185], dtype=np.float32)).tolist()
    if lamp:
        scene.add(Rect([0.0, 1.9, 1], [0.1, 0, 0], [0, 0, 0.1]), Emissive(le))
    frm, to, fov = CAMERAS[camera]
    scene.set(ThinLenCamera(Film(list(size), Uncharted2()), frm, to, fov))
    return scene


def cbox_clutter(size=(48, 48), extra=24, seed=31):
    """cbox (README camera) plus `extra` random primitives of the kinds whose image depends on the accel's test order --
    rotated + scaled Boxes, Spheres, scaled Boxes (bbox.cpp:149-171): what separates pine-BVH order from EmbreeAccel's once a
    scene has more primitives than one BVH8 node (tools/embree_order_distance.py, tests/golden/film_embree_clutter*)."""
    rng = np.random.default_rng(seed)
    scene = cbox(size, "readme")
    for i in range(extra):
        c = rng.uniform([-0.8, 0.1, 0.3], [0.8, 1.6, 1.8]).tolist()
        if i % 3 == 0:
            scene.add(Box(AABB([0, 0, 0], [1, 1, 1]),
                          translate(c) * rotate_y(float(rng.uniform(-1, 1))) * scale(rng.uniform(0.1, 0.35, 3).tolist())), "floor")
        elif i % 3 == 1:
            scene.add(Sphere(c, float(rng.uniform(0.05, 0.2))), "red")
        else:
            scene.add(Box(AABB([0, 0, 0], [1, 1, 1]), translate(c) * scale(rng.uniform(0.1, 0.3, 3).tolist())), "green")
    return scene


def classic_cones(size=(720, 360), n=100, with_spheres=True, checker_floor=False):
    """Config C4: scenes/classic.pine:4-18 materials/shapes, plus n x n procedurally
    placed cones Cone([x,0,z], Y, 0.05, 0.05).  checker_floor=True keeps the script's node-graph floor
    (lerp over Checkerboard(UV(), 0.95) for albedo and roughness, classic.pine:4-7); the default is the
    constant-node floor the round-1 fixtures were rendered with."""
    scene = Scene()
    if checker_floor:
        from .api import Checkerboard, UV, lerp
        scene.add("floor", Uber(lerp(Checkerboard(UV(), 0.95), [0.5, 0.7, 1.0], [0.01, 0.02, 0.03]),
                                lerp(Checkerboard(UV(), 0.95), 0.0, 0.4)))
    else:
        scene.add("floor", Uber([0.5, 0.7, 1.0], 0.4))
    scene.add("diffuse", Diffuse([0.8, 0.8, 0.8]))
    scene.add("metal", Uber([1.0, 1.0, 1.0], 0.0, 1.0))
    scene.add("glossy", Uber([0.98, 0.55, 0.02], 0.0, 0.0))
    scene.add(Disk([0, 0, 0], [0, 1, 0], 100), "floor")
    if with_spheres:
        scene.add(Sphere([-3, 1, 0], 1), "metal")
        scene.add(Sphere([0, 1, 0], 1), "diffuse")
        scene.add(Sphere([3, 1, 0], 1), "glossy")
    le = (np.array([1, 1, 1], dtype=np.float32) * np.float32(160)).tolist()
    scene.add(Rect([-1, 3, -1], [1, 0, 0], [0, 0, 1]), Emissive(le))
    half = np.float32(0.1) * np.float32(n) / np.float32(2)
    for i in range(n):
        for j in range(n):
            x = float(np.float32(-half + np.float32(0.05)) + np.float32(0.1) * np.float32(i))
            z = float(np.float32(-half + np.float32(0.05)) + np.float32(0.1) * np.float32(j))
            scene.add(Cone([x, 0, z], [0, 1, 0], 0.05, 0.05), "diffuse")
    scene.set(ThinLenCamera(Film(list(size)), [0, 4, -8], [0, 1, 0], 0.3))
    return scene


def icosphere(subdiv=3, radius=0.4, center=(0.0, 0.5, 1.0)):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    verts = (np.array(v) * radius + np.array(center)).astype(np.float32)
    return verts, np.array(f, dtype=np.uint32)


def sss(size=(640, 640), subdiv=3, camera="readme", skin=None, emissive_mesh=False):
    """Config C5: Rect-only cbox room + one closed icosphere mesh (1280 triangles at
    subdiv=3) with Subsurface([1,1,1], 0.0, [40,40,40]) + the cbox emissive Rect."""
    scene = Scene()
    scene.add("floor", Diffuse([0.9, 0.9, 0.9]))
    scene.add("blue", Diffuse([0.2, 0.5, 0.9]))
    scene.add("red", Diffuse([0.9, 0.1, 0.05]))
    scene.add("green", Diffuse([0.2, 0.9, 0.05]))
    scene.add("skin", skin if skin is not None else Subsurface([1, 1, 1], 0.0, [40, 40, 40]))
    scene.add(Rect([0, 0, 1], [2, 0, 0], [0, 0, 2], True), "floor")
    scene.add(Rect([0, 2, 1], [2, 0, 0], [0, 0, 2]), "floor")
    scene.add(Rect([-1, 1, 1], [0, 0, 2], [0, 2, 0], True), "red")
    scene.add(Rect([1, 1, 1], [0, 0, 2], [0, 2, 0]), "green")
    scene.add(Rect([0, 1, 2], [2, 0, 0], [0, 2, 0], True), "blue")
    verts, faces = icosphere(subdiv)
    scene.add(Mesh(verts, faces), "skin")
    if emissive_mesh:  # a second, small mesh as an area light (Mesh::sample picks a triangle, geometry.h:170-178)
        v2, f2 = icosphere(1, 0.12, (-0.5, 1.5, 0.6))
        scene.add(Mesh(v2, f2), Emissive([30.0, 24.0, 12.0]))
    le = (np.float32(600) * np.array([1.0, 0.64, 0.185], dtype=np.float32)).tolist()
    scene.add(Rect([0.0, 1.9, 1], [0.1, 0, 0], [0, 0, 0.1]), Emissive(le))
    frm, to, fov = CAMERAS[camera]
    scene.set(ThinLenCamera(Film(list(size), Uncharted2()), frm, to, fov))
    return scene


def shapes_zoo(size=(64, 64)):
    """One of every supported analytic shape + materials: used by the per-shape parity fixtures."""
    scene = Scene()
    scene.add("d", Diffuse([0.8, 0.7, 0.6]))
    scene.add("u", Uber([0.9, 0.6, 0.3], 0.3, 0.0, 0.0))
    scene.add(Rect([0, 0, 1], [2, 0, 0], [0, 0, 2], True), "d")
    scene.add(Box([-0.9, 0.0, 0.2], [-0.5, 0.5, 0.6]), "d")
    scene.add(Box(AABB([0, 0, 0], [1, 1, 1]),
                  translate([0.0, 0.0, 0.6]) * rotate_y(0.4) * scale([0.6, 0.6, 0.6])), "u")
    scene.add(Sphere([0.5, 0.3, 1.2], 0.3), "u")
    scene.add(Disk([0.0, 1.5, 1.0], [0.2, -1.0, 0.1], 0.4), "d")
    scene.add(Cone([-0.3, 0.0, 1.4], [0, 1, 0], 0.2, 0.5), "d")
    le = (np.float32(600) * np.array([1.0, 0.64, 0.185], dtype=np.float32)).tolist()
    scene.add(Rect([0.0, 1.9, 1], [0.1, 0, 0], [0, 0, 0.1]), Emissive(le))
    scene.set(ThinLenCamera(Film(list(size), Uncharted2()), [0, 1, -4], [0, 1, 0], 0.25))
    return scene


def materials_zoo(size=(64, 64)):
    """Every material kind of material.h:18-131 except Subsurface, with node-graph parameters: a Rect-only
    room whose floor is a checkerboard over Position, Metal / Glossy / Glass spheres, an Uber sphere whose
    roughness and metallic vary with the normal, a Diffuse back wall tinted by fract(Position)."""
    from .api import (Checkerboard, Position, Normal, UV, Metal, Glossy, Glass, lerp, node_fract, node_abs, Vec3)
    scene = Scene()
    scene.add("floor", Diffuse(lerp(Checkerboard(Position() * 2.0, 0.5), [0.85, 0.85, 0.85], [0.15, 0.2, 0.6])))
    scene.add("wall", Diffuse([0.8, 0.8, 0.8]))
    # (UV is read on a Rect only: Rect / Disk / triangle uv are plain arithmetic, a Sphere's uv goes through
    #  atan2f / acosf, where the device libm may differ from glibc in the last bit -- DESIGN.md 1.3)
    scene.add("back", Diffuse(lerp(Checkerboard(UV() * 4.0, 0.5), node_fract(Position() * 1.5) * 0.5 + [0.3, 0.3, 0.3],
                                   [0.1, 0.1, 0.1])))
    scene.add("metal", Metal([1.0, 0.8, 0.6], 0.2))
    scene.add("glossy", Glossy([0.9, 0.25, 0.2] * node_abs(Normal())[1] + [0.05, 0.05, 0.05], 0.1, 1.5))
    scene.add("glass", Glass([1.0, 1.0, 1.0], 0.0, 1.5))
    absn = node_abs(Normal())
    scene.add("uber", Uber(Vec3(absn[0] * 0.15, 0.5, absn[2] * 0.3) + [0.2, 0.2, 0.2], absn[0] * 0.5 + 0.05,
                           Checkerboard(Position() * 3.0, 0.5)))
    scene.add(Rect([0, 0, 1], [2, 0, 0], [0, 0, 2], True), "floor")
    scene.add(Rect([0, 2, 1], [2, 0, 0], [0, 0, 2]), "wall")
    scene.add(Rect([-1, 1, 1], [0, 0, 2], [0, 2, 0], True), "wall")
    scene.add(Rect([1, 1, 1], [0, 0, 2], [0, 2, 0]), "wall")
    scene.add(Rect([0, 1, 2], [2, 0, 0], [0, 2, 0], True), "back")
    scene.add(Sphere([-0.55, 0.3, 1.2], 0.3), "metal")
    scene.add(Sphere([0.1, 0.25, 0.8], 0.25), "glass")
    scene.add(Sphere([0.6, 0.3, 1.3], 0.3), "glossy")
    scene.add(Sphere([-0.1, 0.9, 1.5], 0.3), "uber")
    scene.add(Rect([0.0, 1.9, 1], [0.5, 0, 0], [0, 0, 0.5]), Emissive([20.0, 18.0, 15.0]))
    scene.set(ThinLenCamera(Film(list(size)), [0, 1, -4], [0, 1, 0], 0.25))
    return scene


def lights_zoo(size=(64, 64), with_sky=True):
    """cbox geometry (README camera) lit by every light kind of light.h: the emissive Rect (area light), a
    PointLight added BEFORE it, a SpotLight and a DirectionalLight after, and the Sky environment light --
    the light sampler's list order is the add order, the environment light last (lightsampler.cpp:6-10)."""
    from .api import PointLight, SpotLight, DirectionalLight, Sky
    scene = cbox(size, "readme", lamp=False)
    scene.add(PointLight([0.5, 1.5, 0.5], [2.5, 2.0, 1.5]))
    scene.add(Rect([0.0, 1.9, 1], [0.1, 0, 0], [0, 0, 0.1]), Emissive((np.array([1.0, 0.64, 0.185], dtype=np.float32) * np.float32(600)).tolist()))
    scene.add(SpotLight([-0.5, 1.75, 1.0], [0.25, -1.0, 0.125], [5.0, 5.0, 4.0], 0.4, 0.2))
    scene.add(DirectionalLight([0.5, 1.0, -1.0], [0.4, 0.4, 0.6]))
    if with_sky:
        scene.set(Sky([0.9, 0.9, 1.0]))
    return scene


def xshapes_zoo(size=(64, 64), extra_lights=True):
    """The analytic shapes of geometry.h that no BASELINE scene uses: a Plane floor (UV checkerboard), a
    Cylinder (side surface only; its bounding box is the reference's Sphere(p0, r) box, geometry.h:141), a
    thick Line, a stand-alone Triangle -- lit by a Rect lamp and, with extra_lights, by an emissive Triangle,
    an emissive Line and a dim emissive Plane above (Plane::sample / Line::sample / Triangle::sample)."""
    from .api import Plane, Line, Cylinder, Triangle, Metal, Checkerboard, UV, lerp
    scene = Scene()
    scene.add("floor", Diffuse(lerp(Checkerboard(UV() * 2.0, 0.5), [0.8, 0.8, 0.8], [0.2, 0.3, 0.6])))
    scene.add("white", Diffuse([0.85, 0.85, 0.85]))
    scene.add("red", Diffuse([0.9, 0.15, 0.1]))
    scene.add("metal", Metal([0.9, 0.8, 0.5], 0.15))
    scene.add(Plane([0, 0, 0], [0, 1, 0]), "floor")
    scene.add(Rect([0, 1, 2.2], [3, 0, 0], [0, 2, 0], True), "white")
    scene.add(Cylinder([-0.6, 0.25, 1.2], [-0.6, 1.25, 1.2], 0.25), "red")
    scene.add(Cylinder([0.75, 0.3, 0.6], [0.55, 0.3, 0.9], 0.3), "metal")
    scene.add(Line([0.1, 0.1, 0.9], [0.8, 0.9, 1.5], 0.06), "metal")
    scene.add(Line([-0.9, 0.05, 0.5], [0.9, 0.05, 0.4], 0.04), "white")
    scene.add(Triangle([-0.3, 0.0, 1.8], [0.5, 0.0, 1.9], [0.1, 1.1, 1.7]), "red")
    scene.add(Sphere([0.0, 0.2, 1.0], 0.2), "white")
    scene.add(Rect([0.0, 1.9, 1], [0.5, 0, 0], [0, 0, 0.5]), Emissive([20.0, 18.0, 15.0]))
    if extra_lights:
        scene.add(Triangle([-1.0, 1.2, 1.0], [-1.0, 1.6, 1.4], [-1.0, 1.6, 0.8]), Emissive([12.0, 4.0, 2.0]))
        scene.add(Line([0.9, 0.6, 1.9], [0.9, 1.4, 1.9], 0.03), Emissive([2.0, 8.0, 14.0]))
        scene.add(Plane([0, 6, 0], [0, -1, 0]), Emissive([0.15, 0.2, 0.3]))
    scene.set(ThinLenCamera(Film(list(size)), [0, 1, -4], [0, 1, 0], 0.25))
    return scene


def random_scene(seed, variety=False):
    """A seeded random mix of every shape, material and light kind in a two-wall room (fuzzing: tools/fuzz_scenes.py,
    tests): returns (scene, spp, max_path_length)."""
    import pine_amd as pa
    r = np.random.default_rng(seed)
    f = lambda lo, hi: float(np.float32(r.uniform(lo, hi)))
    v = lambda lo, hi: [f(lo, hi), f(lo, hi), f(lo, hi)]
    s = pa.Scene()
    mats = []
    nodes = r.random() < 0.4
    for i in range(int(r.integers(2, 6))):
        k = int(r.integers(0, 6))
        col = v(0.1, 0.95)
        if nodes and r.random() < 0.5:
            col = pa.lerp(pa.Checkerboard(pa.Position() * f(1, 4), 0.5), col, v(0.05, 0.9))
        if k == 0: m = pa.Diffuse(col)
        elif k == 1: m = pa.Metal(col, f(0.0, 0.6))
        elif k == 2: m = pa.Glossy(col, f(0.0, 0.5), f(1.1, 1.8))
        elif k == 3: m = pa.Glass(col, f(0.0, 0.3), f(1.1, 1.8))
        elif k == 4 and variety == 2:  # fractional metallic / transmission: the lobe choice draws from the pixel's RNG (sampler.h:317-324)
            m = pa.Uber(col, f(0.0, 0.8), f(0.0, 1.0), f(0.0, 1.0), f(1.1, 1.8))
        elif k == 4: m = pa.Uber(col, f(0.0, 0.8), float(r.integers(0, 2)), 0.0)
        else: m = pa.Diffuse(col)
        s.add(f"m{i}", m)
        mats.append(f"m{i}")
    pick = lambda: mats[int(r.integers(0, len(mats)))]
    # a room so that paths bounce
    s.add(pa.Rect([0, 0, 1], [3, 0, 0], [0, 0, 3], True), pick())
    s.add(pa.Rect([0, 1, 2.5], [3, 0, 0], [0, 2, 0], True), pick())
    for i in range(int(r.integers(3, 10))):
        k = int(r.integers(0, 11))
        c = [f(-1, 1), f(0.1, 1.2), f(0.6, 2.0)]
        m = pick()
        if k == 0: s.add(pa.Sphere(c, f(0.1, 0.4)), m)
        elif k == 1: s.add(pa.Disk(c, v(-1, 1), f(0.1, 0.5)), m)
        elif k == 2: s.add(pa.Cone([c[0], 0.0, c[2]], [0, 1, 0], f(0.1, 0.3), f(0.2, 0.8)), m)
        elif k == 3: s.add(pa.AABB(c, [c[0] + f(0.1, 0.5), c[1] + f(0.1, 0.5), c[2] + f(0.1, 0.5)]), m)
        elif k == 4: s.add(pa.Box(pa.AABB([0, 0, 0], [1, 1, 1]), pa.translate(c) * pa.rotate_y(f(-1, 1)) * pa.scale(v(0.2, 0.6))), m)
        elif k == 5: s.add(pa.Rect(c, v(-0.6, 0.6), v(-0.6, 0.6)), m) if True else None
        elif k == 6: s.add(pa.Line(c, [c[0] + f(-0.8, 0.8), c[1] + f(-0.3, 0.8), c[2] + f(-0.5, 0.5)], f(0.02, 0.08)), m)
        elif k == 7: s.add(pa.Cylinder(c, [c[0] + f(-0.3, 0.3), c[1] + f(0.2, 0.8), c[2] + f(-0.3, 0.3)], f(0.1, 0.3)), m)
        elif k == 8: s.add(pa.Triangle(c, [c[0] + f(0.2, 0.8), c[1], c[2] + f(-0.3, 0.3)], [c[0] + f(-0.2, 0.4), c[1] + f(0.3, 0.9), c[2]]), m)
        elif k == 9:
            vs = np.float32([c, [c[0] + 0.5, c[1], c[2]], [c[0] + 0.5, c[1] + 0.5, c[2] + 0.1], [c[0], c[1] + 0.5, c[2] + 0.1]])
            s.add(pa.Mesh(vs, np.uint32([[0, 1, 2], [0, 2, 3]])), m)
        else: s.add(pa.Plane([0, f(-0.2, 0.0), 0], [f(-0.1, 0.1), 1, f(-0.1, 0.1)]), m)
    if variety == 2 and r.random() < 0.6:  # a Subsurface icosphere (BSSRDF random walk inside a mesh: megakernel path)
        vs, fs = icosphere(1, f(0.2, 0.4), (f(-0.6, 0.6), f(0.4, 0.9), f(0.8, 1.6)))
        s.add(pa.Mesh(vs, fs), pa.Subsurface(v(0.5, 0.95), f(0.1, 0.5), v(5.0, 40.0)))
    # lights: an area lamp always, others sometimes (add order matters to the light sampler)
    if r.random() < 0.4: s.add(pa.PointLight(v(-0.8, 1.8), v(1, 4)))
    s.add(pa.Rect([f(-0.5, 0.5), 1.9, f(0.6, 1.6)], [f(0.2, 0.6), 0, 0], [0, 0, f(0.2, 0.6)]), pa.Emissive(v(8, 25)))
    if r.random() < 0.3: s.add(pa.SpotLight(v(-0.8, 1.8), [f(-0.3, 0.3), -1.0, f(-0.3, 0.3)], v(2, 6), f(0.2, 0.6), f(0.05, 0.2)))
    if r.random() < 0.3: s.add(pa.DirectionalLight([f(-1, 1), 1.0, f(-1, 0)], v(0.2, 0.8)))
    if r.random() < 0.3: s.add(pa.Sphere([f(-0.8, 0.8), f(1.0, 1.6), f(0.8, 1.6)], f(0.05, 0.15)), pa.Emissive(v(5, 20)))
    if r.random() < 0.3: s.set(pa.Sky(v(0.3, 1.0)))
    if variety:  # also: odd film sizes, a thin lens, SobolSampler (returns a fourth element, the sampler name)
        size = [int(r.integers(1, 40)), int(r.integers(1, 30))]
        lens = (f(0.01, 0.08), f(2.0, 6.0)) if r.random() < 0.4 else (0.0, 1.0)
        s.set(pa.ThinLenCamera(pa.Film(size), [f(-0.3, 0.3), f(0.8, 1.2), -4], [0, 1, 0], 0.25, lens[0], lens[1]))
        sampler = "sobol" if (r.random() < 0.4 and variety != 2) else "blue"  # (SobolSampler + Subsurface is refused on the device)
        return s, int(2 ** r.integers(0, 5)), int(r.integers(1, 9)), sampler
    s.set(pa.ThinLenCamera(pa.Film([28, 20]), [f(-0.3, 0.3), f(0.8, 1.2), -4], [0, 1, 0], 0.25))
    return s, int(2 ** r.integers(1, 4)), int(r.integers(2, 7))
