"""Host-side mirror of the reference's script-visible interface for the PathIntegrator path.

Names, argument order and error behaviour follow what a `.pine` script sees (registered in
src/pine/core/program_context.cpp:23-125 and the *_context functions it calls), e.g.

    scene = Scene()
    scene.add("floor", Diffuse([0.9, 0.9, 0.9]))
    scene.add(Rect([0, 0, 1], [2, 0, 0], [0, 0, 2], True), "floor")
    scene.set(ThinLenCamera(Film([640, 640], Uncharted2()), [0, 0, 0], [0, 0, 1], 0.4))
    PathIntegrator(BlueSampler(256), 8).render(scene)
    scene.camera.film().save("images/cbox.png")

Everything here is plumbing over the C ABI (include/pine_gpu.h); all arithmetic that feeds the
render (shape constructors, matrices, camera, BVH) happens in libpine_gpu.so.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import lib, check, PineError, f3, f16


def _v3(v):
    v = list(v)
    if len(v) != 3:
        raise PineError("expected a vec3")
    return f3(*[float(x) for x in v])


# ---- mat4 (PRL builtins translate/rotate_*/scale/look_at, vecmath.h:1102-1180) -----------------
class mat4:
    """Column-vector 4x4 float matrix in the reference's storage order (m[c*4 + r])."""

    def __init__(self, storage=None):
        self.s = f16()
        if storage is None:
            lib.pine_gpu_mat4_identity(self.s)
        else:
            for i, x in enumerate(storage):
                self.s[i] = float(x)

    def __mul__(self, other):
        out = mat4()
        lib.pine_gpu_mat4_mul(self.s, other.s, out.s)
        return out

    def numpy(self):
        """4x4 array indexed [row, col]."""
        return np.array(list(self.s), dtype=np.float32).reshape(4, 4).T.copy()


def translate(v):
    out = mat4()
    lib.pine_gpu_mat4_translate(_v3(v), out.s)
    return out


def scale(v):
    out = mat4()
    lib.pine_gpu_mat4_scale(_v3(v), out.s)
    return out


def rotate_x(rad):
    out = mat4()
    lib.pine_gpu_mat4_rotate_x(float(rad), out.s)
    return out


def rotate_y(rad):
    out = mat4()
    lib.pine_gpu_mat4_rotate_y(float(rad), out.s)
    return out


def rotate_z(rad):
    out = mat4()
    lib.pine_gpu_mat4_rotate_z(float(rad), out.s)
    return out


def inverse(m):
    out = mat4()
    lib.pine_gpu_mat4_inverse(m.s, out.s)
    return out


def look_at(frm, to):
    out = mat4()
    lib.pine_gpu_mat4_look_at(_v3(frm), _v3(to), out.s)
    return out


# ---- materials (src/pine/core/material.cpp:46-62) ---------------------------------------------
# ---- shading nodes (src/pine/core/node.h:13-297, node.cpp:29-116) --------------------------------
class Node:
    """A Nodef (is_vec3 False) or Node3f (True) expression; built lazily, instantiated in a scene's node
    table when a material using it is added.  Operators compose nodes the way node.cpp:70-102 does."""
    is_vec3 = False

    def __init__(self, kind, *args, is_vec3=False):
        self.kind, self.args, self.is_vec3 = kind, args, is_vec3

    @staticmethod
    def of(x, want_vec3=None):
        if isinstance(x, Node):
            n = x
        elif isinstance(x, (int, float, np.floating)):
            n = Node("constf", float(np.float32(x)))
        else:
            n = Node("const3", [float(np.float32(v)) for v in x], is_vec3=True)
        if want_vec3 is True and not n.is_vec3:
            n = Node("splat", n, is_vec3=True)     # Node3f(Nodef): vec3{x.eval} (node.h:78,291-293)
        if want_vec3 is False and n.is_vec3:
            raise PineError("a Nodef is required here")
        return n

    def _bin(self, op, other, swap=False):
        o = Node.of(other)
        a, b = (o, self) if swap else (self, o)
        v = a.is_vec3 or b.is_vec3
        return Node("bin", op, Node.of(a, v), Node.of(b, v), is_vec3=v)

    def __add__(self, o): return self._bin("+", o)
    def __radd__(self, o): return self._bin("+", o, True)
    def __sub__(self, o): return self._bin("-", o)
    def __rsub__(self, o): return self._bin("-", o, True)
    def __mul__(self, o): return self._bin("*", o)
    def __rmul__(self, o): return self._bin("*", o, True)
    def __truediv__(self, o): return self._bin("/", o)
    def __rtruediv__(self, o): return self._bin("/", o, True)
    def __pow__(self, o): return self._bin("^", o)
    def __neg__(self): return Node("un", "-", self, is_vec3=self.is_vec3)
    def __getitem__(self, n): return Node("comp", self, int(n))

    def _instantiate(self, scene, memo):
        if id(self) in memo:
            return memo[id(self)][1]
        h, k, a = scene._h, self.kind, self.args
        sub = lambda x: x._instantiate(scene, memo)  # noqa: E731
        if k == "constf":
            r = lib.pine_gpu_scene_node_constf(h, float(a[0]))
        elif k == "const3":
            r = lib.pine_gpu_scene_node_const3(h, _v3(a[0]))
        elif k in ("position", "normal", "uv"):
            r = lib.pine_gpu_scene_node_input(h, {"position": 0, "normal": 1, "uv": 2}[k])
        elif k == "bin":
            r = lib.pine_gpu_scene_node_binary(h, ord(a[0]), sub(a[1]), sub(a[2]))
        elif k == "un":
            r = lib.pine_gpu_scene_node_unary(h, ord(a[0]), sub(a[1]))
        elif k == "comp":
            r = lib.pine_gpu_scene_node_component(h, sub(a[0]), a[1])
        elif k == "tovec3":
            ids = [sub(x) for x in a]
            r = lib.pine_gpu_scene_node_to_vec3(h, ids[0], ids[1] if len(ids) == 3 else -1, ids[2] if len(ids) == 3 else -1)
        elif k == "checker":
            r = lib.pine_gpu_scene_node_checkerboard(h, sub(a[0]), float(a[1]))
        elif k == "splat":
            r = lib.pine_gpu_scene_node_splat(h, sub(a[0]))
        else:
            raise PineError("unknown node kind " + k)
        memo[id(self)] = (self, check(r, "node"))  # (keeps the node alive: ids of dead temporaries get reused)
        return memo[id(self)][1]


def Position(): return Node("position", is_vec3=True)
def Normal(): return Node("normal", is_vec3=True)
def UV(): return Node("uv", is_vec3=True)
def Checkerboard(p, ratio=0.5): return Node("checker", Node.of(p, True), float(np.float32(ratio)))
def Vec3(x, y=None, z=None):
    if y is None:
        return Node("tovec3", Node.of(x, False), is_vec3=True)
    return Node("tovec3", Node.of(x, False), Node.of(y, False), Node.of(z, False), is_vec3=True)
def node_abs(x): return Node("un", "a", Node.of(x), is_vec3=Node.of(x).is_vec3)
def node_sqr(x): return Node("un", "s", Node.of(x), is_vec3=Node.of(x).is_vec3)
def node_sqrt(x): return Node("un", "r", Node.of(x), is_vec3=Node.of(x).is_vec3)
def node_fract(x): return Node("un", "f", Node.of(x), is_vec3=Node.of(x).is_vec3)


def lerp(t, a, b):
    """lerp over nodes exactly as node.cpp:88-102 composes it (plain numbers lerp numerically)."""
    if not any(isinstance(x, Node) for x in (t, a, b)):
        raise PineError("lerp: at least one argument must be a node (use numpy for plain numbers)")
    t, a, b = Node.of(t), Node.of(a), Node.of(b)
    one = Node("constf", 1.0)
    if not t.is_vec3 and not a.is_vec3 and not b.is_vec3:      # (Nodef, Nodef, Nodef)
        return (t * b) + ((one - t) * a)
    if not t.is_vec3:                                           # (Nodef, Node3f, Node3f)
        return (Vec3(t) * Node.of(b, True)) + (Vec3(one - t) * Node.of(a, True))
    one3 = Node("const3", [1.0, 1.0, 1.0], is_vec3=True)        # (Node3f, Node3f, Node3f)
    return (t * Node.of(b, True)) + ((one3 - t) * Node.of(a, True))


class Material:
    pass


class Emissive(Material):
    def __init__(self, color):
        self.color = color


class Diffuse(Material):
    def __init__(self, albedo):
        self.albedo = albedo


class Uber(Material):
    def __init__(self, albedo, roughness, metallic=0.0, transmission=0.0, ior=1.45):
        self.albedo, self.roughness, self.metallic, self.transmission, self.ior = albedo, roughness, metallic, transmission, ior


class Subsurface(Material):
    def __init__(self, albedo, roughness, sigma_s):
        self.albedo, self.roughness, self.sigma_s = albedo, roughness, sigma_s


class Metal(Material):       # material.h:39-50
    def __init__(self, albedo, roughness):
        self.albedo, self.roughness = albedo, roughness


class Glossy(Material):      # material.h:52-64
    def __init__(self, albedo, roughness, ior=1.4):
        self.albedo, self.roughness, self.ior = albedo, roughness, ior


class Glass(Material):       # material.h:66-78
    def __init__(self, albedo, roughness, ior=1.4):
        self.albedo, self.roughness, self.ior = albedo, roughness, ior


# ---- lights other than emissive geometry (src/pine/core/light.h:21-67, light.cpp:173-186) ---------
class Light:
    pass


class PointLight(Light):
    def __init__(self, position, color):
        self.position, self.color = position, color


class SpotLight(Light):
    def __init__(self, position, direction, color, falloff_radian, cutoff_additional_radian=0.0):
        self.position, self.direction, self.color = position, direction, color
        self.falloff, self.cutoff_additional = falloff_radian, cutoff_additional_radian


class DirectionalLight(Light):
    def __init__(self, direction, color):
        self.direction, self.color = direction, color


class Sky:
    """EnvironmentLight Sky(sun_color): scene.set(Sky(...))."""
    def __init__(self, sun_color):
        self.sun_color = sun_color


# ---- shapes (src/pine/core/geometry.cpp:901-946) ------------------------------------------------
class Shape:
    pass


class Rect(Shape):
    def __init__(self, position, ex, ey, flip_normal=False):
        self.position, self.ex, self.ey, self.flip_normal = position, ex, ey, flip_normal


class AABB(Shape):
    def __init__(self, lower, upper):
        self.lower, self.upper = lower, upper


class OBB(Shape):
    def __init__(self, aabb, m):
        self.aabb, self.m = aabb, m


def Box(*args):
    """Box(lower, upper) | Box(AABB, mat4) | Box(lower, upper, mat4) -- geometry.cpp:913-918."""
    if len(args) == 2 and isinstance(args[0], AABB):
        return OBB(args[0], args[1])
    if len(args) == 2:
        return AABB(args[0], args[1])
    if len(args) == 3:
        return OBB(AABB(args[0], args[1]), args[2])
    raise PineError("Box: no matching overload")


class Sphere(Shape):
    def __init__(self, center, radius):
        self.center, self.radius = center, radius


class Disk(Shape):
    def __init__(self, position, normal, radius):
        self.position, self.normal, self.radius = position, normal, radius


class Cone(Shape):
    def __init__(self, position, normal, radius, height):
        self.position, self.normal, self.radius, self.height = position, normal, radius, height


class Plane(Shape):
    """Plane(position, normal) geometry.cpp:31-34 (its bounding box is position +-100, geometry.cpp:52)"""
    def __init__(self, position, normal):
        self.position, self.normal = position, normal


class Line(Shape):
    """Line(p0, p1, thickness) geometry.cpp:171-179"""
    def __init__(self, p0, p1, thickness):
        self.p0, self.p1, self.thickness = p0, p1, thickness


class Cylinder(Shape):
    """Cylinder(p0, p1, r) geometry.h:139-157: side surface only, never a light"""
    def __init__(self, p0, p1, radius):
        self.p0, self.p1, self.radius = p0, p1, radius


class Triangle(Shape):
    """Triangle(v0, v1, v2) geometry.cpp:528-531"""
    def __init__(self, v0, v1, v2):
        self.v0, self.v1, self.v2 = v0, v1, v2


class Mesh(Shape):
    """Mesh(vertices, indices[, texcoords, normals]) geometry.cpp:596-604: per-vertex normals give the interpolated shading
    normal, per-vertex texcoords replace the barycentric uv (geometry.h:199-210)."""
    def __init__(self, vertices, indices, texcoords=None, normals=None):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.texcoords = None if texcoords is None else np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
        for name, a in (("normals", self.normals), ("texcoords", self.texcoords)):
            if a is not None and len(a) != len(self.vertices):
                raise PineError(f"Mesh: {len(self.vertices)} vertices but {len(a)} {name}")  # CHECK_EQ geometry.cpp:602-603


# ---- film / camera / sampler -------------------------------------------------------------------
class Uncharted2:
    code = 0


class ACES:
    code = 1


class Film:
    """Film(size[, tonemapper]) -- src/pine/core/film.h:24-27.  pixels: H x W x 4 float32, row 0 first."""

    def __init__(self, size, tone_mapper=None):
        self.size = (int(size[0]), int(size[1]))
        self.tone_mapper = tone_mapper or Uncharted2()
        self.pixels = np.zeros((self.size[1], self.size[0], 4), dtype=np.float32)

    def width(self):
        return self.size[0]

    def height(self):
        return self.size[1]

    def finalize_u8(self):
        """film.finalize() + gamma + y-flip as save() does (film.cpp:12-27, fileio.cpp:42-54)."""
        w, h = self.size
        out = np.zeros((h, w, 4), dtype=np.uint8)
        src = np.ascontiguousarray(self.pixels, dtype=np.float32)
        check(lib.pine_gpu_film_finalize_u8(src.ctypes.data_as(_lib.c_f_p), w, h, self.tone_mapper.code,
                                            out.ctypes.data_as(C.POINTER(C.c_uint8))), "film.finalize")
        return out

    def save(self, filename):
        """scene.camera.film().save(path): 8-bit PNG of the tone-mapped film."""
        from . import png
        png.write_png(filename, self.finalize_u8())


class ThinLenCamera:
    def __init__(self, film, frm, to, fov, len_radius=0.0, focus_distance=1.0):
        self._film, self.frm, self.to, self.fov = film, frm, to, fov
        self.len_radius, self.focus_distance = len_radius, focus_distance

    def film(self):
        return self._film


class BlueSampler:
    """BlueSobolSampler: spp rounded up to a power of two and clamped to 256 (sampler.cpp:115-121)."""
    kind = 0  # PINE_GPU_SAMPLER_BLUE

    def __init__(self, samples_per_pixel):
        if samples_per_pixel <= 0:
            raise PineError("`BlueSampler` should have positive samples per pixel")
        self.requested = int(samples_per_pixel)

    def spp(self):
        n = min(self.requested, 256)
        p = 1
        while p < n:
            p *= 2
        return p


class SobolSampler:
    """SobolSampler(spp) (sampler.h:83-164): spp is used as given -- no rounding, no clamp to 256.  On the
    device any count up to 4096."""
    kind = 1  # PINE_GPU_SAMPLER_SOBOL

    def __init__(self, samples_per_pixel):
        self.requested = int(samples_per_pixel)

    def spp(self):
        return self.requested


class HaltonSampler:
    """HaltonSampler(spp) (sampler.h:40-81): scrambled radical inverses over the first primes, the pixel's place in the
    sequence from its coordinates modulo 128; spp as given (on the device up to 4096)."""
    kind = 2  # PINE_GPU_SAMPLER_HALTON

    def __init__(self, samples_per_pixel):
        if samples_per_pixel <= 0:
            raise PineError("`HaltonSampler` should have positive samples per pixel")
        self.requested = int(samples_per_pixel)

    def spp(self):
        return self.requested


# ---- Scene (src/pine/core/scene.cpp:64-79) -----------------------------------------------------
class Scene:
    def __init__(self):
        self._h = C.c_void_p(check(lib.pine_gpu_scene_create(), "Scene()"))
        self.camera = None
        self._anon = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.pine_gpu_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # scene.add(name, material) | scene.add(shape, material | name)
    def add(self, a, b=None):
        if isinstance(a, str) and isinstance(b, Material):
            return self._add_material(a, b)
        if isinstance(a, Light) and b is None:
            h = self._h
            if isinstance(a, PointLight):
                return check(lib.pine_gpu_scene_add_light_point(h, _v3(a.position), _v3(a.color)), "PointLight")
            if isinstance(a, SpotLight):
                return check(lib.pine_gpu_scene_add_light_spot(h, _v3(a.position), _v3(a.direction), _v3(a.color),
                                                               float(a.falloff), float(a.cutoff_additional)), "SpotLight")
            return check(lib.pine_gpu_scene_add_light_directional(h, _v3(a.direction), _v3(a.color)), "DirectionalLight")
        if isinstance(a, Shape):
            if isinstance(b, str):
                mid = check(lib.pine_gpu_scene_find_material(self._h, b.encode()), "scene.add")
            elif isinstance(b, Material):
                mid = self._add_material("", b)
            else:
                raise PineError("scene.add: second argument must be a material or a material name")
            return self._add_shape(a, mid)
        raise PineError("scene.add: no matching overload")

    def set(self, camera):
        if isinstance(camera, Sky):
            check(lib.pine_gpu_scene_set_env_sky(self._h, _v3(camera.sun_color)), "scene.set")
            return camera
        if not isinstance(camera, ThinLenCamera):
            raise PineError("scene.set: expected a camera")
        f = camera.film()
        check(lib.pine_gpu_scene_set_camera_thinlens(self._h, f.size[0], f.size[1], f.tone_mapper.code,
                                                     _v3(camera.frm), _v3(camera.to), float(camera.fov),
                                                     float(camera.len_radius), float(camera.focus_distance)),
              "scene.set")
        self.camera = camera
        return camera

    def _add_material(self, name, m):
        n = name.encode()
        if isinstance(m, Emissive):
            return check(lib.pine_gpu_scene_add_material_emissive(self._h, n, _v3(m.color)), "Emissive")
        memo = {}
        nid = lambda x, v: Node.of(x, v)._instantiate(self, memo)  # noqa: E731
        if isinstance(m, Diffuse) and isinstance(m.albedo, Node):
            return check(lib.pine_gpu_scene_add_material_diffuse_n(self._h, n, nid(m.albedo, True)), "Diffuse")
        if isinstance(m, Uber) and any(isinstance(x, Node) for x in (m.albedo, m.roughness, m.metallic, m.transmission)):
            return check(lib.pine_gpu_scene_add_material_uber_n(self._h, n, nid(m.albedo, True), nid(m.roughness, False),
                                                                nid(m.metallic, False), nid(m.transmission, False), float(m.ior)), "Uber")
        if isinstance(m, Metal):
            return check(lib.pine_gpu_scene_add_material_metal(self._h, n, nid(m.albedo, True), nid(m.roughness, False)), "Metal")
        if isinstance(m, Glossy):
            return check(lib.pine_gpu_scene_add_material_glossy(self._h, n, nid(m.albedo, True), nid(m.roughness, False),
                                                                nid(m.ior, False)), "Glossy")
        if isinstance(m, Glass):
            return check(lib.pine_gpu_scene_add_material_glass(self._h, n, nid(m.albedo, True), nid(m.roughness, False),
                                                               nid(m.ior, False)), "Glass")
        if isinstance(m, Diffuse):
            return check(lib.pine_gpu_scene_add_material_diffuse(self._h, n, _v3(m.albedo)), "Diffuse")
        if isinstance(m, Uber):
            return check(lib.pine_gpu_scene_add_material_uber(self._h, n, _v3(m.albedo), float(m.roughness),
                                                              float(m.metallic), float(m.transmission), float(m.ior)), "Uber")
        if isinstance(m, Subsurface):
            return check(lib.pine_gpu_scene_add_material_subsurface(self._h, n, _v3(m.albedo), float(m.roughness),
                                                                    _v3(m.sigma_s)), "Subsurface")
        raise PineError("unsupported material")

    def _add_shape(self, s, mid):
        h = self._h
        if isinstance(s, Rect):
            return check(lib.pine_gpu_scene_add_rect(h, _v3(s.position), _v3(s.ex), _v3(s.ey), int(bool(s.flip_normal)), mid), "Rect")
        if isinstance(s, AABB):
            return check(lib.pine_gpu_scene_add_aabb(h, _v3(s.lower), _v3(s.upper), mid), "Box")
        if isinstance(s, OBB):
            return check(lib.pine_gpu_scene_add_obb(h, _v3(s.aabb.lower), _v3(s.aabb.upper), s.m.s, mid), "Box")
        if isinstance(s, Sphere):
            return check(lib.pine_gpu_scene_add_sphere(h, _v3(s.center), float(s.radius), mid), "Sphere")
        if isinstance(s, Disk):
            return check(lib.pine_gpu_scene_add_disk(h, _v3(s.position), _v3(s.normal), float(s.radius), mid), "Disk")
        if isinstance(s, Cone):
            return check(lib.pine_gpu_scene_add_cone(h, _v3(s.position), _v3(s.normal), float(s.radius), float(s.height), mid), "Cone")
        if isinstance(s, Plane):
            return check(lib.pine_gpu_scene_add_plane(h, _v3(s.position), _v3(s.normal), mid), "Plane")
        if isinstance(s, Line):
            return check(lib.pine_gpu_scene_add_line(h, _v3(s.p0), _v3(s.p1), float(s.thickness), mid), "Line")
        if isinstance(s, Cylinder):
            return check(lib.pine_gpu_scene_add_cylinder(h, _v3(s.p0), _v3(s.p1), float(s.radius), mid), "Cylinder")
        if isinstance(s, Triangle):
            return check(lib.pine_gpu_scene_add_triangle(h, _v3(s.v0), _v3(s.v1), _v3(s.v2), mid), "Triangle")
        if isinstance(s, Mesh):
            if s.normals is not None or s.texcoords is not None:
                return check(lib.pine_gpu_scene_add_mesh_full(
                    h, s.vertices.ctypes.data_as(_lib.c_f_p), len(s.vertices), s.indices.ctypes.data_as(C.POINTER(C.c_uint32)), len(s.indices),
                    s.normals.ctypes.data_as(_lib.c_f_p) if s.normals is not None else None,
                    s.texcoords.ctypes.data_as(_lib.c_f_p) if s.texcoords is not None else None, mid), "Mesh")
            return check(lib.pine_gpu_scene_add_mesh(h, s.vertices.ctypes.data_as(_lib.c_f_p), len(s.vertices),
                                                     s.indices.ctypes.data_as(C.POINTER(C.c_uint32)), len(s.indices), mid), "Mesh")
        raise PineError("unsupported shape")

    def describe(self) -> str:
        """The scene as .pscene text (exchange format shared with the oracle and the reference driver)."""
        n = check(lib.pine_gpu_scene_describe(self._h, None, 0), "describe")
        buf = C.create_string_buffer(n + 1)
        lib.pine_gpu_scene_describe(self._h, buf, n + 1)
        return buf.value.decode()


# ---- PathIntegrator (program_context.cpp:76-81) -------------------------------------------------
def _specialize_flags(specialize):
    if specialize is None:
        return 0
    return _lib.FLAG_SPECIALIZE if specialize else _lib.FLAG_NO_SPECIALIZE


def _order_flags(order):
    if order in (None, "pine", "bvh"):
        return 0
    if order in ("embree", "nearest"):  # ("nearest": this order's former name)
        return _lib.FLAG_ORDER_EMBREE
    raise PineError(f"unknown traversal order {order!r} (pine | embree)")


class Plan:
    """A PathIntegrator bound to a scene with all device state resident (bench / multi-GPU)."""

    def __init__(self, scene, spp, max_path_length, device=0, shard_rank=0, shard_world=1,
                 samples_per_item=0, timing=False, sampler="blue", flags=0, specialize=None, order="pine"):
        """spp: an int (BlueSampler(spp), or SobolSampler(spp) with sampler="sobol") or a sampler object.
        specialize: None -- the library's default: the scene's own kernel from the cache, else compiled in the background
        while the precompiled kernel renders; True -- PINE_GPU_FLAG_SPECIALIZE: wait for the compiler at plan creation, fail if
        the kernel cannot be built; False -- PINE_GPU_FLAG_NO_SPECIALIZE: precompiled kernels only (stats().specialized tells).
        order: "pine" -- closest hits in pine-BVH order, Accel(BVH()) (the default and the parity gate); "embree" --
        PINE_GPU_FLAG_ORDER_EMBREE: the order of the reference's default accel, EmbreeAccel (order-dependent shapes appear as under it)."""
        flags = int(flags) | _specialize_flags(specialize) | _order_flags(order)
        if scene.camera is None:
            raise PineError("scene has no camera")
        self.scene = scene
        kind = {"sobol": 1, "halton": 2}.get(sampler, 0)
        if hasattr(spp, "requested"):
            kind, spp = getattr(spp, "kind", 0), spp.requested
        self.params = _lib.RenderParams(int(spp), int(max_path_length), int(device), int(shard_rank),
                                        int(shard_world), int(samples_per_item),
                                        (_lib.FLAG_TIMING if timing else 0) | int(flags), kind)
        h = lib.pine_gpu_plan_create(scene._h, C.byref(self.params))
        if not h:
            raise PineError("PathIntegrator: " + _lib.last_error())
        self._h = C.c_void_p(h)

    def launch(self, film_dev_ptr, stream_ptr=0):
        check(lib.pine_gpu_plan_launch(self._h, C.c_void_p(film_dev_ptr), C.c_void_p(stream_ptr)), "render")

    def launch_packed(self, slab_dev_ptr, stream_ptr=0):
        """Multi-GPU form: write only this rank's tiles, tile-major, into a slab of slab_floats() floats."""
        check(lib.pine_gpu_plan_launch_packed(self._h, C.c_void_p(slab_dev_ptr), C.c_void_p(stream_ptr)), "render")

    def slab_floats(self):
        w, h = self.scene.camera.film().size
        return int(lib.pine_gpu_packed_slab_floats(w, h, self.params.shard_world))

    def check(self):
        """Wait for the last launch; raises PineError if its path kernel bailed out (incomplete film)."""
        check(lib.pine_gpu_plan_check(self._h), "render")

    def stats(self):
        st = _lib.PlanStats()
        check(lib.pine_gpu_plan_stats_get(self._h, C.byref(st)), "stats")
        return st

    def read_samples(self):
        w, h = self.scene.camera.film().size
        spp = self.stats().spp_effective
        out = np.zeros((h, w, spp, 4), dtype=np.float32)
        check(lib.pine_gpu_plan_read_samples(self._h, out.ctypes.data_as(_lib.c_f_p), out.size), "read_samples")
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib.pine_gpu_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def film_unpack(size, world, slabs_dev_ptr, film_dev_ptr, device=0, stream_ptr=0):
    """Scatter gathered per-rank slabs ([rank][slab]) into the row-major film (device pointers)."""
    check(lib.pine_gpu_film_unpack(int(size[0]), int(size[1]), int(world), int(device), C.c_void_p(slabs_dev_ptr),
                                   C.c_void_p(film_dev_ptr), C.c_void_p(stream_ptr)), "film_unpack")


def packed_offset(size, world, x, y):
    """(rank, float4 index inside that rank's slab) of pixel (x, y): host statement of the slab layout."""
    r, o = C.c_int(0), C.c_int64(0)
    check(lib.pine_gpu_packed_offset(int(size[0]), int(size[1]), int(world), int(x), int(y), C.byref(r), C.byref(o)), "packed_offset")
    return r.value, o.value


class PathIntegrator:
    """PathIntegrator(sampler, max_path_length).render(scene) -- the convenience overload a .pine
    script uses (program_context.cpp:79-81); pine-BVH traversal order, UniformLightSampler."""

    def __init__(self, sampler, max_path_length, device=0, flags=0, devices=None, specialize=None, order="pine"):
        """devices: a list of HIP device ordinals -- the film is rendered by all of them from this one process
        (pine_gpu_path_render_devices); default: the single `device`.
        specialize: None / True / False as for Plan (the scene's own kernel: automatic / required / never; same film).
        order: "pine" (Accel(BVH()), the default) or "embree" (PINE_GPU_FLAG_ORDER_EMBREE: what the reference's EmbreeAccel does)."""
        flags = int(flags) | _specialize_flags(specialize) | _order_flags(order)
        if max_path_length <= 0:  # path.cpp:12-13
            raise PineError(f"`PathIntegrator` expect `max_path_length` to be positive, get {max_path_length}")
        self.sampler, self.max_path_length, self.device, self.flags = sampler, int(max_path_length), device, int(flags)
        self.devices = list(devices) if devices else None

    def render(self, scene):
        if scene.camera is None:
            raise PineError("scene has no camera")
        film = scene.camera.film()
        prm = _lib.RenderParams(self.sampler.requested, self.max_path_length, self.device, 0, 1, 0, self.flags,
                                getattr(self.sampler, "kind", 0))
        out = np.zeros((film.size[1], film.size[0], 4), dtype=np.float32)
        if self.devices:
            arr = (C.c_int * len(self.devices))(*self.devices)
            check(lib.pine_gpu_path_render_devices(scene._h, C.byref(prm), arr, len(self.devices), out.ctypes.data_as(_lib.c_f_p)),
                  "PathIntegrator.render")
            film.pixels = out
            return film
        check(lib.pine_gpu_path_render(scene._h, C.byref(prm), out.ctypes.data_as(_lib.c_f_p)), "PathIntegrator.render")
        film.pixels = out
        return film
