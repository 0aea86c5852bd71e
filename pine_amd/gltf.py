"""glTF (.glb / .gltf) scene import -- what `load(scene, "file.glb")` does in the reference
(src/pine/core/fileio.cpp:146-330, through tinygltf): every mesh primitive of every node of every scene becomes a
Mesh(vertices, indices, texcoords, normals) with the node's accumulated transform applied (Mesh::apply), its material an
Uber(baseColor, roughness, metallic, transmission, ior) from the pbrMetallicRoughness factors and the KHR transmission / ior
extensions, or an Emissive(emissiveFactor * emissiveStrength) when that product is not zero; a node with a camera sets
ThinLenCamera(Film([640 * aspect, 640]), position, position + R * (0, 0, -1), yfov / 2).

SURVEY.md 8(f) rank 4.  All matrix arithmetic goes through the library's host math (binary32, the reference's operand
order).  Image textures (baseColorTexture, metallicRoughnessTexture -> NodeImage) are not supported: such a material is
refused by name."""
import base64
import json
import os
import struct
import ctypes as C

import numpy as np

from . import _lib
from .api import Scene, Mesh, Uber, Emissive, Film, ThinLenCamera, PineError

lib = _lib.lib
_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32, 5130: np.float64}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


class _M4:
    """mat4 in the reference's storage order, operations by the library (float32, same operand order)."""

    def __init__(self, a=None):
        self.a = _lib.f16()
        if a is None:
            lib.pine_gpu_mat4_identity(self.a)
        else:
            self.a[:] = a

    def __mul__(self, o):
        r = _M4()
        lib.pine_gpu_mat4_mul(self.a, o.a, r.a)
        return r


def _translate(v):
    r = _M4()
    lib.pine_gpu_mat4_translate(_lib.f3(*map(float, v)), r.a)
    return r


def _scale(v):
    r = _M4()
    lib.pine_gpu_mat4_scale(_lib.f3(*map(float, v)), r.a)
    return r


def _q2m(w, x, y, z):
    r = _M4()
    lib.pine_gpu_mat4_from_quaternion(float(np.float32(w)), float(np.float32(x)), float(np.float32(y)), float(np.float32(z)), r.a)
    return r


def _node_matrix(m):  # transpose(mat4(m[0], ..., m[15])): the scalar constructor takes row-major arguments (fileio.cpp:162-164)
    rows, t = _M4(), _M4()
    lib.pine_gpu_mat4_from_rows(_lib.f16(*[float(np.float32(v)) for v in m]), rows.a)
    lib.pine_gpu_mat4_transpose(rows.a, t.a)
    return t


def _read(path):
    data = open(path, "rb").read()
    if data[:4] == b"glTF":
        _, _, total = struct.unpack("<4sII", data[:12])
        pos, doc, blob = 12, None, b""
        while pos < total:
            n, kind = struct.unpack("<II", data[pos:pos + 8])
            chunk = data[pos + 8:pos + 8 + n]
            if kind == 0x4E4F534A:
                doc = json.loads(chunk.decode("utf-8"))
            elif kind == 0x004E4942:
                blob = chunk
            pos += 8 + n
        if doc is None:
            raise PineError("Unable to create scene from GLTF file")
        return doc, blob
    return json.loads(data.decode("utf-8")), b""


def load(path, scene=None, transform=None):
    """Import `path` into `scene` (a new Scene by default); returns the scene."""
    doc, glb_blob = _read(path)
    if not isinstance(doc, dict) or "version" not in (doc.get("asset") or {}):  # (tinygltf's REQUIRE_VERSION, its default)
        raise PineError("Unable to create scene from GLTF file (no asset.version)")
    base = os.path.dirname(os.path.abspath(path))
    buffers = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            buffers.append(glb_blob)
        elif uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            buffers.append(open(os.path.join(base, uri), "rb").read())

    def accessor(index):
        acc = doc["accessors"][index]
        view = doc["bufferViews"][acc["bufferView"]]
        dt, nc = np.dtype(_COMPONENT[acc["componentType"]]), _NCOMP[acc["type"]]
        start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
        stride = view.get("byteStride", 0) or dt.itemsize * nc
        raw = buffers[view["buffer"]]
        if stride == dt.itemsize * nc:
            arr = np.frombuffer(raw, dtype=dt, count=acc["count"] * nc, offset=start).reshape(acc["count"], nc)
        else:
            arr = np.stack([np.frombuffer(raw, dtype=dt, count=nc, offset=start + i * stride) for i in range(acc["count"])])
        return acc, arr

    scene = scene or Scene()
    counter = [0]

    def material_of(prim):
        basecolor, roughness, metallic, transmission = [1.0, 1.0, 1.0], 1.0, 0.0, 0.0
        emission_color, emission_strength, ior = [1.0, 1.0, 1.0], 0.0, 1.45
        mi = prim.get("material", -1)
        if mi is not None and mi >= 0:
            mat = doc["materials"][mi]
            ext = mat.get("extensions", {})
            if "KHR_materials_transmission" in ext:
                transmission = ext["KHR_materials_transmission"].get("transmissionFactor", 0.0)
            if "KHR_materials_ior" in ext:
                ior = ext["KHR_materials_ior"].get("ior", 1.5)
            if "KHR_materials_emissive_strength" in ext:
                emission_strength = ext["KHR_materials_emissive_strength"].get("emissiveStrength", 1.0)
            pbr = mat.get("pbrMetallicRoughness", {})
            if "baseColorTexture" in pbr or "metallicRoughnessTexture" in pbr:
                raise PineError("glTF material `%s` uses image textures (NodeImage): not supported" % mat.get("name", mi))
            basecolor = list(pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0]))[:3]   # tinygltf defaults
            metallic = pbr.get("metallicFactor", 1.0)
            roughness = pbr.get("roughnessFactor", 1.0)
            emission_color = list(mat.get("emissiveFactor", [0.0, 0.0, 0.0]))
        f32 = lambda v: float(np.float32(v))
        emission = [f32(np.float32(c) * np.float32(emission_strength)) for c in emission_color]
        counter[0] += 1
        if all(e == 0.0 for e in emission):
            return Uber([f32(c) for c in basecolor], f32(roughness), f32(metallic), f32(transmission), f32(ior))
        return Emissive(emission)

    def process(node_index, xf):
        node = doc["nodes"][node_index]
        if len(node.get("matrix", [])) == 16:
            xf = xf * _node_matrix(node["matrix"])
        if len(node.get("translation", [])) == 3:
            xf = xf * _translate([np.float32(v) for v in node["translation"]])
        if len(node.get("rotation", [])) == 4:
            r = node["rotation"]
            xf = xf * _q2m(r[3], r[0], r[1], r[2])
        if len(node.get("scale", [])) == 3:
            xf = xf * _scale([np.float32(v) for v in node["scale"]])
        if node.get("mesh", -1) is not None and node.get("mesh", -1) >= 0:
            for prim in doc["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    raise PineError("only TRIANGLES primitives are supported (as in the reference)")
                iacc, idx = accessor(prim["indices"])
                if idx.dtype.itemsize not in (2, 4):
                    raise PineError("index byte size must be 2 or 4 (fileio.cpp:181)")
                faces = np.ascontiguousarray(idx.reshape(-1)[: (idx.size // 3) * 3].astype(np.uint32).reshape(-1, 3))
                verts = normals = uvs = None
                for name, ai in prim["attributes"].items():
                    acc, arr = accessor(ai)
                    if name == "POSITION":
                        verts = np.ascontiguousarray(arr.astype(np.float32))
                    elif name == "NORMAL":
                        normals = np.ascontiguousarray(arr.astype(np.float32))
                    elif name == "TEXCOORD_0":
                        uvs = np.ascontiguousarray(arr.astype(np.float32))
                material = material_of(prim)
                # Mesh::apply(transform) (geometry.cpp:647-653) on the host, in the reference's operand order
                _lib.check(lib.pine_gpu_mesh_apply(verts.ctypes.data_as(_lib.c_f_p), len(verts),
                                                   normals.ctypes.data_as(_lib.c_f_p) if normals is not None else None, xf.a), "Mesh.apply")
                scene.add(Mesh(verts, faces, normals=normals, texcoords=uvs), material)
        for child in node.get("children", []):
            process(child, xf)

    root = _M4(list(transform.s)) if transform is not None else _M4()  # (a pine_amd.mat4)
    for sc in doc.get("scenes", []):
        for ni in sc.get("nodes", []):
            process(ni, root)

    for node in doc.get("nodes", []):
        ci = node.get("camera", -1)
        if ci is not None and ci >= 0:
            cam = doc["cameras"][ci]["perspective"]
            P, R = node["translation"], node["rotation"]  # (the reference CHECKs both are present)
            pos = np.float32(P)
            rot = _q2m(R[3], R[0], R[1], R[2])
            m = np.array(list(rot.a), dtype=np.float32).reshape(4, 4)  # m[c][r]
            # at = pos + mat3(rot) * (0, 0, -1) = pos + (x*0 + y*0 + z*(-1))  (operator*(mat3, vec3) vecmath.h:695)
            col = lambda c: m[c, :3]
            prod = col(0) * np.float32(0) + col(1) * np.float32(0) + col(2) * np.float32(-1)
            at = pos + prod.astype(np.float32)
            aspect = cam.get("aspectRatio", 0.0)
            w = int(640 * aspect)  # vec2i(640 * cam.perspective.aspectRatio, 640): double product truncated
            scene.set(ThinLenCamera(Film([w, 640]), [float(v) for v in pos], [float(v) for v in at], float(np.float32(cam["yfov"] / 2))))
    return scene
