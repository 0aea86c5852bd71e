"""Writes tests/golden/import_test.glb: a small binary glTF that exercises what the reference's importer reads
(src/pine/core/fileio.cpp:146-330) -- node hierarchy with matrix / translation / rotation / scale, u16 and u32 indices,
POSITION + NORMAL + TEXCOORD_0, pbrMetallicRoughness factors, the transmission / ior / emissive-strength extensions, a
camera node.  The file is DATA (a fixture); tools/make_golden.py --gltf renders it with the reference."""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pine_amd.scenes import icosphere  # noqa: E402


def main(out):
    blob = bytearray()
    views, accessors = [], []

    def add(arr, kind, ctype, target=None):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes, **({"target": target} if target else {})})
        blob.extend(arr.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": len(arr) if kind != "SCALAR" else arr.size, "type": kind}
        if kind == "VEC3" and ctype == 5126:
            acc["min"], acc["max"] = arr.min(axis=0).tolist(), arr.max(axis=0).tolist()
        accessors.append(acc)
        return len(accessors) - 1

    def quad(p0, ex, ey):
        p0, ex, ey = map(np.float32, (p0, ex, ey))
        v = np.float32([p0, p0 + ex, p0 + ex + ey, p0 + ey])
        n = np.cross(ex, ey)
        n = np.float32(n / np.linalg.norm(n))
        return v, np.float32([n] * 4), np.float32([[0, 0], [1, 0], [1, 1], [0, 1]]), np.uint16([0, 1, 2, 0, 2, 3])

    meshes = []

    def mesh(v, n, t, idx, material, u32=False):
        attrs = {"POSITION": add(np.float32(v), "VEC3", 5126, 34962)}
        if n is not None:
            attrs["NORMAL"] = add(np.float32(n), "VEC3", 5126, 34962)
        if t is not None:
            attrs["TEXCOORD_0"] = add(np.float32(t), "VEC2", 5126, 34962)
        i = add(np.uint32(idx).reshape(-1) if u32 else np.uint16(idx).reshape(-1), "SCALAR", 5125 if u32 else 5123, 34963)
        meshes.append({"primitives": [{"attributes": attrs, "indices": i, "material": material, "mode": 4}]})
        return len(meshes) - 1

    materials = [
        {"name": "wall", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.78, 0.7, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.9}},
        {"name": "ball", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.45, 0.1, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.35},
         "extensions": {"KHR_materials_ior": {"ior": 1.6}}},
        {"name": "lamp", "emissiveFactor": [1.0, 0.9, 0.75], "pbrMetallicRoughness": {"metallicFactor": 0.0},
         "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 18.0}}},
        {"name": "glassy", "pbrMetallicRoughness": {"baseColorFactor": [0.7, 0.9, 0.95, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.05},
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.45}}},
        {"name": "metal", "pbrMetallicRoughness": {"baseColorFactor": [0.95, 0.8, 0.5, 1.0], "metallicFactor": 1.0, "roughnessFactor": 0.25}},
    ]
    floor = mesh(*quad([-2, 0, -2], [0, 0, 4], [4, 0, 0]), 0)
    back = mesh(*quad([-2, 0, 2], [4, 0, 0], [0, 3, 0]), 0)
    side = mesh(*quad([-2, 0, -2], [0, 3, 0], [0, 0, 4]), 0)
    lamp = mesh(*quad([-0.5, 2.9, -0.5], [1, 0, 0], [0, 0, 1]), 2)
    sv, sf = icosphere(2, 1.0, (0.0, 0.0, 0.0))
    sn = np.float32(sv / np.linalg.norm(sv, axis=1, keepdims=True))
    ball = mesh(sv, sn, None, sf, 1, u32=True)
    cube_v = np.float32([[x, y, z] for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)])
    cube_f = np.uint16([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]])
    cube = mesh(cube_v, None, None, cube_f, 3)      # no normals: the geometric-normal path
    cube2 = mesh(cube_v * np.float32(0.8), None, np.float32(cube_v[:, :2] + 0.5), cube_f, 4)  # texcoords only
    s = np.sin(0.35), np.cos(0.35)
    nodes = [
        {"name": "room", "children": [1, 2, 3, 4]},
        {"name": "floor", "mesh": floor},
        {"name": "back", "mesh": back},
        {"name": "side", "mesh": side},
        {"name": "lamp", "mesh": lamp},
        {"name": "ball", "mesh": ball, "translation": [0.6, 0.55, 0.3], "rotation": [0.0, float(np.sin(0.4)), 0.0, float(np.cos(0.4))], "scale": [0.55, 0.55, 0.4]},
        {"name": "group", "matrix": [s[1], 0, -s[0], 0, 0, 1, 0, 0, s[0], 0, s[1], 0, -0.8, 0.0, 0.6, 1], "children": [7, 8]},
        {"name": "cube", "mesh": cube, "translation": [0.0, 0.4, 0.0], "scale": [0.8, 0.8, 0.8]},
        {"name": "cube2", "mesh": cube2, "translation": [0.1, 1.2, 0.1], "rotation": [float(np.sin(0.3)), 0.0, 0.0, float(np.cos(0.3))]},
        {"name": "camera", "camera": 0, "translation": [0.3, 1.4, -4.2], "rotation": [0.0, 1.0, 0.0, 0.0]},
    ]
    doc = {"asset": {"version": "2.0", "generator": "pine-mi355x tools/make_test_glb.py"}, "scene": 0, "scenes": [{"nodes": [0, 5, 6, 9]}],
           "nodes": nodes, "meshes": meshes, "materials": materials, "accessors": accessors, "bufferViews": views,
           "buffers": [{"byteLength": len(blob)}], "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7, "aspectRatio": 1.0, "znear": 0.1}}],
           "extensionsUsed": ["KHR_materials_ior", "KHR_materials_transmission", "KHR_materials_emissive_strength"]}
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    while len(blob) % 4:
        blob.append(0)
    with open(out, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + bytes(blob))
    print(out, 12 + 16 + len(js) + len(blob), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "import_test.glb"))
