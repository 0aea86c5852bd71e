"""ctypes binding of libpine_gpu.so (include/pine_gpu.h).

The library is built in-tree by `make -C pine_amd/csrc` (or `__graft_entry__.build()`); it is the
product: if it is missing, importing this module raises -- there is no Python/CPU fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PINE_GPU_LIB") or os.path.join(_HERE, "lib", "libpine_gpu.so")
TABLE_PATH = os.path.join(_HERE, "data", "bluesobol_u8.bin")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C pine_amd/csrc). pine_amd has no CPU fallback."
    )
lib = C.CDLL(LIB_PATH)

f3 = C.c_float * 3
f16 = C.c_float * 16
c_f_p = C.POINTER(C.c_float)


class RenderParams(C.Structure):
    _fields_ = [
        ("spp", C.c_int32),
        ("max_path_length", C.c_int32),
        ("device", C.c_int32),
        ("shard_rank", C.c_int32),
        ("shard_world", C.c_int32),
        ("samples_per_item", C.c_int32),
        ("flags", C.c_int32),
        ("sampler", C.c_int32),
    ]


class PlanStats(C.Structure):
    _fields_ = [
        ("camera_samples", C.c_uint64),
        ("vertices", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("trace_ms", C.c_float),
        ("resolve_ms", C.c_float),
        ("prepass_ms", C.c_float),
        ("spp_effective", C.c_int32),
        ("samples_per_item", C.c_int32),
        ("grid_blocks", C.c_int32),
        ("block_threads", C.c_int32),
        ("lds_bytes", C.c_int32),
        ("timed_launches", C.c_int32),
        ("walk_steps", C.c_uint64),
        ("accel_build_ms", C.c_float),
        ("upload_ms", C.c_float),
        ("accel_built_on_device", C.c_int32),
        ("serial_tiles", C.c_int32),
        ("specialized", C.c_int32),
        ("specialize_ms", C.c_float),
        ("kernel_features", C.c_uint32),
        ("specialize_source", C.c_int32),
        ("specialize_pending", C.c_int32),
        ("reserved", C.c_int32),
    ]


FLAG_TIMING = 1
FLAG_PROGRESS = 2
FLAG_FAST = 4
FLAG_DEVICE_BVH = 8
FLAG_DEBUG_FORCE_BAIL = 0x100
FLAG_VERTEX_LOG = 0x200
FLAG_SPECIALIZE = 0x400
FLAG_SPECIALIZE_NO_BAKE = 0x800
FLAG_SPECIALIZE_ASYNC = 0x1000
FLAG_NO_SPECIALIZE = 0x2000
FLAG_ORDER_EMBREE = 0x4000
FLAG_ORDER_NEAREST = FLAG_ORDER_EMBREE  # (former name)

# every symbol include/pine_gpu.h declares, with its signature
SIGNATURES = {
    "pine_gpu_last_error": (C.c_char_p, []),
    "pine_gpu_progress": (C.c_float, []),
    "pine_gpu_release_cached_memory": (None, []),
    "pine_gpu_abi_version": (C.c_int, []),
    "pine_gpu_plan_test_traverse_baked": (C.c_int, [C.c_void_p, c_f_p, C.c_int64, C.POINTER(C.c_uint32)]),
    "pine_gpu_test_specialize_compile": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_char_p, C.c_char_p, C.c_int64]),
    "pine_gpu_scene_specialized_source": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pine_gpu_mat4_identity": (None, [f16]),
    "pine_gpu_mat4_translate": (None, [f3, f16]),
    "pine_gpu_mat4_scale": (None, [f3, f16]),
    "pine_gpu_mat4_rotate_x": (None, [C.c_float, f16]),
    "pine_gpu_mat4_rotate_y": (None, [C.c_float, f16]),
    "pine_gpu_mat4_rotate_z": (None, [C.c_float, f16]),
    "pine_gpu_mat4_mul": (None, [f16, f16, f16]),
    "pine_gpu_mat4_inverse": (None, [f16, f16]),
    "pine_gpu_mat4_look_at": (None, [f3, f3, f16]),
    "pine_gpu_mat4_from_quaternion": (None, [C.c_float, C.c_float, C.c_float, C.c_float, f16]),
    "pine_gpu_mat4_from_rows": (None, [f16, f16]),
    "pine_gpu_mat4_transpose": (None, [f16, f16]),
    "pine_gpu_scene_create": (C.c_void_p, []),
    "pine_gpu_scene_destroy": (None, [C.c_void_p]),
    "pine_gpu_scene_add_material_emissive": (C.c_int, [C.c_void_p, C.c_char_p, f3]),
    "pine_gpu_scene_add_material_diffuse": (C.c_int, [C.c_void_p, C.c_char_p, f3]),
    "pine_gpu_scene_add_material_uber": (C.c_int, [C.c_void_p, C.c_char_p, f3, C.c_float, C.c_float, C.c_float, C.c_float]),
    "pine_gpu_scene_add_material_subsurface": (C.c_int, [C.c_void_p, C.c_char_p, f3, C.c_float, f3]),
    "pine_gpu_scene_find_material": (C.c_int, [C.c_void_p, C.c_char_p]),
    "pine_gpu_scene_add_light_point": (C.c_int, [C.c_void_p, f3, f3]),
    "pine_gpu_scene_add_light_spot": (C.c_int, [C.c_void_p, f3, f3, f3, C.c_float, C.c_float]),
    "pine_gpu_scene_add_light_directional": (C.c_int, [C.c_void_p, f3, f3]),
    "pine_gpu_scene_set_env_sky": (C.c_int, [C.c_void_p, f3]),
    "pine_gpu_scene_node_constf": (C.c_int, [C.c_void_p, C.c_float]),
    "pine_gpu_scene_node_const3": (C.c_int, [C.c_void_p, f3]),
    "pine_gpu_scene_node_input": (C.c_int, [C.c_void_p, C.c_int]),
    "pine_gpu_scene_node_binary": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pine_gpu_scene_node_unary": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pine_gpu_scene_node_component": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pine_gpu_scene_node_to_vec3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pine_gpu_scene_node_checkerboard": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "pine_gpu_scene_node_splat": (C.c_int, [C.c_void_p, C.c_int]),
    "pine_gpu_scene_node_is_vec3": (C.c_int, [C.c_void_p, C.c_int]),
    "pine_gpu_scene_add_material_diffuse_n": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "pine_gpu_scene_add_material_uber_n": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "pine_gpu_scene_add_material_metal": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "pine_gpu_scene_add_material_glossy": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "pine_gpu_scene_add_material_glass": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "pine_gpu_scene_add_rect": (C.c_int, [C.c_void_p, f3, f3, f3, C.c_int, C.c_int]),
    "pine_gpu_scene_add_aabb": (C.c_int, [C.c_void_p, f3, f3, C.c_int]),
    "pine_gpu_scene_add_obb": (C.c_int, [C.c_void_p, f3, f3, f16, C.c_int]),
    "pine_gpu_scene_add_sphere": (C.c_int, [C.c_void_p, f3, C.c_float, C.c_int]),
    "pine_gpu_scene_add_disk": (C.c_int, [C.c_void_p, f3, f3, C.c_float, C.c_int]),
    "pine_gpu_scene_add_cone": (C.c_int, [C.c_void_p, f3, f3, C.c_float, C.c_float, C.c_int]),
    "pine_gpu_scene_add_plane": (C.c_int, [C.c_void_p, f3, f3, C.c_int]),
    "pine_gpu_scene_add_line": (C.c_int, [C.c_void_p, f3, f3, C.c_float, C.c_int]),
    "pine_gpu_scene_add_cylinder": (C.c_int, [C.c_void_p, f3, f3, C.c_float, C.c_int]),
    "pine_gpu_scene_add_triangle": (C.c_int, [C.c_void_p, f3, f3, f3, C.c_int]),
    "pine_gpu_scene_add_mesh_full": (C.c_int, [C.c_void_p, c_f_p, C.c_int, C.POINTER(C.c_uint32), C.c_int, c_f_p, c_f_p, C.c_int]),
    "pine_gpu_mesh_apply": (C.c_int, [c_f_p, C.c_int, c_f_p, f16]),
    "pine_gpu_scene_add_rect_state": (C.c_int, [C.c_void_p, f3, f3, f3, f3, C.c_float, C.c_float, f3, f3, C.c_int]),
    "pine_gpu_scene_add_disk_state": (C.c_int, [C.c_void_p, f3, f3, f3, f3, C.c_float, C.c_int]),
    "pine_gpu_scene_add_plane_state": (C.c_int, [C.c_void_p, f3, f3, f3, f3, C.c_int]),
    "pine_gpu_scene_add_cone_state": (C.c_int, [C.c_void_p, f3, f3, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, f3, C.c_int]),
    "pine_gpu_scene_add_triangle_state": (C.c_int, [C.c_void_p, f3, f3, f3, f3, C.c_int]),
    "pine_gpu_scene_add_mesh": (C.c_int, [C.c_void_p, c_f_p, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int]),
    "pine_gpu_scene_set_camera_thinlens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, f3, f3, C.c_float, C.c_float, C.c_float]),
    "pine_gpu_scene_set_camera_thinlens_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, f3, C.c_float * 9, C.c_float * 2, C.c_float, C.c_float]),
    "pine_gpu_scene_shape_record": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "pine_gpu_scene_camera_record": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "pine_gpu_scene_describe": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pine_gpu_scene_build_accel": (C.c_int, [C.c_void_p]),
    "pine_gpu_scene_build_accel_device": (C.c_int, [C.c_void_p, C.c_int]),
    "pine_gpu_scene_accel_dump": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int64]),
    "pine_gpu_shard_of_pixel": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pine_gpu_set_table_path": (C.c_int, [C.c_char_p]),
    "pine_gpu_path_render": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), c_f_p]),
    "pine_gpu_path_render_multi": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_uint64, c_f_p]),
    "pine_gpu_path_render_devices": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.POINTER(C.c_int), C.c_int, c_f_p]),
    "pine_gpu_plan_create": (C.c_void_p, [C.c_void_p, C.POINTER(RenderParams)]),
    "pine_gpu_plan_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pine_gpu_plan_launch_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pine_gpu_packed_slab_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "pine_gpu_packed_offset": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "pine_gpu_film_unpack": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pine_gpu_plan_destroy": (None, [C.c_void_p]),
    "pine_gpu_plan_stats_get": (C.c_int, [C.c_void_p, C.POINTER(PlanStats)]),
    "pine_gpu_plan_check": (C.c_int, [C.c_void_p]),
    "pine_gpu_plan_read_samples": (C.c_int, [C.c_void_p, c_f_p, C.c_int64]),
    "pine_gpu_plan_debug_sections": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "pine_gpu_plan_vertex_log": (C.c_int64, [C.c_void_p, c_f_p, C.c_int64]),
    "pine_gpu_test_lomuto": (C.c_int, [C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pine_gpu_test_sampler": (C.c_int, [C.c_int, C.c_int, c_f_p, C.c_int64]),
    "pine_gpu_test_rng": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.c_int64]),
    "pine_gpu_test_sincos": (C.c_int, [C.c_int, c_f_p, C.c_int64, c_f_p, c_f_p]),
    "pine_gpu_test_powlog": (C.c_int, [C.c_int, c_f_p, c_f_p, C.c_int64, c_f_p, c_f_p]),
    "pine_gpu_test_atan": (C.c_int, [C.c_int, c_f_p, c_f_p, C.c_int64, c_f_p, c_f_p]),
    "pine_gpu_test_embree_tree": (C.c_int, [c_f_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "pine_gpu_test_traverse": (C.c_int, [C.c_void_p, C.c_int, c_f_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "pine_gpu_scene_accel_bvhs": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int64]),
    "pine_gpu_test_shapes": (C.c_int, [C.c_void_p, C.c_int, c_f_p, C.c_int64, c_f_p, C.c_int64]),
    "pine_gpu_film_finalize_u8": (C.c_int, [c_f_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args

lib.pine_gpu_set_table_path(TABLE_PATH.encode())


class PineError(RuntimeError):
    """Raised where the reference would SEVERE()/abort (src/pine/core/log.h:45-51)."""


def last_error() -> str:
    return (lib.pine_gpu_last_error() or b"").decode()


def check(rc, what=""):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise PineError(f"{what}: {last_error()}" if what else last_error())
    return rc
