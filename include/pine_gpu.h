/* include/pine_gpu.h -- C ABI of the MI355X-native PathIntegrator hot path (libpine_gpu.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository wicstas/pine).  The reference exposes the path to scripts through its Context
 * reflection table (src/pine/core/context.h:235-609, registered in
 * src/pine/core/program_context.cpp:23-125); INTEGRATION.md shows the thunks a pine maintainer
 * would register to route `PathIntegrator(...).render(scene)` through this library.
 *
 * Conventions: every function returning int returns 0 (or a non-negative id) on success and a
 * negative value on failure, with a message available from pine_gpu_last_error() (the reference
 * aborts through SEVERE, src/pine/core/log.h:45-51; nothing aborts across this ABI).  Vectors are
 * float[3]; matrices are float[16] in the reference's storage order (column vectors,
 * src/pine/core/vecmath.h:575-640), i.e. m[c*4 + r].  Thread-safety: a scene may be built from one
 * thread at a time; rendering distinct plans from distinct threads is safe; pine_gpu_progress() may
 * be polled from any thread (src/cli/pine.cpp:36-40 does exactly that).
 */
#ifndef PINE_GPU_H
#define PINE_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PINE_GPU_ABI_VERSION 4

typedef struct pine_gpu_scene pine_gpu_scene; /* replaces pine::Scene, src/pine/core/scene.h:14-43 */
typedef struct pine_gpu_plan pine_gpu_plan;   /* a PathIntegrator bound to a scene + device state */

/* ---- errors / progress ------------------------------------------------------------------ */
const char* pine_gpu_last_error(void);  /* thread-local message of the last failing call */
float pine_gpu_progress(void);          /* get_progress(), src/pine/core/integrator.cpp:17-19 */
int pine_gpu_abi_version(void);

/* ---- host math used by scene scripts (PRL builtins) ---------------------------------------
 * translate/scale/rotate_x|y|z/look_at: src/pine/core/vecmath.h:1102-1180; operator*(mat4,mat4)
 * :617-624; inverse(mat4): src/pine/core/vecmath.cpp:103-132.  Same operand order, same libm. */
void pine_gpu_mat4_identity(float out[16]);
void pine_gpu_mat4_translate(const float v[3], float out[16]);
void pine_gpu_mat4_scale(const float v[3], float out[16]);
void pine_gpu_mat4_rotate_x(float rad, float out[16]);
void pine_gpu_mat4_rotate_y(float rad, float out[16]);
void pine_gpu_mat4_rotate_z(float rad, float out[16]);
void pine_gpu_mat4_mul(const float a[16], const float b[16], float out[16]);
void pine_gpu_mat4_inverse(const float m[16], float out[16]);
void pine_gpu_mat4_look_at(const float from[3], const float at[3], float out[16]);
/* what the reference's glTF import composes node transforms from (src/pine/core/fileio.cpp:127-169): q2m(w, x, y, z), the
 * matrix's scalar constructor (row-major arguments) and transpose */
void pine_gpu_mat4_from_quaternion(float w, float x, float y, float z, float out[16]);
void pine_gpu_mat4_from_rows(const float rows[16], float out[16]);
void pine_gpu_mat4_transpose(const float m[16], float out[16]);

/* ---- Scene ---------------------------------------------------------------------------------
 * Scene(): src/pine/core/scene.cpp:64-79 (`Scene` ctor + `add`/`set` methods). */
pine_gpu_scene* pine_gpu_scene_create(void);
void pine_gpu_scene_destroy(pine_gpu_scene* scene);

/* scene.add(name, Material): Scene::add_material src/pine/core/scene.cpp:8-13.  Constant shading
 * nodes only (Node3f/Nodef literals).  Returns the material id.  A later add with the same name
 * shadows the earlier one, as the reference's map assignment does. */
int pine_gpu_scene_add_material_emissive(pine_gpu_scene*, const char* name, const float color[3]);
                                         /* Emissive(Node3f)   src/pine/core/material.h:18-28  */
int pine_gpu_scene_add_material_diffuse(pine_gpu_scene*, const char* name, const float albedo[3]);
                                         /* Diffuse(Node3f)    src/pine/core/material.h:30-37  */
int pine_gpu_scene_add_material_uber(pine_gpu_scene*, const char* name, const float albedo[3],
                                     float roughness, float metallic, float transmission, float ior);
                                         /* Uber(...)          src/pine/core/material.h:80-97  */
int pine_gpu_scene_add_material_subsurface(pine_gpu_scene*, const char* name, const float albedo[3],
                                           float roughness, const float sigma_s[3]);
                                         /* Subsurface(...)    src/pine/core/material.h:99-110 */
int pine_gpu_scene_find_material(pine_gpu_scene*, const char* name);
                                         /* Scene::find_material src/pine/core/scene.cpp:49-54 */

/* scene.add(Light) / scene.set(EnvironmentLight): Scene::add_light, set_env_light src/pine/core/scene.cpp:29-46.
 * PointLight / SpotLight / DirectionalLight (delta lights: sampled without MIS, path.cpp:104-106) and
 * the Sky environment light (added to radiance on a miss, path.cpp:75-81; sampled uniformly over the
 * sphere): src/pine/core/light.h:21-67, light.cpp:11-84.  Lights enter the light sampler's list in
 * add order together with the area lights of emissive geometry; the environment light comes last
 * (lightsampler.cpp:6-10).  Atmosphere and ImageSky are not supported. */
int pine_gpu_scene_add_light_point(pine_gpu_scene*, const float position[3], const float color[3]);
int pine_gpu_scene_add_light_spot(pine_gpu_scene*, const float position[3], const float direction[3],
                                  const float color[3], float falloff_radian, float cutoff_additional_radian);
int pine_gpu_scene_add_light_directional(pine_gpu_scene*, const float direction[3], const float color[3]);
int pine_gpu_scene_set_env_sky(pine_gpu_scene*, const float sun_color[3]);

/* Shading nodes (Nodef / Node3f: src/pine/core/node.h:13-297, registered node.cpp:29-116).  A node
 * lives in the scene's node table; each call returns its id (or < 0).  Supported: constants, the
 * surface inputs Position / Normal / UV, NodeBinary (+ - * / ^), NodeUnary (- abs sqr sqrt fract),
 * NodeComponent, NodeToVec3, NodeCheckerboard; `lerp` etc. are compositions of these exactly as
 * node.cpp:88-102 composes them.  Not supported: noise, image and function nodes.  Subtrees that do
 * not read the surface are folded to literals on the host with the same float operations; the rest
 * run as small postfix programs in the shading stage of the kernels. */
int pine_gpu_scene_node_constf(pine_gpu_scene*, float value);                 /* Nodef(float)   node.h:262 */
int pine_gpu_scene_node_const3(pine_gpu_scene*, const float value[3]);        /* Node3f(vec3)   node.h:281 */
int pine_gpu_scene_node_input(pine_gpu_scene*, int which);                    /* 0 Position, 1 Normal, 2 UV  node.h:113-127 */
int pine_gpu_scene_node_binary(pine_gpu_scene*, int op, int a, int b);        /* op: one of + - * / ^ (as a char); both Nodef or both Node3f  node.h:129-152 */
int pine_gpu_scene_node_unary(pine_gpu_scene*, int op, int a);                /* '-' neg, 'a' abs, 's' sqr, 'r' sqrt, 'f' fract  node.h:154-178 */
int pine_gpu_scene_node_component(pine_gpu_scene*, int a, int component);     /* NodeComponent(Node3f, 0..2) node.h:180-193 */
int pine_gpu_scene_node_to_vec3(pine_gpu_scene*, int x, int y, int z);        /* NodeToVec3(x) when y = z = -1  node.h:195-210 */
int pine_gpu_scene_node_checkerboard(pine_gpu_scene*, int p, float ratio);    /* NodeCheckerboard node.h:232-239, node.cpp:15-18 */
int pine_gpu_scene_node_splat(pine_gpu_scene*, int a);                        /* a Nodef where a Node3f is expected  node.h:78,291-293 */
int pine_gpu_scene_node_is_vec3(pine_gpu_scene*, int id);                     /* 1 Node3f, 0 Nodef */
/* Materials whose parameters are nodes (albedo: Node3f; the others: Nodef). */
int pine_gpu_scene_add_material_diffuse_n(pine_gpu_scene*, const char* name, int albedo);
int pine_gpu_scene_add_material_uber_n(pine_gpu_scene*, const char* name, int albedo, int roughness, int metallic,
                                       int transmission, float ior);
int pine_gpu_scene_add_material_metal(pine_gpu_scene*, const char* name, int albedo, int roughness);
                                         /* Metal(Node3f, Nodef)         src/pine/core/material.h:39-50 */
int pine_gpu_scene_add_material_glossy(pine_gpu_scene*, const char* name, int albedo, int roughness, int ior);
                                         /* Glossy(Node3f, Nodef, Nodef) src/pine/core/material.h:52-64 */
int pine_gpu_scene_add_material_glass(pine_gpu_scene*, const char* name, int albedo, int roughness, int ior);
                                         /* Glass(Node3f, Nodef, Nodef)  src/pine/core/material.h:66-78 */

/* scene.add(Shape, material): Scene::add_geometry src/pine/core/scene.cpp:14-22 (emissive geometry
 * becomes an AreaLight automatically).  Returns the geometry index.  Shape constructors:
 * src/pine/core/geometry.cpp:901-946. */
int pine_gpu_scene_add_rect(pine_gpu_scene*, const float position[3], const float ex[3],
                            const float ey[3], int flip_normal, int material);
                                         /* Rect(vec3,vec3,vec3,bool) geometry.cpp:255-267 */
int pine_gpu_scene_add_aabb(pine_gpu_scene*, const float lower[3], const float upper[3], int material);
                                         /* Box(vec3,vec3) = AABB     bbox.h:29-33         */
int pine_gpu_scene_add_obb(pine_gpu_scene*, const float lower[3], const float upper[3],
                           const float m[16], int material);
                                         /* Box(AABB,mat4) = OBB      bbox.cpp:144         */
int pine_gpu_scene_add_sphere(pine_gpu_scene*, const float center[3], float radius, int material);
                                         /* Sphere(vec3,float)        geometry.cpp:72      */
int pine_gpu_scene_add_disk(pine_gpu_scene*, const float position[3], const float normal[3],
                            float radius, int material);
                                         /* Disk(vec3,vec3,float)     geometry.cpp:123-127 */
int pine_gpu_scene_add_cone(pine_gpu_scene*, const float position[3], const float normal[3],
                            float radius, float height, int material);
                                         /* Cone(vec3,vec3,float,float) geometry.cpp:409-414 */
int pine_gpu_scene_add_plane(pine_gpu_scene*, const float position[3], const float normal[3], int material);
                                         /* Plane(vec3,vec3)          geometry.cpp:31-34   */
int pine_gpu_scene_add_line(pine_gpu_scene*, const float p0[3], const float p1[3], float thickness,
                            int material);
                                         /* Line(vec3,vec3,float)     geometry.cpp:171-179 */
int pine_gpu_scene_add_cylinder(pine_gpu_scene*, const float p0[3], const float p1[3], float radius,
                                int material);
                                         /* Cylinder(vec3,vec3,float) geometry.h:140-141 (side surface only;
                                            not usable as a light, as in the reference)    */
int pine_gpu_scene_add_triangle(pine_gpu_scene*, const float v0[3], const float v1[3], const float v2[3],
                                int material);
                                         /* Triangle(vec3,vec3,vec3)  geometry.cpp:528-531 */
/* Mesh(vertices, indices, texcoords, normals) geometry.cpp:596-604: per-vertex normals (interpolated shading normal,
 * geometry.h:199-204) and / or texture coordinates (:205-210), as the reference's glTF import produces them; either may be
 * null.  pine_gpu_mesh_apply: Mesh::apply(mat4) geometry.cpp:647-653 in place on caller arrays (the import applies a node's
 * accumulated transform to its mesh before adding it). */
int pine_gpu_scene_add_mesh_full(pine_gpu_scene*, const float* vertices, int num_vertices, const uint32_t* indices, int num_triangles,
                                 const float* normals, const float* texcoords, int material);
int pine_gpu_mesh_apply(float* vertices, int num_vertices, float* normals, const float m[16]);

/* State-level forms, for a binding that walks an already constructed pine::Scene (examples/adapter):
 * the members the reference's shape object keeps, exactly as stored -- a constructed Rect / Disk / Plane / Cone holds
 * NORMALISED axes and derived lengths that do not invert to its constructor arguments bit for bit.  (Sphere, Box,
 * Line, Cylinder and Mesh store their constructor arguments: use the calls above.)  Member lists:
 * Rect geometry.h:92-96, Disk :56-60, Plane :23-25, Cone :134-140 (bottom_position = its Disk's centre), Triangle :115-117. */
int pine_gpu_scene_add_rect_state(pine_gpu_scene*, const float position[3], const float ex[3], const float ey[3], const float n[3],
                                  float lx, float ly, const float rx[3], const float ry[3], int material);
int pine_gpu_scene_add_disk_state(pine_gpu_scene*, const float position[3], const float n[3], const float u[3], const float v[3],
                                  float r, int material);
int pine_gpu_scene_add_plane_state(pine_gpu_scene*, const float position[3], const float n[3], const float u[3], const float v[3],
                                   int material);
int pine_gpu_scene_add_cone_state(pine_gpu_scene*, const float apex[3], const float n[3], float r, float h, float A, float A2, float S,
                                  const float bottom_position[3], int material);
int pine_gpu_scene_add_triangle_state(pine_gpu_scene*, const float v0[3], const float v1[3], const float v2[3], const float n[3],
                                      int material);
int pine_gpu_scene_add_mesh(pine_gpu_scene*, const float* vertices, int num_vertices,
                            const uint32_t* indices, int num_triangles, int material);
                                         /* Mesh(vertices, indices)   geometry.cpp:601-609 */

/* scene.set(ThinLenCamera(Film(size, tonemapper), from, to, fov[, len_radius, focus_distance])):
 * src/pine/core/camera.cpp:7-16,40-45; Film src/pine/core/film.h:24-27.
 * tonemapper: 0 = Uncharted2, 1 = ACES (used only by pine_gpu_film_finalize). */
int pine_gpu_scene_set_camera_thinlens(pine_gpu_scene*, int film_w, int film_h, int tonemapper,
                                       const float from[3], const float to[3], float fov,
                                       float len_radius, float focus_distance);

/* State-level form: the members of a constructed ThinLenCamera (src/pine/core/camera.h:21-26); c2w = its mat3, 9 floats,
 * columns x, y, z. */
int pine_gpu_scene_set_camera_thinlens_state(pine_gpu_scene*, int film_w, int film_h, int tonemapper, const float position[3],
                                             const float c2w[9], const float fov2d[2], float len_radius, float focus_distance);

/* Test hooks: the 128-byte device record of geometry `index` (30 floats of state, kind, material: pine_types.h DShape)
 * and the camera record -- what constructor-level and state-level calls must agree on. */
int pine_gpu_scene_shape_record(pine_gpu_scene*, int index, float out[32]);
int pine_gpu_scene_camera_record(pine_gpu_scene*, float out[20]);

/* Text dump of the scene as it was built (the .pscene exchange format, pine_amd/scene_io.py).
 * Returns the number of bytes needed (excluding NUL); writes at most `capacity` bytes. */
int64_t pine_gpu_scene_describe(pine_gpu_scene*, char* buf, int64_t capacity);

/* Host-side accel build only (no GPU): BVH::build src/pine/impl/accel/bvh.cpp:453-495.
 * Returns node count; optional dumps for tests: nodes as 16 x int32/float32 words each. */
int pine_gpu_scene_build_accel(pine_gpu_scene*);
int64_t pine_gpu_scene_accel_dump(pine_gpu_scene*, void* nodes_out, int64_t node_capacity_bytes,
                                  int32_t* prims_out, int64_t prim_capacity);

/* ---- PathIntegrator ------------------------------------------------------------------------
 * PathIntegrator(Accel, Sampler, LightSampler, int) + render(Scene&):
 * src/pine/impl/integrator/path.cpp:7-41, registered program_context.cpp:76-81.  The sampler is
 * BlueSampler(spp) (src/pine/core/sampler.cpp:115-121: spp rounded up to a power of two, clamped
 * to 256) or SobolSampler(spp), the light sampler UniformLightSampler, the accel pine's BVH order.
 */
typedef struct {
  int32_t spp;             /* requested samples per pixel (BlueSampler argument)            */
  int32_t max_path_length; /* PathIntegrator depth argument (> 0)                           */
  int32_t device;          /* HIP device ordinal                                            */
  int32_t shard_rank;      /* this process renders tiles t with t % shard_world == rank     */
  int32_t shard_world;     /* 1 = whole film                                                */
  int32_t samples_per_item;/* 0 = auto; work item = this many consecutive samples of a pixel */
  int32_t flags;           /* PINE_GPU_FLAG_*                                               */
  int32_t sampler;         /* PINE_GPU_SAMPLER_*: which Sampler the PathIntegrator is constructed with   */
} pine_gpu_render_params;

#define PINE_GPU_SAMPLER_BLUE  0 /* BlueSampler(spp)  src/pine/core/sampler.h:166-201 (the default)           */
#define PINE_GPU_SAMPLER_SOBOL 1 /* SobolSampler(spp) src/pine/core/sampler.h:83-164, sampler.cpp:81-113: spp
                                    as given (no clamp to 256, any count); on the device up to 4096 (a count that is
                                    not a power of two renders one work item per pixel)                      */
#define PINE_GPU_SAMPLER_HALTON 2 /* HaltonSampler(spp) src/pine/core/sampler.h:40-81, sampler.cpp:39-79,
                                    lowdiscrepancy.h:26-51: scrambled radical inverses over the first primes; on the
                                    device under SobolSampler's limit (4096 samples per pixel)               */

#define PINE_GPU_FLAG_TIMING 1 /* record per-kernel HIP-event timings for the roofline report */
#define PINE_GPU_FLAG_PROGRESS 2 /* the kernels post the claimed work-item count to host memory now and then, so that
                                    pine_gpu_progress() moves while a launch runs (pine_gpu_path_render sets it itself) */
#define PINE_GPU_FLAG_FAST 4 /* declared-tolerance arithmetic instead of bit-exact parity with the reference: contracted
                                    multiply-adds, 1-ulp hardware reciprocal / square root / division, the device's native
                                    sin / cos / pow / log.  Integer work -- sampler, RNG, hash -- stays exact.  Declared tolerance
                                    (tests/test_gpu_parity.py FAST_TOLERANCE): where the path depends continuously on the
                                    float bits (Rect-only cbox) >= 99.9 % of the pixels are within relative L2 1e-4 of the exact film (RMSE 4e-6); where the
                                    reference's algorithm decides on nearly equal numbers (scaled OBBs, grazing cones, near-delta
                                    lobes) 0.8 - 2 % of the samples take another path and the films agree as Monte-Carlo estimates:
                                    RMSE at 256 spp <= 4e-3 (cbox) / 2e-2 (10 000 cones), bias of the image mean <= 5e-3.  Only
                                    scenes one of the fast variants covers (the BASELINE scenes' feature sets); others fail with a
                                    message.  Never the default, never the parity gate. */
#define PINE_GPU_FLAG_DEVICE_BVH 8 /* build the BVH on the GPU (the same level-synchronous binned-SAH build as on the host, same tree, same
                                    primitive order: pine_amd/csrc/pine_bvh_build_device.h) when the plan is the first to need the scene's
                                    accel; also $PINE_GPU_DEVICE_BVH=1.  pine_gpu_plan_stats.accel_built_on_device says what happened. */
#define PINE_GPU_FLAG_DEBUG_FORCE_BAIL 0x100 /* test hook: the stage-queued path kernel raises its protocol-failure
                                    bail-out at once; every synchronising entry point must then FAIL (never return the film) */
/* Scene-specialised kernels.  The path kernel can be compiled FOR THE SCENE: (1) with exactly the scene's
 * feature set (shape kinds, material lobes, node programs, light kinds, sampler) instead of the nearest precompiled superset;
 * (2) small scenes without meshes (at most 10 primitives: cbox-class) get their BVH and every primitive record baked into the
 * kernel as immediates, the traversal fully unrolled; a small top level around ONE mesh becomes code run where a ray is
 * created, and only rays that reach the mesh enter the traversal stages.  Same tests in pine's order: bit-identical films;
 * cbox 24 % faster, the Subsurface icosphere 46 %.  A build is one `hipcc --genco` run in a child process (seconds), cached
 * on disk by content ($PINE_GPU_CACHE_DIR, else ~/.cache/pine_gpu); it needs hipcc and this library's device headers.
 *
 * DEFAULT (no flag) -- automatic, never in the caller's way: a code object already in the cache is loaded at plan creation
 * (about a millisecond); otherwise the compiler runs in the BACKGROUND while the precompiled kernel renders, and the first
 * launch after it has finished -- of this plan or of any later plan or pine_gpu_path_render call on the same geometry -- runs
 * the scene's own kernel.  Nothing fails because of it: no compiler, no headers, no cache directory or a full compile queue
 * leave the precompiled kernel in place.  What ran is in pine_gpu_plan_stats (`specialized`, `specialize_source`,
 * `specialize_pending`).  $PINE_GPU_SPECIALIZE=0 (or PINE_GPU_FLAG_NO_SPECIALIZE): precompiled kernels only. */
#define PINE_GPU_FLAG_SPECIALIZE 0x400 /* the caller WANTS the scene's kernel from the first launch: plan creation waits for the
                                    compiler, and a kernel that cannot be built fails the plan.  A scene with nothing to gain
                                    renders with the precompiled kernel.  Also $PINE_GPU_SPECIALIZE=1 (every plan). */
#define PINE_GPU_FLAG_SPECIALIZE_NO_BAKE 0x800 /* (either mode) the exact feature set only, never the baked scene --
                                    for geometry that changes from render to render (an animation): a baked kernel is keyed by
                                    the geometry and would be compiled per frame, a feature-set kernel once */
#define PINE_GPU_FLAG_SPECIALIZE_ASYNC 0x1000 /* with PINE_GPU_FLAG_SPECIALIZE: plan creation does not wait for the compiler -- the
                                    background build of the default mode, but a build that fails is reported (plan stats:
                                    specialized == -1) instead of passing silently */
#define PINE_GPU_FLAG_NO_SPECIALIZE 0x2000 /* the precompiled kernel table only: no cache lookup, no background compiler */
#define PINE_GPU_FLAG_ORDER_EMBREE 0x4000 /* closest-hit queries hand the scene's non-mesh shapes to their tests in the order of the reference's
                                    DEFAULT accel, EmbreeAccel (src/pine/impl/accel/embree.cpp:101-143,195-257; what a .pine script's
                                    PathIntegrator(sampler, n) gets on real pine, program_context.cpp:79-81), instead of pine-BVH order.  pine
                                    registers every such shape as one Embree user primitive with pine's own bounds and intersect callbacks, so
                                    Embree decides only the ORDER -- and pine has shapes whose answer depends on it: the scaled
                                    Box(AABB, mat4) (src/pine/core/bbox.cpp:149-171), Plane's finite bounds, Line, Cylinder.  The order is
                                    restated from the vendored Embree 4.3.1's BVH8 builder and single-ray traverser as an AVX2 x86 host runs
                                    them (pine_amd/csrc/pine_embree_order.h, scene_traverse_embree); the films of the real reference built
                                    with EmbreeAccel are reproduced bit for bit (tests/golden/film_embree_*, any number of shapes), without
                                    the flag (the default, and the parity gate) those of Accel(BVH()).  Meshes are Embree triangle
                                    geometry there: they are asked first, through Embree's own Moeller-Trumbore test and barycentrics
                                    (restated too); only an exact tie in t between coplanar triangles of different meshes is decided by
                                    Embree's own triangle hierarchy, which is not.  Not with _FAST. */
#define PINE_GPU_FLAG_ORDER_NEAREST PINE_GPU_FLAG_ORDER_EMBREE /* (the name of this flag before the order was Embree's own for any shape count) */
#define PINE_GPU_FLAG_VERTEX_LOG 0x200 /* test hook: choose the kernel variant compiled with the per-vertex log (pine_gpu_plan_vertex_log) */

/* Multi-GPU partition: rank that owns pixel (x, y) of a film_w-wide film when 8x8-pixel tiles are
 * dealt round-robin to `world` ranks (host-side helper; the kernels use the same mapping). */
int pine_gpu_shard_of_pixel(int film_w, int x, int y, int world);

/* Path to the packed BlueSobol tables (pine_amd/data/bluesobol_u8.bin).  Optional: without it the library looks at
 * $PINE_GPU_TABLES, then at ../data/bluesobol_u8.bin relative to the directory libpine_gpu.so was loaded from. */
int pine_gpu_set_table_path(const char* path);

/* One-shot, drop-in form: render into a HOST film of W*H float4 (row 0 first, exactly
 * Array2d<vec4>, src/pine/core/array.h:51-55); includes upload + download.  Fails if no GPU. */
int pine_gpu_path_render(pine_gpu_scene*, const pine_gpu_render_params*, float* film_out_host);

/* The same on several devices of one node from ONE process (`pine_gpu_path_render(..., device_mask, ...)`):
 * shard r of n -- 8x8-pixel tiles dealt round-robin, SURVEY.md 8(e) -- renders on the r-th selected device; the per-device
 * tile slabs travel to the first device with peer copies (xGMI), are scattered into the film there and downloaded.  The film
 * is bit-identical to the one-device film.  `prm->device / shard_rank / shard_world` are ignored.  The reference's
 * counterpart is its thread pool over one shared film (src/pine/core/parallel.h:19-67, path.cpp:31-39).
 * _multi: bit d of `device_mask` selects HIP device d.  _devices: an explicit list (a device may appear more than once). */
int pine_gpu_path_render_multi(pine_gpu_scene*, const pine_gpu_render_params*, uint64_t device_mask, float* film_out_host);
int pine_gpu_path_render_devices(pine_gpu_scene*, const pine_gpu_render_params*, const int* devices, int num_devices,
                                 float* film_out_host);

/* Resident form (bench / multi-GPU): build device state once, launch many times.
 * `film_dev` is a DEVICE pointer to W*H float4; `stream` is a hipStream_t (0 = default stream).
 * Pixels outside this rank's shard are written as zeros, so a sum-reduce over ranks is exact. */
pine_gpu_plan* pine_gpu_plan_create(pine_gpu_scene*, const pine_gpu_render_params*);
/* Device memory of destroyed plans (the per-sample radiance buffer of a 640 x 640 x 256 render is 1.7 GB) is kept for the next
 * plan instead of being returned to the driver: a one-shot pine_gpu_path_render spends most of its time outside the kernels in
 * hipMalloc / hipFree otherwise.  At most $PINE_GPU_POOL_MB (default 16 384; 0 = keep nothing) per process; this call
 * returns all of it, and drops the process's table of loaded scene kernels (the last 16 code objects stay loaded per device so
 * that the next plan of the same geometry does not load its kernel again).  The reference has no counterpart (its film and per-thread state live in host memory). */
void pine_gpu_release_cached_memory(void);
int pine_gpu_plan_launch(pine_gpu_plan*, void* film_dev, void* stream);
void pine_gpu_plan_destroy(pine_gpu_plan*);

/* Multi-GPU without moving the zeros: the rank writes only its own tiles, tile-major, into a slab
 * of pine_gpu_packed_slab_floats(...) floats ([local tile][pixel in 8x8 tile] float4; equal size on
 * every rank, the last tile may be padding); the slabs are gathered to one rank ([rank][slab]) and
 * pine_gpu_film_unpack scatters them into the row-major W*H float4 film there.  This replaces the
 * reference's shared-memory film (src/pine/impl/integrator/path.cpp:38 writes film[p] from any
 * thread) across devices.  pine_gpu_packed_offset is the host-side statement of the same mapping. */
int pine_gpu_plan_launch_packed(pine_gpu_plan*, void* slab_dev, void* stream);
int64_t pine_gpu_packed_slab_floats(int film_w, int film_h, int world);
int pine_gpu_packed_offset(int film_w, int film_h, int world, int x, int y, int* rank_out, int64_t* float4_index_out);
int pine_gpu_film_unpack(int film_w, int film_h, int world, int device, const void* slabs_dev, void* film_dev, void* stream);

typedef struct {
  uint64_t camera_samples;   /* samples this plan renders per launch (its shard)              */
  uint64_t vertices;         /* radiance() invocations of the last launch    */
  uint64_t shadow_rays;
  float trace_ms;            /* path kernel, HIP events on the launch stream: mean over the launches   */
  float resolve_ms;          /* ordered per-pixel sum kernel                since the previous stats_get */
  float prepass_ms;          /* RNG checkpoint kernel                       (at most the last 64)       */
  int32_t spp_effective;
  int32_t samples_per_item;
  int32_t grid_blocks;
  int32_t block_threads;
  int32_t lds_bytes;
  int32_t timed_launches;    /* number of launches the three timings are averaged over        */
  uint64_t walk_steps;       /* BSSRDF random-walk steps of the last launch (bxdf.cpp:340-351; stage-queued kernel) */
  float accel_build_ms;      /* host: BVH build + flattening for this plan (0 if the scene's accel was already built) */
  float upload_ms;           /* host: device allocation + upload of scene, tables and work buffers at plan creation  */
  int32_t accel_built_on_device; /* 1: that build ran on the GPU (PINE_GPU_FLAG_DEVICE_BVH)                            */
  int32_t serial_tiles;      /* tile classes (Subsurface scenes): 8x8 tiles of this shard whose pixels are one whole-pixel item each
                              * because a camera ray of theirs can reach a Subsurface shape; the others' samples are independent
                              * items of samples_per_item samples.  0: one class (samples_per_item describes every item) */
  int32_t specialized;       /* 0 a precompiled kernel runs (no gain possible, specialisation off, or a background build is still
                              * running); 1 a kernel compiled for this scene's exact feature set; 2 ... with the scene's BVH and
                              * primitive records baked in as well; -1 a background build failed, the precompiled kernel keeps running */
  float specialize_ms;       /* host: generating + compiling (or fetching from the cache) + loading that kernel at plan creation */
  uint32_t kernel_features;  /* feature bits (pine_device.h F_*) of the path kernel in use */
  int32_t specialize_source; /* where the scene's kernel came (or will come) from: 0 none; 1 the on-disk cache; 2 compiled at plan
                              * creation (PINE_GPU_FLAG_SPECIALIZE); 3 compiled in the background by this process           */
  int32_t specialize_pending;/* 1: a background build is still running (the precompiled kernel renders meanwhile)             */
  int32_t reserved;
} pine_gpu_plan_stats;
/* PINE_GPU_FLAG_SPECIALIZE, host half: the text plan creation would compile for this scene (its BVH as straight-line code,
 * boxes and primitive records as hexadecimal float literals), NUL-terminated into out[0..cap) when it fits; returns its
 * length, 0 when the scene does not qualify (meshes, a BVH too large to unroll, a non-finite record), < 0 on error.
 * Needs no GPU: a way to see what was compiled, and the CPU tests' handle on the generator. */
int64_t pine_gpu_scene_specialized_source(pine_gpu_scene*, char* out, int64_t cap);
/* ... and the compile step on its own (no GPU needed: hipcc cross-compiles): the scene's kernel for the stage-queued variant
 * <features, ctx> (pine_variants.h; the kernel's name carries features | F_BAKED) and `arch` ("gfx950"), through the same cache
 * plan creation uses; the code object's path into path_out.  A null scene compiles level 1: <features, ctx> as given, generic
 * traversal (tools/compile_sweep.py walks feature combinations with it).  Returns 1 on a cache hit, 0 after a compiler run, < 0 on failure.  Build check + tests. */
int pine_gpu_test_specialize_compile(pine_gpu_scene*, uint32_t features, int ctx, const char* arch, char* path_out, int64_t cap);
/* The scene's BVH built on HIP device `device` right now (the accel is then reused by every later plan); returns the node
 * count, < 0 on failure.  pine_gpu_scene_accel_dump shows the result: identical to the host build's. */
int pine_gpu_scene_build_accel_device(pine_gpu_scene*, int device);
/* Blocks until the last launch has finished (needed to read the device-side counters).  Fails (< 0) if the
 * path kernel of that launch bailed out of a bounded wait: its film is incomplete. */
int pine_gpu_plan_stats_get(pine_gpu_plan*, pine_gpu_plan_stats* out);
/* For callers that launch asynchronously and read the film themselves: blocks until the plan's last launch
 * has finished; 0 = it completed, < 0 = its path kernel bailed out (pine_gpu_last_error() names the wait that
 * ran out) and the film must be discarded.  The reference has no counterpart: its render() cannot fail part-way. */
int pine_gpu_plan_check(pine_gpu_plan*);

/* Diagnostic builds only (-DPINE_PROFILE_SECTIONS): per-section wave-cycle sums of the last launch
 * (all zeros in the product build). */
int pine_gpu_plan_debug_sections(pine_gpu_plan*, uint64_t out[16]);
/* Test hook (stage-queued kernel): the per-vertex terms of path.cpp:98-121 of every path of a small film.  Called with
 * out == NULL it switches the log on for the plan's next launches and returns the number of floats; called with a buffer
 * after a launch it copies the log: 16 floats per radiance() invocation at
 * [((y * W + x) * spp + sample) * max_path_length + level] = kind (0 miss, 1 emissive, 2 path-length limit, 3 shaded) |
 * length | direct term (3) | bs.f (3) | cosine | bs.pdf | is_delta | mis | returned light pdf (-1: none) | returned Lo (3)
 * -- the record `pine_ref vertices` writes from the reference's own objects (tests/golden/vertices_*.npz).  The plan must have
 * been created with PINE_GPU_FLAG_VERTEX_LOG: only two kernel variants (cbox's kinds; everything but Subsurface) carry the hook. */
int64_t pine_gpu_plan_vertex_log(pine_gpu_plan*, float* out, int64_t capacity_floats);

/* Per-sample radiance of the last launch: copies spp_eff*W*H float4 (r,g,b,vertices) to host,
 * layout [(y*W+x)*spp + s].  Test/debug aid. */
int pine_gpu_plan_read_samples(pine_gpu_plan*, float* out_host, int64_t capacity_floats);

/* Host test hook: the reference's partition (src/psl/algorithm.h:394-402) as its sequential swap loop and as the data-parallel
 * formulation of the device BVH build; both permutations out, < 0 if they disagree. */
int pine_gpu_test_lomuto(const unsigned char* pred, int n, int* perm_sequential, int* perm_parallel);

/* Device-side unit-test hooks: run the device implementations on arrays (parity vs oracle). */
int pine_gpu_test_sampler(int device, int spp, float* out_host, int64_t capacity);  /* layout of oracle_sampler_stream */
int pine_gpu_test_rng(int device, uint64_t* out_host, int64_t capacity);            /* layout of oracle_rng_stream */
int pine_gpu_test_sincos(int device, const float* x_host, int64_t n, float* sin_out, float* cos_out);
int pine_gpu_test_powlog(int device, const float* x_host, const float* y_host, int64_t n, float* pow_out,
                         float* log_out);                                           /* powf(x, y), logf(x) */
int pine_gpu_test_atan(int device, const float* y_host, const float* x_host, int64_t n, float* atan2_out,
                       float* acos_out);                                            /* atan2f(y, x), acosf(x) */
/* BVH traversal (bvh.cpp:321-451, 497-548) of the scene's accel for each ray (8 floats: o, d, tmin, tmax), by the nested
 * loops of the scene-in-LDS kernel variants (flat = 0) or the flat state machine of the others (flat = 1).  Per ray
 * 2 * cap + 5 words: [count, count test words ...] (cap words), hit, geometry, triangle, tmax bits of the closest-hit query,
 * then [count, words ...] (cap words) and hit of the any-hit query.  A test word is a top-level primitive's geometry index
 * or 0x40000000 | triangle index within the mesh entered last: the layout of `pine_ref bvh` (tests/golden/bvh_*.npz). */
int pine_gpu_test_traverse(pine_gpu_scene*, int device, const float* rays_host, int64_t nrays, int flat, int cap, uint32_t* out_host);
/* (flat = 2: the closest-hit query in EmbreeAccel's order, PINE_GPU_FLAG_ORDER_EMBREE: the geometry indices handed to their tests)
 * ... and the hierarchy of that order over n boxes (6 floats each), HOST code only: the root's child word, then 8 child words per
 * node in creation order (>= 0 a node, < 0 the complement of a box index, INT32_MIN unused).  -> words written, < 0 on failure. */
int pine_gpu_test_embree_tree(const float* boxes, int n, int* words, int cap);
/* ... and the BAKED traversal of a plan created with PINE_GPU_FLAG_SPECIALIZE (plan stats: specialized == 2) on the same kind of
 * rays: per ray 4 words -- hit, geometry, tmax bits of the closest-hit query, result of the any-hit query.  Must equal the
 * corresponding words of pine_gpu_test_traverse for every finite ray (tests/test_specialize.py: axis-parallel and grazing rays,
 * origins on the walls, zero direction components). */
int pine_gpu_plan_test_traverse_baked(pine_gpu_plan*, const float* rays_host, int64_t nrays, uint32_t* out_host);
/* The scene's BVHs after pine_gpu_scene_build_accel: per BVH (top level first, then one per mesh in geometry order) 5 words:
 * root, root_start, root_count, prim_base (DBvh) and the geometry index of its mesh (-1 for the top level).  Returns their
 * number, or < 0.  With pine_gpu_scene_accel_dump this is the whole tree (tests compare it with the reference's own). */
int pine_gpu_scene_accel_bvhs(pine_gpu_scene*, int32_t* out, int64_t capacity_words);
int pine_gpu_test_shapes(pine_gpu_scene*, int device, const float* rays_host, int64_t nrays,
                         float* out_host, int64_t capacity);                        /* layout of oracle_shapes */

/* ---- Film (host side, after the hot path) ----------------------------------------------------
 * Film::finalize + tone mapping + to_uint8_array: src/pine/core/film.cpp:12-27,66-68,
 * src/pine/core/color.cpp:6-23, src/pine/core/fileio.cpp:42-54.  rgba_out: W*H*4 bytes, y-flipped. */
int pine_gpu_film_finalize_u8(const float* film_host, int w, int h, int tonemapper, uint8_t* rgba_out);

#ifdef __cplusplus
}
#endif
#endif /* PINE_GPU_H */
