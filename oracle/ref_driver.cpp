// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Our own driver around the REAL reference (wicstas/pine), compiled against the reference sources
// where they lie under /root/reference (see oracle/Makefile).  It builds a Scene through the
// reference's C++ API from this repo's neutral scene description (.pscene, see
// pine_amd/scene_io.py), runs the reference's PathIntegrator with pine's own BVH accel
// (SURVEY.md oracle variant O-gcc-bvh) and dumps raw outputs that become golden fixtures under
// tests/golden/.  Nothing from the reference is copied here: only its public API is called.
//
//   pine_ref render  <scene.pscene> <spp> <depth> <out.film>      raw vec4 film, W*H*16 bytes
//   pine_ref sampler <spp> <out.bin>     BlueSobolSampler get1d/get2d streams (fixture 1)
//   pine_ref rng     <out.bin>           hash()/RNG known answers (fixture 1)
//   pine_ref host    <out.bin>           host-side math known answers (matrices, look_at, ctors)
//   pine_ref shapes  <scene.pscene> <rays.bin> <out.bin>   per-shape hit/intersect records
//   pine_ref gltf <file.glb> <spp> <depth> <out.film>       the reference's own glTF import + PathIntegrator(BVH)
//   pine_ref finalize <film.bin> <W> <H> <tonemapper> <out.u8> [out.png]   Film::finalize + flip + gamma + x256 (film.save)
//   pine_ref bvh     <scene.pscene> <rays.bin> <tree.bin> <trav.bin>   the reference's own BVH (bvh.cpp:30-147, 453-495) as a
//                                        canonical pre-order stream, and for every ray the primitives BVH::intersect / BVH::hit
//                                        test, in order, with the result (bvh.cpp:321-451, 497-548)
//   pine_ref accelq  <scene.pscene> <rays.bin> <out.bin>   Accel::intersect and Accel::hit of the accel $PINE_REF_ACCEL names (bvh |
//                                        embree: EmbreeAccel, pine_ref_embree only) for every ray: 10 words -- hit, geometry index,
//                                        tmax bits, any-hit result, the surface point and normal (bits).  The ray-level view of what a film is made of.
//   pine_ref vertices <scene.pscene> <spp> <depth> <out.bin>   per-vertex terms of every path of the film (path.cpp:42-124):
//                                        radiance() restated around the reference's OWN intersect / light sampler / bxdf
//                                        objects with a log; the restated loop's film is checked against render()'s
//   pine_ref prl     <literal>...        psl::stof / stoi / to_string of each literal, and the constant
//                                        expressions of a cbox-class script evaluated with psl::stof values
//                                        (pins the PRL front-end's literal and vector arithmetic)
#include <pine/core/fileio.h>
#include <pine/core/film.h>
#include <pine/core/lightsampler.h>
#include <pine/core/sampler.h>
#include <pine/core/scene.h>
#include <pine/core/rng.h>
#include <pine/impl/integrator/path.h>
#include <pine/impl/accel/bvh.h>
// BVHImpl::hit / Intersect are templates over the per-primitive callback whose definitions live in the reference's
// bvh.cpp, not in its header; the `bvh` command instantiates them with logging callbacks, so that translation unit is
// compiled as part of this one -- in place, from where it lies (the archive's own bvh.o is then simply not pulled in).
#include <pine/impl/accel/bvh.cpp>
#ifdef PINE_REF_WITH_EMBREE
#include <pine/impl/accel/embree.h>
#endif

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace pine {
// defined (with external linkage) in the reference's src/pine/core/fileio.cpp:42-54; no header declares it
psl::vector<uint8_t> to_uint8_array(vec2i size, int nchannel, const float* data, bool flip_y, bool apply_gamma);
}
using namespace pine;

static float rdf(std::istream& in) {
  std::string tok;
  in >> tok;
  return strtof(tok.c_str(), nullptr);  // accepts hexfloat
}
static vec3 rd3(std::istream& in) {
  float x = rdf(in), y = rdf(in), z = rdf(in);
  return vec3(x, y, z);
}

struct Loaded {
  Scene scene;
  int W = 0, H = 0;
};

static void load_pscene(const char* path, Loaded& out) {
  std::ifstream f(path);
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path);
    exit(2);
  }
  auto& scene = out.scene;
  // shading-node table of the description: each entry is a Nodef or a Node3f of the reference
  std::vector<psl::optional<Nodef>> nf;
  std::vector<psl::optional<Node3f>> n3;
  auto F = [&](int id) -> Nodef { return *nf[size_t(id)]; };
  auto V = [&](int id) -> Node3f { return *n3[size_t(id)]; };
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream in(line);
    std::string kw;
    in >> kw;
    if (kw == "node") {
      int id, a = -1, b = -1, c = -1;
      std::string kind, op;
      in >> id >> kind;
      nf.emplace_back();
      n3.emplace_back();
      if (kind == "constf") nf[id] = Nodef(rdf(in));
      else if (kind == "const3") n3[id] = Node3f(rd3(in));
      else if (kind == "position") n3[id] = Node3f(NodePosition());
      else if (kind == "normal") n3[id] = Node3f(NodeNormal());
      else if (kind == "uv") n3[id] = Node3f(NodeUV());
      else if (kind == "binf") {
        in >> op >> a >> b;
        switch (op[0]) {
          case '+': nf[id] = Nodef(NodeBinary<float, '+'>(F(a), F(b))); break;
          case '-': nf[id] = Nodef(NodeBinary<float, '-'>(F(a), F(b))); break;
          case '*': nf[id] = Nodef(NodeBinary<float, '*'>(F(a), F(b))); break;
          case '/': nf[id] = Nodef(NodeBinary<float, '/'>(F(a), F(b))); break;
          default: nf[id] = Nodef(NodeBinary<float, '^'>(F(a), F(b))); break;
        }
      } else if (kind == "bin3") {
        in >> op >> a >> b;
        switch (op[0]) {
          case '+': n3[id] = Node3f(NodeBinary<vec3, '+'>(V(a), V(b))); break;
          case '-': n3[id] = Node3f(NodeBinary<vec3, '-'>(V(a), V(b))); break;
          case '*': n3[id] = Node3f(NodeBinary<vec3, '*'>(V(a), V(b))); break;
          case '/': n3[id] = Node3f(NodeBinary<vec3, '/'>(V(a), V(b))); break;
          default: n3[id] = Node3f(NodeBinary<vec3, '^'>(V(a), V(b))); break;
        }
      } else if (kind == "unf") {
        in >> op >> a;
        switch (op[0]) {
          case '-': nf[id] = Nodef(NodeUnary<float, '-'>(F(a))); break;
          case 'a': nf[id] = Nodef(NodeUnary<float, 'a'>(F(a))); break;
          case 's': nf[id] = Nodef(NodeUnary<float, 's'>(F(a))); break;
          case 'r': nf[id] = Nodef(NodeUnary<float, 'r'>(F(a))); break;
          default: nf[id] = Nodef(NodeUnary<float, 'f'>(F(a))); break;
        }
      } else if (kind == "un3") {
        in >> op >> a;
        switch (op[0]) {
          case '-': n3[id] = Node3f(NodeUnary<vec3, '-'>(V(a))); break;
          case 'a': n3[id] = Node3f(NodeUnary<vec3, 'a'>(V(a))); break;
          case 's': n3[id] = Node3f(NodeUnary<vec3, 's'>(V(a))); break;
          case 'r': n3[id] = Node3f(NodeUnary<vec3, 'r'>(V(a))); break;
          default: n3[id] = Node3f(NodeUnary<vec3, 'f'>(V(a))); break;
        }
      } else if (kind == "comp") {
        int n;
        in >> a >> n;
        nf[id] = Nodef(NodeComponent(V(a), n));
      } else if (kind == "tovec3") {
        in >> a;
        if (in >> b >> c) n3[id] = Node3f(NodeToVec3(F(a), F(b), F(c)));
        else n3[id] = Node3f(NodeToVec3(F(a)));
      } else if (kind == "checker") {
        in >> a;
        nf[id] = Nodef(NodeCheckerboard(V(a), rdf(in)));
      } else if (kind == "splat") {
        in >> a;
        n3[id] = Node3f(F(a));  // Mnode<vec3> holding an Mnode<float> (node.h:78)
      } else {
        fprintf(stderr, "unknown node kind %s\n", kind.c_str());
        exit(2);
      }
    } else if (kw == "material") {
      std::string name, kind;
      in >> name >> kind;
      int a, r, m, t, i;
      if (kind == "diffuse_n") {
        in >> a;
        scene.add_material(name.c_str(), Material(DiffuseMaterial(V(a))));
      } else if (kind == "uber_n") {
        in >> a >> r >> m >> t;
        scene.add_material(name.c_str(), Material(UberMaterial(V(a), F(r), F(m), F(t), rdf(in))));
      } else if (kind == "metal") {
        in >> a >> r;
        scene.add_material(name.c_str(), Material(MetalMaterial(V(a), F(r))));
      } else if (kind == "glossy") {
        in >> a >> r >> i;
        scene.add_material(name.c_str(), Material(GlossyMaterial(V(a), F(r), F(i))));
      } else if (kind == "glass") {
        in >> a >> r >> i;
        scene.add_material(name.c_str(), Material(GlassMaterial(V(a), F(r), F(i))));
      } else if (kind == "emissive") {
        scene.add_material(name.c_str(), Material(EmissiveMaterial(rd3(in))));
      } else if (kind == "diffuse") {
        scene.add_material(name.c_str(), Material(DiffuseMaterial(rd3(in))));
      } else if (kind == "uber") {
        auto albedo = rd3(in);
        float rough = rdf(in), metal = rdf(in), trans = rdf(in), ior = rdf(in);
        scene.add_material(name.c_str(), Material(UberMaterial(albedo, rough, metal, trans, ior)));
      } else if (kind == "subsurface") {
        auto albedo = rd3(in);
        float rough = rdf(in);
        auto sigma = rd3(in);
        scene.add_material(name.c_str(), Material(SubsurfaceMaterial(albedo, rough, sigma)));
      } else {
        fprintf(stderr, "unknown material kind %s\n", kind.c_str());
        exit(2);
      }
    } else if (kw == "shape") {
      std::string kind, mat;
      in >> kind >> mat;
      if (kind == "rect") {
        auto p = rd3(in), ex = rd3(in), ey = rd3(in);
        int flip;
        in >> flip;
        scene.add_geometry(Rect(p, ex, ey, flip != 0), psl::string(mat.c_str()));
      } else if (kind == "box") {
        auto lo = rd3(in), hi = rd3(in);
        scene.add_geometry(AABB(lo, hi), psl::string(mat.c_str()));
      } else if (kind == "obb") {
        auto lo = rd3(in), hi = rd3(in);
        float m[16];
        for (auto& v : m) v = rdf(in);
        // row-major constructor arguments, as the reference's mat4 scalar ctor takes them
        auto M = mat4(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11],
                      m[12], m[13], m[14], m[15]);
        scene.add_geometry(OBB(AABB(lo, hi), M), psl::string(mat.c_str()));
      } else if (kind == "sphere") {
        auto c = rd3(in);
        float r = rdf(in);
        scene.add_geometry(Sphere(c, r), psl::string(mat.c_str()));
      } else if (kind == "disk") {
        auto p = rd3(in), n = rd3(in);
        float r = rdf(in);
        scene.add_geometry(Disk(p, n, r), psl::string(mat.c_str()));
      } else if (kind == "cone") {
        auto p = rd3(in), n = rd3(in);
        float r = rdf(in), h = rdf(in);
        scene.add_geometry(Cone(p, n, r, h), psl::string(mat.c_str()));
      } else if (kind == "plane") {
        auto p = rd3(in), n = rd3(in);
        scene.add_geometry(Plane(p, n), psl::string(mat.c_str()));
      } else if (kind == "line") {
        auto a = rd3(in), b = rd3(in);
        float th = rdf(in);
        scene.add_geometry(Line(a, b, th), psl::string(mat.c_str()));
      } else if (kind == "cylinder") {
        auto a = rd3(in), b = rd3(in);
        float r = rdf(in);
        scene.add_geometry(Cylinder(a, b, r), psl::string(mat.c_str()));
      } else if (kind == "triangle") {
        auto a = rd3(in), b = rd3(in), c = rd3(in);
        scene.add_geometry(Triangle(a, b, c), psl::string(mat.c_str()));
      } else if (kind == "mesh" || kind == "mesh_full") {
        int nv, nt;
        in >> nv >> nt;
        psl::vector<vec3> verts;
        psl::vector<vec3u32> idx;
        for (int i = 0; i < nv; i++) verts.push_back(rd3(in));
        for (int i = 0; i < nt; i++) {
          uint32_t a, b, c;
          in >> a >> b >> c;
          idx.push_back(vec3u32(a, b, c));
        }
        psl::vector<vec3> normals;
        psl::vector<vec2> texcoords;
        if (kind == "mesh_full") {  // per-vertex normals / texcoords: Mesh(vertices, indices, texcoords, normals)
          int has_n, has_t;
          in >> has_n >> has_t;
          if (has_n)
            for (int i = 0; i < nv; i++) normals.push_back(rd3(in));
          if (has_t)
            for (int i = 0; i < nv; i++) {
              float tx = rdf(in), ty = rdf(in);
              texcoords.push_back(vec2(tx, ty));
            }
        }
        scene.add_geometry(Mesh(MOVE(verts), MOVE(idx), MOVE(texcoords), MOVE(normals)), psl::string(mat.c_str()));
      } else {
        fprintf(stderr, "unknown shape kind %s\n", kind.c_str());
        exit(2);
      }
    } else if (kw == "light") {
      std::string kind;
      in >> kind;
      if (kind == "point") {
        auto p = rd3(in), c = rd3(in);
        scene.add_light(Light(PointLight(p, c)));
      } else if (kind == "spot") {
        auto p = rd3(in), d = rd3(in), c = rd3(in);
        float falloff = rdf(in), extra = rdf(in);
        scene.add_light(Light(SpotLight(p, d, c, falloff, extra)));
      } else if (kind == "directional") {
        auto d = rd3(in), c = rd3(in);
        scene.add_light(Light(DirectionalLight(d, c)));
      } else {
        fprintf(stderr, "unknown light kind %s\n", kind.c_str());
        exit(2);
      }
    } else if (kw == "envlight") {
      std::string kind;
      in >> kind;
      if (kind != "sky") {
        fprintf(stderr, "unknown environment light %s\n", kind.c_str());
        exit(2);
      }
      scene.set_env_light(EnvironmentLight(Sky(rd3(in))));
    } else if (kw == "camera") {
      std::string kind;
      in >> kind;
      in >> out.W >> out.H;
      auto from = rd3(in), to = rd3(in);
      float fov = rdf(in), lr = rdf(in), fd = rdf(in);
      scene.set_camera(ThinLenCamera(Film(vec2i(out.W, out.H)), from, to, fov, lr, fd));
    } else {
      fprintf(stderr, "unknown keyword %s\n", kw.c_str());
      exit(2);
    }
  }
}

static void write_file(const char* path, const void* p, size_t n) {
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(p, 1, n, f) != n) {
    fprintf(stderr, "cannot write %s\n", path);
    exit(2);
  }
  fclose(f);
}

// ---- reading the reference's private BVH members without touching its sources: explicit instantiation may name
// private members (the usual member-pointer idiom); every object below is the reference's own, built by its own code ----
template <typename Tag, auto Member>
struct Rob {
  friend constexpr auto rob(Tag) { return Member; }
};
struct TagTbvh { friend constexpr auto rob(TagTbvh); };
struct TagLbvh { friend constexpr auto rob(TagLbvh); };
struct TagIndices { friend constexpr auto rob(TagIndices); };
struct TagNodes { friend constexpr auto rob(TagNodes); };
template struct Rob<TagTbvh, &BVH::tbvh>;
template struct Rob<TagLbvh, &BVH::lbvh>;
template struct Rob<TagIndices, &BVH::indices>;
template struct Rob<TagNodes, &BVHImpl::nodes>;

// Canonical pre-order stream of one BVHImpl (independent of how nodes are numbered): an inner node is its two child boxes
// (12 floats: lower, upper of child 0, then of child 1) followed by the two children; a child is either the word
// 0x80000000 | n and n primitive ids (a leaf, in stored = test order), or the word 0x40000000 and an inner node.
// `map` turns the BVH's own primitive index into the id written (top level: geometry index; mesh: triangle index).
template <typename MapFn>
static void bvh_stream(const BVHImpl& impl, MapFn map, std::vector<uint32_t>& out) {
  const auto& nodes = impl.*rob(TagNodes{});
  auto f2u = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  auto leaf = [&](const auto& n) {
    out.push_back(0x80000000u | uint32_t(n.primitiveIndices.size()));
    for (int i : n.primitiveIndices) out.push_back(uint32_t(map(i)));
  };
  if (nodes.size() == 0) {
    out.push_back(0x80000000u);
    return;
  }
  auto rec = [&](auto&& self, int ni) -> void {
    const auto& n = nodes[ni];
    if (n.primitiveIndices.size()) return leaf(n);
    out.push_back(0x40000000u);
    for (int c = 0; c < 2; c++)
      for (int k = 0; k < 3; k++) out.push_back(f2u(n.aabbs[c].lower[k]));
    // (order written: lower0, lower1, then upper0, upper1 -- see below; fixed here once, the reader mirrors it)
    for (int c = 0; c < 2; c++)
      for (int k = 0; k < 3; k++) out.push_back(f2u(n.aabbs[c].upper[k]));
    self(self, n.children[0]);
    self(self, n.children[1]);
  };
  rec(rec, impl.rootIndex);
}

static const int kPixels[][2] = {{0, 0}, {1, 0}, {3, 5}, {127, 127}, {128, 5}, {639, 639}};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::string cmd = argv[1];
  if (cmd == "render" && (argc == 6 || argc == 7)) {
    Loaded L;
    load_pscene(argv[2], L);
    int spp = atoi(argv[3]), depth = atoi(argv[4]);
    const bool sobol = argc == 7 && std::string(argv[6]) == "sobol";  // SobolSampler(spp) instead of BlueSampler(spp)
    // zero the film first: PathIntegrator::render does not clear it (path.cpp:38 plain store)
    L.scene.camera.film().clear();
    const bool halton = argc == 7 && std::string(argv[6]) == "halton";  // HaltonSampler(spp)
    // $PINE_REF_ACCEL=embree (the pine_ref_embree build only): EmbreeAccel, the accel a `.pine` script gets
    // (program_context.cpp:79-81) -- SURVEY.md's oracle variant O-gcc-embree; default: pine's own BVH (O-gcc-bvh)
    const char* accel_env = getenv("PINE_REF_ACCEL");
    const bool embree = accel_env && std::string(accel_env) == "embree";
#ifndef PINE_REF_WITH_EMBREE
    if (embree) {
      fprintf(stderr, "this binary was built without Embree (make -C oracle embree)\n");
      return 2;
    }
#endif
    auto integ = PathIntegrator(
#ifdef PINE_REF_WITH_EMBREE
                                embree ? Accel(EmbreeAccel()) :
#endif
                                Accel(BVH()),
                                sobol ? Sampler(SobolSampler(spp)) : halton ? Sampler(HaltonSampler(spp)) : Sampler(BlueSobolSampler(spp)),
                                UniformLightSampler(), depth);
    auto t0 = std::chrono::steady_clock::now();
    integ.render(L.scene);
    auto t1 = std::chrono::steady_clock::now();
    double sec = std::chrono::duration<double>(t1 - t0).count();
    auto& film = L.scene.camera.film();
    write_file(argv[5], film.data(), size_t(16) * L.W * L.H);
    int eff = (sobol || halton) ? spp : BlueSobolSampler(spp).spp();
    printf("{\"seconds\": %.6f, \"threads\": %u, \"w\": %d, \"h\": %d, \"spp\": %d, \"depth\": %d, "
           "\"msamples_per_s\": %.6f}\n",
           sec, std::thread::hardware_concurrency(), L.W, L.H, eff, depth,
           double(L.W) * L.H * eff / sec * 1e-6);
    return 0;
  }
  if (cmd == "sampler" && argc == 4) {
    // fixture 1: for each pixel in kPixels, sample indices 0..n-1, 130 x get2d (dims 0..259 with
    // the wrap-to-2 at 256) then the same via alternating get1d/get2d to pin the counters.
    int spp = atoi(argv[2]);
    std::vector<float> out;
    for (auto& px : kPixels) {
      auto s = BlueSobolSampler(spp);
      s.start_pixel(vec2i(px[0], px[1]), 0);
      for (int i = 0; i < s.spp(); i++) {
        for (int d = 0; d < 130; d++) {
          auto v = s.get2d();
          out.push_back(v.x);
          out.push_back(v.y);
        }
        s.start_next_sample();
      }
      s.start_pixel(vec2i(px[0], px[1]), 0);
      for (int i = 0; i < s.spp(); i++) {
        for (int d = 0; d < 90; d++) {
          out.push_back(s.get1d());
          auto v = s.get2d();
          out.push_back(v.x);
          out.push_back(v.y);
        }
        s.start_next_sample();
      }
    }
    write_file(argv[3], out.data(), out.size() * 4);
    return 0;
  }
  if (cmd == "rng" && argc == 3) {
    // per pixel: hash (u64), state after seeding (2 x u64), then 16 nextf as raw bits (u32, padded
    // to u64) -- all as u64 words.
    std::vector<uint64_t> out;
    for (auto& px : kPixels) {
      uint64_t h = hash(vec2i(px[0], px[1]), int(0));
      out.push_back(h);
      RNG r(h);
      out.push_back(r.s[0]);
      out.push_back(r.s[1]);
      for (int i = 0; i < 16; i++) {
        float f = r.nextf();
        uint32_t b;
        memcpy(&b, &f, 4);
        out.push_back(b);
      }
    }
    write_file(argv[2], out.data(), out.size() * 8);
    return 0;
  }
  if (cmd == "host" && argc == 3) {
    // host-side math known answers: the two cbox box transforms + inverses, look_at for both
    // cameras, coordinate_system of a few normals.
    std::vector<float> out;
    auto push4 = [&](mat4 m) {
      for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) out.push_back(m[c][r]);  // column-major storage order
    };
    auto push3 = [&](mat3 m) {
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) out.push_back(m[c][r]);
    };
    auto m0 = translate(vec3(0.0f, 0.0f, 0.6f)) * rotate_y(0.4f) * scale(vec3(0.6f, 0.6f, 0.6f));
    auto m1 = translate(vec3(-0.6f, 0.0f, 1.0f)) * rotate_y(-0.4f) * scale(vec3(0.6f, 1.3f, 0.6f));
    push4(m0);
    push4(inverse(m0));
    push4(m1);
    push4(inverse(m1));
    push4(look_at(vec3(0, 0, 0), vec3(0, 0, 1)));
    push4(look_at(vec3(0, 1, -4), vec3(0, 1, 0)));
    push4(look_at(vec3(0, 4, -8), vec3(0, 1, 0)));
    push4(rotate_x(0.3f) * rotate_z(-1.1f));
    vec3 ns[] = {vec3(0, 1, 0), vec3(1, 0, 0), vec3(0, 0, -1), normalize(vec3(1, 2, 3)),
                 normalize(vec3(-3, 2, 0.5f))};
    for (auto n : ns) push3(coordinate_system(n));
    write_file(argv[2], out.data(), out.size() * 4);
    return 0;
  }
  if (cmd == "prl") {
    auto hex = [](float x) {
      char b[64];
      snprintf(b, sizeof b, "%a", double(x));
      return std::string(b);
    };
    for (int i = 2; i < argc; i++) {
      psl::string lit = argv[i];
      const bool is_float = psl::contains(lit, '.');
      if (is_float) {
        const float v = psl::stof(lit);
        printf("literal %s f32 %s str %s\n", argv[i], hex(v).c_str(), psl::to_string(v).c_str());
      } else {
        printf("literal %s i32 %d str %s\n", argv[i], psl::stoi(lit), psl::to_string(psl::stoi(lit)).c_str());
      }
    }
    auto F = [](const char* t) { return psl::stof(psl::string(t)); };
    auto p3 = [&](const char* name, vec3 v) { printf("expr %s vec3 %s %s %s\n", name, hex(v.x).c_str(), hex(v.y).c_str(), hex(v.z).c_str()); };
    auto p16 = [&](const char* name, mat4 m) {
      printf("expr %s mat4", name);
      for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) printf(" %s", hex(m[c][r]).c_str());
      printf("\n");
    };
    p3("600*[1.0,0.64,0.185]", 600 * vec3(F("1.0"), F("0.64"), F("0.185")));
    p3("[1.0,0.64,0.185]*600", vec3(F("1.0"), F("0.64"), F("0.185")) * 600);
    p3("[0.9,0.1,0.05]/3", vec3(F("0.9"), F("0.1"), F("0.05")) / 3);
    p3("2.5/[0.9,0.1,0.05]", F("2.5") / vec3(F("0.9"), F("0.1"), F("0.05")));
    p3("[0.2,0.5,0.9]*[0.9,0.1,0.05]", vec3(F("0.2"), F("0.5"), F("0.9")) * vec3(F("0.9"), F("0.1"), F("0.05")));
    p3("[0.2,0.5,0.9]+[1,2,3]", vec3(F("0.2"), F("0.5"), F("0.9")) + vec3(vec3i(1, 2, 3)));
    {
      vec3i a = vec3i(1, 1, 1) * 160;
      printf("expr [1,1,1]*160 vec3i %d %d %d\n", a.x, a.y, a.z);
      vec2i b = vec2i(256, 256) * 4;
      printf("expr [256,256]*4 vec2i %d %d\n", b.x, b.y);
      vec2i c = 7 / vec2i(2, 3);
      printf("expr 7/[2,3] vec2i %d %d\n", c.x, c.y);
    }
    p16("translate([0.0,0.0,0.6])*rotate_y(0.4)*scale([0.6,0.6,0.6])",
        translate(vec3(F("0.0"), F("0.0"), F("0.6"))) * rotate_y(F("0.4")) * scale(vec3(F("0.6"), F("0.6"), F("0.6"))));
    p16("translate([-0.6,0.0,1.0])*rotate_y(-0.4)*scale([0.6,1.3,0.6])",
        translate(vec3(-F("0.6"), F("0.0"), F("1.0"))) * rotate_y(-F("0.4")) * scale(vec3(F("0.6"), F("1.3"), F("0.6"))));
    p16("rotate_x(0.3)*rotate_z(1.1)*translate(1.5,0.25,3.0)", rotate_x(F("0.3")) * rotate_z(F("1.1")) * translate(F("1.5"), F("0.25"), F("3.0")));
    p16("look_at([0,4,-8],[0,1,0])", look_at(vec3(0, 4, -8), vec3(0, 1, 0)));
    printf("expr 2^10 i32 %d\n", psl::powi(2, 10));
    printf("expr 0.5^2.5 f32 %s\n", hex(psl::pow(F("0.5"), F("2.5"))).c_str());
    printf("expr 7+0.25 f32 %s\n", hex(7 + F("0.25")).c_str());
    printf("expr 0.1+0.2-0.3 f32 %s\n", hex(F("0.1") + (F("0.2") - F("0.3"))).c_str());   // `-` binds before `+` (jit.cpp:1772-1792)
    printf("expr 0.3-0.2+0.1 f32 %s\n", hex((F("0.3") - F("0.2")) + F("0.1")).c_str());
    printf("expr 0.7*0.3/0.9 f32 %s\n", hex(F("0.7") * (F("0.3") / F("0.9"))).c_str());   // `/` binds before `*`
    printf("expr 0.7/0.3*0.9 f32 %s\n", hex((F("0.7") / F("0.3")) * F("0.9")).c_str());
    return 0;
  }
  if (cmd == "bvh" && argc == 6) {
    Loaded L;
    load_pscene(argv[2], L);
    BVH accel;
    accel.build(&L.scene);
    const BVHImpl& tbvh = accel.*rob(TagTbvh{});
    const auto& lbvh = accel.*rob(TagLbvh{});
    const auto& indices = accel.*rob(TagIndices{});
    // tree.bin: number of BVHs, then the top level (ids: geometry indices), then every mesh BVH in lbvh order preceded by
    // the geometry index of its mesh (ids: triangle indices)
    std::vector<uint32_t> tree;
    tree.push_back(uint32_t(1 + lbvh.size()));
    bvh_stream(tbvh, [&](int i) { return indices[i]; }, tree);
    for (size_t m = 0; m < lbvh.size(); m++) {
      tree.push_back(uint32_t(indices[m]));
      bvh_stream(lbvh[m], [](int i) { return i; }, tree);
    }
    write_file(argv[4], tree.data(), tree.size() * 4);
    // trav.bin: per ray, closest hit then any hit: [n, n test words..., hit, geometry, triangle, tmax bits] [n, words..., hit].
    // A test word is the geometry index of a top-level primitive (for a mesh: the entry into its BVH) or
    // 0x40000000 | triangle index inside the mesh entered last.  The callbacks are BVH::intersect's / BVH::hit's own
    // (bvh.cpp:497-548) with the log added.
    std::ifstream rf(argv[3], std::ios::binary);
    rf.seekg(0, std::ios::end);
    size_t nbytes = rf.tellg();
    rf.seekg(0);
    std::vector<float> rays(nbytes / 4);
    rf.read((char*)rays.data(), nbytes);
    std::vector<uint32_t> trav;
    auto& scene = L.scene;
    for (size_t r = 0; r + 8 <= rays.size(); r += 8) {
      const float* q = &rays[r];
      {
        Ray ray(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), q[6], q[7]);
        SurfaceInteraction it;
        std::vector<uint32_t> log;
        uint32_t geom_index = 0, prim_index = 0;
        bool hit = tbvh.Intersect(ray, [&](Ray& ray, int i_lbvh) {
          auto& geometry = scene.geometries[indices[i_lbvh]];
          log.push_back(uint32_t(indices[i_lbvh]));
          if (i_lbvh < int(lbvh.size())) {
            auto& mesh = geometry->shape.as<Mesh>();
            auto h = lbvh[i_lbvh].Intersect(ray, [&](Ray& ray, int index) {
              log.push_back(0x40000000u | uint32_t(index));
              auto hh = mesh.intersect(ray, index);
              if (hh) prim_index = index;
              return hh;
            });
            if (h) geom_index = indices[i_lbvh];
            return h;
          } else {
            auto h = geometry->intersect(ray, it);
            if (h) geom_index = indices[i_lbvh];
            return h;
          }
        });
        trav.push_back(uint32_t(log.size()));
        trav.insert(trav.end(), log.begin(), log.end());
        trav.push_back(hit ? 1u : 0u);
        trav.push_back(hit ? geom_index : 0u);
        trav.push_back(hit ? prim_index : 0u);
        uint32_t tb;
        memcpy(&tb, &ray.tmax, 4);
        trav.push_back(tb);
      }
      {
        Ray ray(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), q[6], q[7]);
        std::vector<uint32_t> log;
        bool hit = tbvh.hit(ray, [&](const Ray& ray, int lbvhIndex) {
          auto& geometry = scene.geometries[indices[lbvhIndex]];
          log.push_back(uint32_t(indices[lbvhIndex]));
          if (lbvhIndex < int(lbvh.size())) {
            return lbvh[lbvhIndex].hit(ray, [&](const Ray& ray, int index) {
              log.push_back(0x40000000u | uint32_t(index));
              return geometry->shape.as<Mesh>().hit(ray, index);
            });
          } else {
            return geometry->hit(ray);
          }
        });
        trav.push_back(uint32_t(log.size()));
        trav.insert(trav.end(), log.begin(), log.end());
        trav.push_back(hit ? 1u : 0u);
      }
    }
    write_file(argv[5], trav.data(), trav.size() * 4);
    printf("{\"bvhs\": %zu, \"tree_words\": %zu, \"rays\": %zu, \"trav_words\": %zu}\n", 1 + lbvh.size(), tree.size(), rays.size() / 8, trav.size());
    return 0;
  }
  if (cmd == "accelq" && argc == 5) {
    Loaded L;
    load_pscene(argv[2], L);
    const char* accel_env = getenv("PINE_REF_ACCEL");
    const bool embree = accel_env && std::string(accel_env) == "embree";
#ifndef PINE_REF_WITH_EMBREE
    if (embree) {
      fprintf(stderr, "this binary was built without Embree (make -C oracle embree)\n");
      return 2;
    }
#endif
    Accel accel =
#ifdef PINE_REF_WITH_EMBREE
        embree ? Accel(EmbreeAccel()) :
#endif
               Accel(BVH());
    accel.build(&L.scene);
    std::ifstream rf(argv[3], std::ios::binary);
    rf.seekg(0, std::ios::end);
    size_t nbytes = rf.tellg();
    rf.seekg(0);
    std::vector<float> rays(nbytes / 4);
    rf.read((char*)rays.data(), nbytes);
    std::vector<uint32_t> out;
    for (size_t r = 0; r + 8 <= rays.size(); r += 8) {
      const float* q = &rays[r];
      Ray ray(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), q[6], q[7]);
      SurfaceInteraction it;
      const bool hit = accel.intersect(ray, it);
      uint32_t geom = 0;
      if (hit)
        for (size_t g = 0; g < L.scene.geometries.size(); g++)
          if (it.shape == &L.scene.geometries[g]->shape) geom = uint32_t(g);
      uint32_t tb;
      memcpy(&tb, &ray.tmax, 4);
      const bool any = accel.hit(Ray(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), q[6], q[7]));
      out.push_back(hit ? 1u : 0u), out.push_back(geom), out.push_back(tb), out.push_back(any ? 1u : 0u);
      const float pn[6] = {it.p.x, it.p.y, it.p.z, it.n.x, it.n.y, it.n.z};  // (zero on a miss)
      for (float f : pn) {
        uint32_t w;
        memcpy(&w, &f, 4);
        out.push_back(hit ? w : 0u);
      }
    }
    write_file(argv[4], out.data(), out.size() * 4);
    printf("{\"rays\": %zu, \"accel\": \"%s\"}\n", rays.size() / 8, embree ? "embree" : "bvh");
    return 0;
  }
  if (cmd == "vertices" && argc == 6) {
    // Per-vertex terms of path.cpp:42-124.  PathIntegrator::radiance keeps them in locals, so the loop is restated here
    // around the reference's own objects (RTIntegrator::intersect / hit, the LightSampler, Material::sample_bxdf, BXDF::f /
    // pdf / sample / sample_p, the Sampler): every number below is computed by the reference's code, only the control flow
    // of radiance() is ours -- and it is checked: the film this loop produces must equal render()'s bit for bit.
    // One record of 16 floats per radiance() invocation, in call order (depth first = path order):
    //   0 kind (0 miss, 1 emissive, 2 path-length limit, 3 shaded)   1 pv.length   2-4 direct term (lo after NEE)
    //   5-7 bs.f   8 cosine   9 bs.pdf   10 bs.is_delta   11 mis applied to the continuation (1 when it returned no light pdf)
    //   12 light_pdf this invocation returns (-1: none)   13-15 Lo this invocation returns
    // (fields that do not exist for a kind are 0; 11 is 0 when the vertex has no continuation)
    Loaded L;
    load_pscene(argv[2], L);
    const int spp_req = atoi(argv[3]), depth = atoi(argv[4]);
    struct Probe : PathIntegrator {
      using PathIntegrator::PathIntegrator;
      int max_len = 0;
      std::vector<float> log;
      struct VX { int length, diffuse_length; float pdf; bool is_delta; };
      struct Res { vec3 Lo; psl::optional<float> light_pdf; };
      void setup(Scene& scene) { RTIntegrator::render(scene); }
      Res trace(Scene& scene, Ray ray, Sampler& sampler, VX pv) {
        const size_t at = log.size();
        log.resize(at + 16, 0.0f);
        auto put = [&](int k, float v) { log[at + k] = v; };
        put(1, float(pv.length));
        put(12, -1.0f);
        auto result = Res();
        auto wi = -ray.d;
        auto& Lo = result.Lo;
        auto finish = [&](int kind) {
          put(0, float(kind));
          put(12, result.light_pdf ? *result.light_pdf : -1.0f);
          put(13, Lo.x), put(14, Lo.y), put(15, Lo.z);
          return result;
        };
        auto it = intersect(ray);
        if (scene.mediums.size()) { fprintf(stderr, "vertices: scenes with media are not supported\n"); exit(2); }
        auto Tr = transmittance(ray.o, ray.d, ray.tmax, sampler);
        if (!it) {
          if (scene.env_light) {
            Lo += Tr * scene.env_light->color(ray.d);
            if (!pv.is_delta) result.light_pdf = scene.env_light->pdf(ray.d);
          }
          return finish(0);
        }
        if (it->material().is<EmissiveMaterial>()) {
          Lo += Tr * it->material().le({*it, wi});
          if (!pv.is_delta) result.light_pdf = light_sampler.pdf(ray, *it);
          return finish(1);
        }
        if (pv.length + 1 >= max_len) return finish(2);
        auto bc = BxdfSampleCtx(*it, wi, 0.6f, pv.diffuse_length > 0);
        auto bxdf = it->material().sample_bxdf(bc, sampler);
        auto beta = vec3(1.0f);
        bxdf.sample_p(beta, bc, sampler);
        auto lo = vec3(0.0f);
        if (!bxdf.is_delta()) {
          if (auto ls = light_sampler.sample(it->p, sampler); ls && !hit(it->spawn_ray(ls->wo, ls->distance))) {
            auto cosine = absdot(ls->wo, it->n);
            auto tr = transmittance(it->p, ls->wo, ls->distance, sampler);
            auto wo = it->to_local(ls->wo);
            if (ls->light->is_delta()) {
              auto f = bxdf.f(wo);
              lo += ls->le * tr * cosine * f / ls->pdf;
            } else {
              auto f = bxdf.f(wo);
              auto mis = balance_heuristic(ls->pdf, bxdf.pdf(wo));
              lo += ls->le * tr * cosine * f / ls->pdf * mis;
            }
          }
        }
        put(2, lo.x), put(3, lo.y), put(4, lo.z);
        if (auto bs = bxdf.sample(bc, sampler)) {
          auto cosine = absdot(bs->wo, it->n);
          auto nv = VX{pv.length + 1, pv.diffuse_length + (bs->is_delta ? 0 : 1), bs->pdf, bs->is_delta};
          auto [Li, light_pdf] = trace(scene, it->spawn_ray(bs->wo), sampler, nv);
          auto mis = light_pdf ? balance_heuristic(bs->pdf, *light_pdf) : 1.0f;
          lo += Li * bs->f * (cosine / bs->pdf * mis);
          put(5, bs->f.x), put(6, bs->f.y), put(7, bs->f.z);
          put(8, cosine), put(9, bs->pdf), put(10, bs->is_delta ? 1.0f : 0.0f), put(11, mis);
        }
        Lo += min(Tr * beta * lo, vec3(8));
        return finish(3);
      }
    };
    L.scene.camera.film().clear();
    Probe probe(Accel(BVH()), Sampler(BlueSobolSampler(spp_req)), UniformLightSampler(), depth);
    probe.max_len = depth;
    probe.setup(L.scene);
    auto& film = L.scene.camera.film();
    const int spp = BlueSobolSampler(spp_req).spp();
    std::vector<float> mine(size_t(L.W) * L.H * 4);
    std::vector<float> out;  // per path: count of records, then the records
    struct Access : Probe { using Probe::samplers; };
    for (int y = 0; y < L.H; y++)
      for (int x = 0; x < L.W; x++) {
        const vec2i p(x, y);
        Sampler& sampler = (probe.*(&Access::samplers))[0].start_pixel(p, 0);
        auto Lsum = vec3(0.0f);
        for (int si = 0; si < spp; si++, sampler.start_next_sample()) {
          auto ray = L.scene.camera.gen_ray((p + sampler.rand2f()) / film.size(), sampler.rand2f());
          probe.log.clear();
          Lsum += probe.trace(L.scene, ray, sampler, Probe::VX{0, 0, 0.0f, true}).Lo;
          out.push_back(float(probe.log.size() / 16));
          out.insert(out.end(), probe.log.begin(), probe.log.end());
        }
        const vec4 px = vec4(Lsum / spp, 1.0f);
        memcpy(&mine[(size_t(y) * L.W + x) * 4], &px, 16);
      }
    // the same film by the reference's own render(): the restated loop must reproduce it bit for bit
    L.scene.camera.film().clear();
    auto integ = PathIntegrator(Accel(BVH()), Sampler(BlueSobolSampler(spp_req)), UniformLightSampler(), depth);
    integ.render(L.scene);
    const bool same = memcmp(mine.data(), L.scene.camera.film().data(), mine.size() * 4) == 0;
    write_file(argv[5], out.data(), out.size() * 4);
    printf("{\"paths\": %d, \"floats\": %zu, \"restated_loop_equals_render\": %s}\n", L.W * L.H * spp, out.size(), same ? "true" : "false");
    return same ? 0 : 3;
  }
  if (cmd == "shapes" && argc == 5) {
    // For every geometry g in the scene and every ray r (8 floats: o, d, tmin, tmax):
    //   hit(r) -> 1 float (0/1); intersect(r) -> hit flag, tmax after; compute_surface_info at
    //   ray(tmax) -> p(3), n(3), uv(2)   => 11 floats per (g, r)
    Loaded L;
    load_pscene(argv[2], L);
    std::ifstream rf(argv[3], std::ios::binary);
    std::vector<float> rays((std::istreambuf_iterator<char>(rf)), {});
    rays.clear();
    rf.clear();
    rf.seekg(0, std::ios::end);
    size_t nbytes = rf.tellg();
    rf.seekg(0);
    rays.resize(nbytes / 4);
    rf.read((char*)rays.data(), nbytes);
    size_t nr = rays.size() / 8;
    std::vector<float> out;
    for (auto& g : L.scene.geometries) {
      if (g->shape.is<Mesh>()) continue;
      for (size_t i = 0; i < nr; i++) {
        const float* q = &rays[i * 8];
        Ray r(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), q[6], q[7]);
        out.push_back(g->hit(r) ? 1.0f : 0.0f);
        SurfaceInteraction it;
        Ray r2 = r;
        bool h = g->intersect(r2, it);
        out.push_back(h ? 1.0f : 0.0f);
        out.push_back(r2.tmax);
        if (h) g->compute_surface_info(r2(), it);
        out.push_back(it.p.x);
        out.push_back(it.p.y);
        out.push_back(it.p.z);
        out.push_back(it.n.x);
        out.push_back(it.n.y);
        out.push_back(it.n.z);
        out.push_back(it.uv.x);
        out.push_back(it.uv.y);
      }
    }
    write_file(argv[4], out.data(), out.size() * 4);
    return 0;
  }
  if (cmd == "gltf" && argc == 6) {
    // gltf <file.gltf|.glb> <spp> <depth> <out.film>: the reference's OWN importer (load_scene -> scene_from_gltf,
    // fileio.cpp:146-330 through tinygltf) builds the scene -- meshes with node transforms applied, Uber / Emissive materials
    // from the pbrMetallicRoughness factors, the camera node -- and PathIntegrator(BVH, BlueSampler) renders it.  Prints the
    // film size (the importer fixes the height at 640) with the timing.
    Scene scene = load_scene(argv[2]);
    int spp = atoi(argv[3]), depth = atoi(argv[4]);
    scene.camera.film().clear();
    const char* accel_env = getenv("PINE_REF_ACCEL");  // embree (pine_ref_embree only): the accel a .pine script's load() + PathIntegrator(sampler, n) gets
    const bool embree = accel_env && std::string(accel_env) == "embree";
#ifndef PINE_REF_WITH_EMBREE
    if (embree) {
      fprintf(stderr, "this binary was built without Embree (make -C oracle embree)\n");
      return 2;
    }
#endif
    auto integ = PathIntegrator(
#ifdef PINE_REF_WITH_EMBREE
        embree ? Accel(EmbreeAccel()) :
#endif
               Accel(BVH()),
        Sampler(BlueSobolSampler(spp)), UniformLightSampler(), depth);
    auto t0 = std::chrono::steady_clock::now();
    integ.render(scene);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto& film = scene.camera.film();
    write_file(argv[5], film.data(), size_t(16) * film.width() * film.height());
    printf("{\"seconds\": %.6f, \"w\": %d, \"h\": %d, \"geometries\": %d, \"spp\": %d, \"depth\": %d}\n", sec, film.width(), film.height(),
           int(scene.geometries.size()), BlueSobolSampler(spp).spp(), depth);
    return 0;
  }
  if (cmd == "finalize" && (argc == 7 || argc == 8)) {
    // finalize <film.bin> <W> <H> <tonemapper 0|1> <out.u8> [out.png]: what scene.camera.film().save(path) does to a
    // rendered film -- Film::finalize (film.cpp:19-25: scale, w = 1, tone mapping), the y flip of save_image(...,
    // true) (fileio.h:33-38,52-55), gamma 1/2.2 and x 256 clamp (fileio.cpp:42-54) -- through the reference's own functions
    const int w = atoi(argv[3]), h = atoi(argv[4]), tm = atoi(argv[5]);
    Film film(vec2i(w, h), tm == 1 ? ToneMapper(ACESToneMapper()) : ToneMapper(Uncharted2ToneMapper()));
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(film.data(), 16, size_t(w) * h, f) != size_t(w) * h) {
      fprintf(stderr, "cannot read %s\n", argv[2]);
      return 2;
    }
    fclose(f);
    if (argc == 8) save_film_as_image(argv[7], film);  // (takes the film by value: finalizes a copy)
    film.finalize();
    auto flipped = invert_y(film.pixels);
    auto u8 = to_uint8_array(flipped.size(), 4, &flipped.data()[0][0], false, true);
    write_file(argv[6], u8.data(), u8.size());
    return 0;
  }
  fprintf(stderr, "usage: see header of oracle/ref_driver.cpp\n");
  return 2;
}
