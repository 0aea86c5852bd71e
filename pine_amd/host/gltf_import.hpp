// pine_amd/host/gltf_import.hpp -- glTF (.glb / .gltf) scene import over the C ABI: what `load(scene, "file.glb")` does in the
// reference (src/pine/core/fileio.cpp:146-330, which goes through tinygltf).  Header-only, host code; used by the PRL
// front-end's `load` builtin and by the C++ facade (pine::load).  The Python twin is pine_amd/gltf.py; both are pinned by
// tests/golden/import_test.glb, which the real reference imported and rendered (tests/test_gltf.py, tests/test_prl.py).
//
// Every mesh primitive of every node of every scene becomes Mesh(vertices, indices, texcoords, normals) with the node's
// accumulated transform applied (transform * transpose(mat4(matrix)) * translate(T) * q2m(R) * scale(S), Mesh::apply), its
// material Uber(baseColor, roughness, metallic, transmission, ior) from the pbrMetallicRoughness factors and the KHR
// transmission / ior extensions, or Emissive(emissiveFactor * emissiveStrength) when that product is not zero; a node with a
// camera sets ThinLenCamera(Film([640 * aspect, 640]), pos, pos + R * (0, 0, -1), yfov / 2).  All matrix arithmetic is the
// library's host math (binary32, the reference's operand order).  Image textures (NodeImage) are refused by name.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pine_gpu.h"

namespace pine_gltf {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- a small JSON reader (objects, arrays, strings, numbers, true / false / null) ----
struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* find(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  const Json& at(const std::string& k) const {
    if (auto* p = find(k)) return *p;
    throw Error("glTF: missing key `" + k + "`");
  }
  const Json& at(size_t i) const {
    if (kind != Array || i >= arr.size()) throw Error("glTF: array index out of range");
    return arr[i];
  }
  double number(double dflt) const { return kind == Number ? num : dflt; }
  int integer(int dflt) const {
    if (kind != Number) return dflt;
    if (!(num >= -2147483648.0 && num <= 2147483647.0)) throw Error("glTF: integer out of range");
    return int(num);
  }
  size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }
};
class JsonParser {
 public:
  JsonParser(const char* p, size_t n) : p_(p), e_(p + n) {}
  Json parse() {
    Json v = value(0);
    ws();
    if (p_ != e_) fail("trailing characters");
    return v;
  }

 private:
  const char *p_, *e_;
  [[noreturn]] void fail(const char* what) { throw Error(std::string("glTF: malformed JSON (") + what + ")"); }
  void ws() {
    while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) p_++;
  }
  bool lit(const char* s) {
    const size_t n = strlen(s);
    if (size_t(e_ - p_) >= n && memcmp(p_, s, n) == 0) {
      p_ += n;
      return true;
    }
    return false;
  }
  Json value(int depth) {
    if (depth > 64) fail("nesting too deep");
    ws();
    if (p_ >= e_) fail("unexpected end");
    Json v;
    if (*p_ == '{') {
      p_++;
      v.kind = Json::Object;
      ws();
      if (p_ < e_ && *p_ == '}') {
        p_++;
        return v;
      }
      while (true) {
        ws();
        if (p_ >= e_ || *p_ != '"') fail("expected a key");
        std::string k = string();
        ws();
        if (p_ >= e_ || *p_ != ':') fail("expected ':'");
        p_++;
        v.obj.emplace_back(std::move(k), value(depth + 1));
        ws();
        if (p_ < e_ && *p_ == ',') {
          p_++;
          continue;
        }
        if (p_ < e_ && *p_ == '}') {
          p_++;
          return v;
        }
        fail("expected ',' or '}'");
      }
    }
    if (*p_ == '[') {
      p_++;
      v.kind = Json::Array;
      ws();
      if (p_ < e_ && *p_ == ']') {
        p_++;
        return v;
      }
      while (true) {
        v.arr.push_back(value(depth + 1));
        ws();
        if (p_ < e_ && *p_ == ',') {
          p_++;
          continue;
        }
        if (p_ < e_ && *p_ == ']') {
          p_++;
          return v;
        }
        fail("expected ',' or ']'");
      }
    }
    if (*p_ == '"') {
      v.kind = Json::String;
      v.str = string();
      return v;
    }
    if (lit("true")) {
      v.kind = Json::Bool;
      v.b = true;
      return v;
    }
    if (lit("false")) {
      v.kind = Json::Bool;
      return v;
    }
    if (lit("null")) return v;
    // a number
    const char* s = p_;
    while (p_ < e_ && (*p_ == '-' || *p_ == '+' || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || (*p_ >= '0' && *p_ <= '9'))) p_++;
    if (p_ == s) fail("unexpected character");
    const std::string t(s, p_);
    char* end = nullptr;
    v.kind = Json::Number;
    v.num = strtod(t.c_str(), &end);
    if (!end || *end) fail("bad number");
    return v;
  }
  std::string string() {
    std::string out;
    p_++;  // opening quote
    while (true) {
      if (p_ >= e_) fail("unterminated string");
      const char c = *p_++;
      if (c == '"') return out;
      if (c != '\\') {
        out.push_back(c);
        continue;
      }
      if (p_ >= e_) fail("unterminated escape");
      const char x = *p_++;
      switch (x) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          if (e_ - p_ < 4) fail("short \\u escape");
          unsigned cp = 0;
          for (int i = 0; i < 4; i++) {
            const char h = *p_++;
            cp = cp * 16 + unsigned(h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : 0);
          }
          if (cp < 0x80) out.push_back(char(cp));
          else if (cp < 0x800) out.push_back(char(0xc0 | (cp >> 6))), out.push_back(char(0x80 | (cp & 0x3f)));
          else out.push_back(char(0xe0 | (cp >> 12))), out.push_back(char(0x80 | ((cp >> 6) & 0x3f))), out.push_back(char(0x80 | (cp & 0x3f)));
          break;
        }
        default: out.push_back(x);  // \" \\ \/
      }
    }
  }
};

inline std::vector<uint8_t> read_file(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw Error("Unable to open file `" + path + "`");
  std::vector<uint8_t> d;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
  fclose(f);
  return d;
}
inline std::vector<uint8_t> base64(const std::string& s) {
  std::vector<uint8_t> out;
  unsigned acc = 0;
  int bits = 0;
  for (char c : s) {
    int v = c >= 'A' && c <= 'Z' ? c - 'A' : c >= 'a' && c <= 'z' ? c - 'a' + 26 : c >= '0' && c <= '9' ? c - '0' + 52 : c == '+' ? 62 : c == '/' ? 63 : -1;
    if (v < 0) continue;
    acc = (acc << 6) | unsigned(v);
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out.push_back(uint8_t(acc >> bits));
    }
  }
  return out;
}

struct Mat4 {
  float a[16];
};
inline Mat4 identity() {
  Mat4 m;
  pine_gpu_mat4_identity(m.a);
  return m;
}
inline Mat4 mul(const Mat4& l, const Mat4& r) {
  Mat4 o;
  pine_gpu_mat4_mul(l.a, r.a, o.a);
  return o;
}

// Adds everything the file holds to `scene`; returns true and fills camera_* when a camera node was found (the caller sets
// it: the PRL front-end keeps its own camera / film objects).
struct ImportedCamera {
  bool present = false;
  int film_w = 0, film_h = 0;
  float from[3] = {0, 0, 0}, to[3] = {0, 0, 0}, fov = 0;
};
inline ImportedCamera import_scene(pine_gpu_scene* scene, const std::string& path, const float* global_transform /* 16 floats or null */) {
  auto check = [](int rc, const char* what) {
    if (rc < 0) throw Error(std::string(what) + ": " + pine_gpu_last_error());
    return rc;
  };
  const std::vector<uint8_t> file = read_file(path);
  Json doc;
  std::vector<uint8_t> glb_blob;
  if (file.size() >= 12 && memcmp(file.data(), "glTF", 4) == 0) {
    uint32_t total;
    memcpy(&total, file.data() + 8, 4);
    if (total > file.size()) throw Error("Unable to create scene from GLTF file (truncated)");
    size_t pos = 12;
    bool have = false;
    while (pos + 8 <= total) {
      uint32_t n, kind;
      memcpy(&n, file.data() + pos, 4);
      memcpy(&kind, file.data() + pos + 4, 4);
      if (pos + 8 + n > total) throw Error("Unable to create scene from GLTF file (bad chunk)");
      if (kind == 0x4E4F534Au) {
        doc = JsonParser(reinterpret_cast<const char*>(file.data() + pos + 8), n).parse();
        have = true;
      } else if (kind == 0x004E4942u) {
        glb_blob.assign(file.begin() + long(pos + 8), file.begin() + long(pos + 8 + n));
      }
      pos += 8 + size_t(n);
    }
    if (!have) throw Error("Unable to create scene from GLTF file");
  } else {
    doc = JsonParser(reinterpret_cast<const char*>(file.data()), file.size()).parse();
  }
  {  // (tinygltf's REQUIRE_VERSION, its default)
    const Json* asset = doc.find("asset");
    if (!asset || !asset->find("version")) throw Error("Unable to create scene from GLTF file (no asset.version)");
  }
  const std::string base = path.find_last_of('/') == std::string::npos ? std::string(".") : path.substr(0, path.find_last_of('/'));
  std::vector<std::vector<uint8_t>> buffers;
  if (auto* bs = doc.find("buffers"))
    for (auto& b : bs->arr) {
      const Json* uri = b.find("uri");
      if (!uri) buffers.push_back(glb_blob);
      else if (uri->str.compare(0, 5, "data:") == 0) buffers.push_back(base64(uri->str.substr(uri->str.find(',') + 1)));
      else buffers.push_back(read_file(base + "/" + uri->str));
    }

  struct View {
    const uint8_t* p;
    size_t count, ncomp, stride, comp_size;
    int ctype;
  };
  auto accessor = [&](int index) {
    const Json& acc = doc.at("accessors").at(size_t(index));
    const Json& view = doc.at("bufferViews").at(size_t(acc.at("bufferView").integer(0)));
    const int ctype = acc.at("componentType").integer(0);
    const size_t cs = ctype == 5120 || ctype == 5121 ? 1 : ctype == 5122 || ctype == 5123 ? 2 : ctype == 5125 || ctype == 5126 ? 4 : ctype == 5130 ? 8 : 0;
    if (!cs) throw Error("glTF: unsupported accessor component type");
    const std::string& type = acc.at("type").str;
    const size_t nc = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
    if (!nc) throw Error("glTF: unsupported accessor type");
    // (offsets, strides and counts are JSON numbers: range-checked as such BEFORE they become sizes -- a negative or
    // huge one must not wrap the bounds check below)
    auto as_size = [](const Json* j) -> size_t {
      if (!j) return 0;
      const double d = j->number(0);
      if (!(d >= 0.0 && d <= 4294967295.0)) throw Error("glTF: offset / stride / count out of range");
      return size_t(d);
    };
    const size_t view_offset = as_size(view.find("byteOffset")), acc_offset = as_size(acc.find("byteOffset"));
    const size_t start = view_offset + acc_offset;
    size_t stride = as_size(view.find("byteStride"));
    if (!stride) stride = cs * nc;
    const int buffer_index = view.at("buffer").integer(-1);
    if (buffer_index < 0 || size_t(buffer_index) >= buffers.size()) throw Error("glTF: bufferView names a buffer that does not exist");
    const std::vector<uint8_t>& raw = buffers[size_t(buffer_index)];
    const size_t count = as_size(&acc.at("count"));
    if (start > raw.size() || (count && (count - 1) * stride + cs * nc > raw.size() - start)) throw Error("glTF: accessor runs past its buffer");
    return View{raw.data() + start, count, nc, stride, cs, ctype};
  };
  auto read_floats = [&](const View& v, size_t want_comp) {
    if (v.ncomp != want_comp || (v.ctype != 5126 && v.ctype != 5130)) throw Error("glTF: expect float vectors for POSITION / NORMAL / TEXCOORD_0");
    std::vector<float> out(v.count * v.ncomp);
    for (size_t i = 0; i < v.count; i++)
      for (size_t c = 0; c < v.ncomp; c++) {
        if (v.ctype == 5126) memcpy(&out[i * v.ncomp + c], v.p + i * v.stride + 4 * c, 4);
        else {
          double d;
          memcpy(&d, v.p + i * v.stride + 8 * c, 8);
          out[i * v.ncomp + c] = float(d);
        }
      }
    return out;
  };

  auto material_of = [&](const Json& prim) {
    float basecolor[3] = {1, 1, 1}, roughness = 1.0f, metallic = 0.0f, transmission = 0.0f, ior = 1.45f;
    float emission_color[3] = {1, 1, 1}, emission_strength = 0.0f;
    const Json* mi = prim.find("material");
    if (mi && mi->integer(-1) >= 0) {
      const Json& mat = doc.at("materials").at(size_t(mi->integer(0)));
      if (const Json* ext = mat.find("extensions")) {
        if (const Json* e = ext->find("KHR_materials_transmission")) transmission = float(e->find("transmissionFactor") ? e->at("transmissionFactor").number(0) : 0.0);
        if (const Json* e = ext->find("KHR_materials_ior")) ior = float(e->find("ior") ? e->at("ior").number(1.5) : 1.5);
        if (const Json* e = ext->find("KHR_materials_emissive_strength")) emission_strength = float(e->find("emissiveStrength") ? e->at("emissiveStrength").number(1) : 1.0);
      }
      const Json* pbr = mat.find("pbrMetallicRoughness");
      if (pbr && (pbr->find("baseColorTexture") || pbr->find("metallicRoughnessTexture")))
        throw Error("glTF material `" + (mat.find("name") ? mat.at("name").str : std::string("?")) + "` uses image textures (NodeImage): not supported");
      for (int k = 0; k < 3; k++) basecolor[k] = 1.0f;
      metallic = 1.0f, roughness = 1.0f;  // tinygltf's defaults once a material is present
      if (pbr) {
        if (const Json* c = pbr->find("baseColorFactor"))
          for (int k = 0; k < 3; k++) basecolor[k] = float(c->at(size_t(k)).number(1));
        if (const Json* v = pbr->find("metallicFactor")) metallic = float(v->number(1));
        if (const Json* v = pbr->find("roughnessFactor")) roughness = float(v->number(1));
      }
      for (int k = 0; k < 3; k++) emission_color[k] = 0.0f;
      if (const Json* e = mat.find("emissiveFactor"))
        for (int k = 0; k < 3; k++) emission_color[k] = float(e->at(size_t(k)).number(0));
    }
    const float emission[3] = {emission_color[0] * emission_strength, emission_color[1] * emission_strength, emission_color[2] * emission_strength};
    const std::string name;  // (anonymous: the library numbers it, as for scene.add(shape, material))
    if (emission[0] == 0.0f && emission[1] == 0.0f && emission[2] == 0.0f)
      return check(pine_gpu_scene_add_material_uber(scene, name.c_str(), basecolor, roughness, metallic, transmission, ior), "Uber");
    return check(pine_gpu_scene_add_material_emissive(scene, name.c_str(), emission), "Emissive");
  };

  struct Walker {
    const Json& doc;
    pine_gpu_scene* scene;
    decltype(accessor)& acc;
    decltype(read_floats)& floats;
    decltype(material_of)& material;
    decltype(check)& ok;
    void node(int index, Mat4 xf, int depth) {
      if (depth > 256) throw Error("glTF: node hierarchy too deep (a cycle?)");
      const Json& n = doc.at("nodes").at(size_t(index));
      if (const Json* m = n.find("matrix"); m && m->size() == 16) {
        float rows[16];
        for (int i = 0; i < 16; i++) rows[i] = float(m->at(size_t(i)).number(0));
        Mat4 r, t;
        pine_gpu_mat4_from_rows(rows, r.a);  // mat4(m[0], ..., m[15]): row-major arguments ...
        pine_gpu_mat4_transpose(r.a, t.a);   // ... transposed (fileio.cpp:162-164)
        xf = mul(xf, t);
      }
      if (const Json* T = n.find("translation"); T && T->size() == 3) {
        const float v[3] = {float(T->at(0).number(0)), float(T->at(1).number(0)), float(T->at(2).number(0))};
        Mat4 t;
        pine_gpu_mat4_translate(v, t.a);
        xf = mul(xf, t);
      }
      if (const Json* R = n.find("rotation"); R && R->size() == 4) {
        Mat4 q;
        pine_gpu_mat4_from_quaternion(float(R->at(3).number(1)), float(R->at(0).number(0)), float(R->at(1).number(0)), float(R->at(2).number(0)), q.a);
        xf = mul(xf, q);
      }
      if (const Json* Sc = n.find("scale"); Sc && Sc->size() == 3) {
        const float v[3] = {float(Sc->at(0).number(1)), float(Sc->at(1).number(1)), float(Sc->at(2).number(1))};
        Mat4 s;
        pine_gpu_mat4_scale(v, s.a);
        xf = mul(xf, s);
      }
      if (const Json* mi = n.find("mesh"); mi && mi->integer(-1) >= 0)
        for (const Json& prim : doc.at("meshes").at(size_t(mi->integer(0))).at("primitives").arr) {
          if (const Json* mode = prim.find("mode"); mode && mode->integer(4) != 4) throw Error("only TRIANGLES primitives are supported (as in the reference)");
          const auto iv = acc(prim.at("indices").integer(0));
          if (iv.comp_size != 2 && iv.comp_size != 4) throw Error("index byte size must be 2 or 4 (fileio.cpp:181)");
          std::vector<uint32_t> faces((iv.count / 3) * 3);
          for (size_t i = 0; i < faces.size(); i++) {
            if (iv.comp_size == 2) {
              uint16_t v;
              memcpy(&v, iv.p + i * iv.stride, 2);
              faces[i] = v;
            } else {
              memcpy(&faces[i], iv.p + i * iv.stride, 4);
            }
          }
          std::vector<float> verts, normals, uvs;
          for (auto& kv : prim.at("attributes").obj) {
            if (kv.first == "POSITION") verts = floats(acc(kv.second.integer(0)), 3);
            else if (kv.first == "NORMAL") normals = floats(acc(kv.second.integer(0)), 3);
            else if (kv.first == "TEXCOORD_0") uvs = floats(acc(kv.second.integer(0)), 2);
          }
          if (verts.empty()) throw Error("glTF: a primitive without POSITION");
          if ((!normals.empty() && normals.size() != verts.size()) || (!uvs.empty() && uvs.size() / 2 != verts.size() / 3))
            throw Error("glTF: attribute counts differ from the vertex count");
          const int mat = material(prim);
          ok(pine_gpu_mesh_apply(verts.data(), int(verts.size() / 3), normals.empty() ? nullptr : normals.data(), xf.a), "Mesh.apply");
          ok(pine_gpu_scene_add_mesh_full(scene, verts.data(), int(verts.size() / 3), faces.data(), int(faces.size() / 3), normals.empty() ? nullptr : normals.data(),
                                          uvs.empty() ? nullptr : uvs.data(), mat),
             "Mesh");
        }
      if (const Json* ch = n.find("children"))
        for (const Json& c : ch->arr) node(c.integer(0), xf, depth + 1);
    }
  } walker{doc, scene, accessor, read_floats, material_of, check};

  Mat4 root = identity();
  if (global_transform) memcpy(root.a, global_transform, 64);
  if (const Json* scenes = doc.find("scenes"))
    for (const Json& sc : scenes->arr)
      if (const Json* ns = sc.find("nodes"))
        for (const Json& ni : ns->arr) walker.node(ni.integer(0), root, 0);

  ImportedCamera cam;
  if (const Json* nodes = doc.find("nodes"))
    for (const Json& n : nodes->arr) {
      const Json* ci = n.find("camera");
      if (!ci || ci->integer(-1) < 0) continue;
      const Json& persp = doc.at("cameras").at(size_t(ci->integer(0))).at("perspective");
      const Json &P = n.at("translation"), &R = n.at("rotation");  // (the reference CHECKs both are present)
      if (P.size() != 3 || R.size() != 4) throw Error("glTF: a camera node needs translation and rotation");
      Mat4 rot;
      pine_gpu_mat4_from_quaternion(float(R.at(3).number(1)), float(R.at(0).number(0)), float(R.at(1).number(0)), float(R.at(2).number(0)), rot.a);
      for (int k = 0; k < 3; k++) cam.from[k] = float(P.at(size_t(k)).number(0));
      // at = pos + mat3(rot) * (0, 0, -1) = pos + (x * 0 + y * 0 + z * (-1))   (operator*(mat3, vec3) vecmath.h:695)
      for (int k = 0; k < 3; k++) cam.to[k] = cam.from[k] + ((rot.a[0 * 4 + k] * 0.0f + rot.a[1 * 4 + k] * 0.0f) + rot.a[2 * 4 + k] * -1.0f);
      cam.film_w = int(640 * (persp.find("aspectRatio") ? persp.at("aspectRatio").number(0) : 0.0));
      cam.film_h = 640;
      cam.fov = float(persp.at("yfov").number(0) / 2);
      cam.present = true;
    }
  return cam;
}

}  // namespace pine_gltf
