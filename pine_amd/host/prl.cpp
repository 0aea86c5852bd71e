// pine_amd/host/prl.cpp -- minimal PRL front-end (libpine_prl.so, C ABI in include/pine_prl.h).
//
// A tree-walking interpreter for the subset of pine's scripting language that cbox-class scene
// scripts use.  It follows the reference where a script can observe it:
//   * grammar: block / statement / expression structure of the hand-written parser
//     (src/pine/core/jit.cpp:1467-2217), including its operator-precedence table whose entries are
//     OCTAL literals except the multiplicative ones (jit.cpp:1772-1792): reduction picks the highest
//     code first, first occurrence on ties, so `a + b - c` groups as a + (b - c) and `a * b / c`
//     as a * (b / c);
//   * literals: integers and floats are converted with psl::stoi / psl::stof
//     (src/psl/string.cpp:158-201), whose float accumulation (digit * 0.1f^k summed in binary32)
//     differs from strtof in the last bit for many decimal strings ("0.64", "0.9", ...): a scene
//     built from a .pine script therefore differs in those constants from one built through the C++
//     API with C++ literals -- and this front-end reproduces the script's values;
//   * typing: `[a, b, c]` is vecN if any element is f32, else vecNi (jit.cpp:1014-1023); every
//     operator and call is resolved by name over a function table with the reference's rule: same
//     arity, exact type match or ONE registered implicit conversion per argument, fewest conversions
//     wins, ties are ambiguous (src/pine/core/context.cpp:143-215);
//   * the table itself: names, overloads and conversions of setup_program_context()
//     (src/pine/core/program_context.cpp:23-125) that the PathIntegrator path needs -- cited at each
//     registration below.
// Not supported (reported as errors): fn / class / lambdas, node graphs beyond constants, the other
// integrators, media, lights other than emissive geometry, mesh import.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pine_gpu.h"
#include "../../include/pine_prl.h"
#include "gltf_import.hpp"
#include "png_writer.hpp"

namespace prl {

struct Error {
  std::string msg;
};
[[noreturn]] static void fail(const std::string& m) { throw Error{m}; }

// ------------------------------------------------------------------------------------------------
// psl number conversions (src/psl/string.cpp:158-201, :118-136) -- restated, binary32 arithmetic
// ------------------------------------------------------------------------------------------------
static int psl_stoi(const std::string& str) {
  // (the reference accumulates in a signed int; a literal beyond 2^31 overflows it -- undefined in C++, two's-complement
  //  wrap-around in the reference's x86-64 build, which is what the unsigned arithmetic here defines and reproduces)
  unsigned number = 0;
  bool neg = false;
  for (size_t j = 0; j < str.size(); j++) {
    if (str[j] == '.') break;
    if (j == 0 && str[j] == '-') neg = true;
    else number = number * 10u + unsigned(int(str[j])) - unsigned('0');
  }
  return int(neg ? 0u - number : number);
}
static float psl_stof(const std::string& str) {
  float number = 0.0f;
  bool neg = false, pass = false;
  float scale = 0.1f;
  for (size_t j = 0; j < str.size(); j++) {
    if (j == 0 && str[j] == '-') neg = true;
    else if (!pass && str[j] == '.') pass = true;
    else if (!pass) number = number * 10 + float(str[j] - '0');
    else {
      number += float(str[j] - '0') * scale;
      scale *= 0.1f;
    }
  }
  return neg ? -number : number;
}
static std::string psl_to_string(int64_t x) { return std::to_string(x); }
static std::string psl_to_string(float x) {  // integer part, '.', four truncated digits
  if (std::isnan(x)) return "nan";
  if (x > 3.40282346638528859812e+38f) return "inf";
  if (x < -3.40282346638528859812e+38f) return "-inf";
  std::string s = x < 0 ? "-" : "";
  x = std::fabs(x);
  s += psl_to_string(int64_t(x)) + ".";
  x = x - std::floor(x);
  for (int i = 0; i < 4; i++) {
    x *= 10;
    s.push_back(char('0' + int(x)));
    x = std::fabs(x - std::floor(x));
  }
  return s;
}

// ------------------------------------------------------------------------------------------------
// Values
// ------------------------------------------------------------------------------------------------
struct Object {
  virtual ~Object() {}
};
struct Value {
  std::string type = "void";
  bool b = false;
  int32_t i[4] = {0, 0, 0, 0};
  float f[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  std::string s;
  std::shared_ptr<Object> o;
};
using Cell = std::shared_ptr<Value>;
static Cell cell(Value v) { return std::make_shared<Value>(std::move(v)); }

struct FilmObj : Object {
  int w = 0, h = 0, tone = 0;
  std::vector<float> pixels;
};
struct CameraObj : Object {
  std::shared_ptr<FilmObj> film;
  float from[3], to[3], fov, len_radius = 0.0f, focus = 1.0f;
};
// A shading-node expression (Nodef / Node3f, node.h:13-297).  Every float-valued node expression is
// typed "Nodef" here and every vec3-valued one "Node3f": the reference gives each node struct its own
// type name ("Addf", "Checkerboard", ...) that converts in one step to Nodef / Node3f and to nothing
// else, so overload resolution picks the same function either way.
struct NodeObj : Object {
  std::string kind;  // constf const3 position normal uv bin un comp tovec3 checker splat
  char op = 0;
  int n = 0;
  float f = 0, v[3] = {0, 0, 0};
  bool is3 = false;
  std::shared_ptr<NodeObj> a, b, c;
  bool constant() const { return kind == "constf" || kind == "const3"; }
};
using NodeP3 = std::shared_ptr<NodeObj>;
struct MaterialObj : Object {
  std::string kind;  // Emissive | Diffuse | Uber | Subsurface | Metal | Glossy | Glass
  NodeP3 albedo, roughness, metallic, transmission, ior_node;
  float ior = 1.45f;
  float sigma_s[3] = {0, 0, 0};
};
struct ShapeObj : Object {
  std::string kind;  // Rect | AABB | OBB | Sphere | Disk | Cone
  std::vector<float> p;
  bool flag = false;
};
struct SceneObj : Object {
  pine_gpu_scene* h = nullptr;
  Cell camera;  // type "Camera"
  SceneObj() : h(pine_gpu_scene_create()) {}
  ~SceneObj() override { pine_gpu_scene_destroy(h); }
};
struct LightObj : Object {
  std::string kind;  // PointLight | SpotLight | DirectionalLight | Sky
  float a[3] = {0, 0, 0}, b[3] = {0, 0, 0}, c[3] = {0, 0, 0};
  float falloff = 0, extra = 0;
};
struct IntegratorObj : Object {
  int spp = 0, depth = 0, sampler = 0;
  int accel = 0;  // 0: the two-argument constructor's default; 1: Accel(BVH()) -- pine-BVH order; 2: Accel(Embree()) -- EmbreeAccel's order
};

static Value mk_i32(int v) {
  Value r;
  r.type = "i32";
  r.i[0] = v;
  return r;
}
static Value mk_f32(float v) {
  Value r;
  r.type = "f32";
  r.f[0] = v;
  return r;
}
static Value mk_bool(bool v) {
  Value r;
  r.type = "bool";
  r.b = v;
  return r;
}
static Value mk_str(std::string v) {
  Value r;
  r.type = "str";
  r.s = std::move(v);
  return r;
}
static Value mk_vecf(int n, const float* v) {
  Value r;
  r.type = "vec" + std::to_string(n);
  for (int k = 0; k < n; k++) r.f[k] = v[k];
  return r;
}
static Value mk_veci(int n, const int* v) {
  Value r;
  r.type = "vec" + std::to_string(n) + "i";
  for (int k = 0; k < n; k++) r.i[k] = v[k];
  return r;
}
static Value mk_obj(const std::string& type, std::shared_ptr<Object> o) {
  Value r;
  r.type = type;
  r.o = std::move(o);
  return r;
}
static Value retype(Value v, const std::string& type) {
  v.type = type;
  return v;
}

// ------------------------------------------------------------------------------------------------
// Function table + overload resolution (context.cpp:143-215)
// ------------------------------------------------------------------------------------------------
struct Interp;
using Impl = std::function<Value(Interp&, std::vector<Cell>&)>;
struct Fn {
  std::vector<std::string> ptypes;  // "T" (by value / const ref: convertible) or "T&" (exact)
  std::string rtype;
  Impl impl;
};
struct Registry {
  std::multimap<std::string, Fn> fns;
  std::map<std::string, bool> types;
  std::map<std::string, Value> constants;
  void def(const std::string& name, std::vector<std::string> ptypes, const std::string& rtype, Impl impl) {
    fns.insert({name, Fn{std::move(ptypes), rtype, std::move(impl)}});
  }
  // T.ctor_variant<A...>: implicit conversion + explicit constructor (context.h:286-301)
  void convert(const std::string& from, const std::string& to, std::function<Value(const Value&)> f) {
    types[to] = true;
    def("@convert." + from + "." + to, {from}, to, [f](Interp&, std::vector<Cell>& a) { return f(*a[0]); });
    def(to, {from}, to, [f](Interp&, std::vector<Cell>& a) { return f(*a[0]); });
  }
  const Fn* unique(const std::string& name) const {
    auto r = fns.equal_range(name);
    if (r.first == r.second) return nullptr;
    auto n = r.first;
    if (++n != r.second) return nullptr;
    return &r.first->second;
  }
};

struct Resolved {
  const Fn* fn = nullptr;
  std::vector<std::pair<size_t, const Fn*>> converts;
};

// ------------------------------------------------------------------------------------------------
// AST
// ------------------------------------------------------------------------------------------------
struct Node;
using NodeP = std::shared_ptr<Node>;
struct Node {
  enum Kind { Num, Bool, Str, Vec, Id, Call, Member, Unary, Binary, Decl, Block, While, For, If, ExprStmt, Break, Continue, Empty, FnDef, Return } kind;
  int line = 0, col = 0;
  std::string text;          // literal text / identifier / function name / operator
  std::vector<NodeP> kids;   // operands / arguments / statements
  int flag = 0;              // Decl: 0 `:=`, 1 `&=`, 2 `=`;  Bool: value
  NodeP a, b, c, d;          // For: init, cond, inc, body; If: cond(a) body(b) else(c); While: cond(a) body(b)
  std::vector<std::pair<std::string, std::string>> params;  // FnDef: (name, type) pairs; text = name, rtype = return type, b = body
  std::string rtype;
};
static NodeP mk(Node::Kind k, int line, int col) {
  auto n = std::make_shared<Node>();
  n->kind = k;
  n->line = line;
  n->col = col;
  return n;
}

// ------------------------------------------------------------------------------------------------
// Parser (structure of jit.cpp:1467-2217)
// ------------------------------------------------------------------------------------------------
struct Parser {
  std::string src;
  size_t pos = 0;
  explicit Parser(std::string s) : src(std::move(s)) { skip(); }

  void loc(size_t p, int& line, int& col) const {
    line = 1;
    col = 1;
    for (size_t k = 0; k < p && k < src.size(); k++) {
      if (src[k] == '\n') line++, col = 1;
      else col++;
    }
  }
  [[noreturn]] void error(const std::string& m) const {
    int l, c;
    loc(pos, l, c);
    fail(std::to_string(l) + ":" + std::to_string(c) + ": " + m);
  }
  bool eof() const { return pos >= src.size(); }
  void skip() {  // spaces and `#` comments
    while (pos < src.size()) {
      if (src[pos] == '#') {
        while (pos < src.size() && src[pos] != '\n') pos++;
      } else if (isspace((unsigned char)src[pos])) pos++;
      else break;
    }
  }
  bool expect(const std::string& s) const { return src.compare(pos, s.size(), s) == 0; }
  bool accept(const std::string& s, bool trim = true) {
    if (!expect(s)) return false;
    pos += s.size();
    if (trim) skip();
    return true;
  }
  void consume(const std::string& s, const std::string& what = "") {
    if (!expect(s)) error("Expect `" + s + "` " + what);
    pos += s.size();
    skip();
  }
  static bool idstart(char c) { return isalpha((unsigned char)c) || c == '_'; }
  NodeP here(Node::Kind k) const {
    int l, c;
    loc(pos, l, c);
    return mk(k, l, c);
  }

  NodeP block(bool top) {
    skip();
    if (top) accept("{");
    else consume("{", "to begin block");
    auto n = here(Node::Block);
    while (!expect("}") && !eof()) n->kids.push_back(block_elem());
    if (top) accept("}");
    else consume("}", "to end block");
    return n;
  }
  NodeP block_elem() {
    if (expect("{")) return block(false);
    if (expect("while")) return while_();
    if (expect("for")) return for_();
    if (expect("if")) return if_chain();
    if (expect("fn")) return fn_def();
    if (expect("class")) error("`class` definitions are not supported by this front-end");
    return stmt();
  }
  // fn name(a: T, b: U): R { ... }   (jit.cpp:1695-1706, 1935-1969; function-typed parameters are lambdas'
  // business and stay unsupported)
  std::string type_name() {
    skip();
    if (expect("(")) error("function types (lambdas) are not supported by this front-end");
    std::string t = id();
    if (accept("&")) t += "&";
    return t;
  }
  NodeP fn_def() {
    auto n = here(Node::FnDef);
    consume("fn", "to start function definition");
    n->text = id();
    consume("(", "to begin parameter definition");
    if (!expect(")"))
      while (true) {
        std::string name = id();
        consume(":", "to specify its type");
        n->params.push_back({name, type_name()});
        if (expect(")")) break;
        consume(",", "to continue specify parameter");
      }
    consume(")", "to end parameter definition");
    consume(":", "to specify return type");
    n->rtype = type_name();
    fn_depth++;
    n->b = block(false);
    fn_depth--;
    return n;
  }
  int fn_depth = 0;
  NodeP while_() {
    auto n = here(Node::While);
    consume("while");
    n->a = expr();
    n->b = block(false);
    return n;
  }
  NodeP decl(const std::string& name, NodeP e, int flag, NodeP at) {
    auto d = mk(Node::Decl, at->line, at->col);
    d->text = name;
    d->a = std::move(e);
    d->flag = flag;
    return d;
  }
  NodeP idnode(const std::string& name, NodeP at) {
    auto d = mk(Node::Id, at->line, at->col);
    d->text = name;
    return d;
  }
  NodeP binary(const std::string& op, NodeP a, NodeP b) {
    auto n = mk(Node::Binary, a->line, a->col);
    n->text = op;
    n->a = std::move(a);
    n->b = std::move(b);
    return n;
  }
  NodeP for_() {  // jit.cpp:1519-1557
    auto n = here(Node::For);
    consume("for");
    const size_t save = pos;
    if (!eof() && idstart(src[pos])) {
      auto at = here(Node::Id);
      std::string name = id();
      if (accept("in")) {
        NodeP begin = expr();
        if (accept("..")) {
          NodeP end = expr();
          n->a = decl(name, begin, 0, at);
          n->b = binary("<", idnode(name, at), end);
          auto inc = mk(Node::Unary, at->line, at->col);
          inc->text = "++x";
          inc->a = idnode(name, at);
          n->c = inc;
        } else {
          consume("~", "or .. to specify range");
          NodeP step = expr();
          consume("~", "to specify range end");
          NodeP end = expr();
          n->a = decl(name, begin, 0, at);
          n->b = binary("<=", idnode(name, at), end);
          n->c = binary("+=", idnode(name, at), step);
        }
        n->d = block(false);
        return n;
      }
    }
    pos = save;
    n->a = stmt();
    n->b = expr();
    consume(";");
    n->c = expr();
    n->d = block(false);
    return n;
  }
  NodeP if_chain() {  // if / else if / else
    auto n = here(Node::If);
    consume("if");
    n->a = expr();
    n->b = block(false);
    if (expect("else")) {
      const size_t save = pos;
      accept("else");
      if (expect("if")) n->c = if_chain();
      else {
        pos = save;
        consume("else");
        n->c = block(false);
      }
    }
    return n;
  }
  NodeP stmt() {  // jit.cpp:1707-1744
    auto at = here(Node::Empty);
    if (accept(";")) return at;
    NodeP s;
    if (accept("break")) s = mk(Node::Break, at->line, at->col);
    else if (accept("continue")) s = mk(Node::Continue, at->line, at->col);
    else if (accept("return")) {
      if (fn_depth == 0) error("`return` can only be used inside a function");
      s = mk(Node::Return, at->line, at->col);
      if (!expect(";")) s->a = expr();
    }
    else {
      if (!eof() && idstart(src[pos])) {
        const size_t save = pos;
        std::string name = id();
        if (accept("=")) s = decl(name, expr(), 2, at);
        else if (accept(":=")) s = decl(name, expr(), 0, at);
        else if (accept("&=")) s = decl(name, expr(), 1, at);
        else {
          pos = save;
          s = mk(Node::ExprStmt, at->line, at->col);
          s->a = expr();
        }
      } else {
        s = mk(Node::ExprStmt, at->line, at->col);
        s->a = expr();
      }
    }
    consume(";", "to end statement");
    return s;
  }

  NodeP expr() {  // jit.cpp:1746-1820
    std::vector<NodeP> exprs;
    std::vector<std::pair<long, std::string>> ops;
    if (expect("(")) {
      // `()` or `(id :` would start a lambda
      size_t p = pos + 1;
      while (p < src.size() && isspace((unsigned char)src[p])) p++;
      if (p < src.size() && src[p] == ')') error("lambda expressions are not supported by this front-end");
      if (p < src.size() && idstart(src[p])) {
        size_t q = p;
        while (q < src.size() && (idstart(src[q]) || isdigit((unsigned char)src[q]))) q++;
        while (q < src.size() && isspace((unsigned char)src[q])) q++;
        if (q < src.size() && src[q] == ':' && !(q + 1 < src.size() && src[q + 1] == '='))
          error("lambda expressions are not supported by this front-end");
      }
      consume("(");
      exprs.push_back(expr());
      consume(")", "to balance the parenthesis");
    } else {
      exprs.push_back(expr0());
    }
    // the reference's codes: leading-0 literals are octal, the multiplicative ones decimal
    static const std::pair<const char*, long> table[] = {
        {"+=", 0000100000}, {"-=", 0000100001}, {"*=", 0000100010}, {"/=", 0000100011}, {"%=", 0000100100},
        {"||", 0001000001}, {"&&", 0001000000}, {"!=", 0010000101}, {"==", 0010000100}, {">=", 0010000011},
        {"<=", 0010000010}, {">", 0010000001},  {"<", 0010000000},  {"-", 0100000001},  {"+", 0100000000},
        {"^", 1000000011},  {"%", 1000000010},  {"/", 1000000001},  {"*", 1000000000}};
    while (true) {
      bool found = false;
      for (auto& t : table)
        if (accept(t.first)) {
          ops.push_back({t.second, t.first});
          found = true;
          break;
        }
      if (!found) break;
      if (accept("(")) {
        exprs.push_back(expr());
        accept(")");
      } else {
        exprs.push_back(expr0());
      }
    }
    while (!ops.empty()) {
      long best = 0;
      size_t index = 0;
      for (size_t k = 0; k < ops.size(); k++)
        if (ops[k].first > best) {
          best = ops[k].first;
          index = k;
        }
      NodeP n = binary(ops[index].second, exprs[index], exprs[index + 1]);
      ops.erase(ops.begin() + index);
      exprs.erase(exprs.begin() + index, exprs.begin() + index + 2);
      exprs.insert(exprs.begin() + index, n);
    }
    return exprs[0];
  }
  NodeP unary(const std::string& op, NodeP x, NodeP at) {
    auto n = mk(Node::Unary, at->line, at->col);
    n->text = op;
    n->a = std::move(x);
    return n;
  }
  NodeP expr0() {  // jit.cpp:1821-1843
    auto at = here(Node::Empty);
    if (accept("++")) return unary("++x", pexpr(), at);
    if (accept("--")) return unary("--x", pexpr(), at);
    if (accept("+")) return unary("+x", pexpr(), at);
    if (accept("-")) return unary("-x", pexpr(), at);
    if (accept("!")) return unary("!x", pexpr(), at);
    NodeP p = pexpr();
    if (accept("++")) return unary("x++", p, at);
    if (accept("--")) return unary("x--", p, at);
    return p;
  }
  std::vector<NodeP> arg_list() {
    std::vector<NodeP> args;
    if (!expect(")"))
      while (true) {
        args.push_back(expr());
        if (expect(")")) break;
        consume(",", "to continue specifying argument");
      }
    return args;
  }
  NodeP pexpr() {  // jit.cpp:1844-1882
    NodeP p = pexpr_base();
    while (true) {
      if (expect("[")) {
        auto at = here(Node::Call);
        accept("[");
        at->text = "[]";
        at->kids = {p, expr()};
        consume("]", "to end subscription operator");
        p = at;
      } else if (expect("..")) {
        break;
      } else if (expect(".")) {
        auto at = here(Node::Member);
        accept(".");
        at->text = id();
        at->a = p;
        p = at;
      } else if (expect("(")) {
        if (p->kind == Node::Id) {
          auto at = mk(Node::Call, p->line, p->col);
          consume("(");
          at->text = p->text;
          at->kids = arg_list();
          consume(")");
          p = at;
        } else if (p->kind == Node::Member) {  // a.f(b) == f(a, b)
          auto at = mk(Node::Call, p->line, p->col);
          consume("(");
          at->text = p->text;
          at->kids = arg_list();
          at->kids.insert(at->kids.begin(), p->a);
          consume(")");
          p = at;
        } else {
          error("An identifier must precedes function call operator ()");
        }
      } else {
        break;
      }
    }
    return p;
  }
  NodeP pexpr_base() {  // jit.cpp:1883-1915
    auto at = here(Node::Empty);
    for (int v = 0; v < 2; v++) {
      const std::string w = v ? "true" : "false";
      if (expect(w)) {
        const size_t q = pos + w.size();
        if (q >= src.size() || !idstart(src[q])) {
          accept(w);
          auto n = mk(Node::Bool, at->line, at->col);
          n->flag = v;
          return n;
        }
      }
    }
    if (expect("\"") || expect("'")) return string_literal();
    if (expect("[")) {
      auto n = mk(Node::Vec, at->line, at->col);
      consume("[", "to start short vector definition");
      if (!accept("]"))
        while (true) {
          n->kids.push_back(expr());
          if (accept("]")) break;
          consume(",", "to specify more element");
        }
      return n;
    }
    if (expect("(")) {
      consume("(");
      NodeP e = expr();
      consume(")", "to balance the parenthesis");
      return e;
    }
    if (!eof() && (isdigit((unsigned char)src[pos]) || src[pos] == '-' || src[pos] == '.') && !expect("..")) return number();
    if (!eof() && idstart(src[pos])) {
      auto n = mk(Node::Id, at->line, at->col);
      n->text = id();
      return n;
    }
    error("Expect a primary expression");
  }
  std::string id() {
    if (eof() || !idstart(src[pos])) error("Expect a letter or `_` to start an identifier");
    std::string s;
    while (pos < src.size() && (idstart(src[pos]) || isdigit((unsigned char)src[pos]))) s.push_back(src[pos++]);
    skip();
    return s;
  }
  NodeP number() {  // jit.cpp:2010-2032
    auto n = here(Node::Num);
    bool pass = false;
    std::string s;
    while (true) {
      s.push_back(src[pos++]);
      if (s.back() == '.') pass = true;
      if (expect("..")) break;
      if (eof()) break;
      const char c = src[pos];
      if (!(isdigit((unsigned char)c) || (!pass && c == '.'))) break;
    }
    if (!pass) {
      if (s.size() > 15 || std::stoll(s) > 2147483647LL) error("This number is too large, need to be < 2147483647");
    }
    skip();
    n->text = s;
    return n;
  }
  NodeP string_literal() {  // jit.cpp:2033-2066
    auto n = here(Node::Str);
    const char q = src[pos++];
    std::string s;
    bool escape = false;
    while (pos < src.size()) {
      const char c = src[pos];
      if (escape) {
        if (c == 'n') s.back() = '\n';
        else if (c == 't') s.back() = '\t';
        else if (c == '"') s.back() = '"';
        else error("Unknown escape character");
        escape = false;
        pos++;
        continue;
      }
      if (c == q) break;
      pos++;
      s.push_back(c);
      escape = c == '\\';
    }
    if (pos >= src.size()) error(std::string("Expect `") + q + "` to end string literal");
    pos++;
    skip();
    n->text = s;
    return n;
  }
};

// ------------------------------------------------------------------------------------------------
// Interpreter
// ------------------------------------------------------------------------------------------------
struct BreakSignal {};
struct ContinueSignal {};
struct ReturnSignal {
  Cell value;  // null for `return;`
};

struct Interp {
  Registry R;
  std::vector<std::map<std::string, Cell>> scopes;
  std::string out;
  int flags = 0, device = 0;
  std::shared_ptr<struct FilmObj> last_film;
  // A tree-walking interpreter is ~1000x slower than the reference's JIT on arithmetic loops
  // (scenes/benchmark.pine would run for hours): scripts get a budget of evaluated nodes
  // (PINE_PRL_MAX_STEPS, default 4e8 ~ half a minute) instead of hanging the host.
  unsigned long long steps = 0, max_steps = 400000000ull;

  Interp();
  [[noreturn]] void error(const Node& n, const std::string& m) const { fail(std::to_string(n.line) + ":" + std::to_string(n.col) + ": " + m); }
  void log(const std::string& s) {
    out += s;
    if (flags & PINE_PRL_ECHO) fputs(s.c_str(), stdout);
  }

  Cell find_var(const std::string& name) const {
    for (auto it = scopes.rbegin(); it != scopes.rend(); ++it) {
      auto f = it->find(name);
      if (f != it->end()) return f->second;
    }
    return nullptr;
  }

  static std::string sig(const std::vector<Cell>& args) {
    std::string s;
    for (size_t k = 0; k < args.size(); k++) s += (k ? ", " : "") + args[k]->type;
    return s;
  }
  // Context::find_f
  Resolved resolve(const Node& at, const std::string& name, const std::vector<Cell>& args) const {
    auto range = R.fns.equal_range(name);
    size_t best = args.size();
    std::vector<Resolved> cands;
    for (auto it = range.first; it != range.second; ++it) {
      const Fn& f = it->second;
      if (f.ptypes.size() != args.size()) continue;
      Resolved r;
      r.fn = &f;
      size_t diff = 0;
      for (size_t k = 0; k < args.size(); k++) {
        std::string pt = f.ptypes[k];
        const bool mut_ref = !pt.empty() && pt.back() == '&';
        if (mut_ref) pt.pop_back();
        if (args[k]->type == pt) continue;
        const Fn* cv = (!mut_ref && R.types.count(pt)) ? R.unique("@convert." + args[k]->type + "." + pt) : nullptr;
        if (cv) {
          r.converts.push_back({k, cv});
          diff++;
        } else {
          diff = args.size() + 1;
          break;
        }
      }
      if (diff <= best) {
        if (diff < best) cands.clear();
        best = diff;
        cands.push_back(r);
      }
    }
    if (cands.size() == 1) return cands[0];
    if (cands.size() > 1) {
      std::string c;
      for (auto& r : cands) {
        c += "\n  " + name + "(";
        for (size_t k = 0; k < r.fn->ptypes.size(); k++) c += (k ? ", " : "") + r.fn->ptypes[k];
        c += ")";
      }
      error(at, "Ambiguous function call `" + name + "(" + sig(args) + ")`, candidates:" + c);
    }
    if (range.first != range.second) {
      std::string c;
      for (auto it = range.first; it != range.second; ++it) {
        c += "\n  " + name + "(";
        for (size_t k = 0; k < it->second.ptypes.size(); k++) c += (k ? ", " : "") + it->second.ptypes[k];
        c += ")";
      }
      error(at, "Function `" + name + "(" + sig(args) + ")` is not found, candidates:" + c);
    }
    error(at, "Function `" + name + "(" + sig(args) + ")` is not found");
  }
  Cell call(const Node& at, const std::string& name, std::vector<Cell> args) {
    Resolved r = resolve(at, name, args);
    for (auto& cv : r.converts) {
      std::vector<Cell> one{args[cv.first]};
      args[cv.first] = cell(cv.second->impl(*this, one));
    }
    Value v = r.fn->impl(*this, args);
    if (!r.fn->rtype.empty() && r.fn->rtype.back() == '&') return args[0];  // `T& f(T& self, ...)`: the object itself
    return cell(std::move(v));
  }

  Cell eval(const NodeP& n) {
    if (++steps > max_steps) error(*n, "script exceeded the interpreter's step budget (PINE_PRL_MAX_STEPS)");
    switch (n->kind) {
      case Node::Num:
        if (n->text.find('.') != std::string::npos) return cell(mk_f32(psl_stof(n->text)));
        return cell(mk_i32(psl_stoi(n->text)));
      case Node::Bool: return cell(mk_bool(n->flag != 0));
      case Node::Str: return cell(mk_str(n->text));
      case Node::Vec: {  // jit.cpp:1014-1023
        if (n->kids.size() < 2 || n->kids.size() > 4) error(*n, "Only 2, 3, or 4 items can exist inside []");
        std::vector<Cell> a;
        bool any_f = false;
        for (auto& k : n->kids) {
          a.push_back(eval(k));
          any_f |= a.back()->type == "f32";
        }
        return call(*n, "vec" + std::to_string(a.size()) + (any_f ? "" : "i"), a);
      }
      case Node::Id: {
        if (Cell c = find_var(n->text)) return c;
        auto k = R.constants.find(n->text);
        if (k != R.constants.end()) return cell(k->second);
        error(*n, "Variable `" + n->text + "` is not found");
      }
      case Node::Call: {
        std::vector<Cell> a;
        for (auto& k : n->kids) a.push_back(eval(k));
        return call(*n, n->text, a);
      }
      case Node::Member: {
        Cell x = eval(n->a);
        const Fn* f = R.unique("@ma." + x->type + "." + n->text);
        if (!f) error(*n, "Can't find member `" + n->text + "` in type `" + x->type + "`");
        std::vector<Cell> one{x};
        return cell(f->impl(*this, one));
      }
      case Node::Unary: {
        std::vector<Cell> a{eval(n->a)};
        return call(*n, n->text, a);
      }
      case Node::Binary: {
        std::vector<Cell> a{eval(n->a), eval(n->b)};  // both sides always evaluated (no short circuit: jit.cpp:1185-1188)
        return call(*n, n->text, a);
      }
      default: error(*n, "internal: not an expression");
    }
  }
  bool truth(const NodeP& cond) {
    Cell c = eval(cond);
    if (c->type == "bool") return c->b;
    const Fn* f = R.unique("@convert." + c->type + ".bool");
    if (!f) error(*cond, "Type `" + c->type + "` is not convertible to bool");
    std::vector<Cell> one{c};
    return f->impl(*this, one).b;
  }

  // A script function (jit.cpp FunctionDefinition): registered like a built-in, so calls resolve through the
  // same overload / one-step-conversion rules.  The body sees its parameters only -- in the reference it is a
  // separate JIT'd function and the script's top-level variables are locals of main() -- plus the registered
  // constants and functions.  The returned value converts to the declared type by a registered conversion.
  void define_function(const NodeP& def) {
    std::vector<std::string> ptypes;
    for (auto& pr : def->params) {
      std::string t = pr.second;
      const std::string bare = (!t.empty() && t.back() == '&') ? t.substr(0, t.size() - 1) : t;
      if (!known_type(bare)) error(*def, "Type `" + bare + "` is not found");
      ptypes.push_back(t);
    }
    if (def->rtype != "void" && !known_type(def->rtype)) error(*def, "Type `" + def->rtype + "` is not found");
    NodeP d = def;
    R.def(def->text, ptypes, def->rtype, [d](Interp& in, std::vector<Cell>& args) -> Value {
      std::vector<std::map<std::string, Cell>> saved;
      saved.swap(in.scopes);
      in.scopes.emplace_back();
      for (size_t k = 0; k < d->params.size(); k++) {
        const std::string& t = d->params[k].second;
        // by value unless declared `T&`
        in.scopes.back()[d->params[k].first] = (!t.empty() && t.back() == '&') ? args[k] : cell(*args[k]);
      }
      if (++in.call_depth > 200) {
        in.call_depth--;
        in.scopes.swap(saved);
        in.error(*d, "function calls nested more than 200 deep");
      }
      Cell result;
      try {
        in.exec(d->b);
      } catch (ReturnSignal& r) {
        result = r.value;
      } catch (...) {
        in.call_depth--;
        in.scopes.swap(saved);
        throw;
      }
      in.call_depth--;
      in.scopes.swap(saved);
      if (d->rtype == "void") return Value();
      if (!result) in.error(*d, "function `" + d->text + "` ended without returning a `" + d->rtype + "`");
      if (result->type == d->rtype) return *result;
      if (const Fn* cv = in.R.unique("@convert." + result->type + "." + d->rtype)) {
        std::vector<Cell> one{result};
        return cv->impl(in, one);
      }
      in.error(*d, "function `" + d->text + "` returns `" + result->type + "` where `" + d->rtype + "` is declared");
    });
  }
  bool known_type(const std::string& t) const {
    static const char* basic[] = {"i32", "f32", "bool", "str", "vec2", "vec3", "vec4", "vec2i", "vec3i", "vec4i", "mat4"};
    for (const char* b : basic)
      if (t == b) return true;
    if (R.types.count(t)) return true;
    return R.fns.count(t) != 0;  // a type with a registered constructor of its own name (Scene, Rect, Diffuse, ...)
  }
  int call_depth = 0;
  void exec(const NodeP& n) {
    switch (n->kind) {
      case Node::Empty: return;
      case Node::Block: {
        scopes.emplace_back();
        try {
          for (auto& k : n->kids) exec(k);
        } catch (...) {
          scopes.pop_back();
          throw;
        }
        scopes.pop_back();
        return;
      }
      case Node::ExprStmt: eval(n->a); return;
      case Node::Return: throw ReturnSignal{n->a ? eval(n->a) : nullptr};
      case Node::FnDef: define_function(n); return;
      case Node::Decl: {  // jit.cpp:1252-1264
        if (n->flag == 2)
          if (Cell x = find_var(n->text)) {
            std::vector<Cell> a{x, eval(n->a)};
            call(*n, "=", a);
            return;
          }
        Cell v = eval(n->a);
        if (n->flag == 1 || v.use_count() == 1) scopes.back()[n->text] = v;  // reference, or a temporary
        else scopes.back()[n->text] = cell(*v);                              // copy of an l-value
        return;
      }
      case Node::While: {
        while (truth(n->a)) {
          try {
            exec(n->b);
          } catch (BreakSignal&) {
            break;
          } catch (ContinueSignal&) {
          }
        }
        return;
      }
      case Node::For: {
        scopes.emplace_back();
        try {
          exec(n->a);
          while (truth(n->b)) {
            try {
              exec(n->d);
            } catch (BreakSignal&) {
              break;
            } catch (ContinueSignal&) {
            }
            eval(n->c);
          }
        } catch (...) {
          scopes.pop_back();
          throw;
        }
        scopes.pop_back();
        return;
      }
      case Node::If: {
        if (truth(n->a)) exec(n->b);
        else if (n->c) exec(n->c);
        return;
      }
      case Node::Break: throw BreakSignal{};
      case Node::Continue: throw ContinueSignal{};
      default: eval(n); return;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// The function table
// ------------------------------------------------------------------------------------------------
static std::string describe_scene(pine_gpu_scene* h) {
  const int64_t n = pine_gpu_scene_describe(h, nullptr, 0);
  std::string s(size_t(n) + 1, '\0');
  pine_gpu_scene_describe(h, &s[0], n + 1);
  s.resize(size_t(n));
  return s;
}
template <class T>
static std::shared_ptr<T> obj(const Cell& c) {
  auto p = std::dynamic_pointer_cast<T>(c->o);
  if (!p) fail("internal: object of type `" + c->type + "` has the wrong payload");
  return p;
}
static void gpu_check(int rc, const char* what) {
  if (rc < 0) fail(std::string(what) + ": " + pine_gpu_last_error());
}

// instantiate a node tree in the scene's node table (shared sub-trees once per material)
static int instantiate_node(SceneObj& s, const NodeP3& n, std::map<const NodeObj*, int>& memo) {
  auto it = memo.find(n.get());
  if (it != memo.end()) return it->second;
  int id = -1;
  if (n->kind == "constf") id = pine_gpu_scene_node_constf(s.h, n->f);
  else if (n->kind == "const3") id = pine_gpu_scene_node_const3(s.h, n->v);
  else if (n->kind == "position") id = pine_gpu_scene_node_input(s.h, 0);
  else if (n->kind == "normal") id = pine_gpu_scene_node_input(s.h, 1);
  else if (n->kind == "uv") id = pine_gpu_scene_node_input(s.h, 2);
  else if (n->kind == "bin") {
    const int x = instantiate_node(s, n->a, memo), y = instantiate_node(s, n->b, memo);
    id = pine_gpu_scene_node_binary(s.h, n->op, x, y);
  } else if (n->kind == "un") id = pine_gpu_scene_node_unary(s.h, n->op, instantiate_node(s, n->a, memo));
  else if (n->kind == "comp") id = pine_gpu_scene_node_component(s.h, instantiate_node(s, n->a, memo), n->n);
  else if (n->kind == "tovec3") {
    const int x = instantiate_node(s, n->a, memo);
    const int y = n->b ? instantiate_node(s, n->b, memo) : -1, z = n->c ? instantiate_node(s, n->c, memo) : -1;
    id = pine_gpu_scene_node_to_vec3(s.h, x, y, z);
  } else if (n->kind == "checker") id = pine_gpu_scene_node_checkerboard(s.h, instantiate_node(s, n->a, memo), n->f);
  else if (n->kind == "splat") id = pine_gpu_scene_node_splat(s.h, instantiate_node(s, n->a, memo));
  gpu_check(id, "shading node");
  memo[n.get()] = id;
  return id;
}

Interp::Interp() {
  scopes.emplace_back();
  auto& r = R;
  for (const char* t : {"bool", "i32", "f32", "str", "str_view", "vec2", "vec3", "vec4", "vec2i", "vec3i", "vec4i", "mat4"}) r.types[t] = true;

  // ---- primitives (context.cpp:17-60, context.h:587-607) ----
  auto arith = [&](const std::string& T, bool is_int) {
    auto bin = [&](const std::string& op, std::function<Value(const Value&, const Value&)> f) {
      r.def(op, {T, T}, T, [f](Interp&, std::vector<Cell>& a) { return f(*a[0], *a[1]); });
    };
    auto cmp = [&](const std::string& op, std::function<bool(const Value&, const Value&)> f) {
      r.def(op, {T, T}, "bool", [f](Interp&, std::vector<Cell>& a) { return mk_bool(f(*a[0], *a[1])); });
    };
    auto asg = [&](const std::string& op, std::function<Value(const Value&, const Value&)> f) {
      r.def(op, {T + "&", T}, T + "&", [f](Interp&, std::vector<Cell>& a) {
        *a[0] = f(*a[0], *a[1]);
        return Value();
      });
    };
    if (is_int) {
      auto divz = [](int a, int b) {
        if (b == 0) fail("integer division by zero");
        return b == -1 ? int(0u - unsigned(a)) : a / b;
      };
      auto remz = [](int a, int b) {
        if (b == 0) fail("integer remainder by zero");
        return b == -1 ? 0 : a % b;
      };
      auto I = [](const Value& v) { return v.i[0]; };
      auto wrap = [](int64_t v) { return mk_i32(int32_t(uint32_t(uint64_t(v)))); };
      bin("+", [=](const Value& a, const Value& b) { return wrap(int64_t(I(a)) + I(b)); });
      bin("-", [=](const Value& a, const Value& b) { return wrap(int64_t(I(a)) - I(b)); });
      bin("*", [=](const Value& a, const Value& b) { return wrap(int64_t(I(a)) * I(b)); });
      bin("/", [=](const Value& a, const Value& b) { return mk_i32(divz(I(a), I(b))); });
      bin("%", [=](const Value& a, const Value& b) { return mk_i32(remz(I(a), I(b))); });
      asg("+=", [=](const Value& a, const Value& b) { return wrap(int64_t(I(a)) + I(b)); });
      asg("-=", [=](const Value& a, const Value& b) { return wrap(int64_t(I(a)) - I(b)); });
      asg("*=", [=](const Value& a, const Value& b) { return wrap(int64_t(I(a)) * I(b)); });
      asg("/=", [=](const Value& a, const Value& b) { return mk_i32(divz(I(a), I(b))); });
      asg("%=", [=](const Value& a, const Value& b) { return mk_i32(remz(I(a), I(b))); });
      cmp("==", [=](const Value& a, const Value& b) { return I(a) == I(b); });
      cmp("!=", [=](const Value& a, const Value& b) { return I(a) != I(b); });
      cmp("<", [=](const Value& a, const Value& b) { return I(a) < I(b); });
      cmp(">", [=](const Value& a, const Value& b) { return I(a) > I(b); });
      cmp("<=", [=](const Value& a, const Value& b) { return I(a) <= I(b); });
      cmp(">=", [=](const Value& a, const Value& b) { return I(a) >= I(b); });
      r.def("-x", {T}, T, [=](Interp&, std::vector<Cell>& a) { return wrap(-int64_t(a[0]->i[0])); });
    } else {
      auto F = [](const Value& v) { return v.f[0]; };
      bin("+", [=](const Value& a, const Value& b) { return mk_f32(F(a) + F(b)); });
      bin("-", [=](const Value& a, const Value& b) { return mk_f32(F(a) - F(b)); });
      bin("*", [=](const Value& a, const Value& b) { return mk_f32(F(a) * F(b)); });
      bin("/", [=](const Value& a, const Value& b) { return mk_f32(F(a) / F(b)); });
      asg("+=", [=](const Value& a, const Value& b) { return mk_f32(F(a) + F(b)); });
      asg("-=", [=](const Value& a, const Value& b) { return mk_f32(F(a) - F(b)); });
      asg("*=", [=](const Value& a, const Value& b) { return mk_f32(F(a) * F(b)); });
      asg("/=", [=](const Value& a, const Value& b) { return mk_f32(F(a) / F(b)); });
      cmp("==", [=](const Value& a, const Value& b) { return F(a) == F(b); });
      cmp("!=", [=](const Value& a, const Value& b) { return F(a) != F(b); });
      cmp("<", [=](const Value& a, const Value& b) { return F(a) < F(b); });
      cmp(">", [=](const Value& a, const Value& b) { return F(a) > F(b); });
      cmp("<=", [=](const Value& a, const Value& b) { return F(a) <= F(b); });
      cmp(">=", [=](const Value& a, const Value& b) { return F(a) >= F(b); });
      r.def("-x", {T}, T, [=](Interp&, std::vector<Cell>& a) { return mk_f32(-a[0]->f[0]); });
    }
    r.def("=", {T + "&", T}, T + "&", [](Interp&, std::vector<Cell>& a) {
      *a[0] = *a[1];
      return Value();
    });
  };
  arith("i32", true);
  arith("f32", false);
  r.convert("i32", "f32", [](const Value& v) { return mk_f32(float(v.i[0])); });                     // f32.ctor_variant<int>
  r.def("i32", {"f32"}, "i32", [](Interp&, std::vector<Cell>& a) { return mk_i32(int(a[0]->f[0])); });  // explicit only
  for (const char* op : {"==", "!=", "&&", "||"}) {
    const std::string o = op;
    r.def(o, {"bool", "bool"}, "bool", [o](Interp&, std::vector<Cell>& a) {
      const bool x = a[0]->b, y = a[1]->b;
      return mk_bool(o == "==" ? x == y : o == "!=" ? x != y : o == "&&" ? (x && y) : (x || y));
    });
  }
  r.def("=", {"bool&", "bool"}, "bool&", [](Interp&, std::vector<Cell>& a) {
    *a[0] = *a[1];
    return Value();
  });
  r.def("!x", {"bool"}, "bool", [](Interp&, std::vector<Cell>& a) { return mk_bool(!a[0]->b); });
  r.def("++x", {"i32&"}, "i32&", [](Interp&, std::vector<Cell>& a) {
    a[0]->i[0] = int32_t(uint32_t(a[0]->i[0]) + 1u);
    return Value();
  });
  r.def("--x", {"i32&"}, "i32&", [](Interp&, std::vector<Cell>& a) {
    a[0]->i[0] = int32_t(uint32_t(a[0]->i[0]) - 1u);
    return Value();
  });
  r.def("x++", {"i32&"}, "i32", [](Interp&, std::vector<Cell>& a) {
    Value old = *a[0];
    a[0]->i[0] = int32_t(uint32_t(a[0]->i[0]) + 1u);
    return old;
  });
  r.def("x--", {"i32&"}, "i32", [](Interp&, std::vector<Cell>& a) {
    Value old = *a[0];
    a[0]->i[0] = int32_t(uint32_t(a[0]->i[0]) - 1u);
    return old;
  });
  // mixed int / float arithmetic (context.cpp:38-53): the int operand is converted as C++ does
  for (const char* op : {"+", "-", "*", "/"}) {
    const char o = op[0];
    auto ap = [o](float a, float b) { return o == '+' ? a + b : o == '-' ? a - b : o == '*' ? a * b : a / b; };
    r.def(op, {"i32", "f32"}, "f32", [ap](Interp&, std::vector<Cell>& a) { return mk_f32(ap(float(a[0]->i[0]), a[1]->f[0])); });
    r.def(op, {"f32", "i32"}, "f32", [ap](Interp&, std::vector<Cell>& a) { return mk_f32(ap(a[0]->f[0], float(a[1]->i[0]))); });
    r.def(std::string(op) + "=", {"f32&", "i32"}, "f32&", [ap](Interp&, std::vector<Cell>& a) {
      a[0]->f[0] = ap(a[0]->f[0], float(a[1]->i[0]));
      return Value();
    });
    r.def(std::string(op) + "=", {"i32&", "f32"}, "i32&", [ap](Interp&, std::vector<Cell>& a) {
      a[0]->i[0] = int(ap(float(a[0]->i[0]), a[1]->f[0]));
      return Value();
    });
  }
  r.def("^", {"i32", "i32"}, "i32", [](Interp&, std::vector<Cell>& a) {  // psl::powi (src/psl/math.h)
    int x = a[0]->i[0], e = a[1]->i[0], y = 1;
    for (int k = 0; k < e; k++) y = int(uint32_t(y) * uint32_t(x));
    return mk_i32(y);
  });
  r.def("^", {"f32", "f32"}, "f32", [](Interp&, std::vector<Cell>& a) { return mk_f32(std::pow(a[0]->f[0], a[1]->f[0])); });
  r.constants["Pi"] = mk_f32(3.1415926535897932f);  // math.h, math.cpp:7-8
  r.constants["E"] = mk_f32(2.7182818284590452f);
  {
    const float X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1};  // vecmath.cpp:305-307
    r.constants["X"] = mk_vecf(3, X);
    r.constants["Y"] = mk_vecf(3, Y);
    r.constants["Z"] = mk_vecf(3, Z);
  }
  // scalar math used by scene scripts (math.cpp:9-41); float overloads + int where registered
  auto f1 = [&](const char* name, float (*fn)(float)) {
    r.def(name, {"f32"}, "f32", [fn](Interp&, std::vector<Cell>& a) { return mk_f32(fn(a[0]->f[0])); });
  };
  f1("sqrt", [](float x) { return std::sqrt(x); });
  f1("floor", [](float x) { return std::floor(x); });
  f1("ceil", [](float x) { return std::ceil(x); });
  f1("sin", [](float x) { return std::sin(x); });
  f1("cos", [](float x) { return std::cos(x); });
  f1("tan", [](float x) { return std::tan(x); });
  f1("exp", [](float x) { return std::exp(x); });
  f1("log", [](float x) { return std::log(x); });
  f1("abs", [](float x) { return std::fabs(x); });
  f1("sqr", [](float x) { return x * x; });
  r.def("abs", {"i32"}, "i32", [](Interp&, std::vector<Cell>& a) { return mk_i32(a[0]->i[0] < 0 ? -a[0]->i[0] : a[0]->i[0]); });
  r.def("sqr", {"i32"}, "i32", [](Interp&, std::vector<Cell>& a) { return mk_i32(a[0]->i[0] * a[0]->i[0]); });
  r.def("min", {"f32", "f32"}, "f32", [](Interp&, std::vector<Cell>& a) { return mk_f32(a[0]->f[0] < a[1]->f[0] ? a[0]->f[0] : a[1]->f[0]); });
  r.def("max", {"f32", "f32"}, "f32", [](Interp&, std::vector<Cell>& a) { return mk_f32(a[0]->f[0] > a[1]->f[0] ? a[0]->f[0] : a[1]->f[0]); });
  r.def("min", {"i32", "i32"}, "i32", [](Interp&, std::vector<Cell>& a) { return mk_i32(a[0]->i[0] < a[1]->i[0] ? a[0]->i[0] : a[1]->i[0]); });
  r.def("max", {"i32", "i32"}, "i32", [](Interp&, std::vector<Cell>& a) { return mk_i32(a[0]->i[0] > a[1]->i[0] ? a[0]->i[0] : a[1]->i[0]); });
  r.def("pow", {"f32", "f32"}, "f32", [](Interp&, std::vector<Cell>& a) { return mk_f32(std::pow(a[0]->f[0], a[1]->f[0])); });

  // ---- strings (context.cpp:62-78) ----
  r.convert("bool", "str", [](const Value& v) { return mk_str(v.b ? "true" : "false"); });
  r.convert("i32", "str", [](const Value& v) { return mk_str(psl_to_string(int64_t(v.i[0]))); });
  r.convert("f32", "str", [](const Value& v) { return mk_str(psl_to_string(v.f[0])); });
  r.convert("str", "str_view", [](const Value& v) { return retype(v, "str_view"); });
  r.def("+", {"str", "str"}, "str", [](Interp&, std::vector<Cell>& a) { return mk_str(a[0]->s + a[1]->s); });
  r.def("+=", {"str&", "str"}, "str&", [](Interp&, std::vector<Cell>& a) {
    a[0]->s += a[1]->s;
    return Value();
  });
  r.def("=", {"str&", "str"}, "str&", [](Interp&, std::vector<Cell>& a) {
    *a[0] = *a[1];
    return Value();
  });
  r.def("print", {"str"}, "void", [](Interp& in, std::vector<Cell>& a) {  // program_context.cpp:41-42
    in.log(a[0]->s);
    return Value();
  });
  r.def("println", {"str"}, "void", [](Interp& in, std::vector<Cell>& a) {
    in.log(a[0]->s + "\n");
    return Value();
  });

  // ---- vectors (vecmath.cpp:134-218; Complex type class context.h:587-599) ----
  for (int n = 2; n <= 4; n++) {
    const std::string V = "vec" + std::to_string(n), VI = V + "i";
    std::vector<std::string> ff(size_t(n), "f32"), ii(size_t(n), "i32");
    r.def(V, ff, V, [n](Interp&, std::vector<Cell>& a) {
      Value v;
      v.type = "vec" + std::to_string(n);
      for (int k = 0; k < n; k++) v.f[k] = a[size_t(k)]->f[0];
      return v;
    });
    r.def(VI, ii, VI, [n](Interp&, std::vector<Cell>& a) {
      Value v;
      v.type = "vec" + std::to_string(n) + "i";
      for (int k = 0; k < n; k++) v.i[k] = a[size_t(k)]->i[0];
      return v;
    });
    for (const char* op : {"+", "-", "*", "/"}) {
      const char o = op[0];
      r.def(op, {V, V}, V, [n, o](Interp&, std::vector<Cell>& a) {
        Value v = *a[0];
        for (int k = 0; k < n; k++) {
          const float x = a[0]->f[k], y = a[1]->f[k];
          v.f[k] = o == '+' ? x + y : o == '-' ? x - y : o == '*' ? x * y : x / y;
        }
        return v;
      });
      r.def(op, {VI, VI}, VI, [n, o](Interp&, std::vector<Cell>& a) {
        Value v = *a[0];
        for (int k = 0; k < n; k++) {
          const int x = a[0]->i[k], y = a[1]->i[k];
          if (o == '/' && y == 0) fail("integer division by zero");
          v.i[k] = o == '+' ? x + y : o == '-' ? x - y : o == '*' ? x * y : x / y;
        }
        return v;
      });
    }
    r.def("-x", {V}, V, [n](Interp&, std::vector<Cell>& a) {
      Value v = *a[0];
      for (int k = 0; k < n; k++) v.f[k] = -v.f[k];
      return v;
    });
    r.def("-x", {VI}, VI, [n](Interp&, std::vector<Cell>& a) {
      Value v = *a[0];
      for (int k = 0; k < n; k++) v.i[k] = -v.i[k];
      return v;
    });
    for (const std::string& T : {V, VI})
      r.def("=", {T + "&", T}, T + "&", [](Interp&, std::vector<Cell>& a) {
        *a[0] = *a[1];
        return Value();
      });
    // vector (x) scalar, both orders (vecmath.cpp:207-214); int scalars with float vectors convert per element
    for (const char* op : {"*", "/"}) {
      const bool mul = op[0] == '*';
      if (n <= 3) {
        r.def(op, {VI, "i32"}, VI, [n, mul](Interp&, std::vector<Cell>& a) {
          Value v = *a[0];
          for (int k = 0; k < n; k++) {
            if (!mul && a[1]->i[0] == 0) fail("integer division by zero");
            v.i[k] = mul ? v.i[k] * a[1]->i[0] : v.i[k] / a[1]->i[0];
          }
          return v;
        });
        r.def(op, {"i32", VI}, VI, [n, mul](Interp&, std::vector<Cell>& a) {
          Value v = *a[1];
          for (int k = 0; k < n; k++) {
            if (!mul && v.i[k] == 0) fail("integer division by zero");
            v.i[k] = mul ? a[0]->i[0] * v.i[k] : a[0]->i[0] / v.i[k];
          }
          return v;
        });
      }
      r.def(op, {V, "f32"}, V, [n, mul](Interp&, std::vector<Cell>& a) {
        Value v = *a[0];
        for (int k = 0; k < n; k++) v.f[k] = mul ? v.f[k] * a[1]->f[0] : v.f[k] / a[1]->f[0];
        return v;
      });
      r.def(op, {"f32", V}, V, [n, mul](Interp&, std::vector<Cell>& a) {
        Value v = *a[1];
        for (int k = 0; k < n; k++) v.f[k] = mul ? a[0]->f[0] * v.f[k] : a[0]->f[0] / v.f[k];
        return v;
      });
      r.def(op, {"i32", V}, V, [n, mul](Interp&, std::vector<Cell>& a) {
        Value v = *a[1];
        const float s = float(a[0]->i[0]);
        for (int k = 0; k < n; k++) v.f[k] = mul ? s * v.f[k] : s / v.f[k];
        return v;
      });
    }
    // members x y z w
    for (int k = 0; k < n; k++) {
      const std::string m(1, "xyzw"[k]);
      r.def("@ma." + V + "." + m, {V}, "f32", [k](Interp&, std::vector<Cell>& a) { return mk_f32(a[0]->f[k]); });
      r.def("@ma." + VI + "." + m, {VI}, "i32", [k](Interp&, std::vector<Cell>& a) { return mk_i32(a[0]->i[k]); });
    }
    r.def("[]", {V, "i32"}, "f32", [n](Interp&, std::vector<Cell>& a) {
      if (a[1]->i[0] < 0 || a[1]->i[0] >= n) fail("vector index out of range");
      return mk_f32(a[0]->f[a[1]->i[0]]);
    });
    auto vstr = [n](const Value& v, bool is_int) {
      std::string s = "[";
      for (int k = 0; k < n; k++) s += (k ? " " : "") + (is_int ? psl_to_string(int64_t(v.i[k])) : psl_to_string(v.f[k]));
      return mk_str(s + "]");
    };
    r.convert(V, "str", [vstr](const Value& v) { return vstr(v, false); });
    r.convert(VI, "str", [vstr](const Value& v) { return vstr(v, true); });
  }
  r.convert("vec2i", "vec2", [](const Value& v) {
    const float f[2] = {float(v.i[0]), float(v.i[1])};
    return mk_vecf(2, f);
  });
  r.convert("f32", "vec2", [](const Value& v) {
    const float f[2] = {v.f[0], v.f[0]};
    return mk_vecf(2, f);
  });
  r.convert("vec3i", "vec3", [](const Value& v) {
    const float f[3] = {float(v.i[0]), float(v.i[1]), float(v.i[2])};
    return mk_vecf(3, f);
  });
  r.convert("f32", "vec3", [](const Value& v) {
    const float f[3] = {v.f[0], v.f[0], v.f[0]};
    return mk_vecf(3, f);
  });
  r.convert("f32", "vec4", [](const Value& v) {
    const float f[4] = {v.f[0], v.f[0], v.f[0], v.f[0]};
    return mk_vecf(4, f);
  });
  r.def("vec2i", {"i32"}, "vec2i", [](Interp&, std::vector<Cell>& a) {
    const int i[2] = {a[0]->i[0], a[0]->i[0]};
    return mk_veci(2, i);
  });
  r.def("vec3i", {"i32"}, "vec3i", [](Interp&, std::vector<Cell>& a) {
    const int i[3] = {a[0]->i[0], a[0]->i[0], a[0]->i[0]};
    return mk_veci(3, i);
  });
  r.def("vec2i", {"vec2"}, "vec2i", [](Interp&, std::vector<Cell>& a) {  // ctor_variant_explicit
    const int i[2] = {int(a[0]->f[0]), int(a[0]->f[1])};
    return mk_veci(2, i);
  });
  r.def("vec3i", {"vec3"}, "vec3i", [](Interp&, std::vector<Cell>& a) {
    const int i[3] = {int(a[0]->f[0]), int(a[0]->f[1]), int(a[0]->f[2])};
    return mk_veci(3, i);
  });

  // ---- mat4 and the transform builders (vecmath.cpp:194-206,264-273; vecmath.h:1102-1180 via the C ABI) ----
  auto mk_mat = [](const float* m) {
    Value v;
    v.type = "mat4";
    memcpy(v.f, m, sizeof v.f);
    return v;
  };
  r.def("*", {"mat4", "mat4"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    pine_gpu_mat4_mul(a[0]->f, a[1]->f, m);
    return mk_mat(m);
  });
  r.def("=", {"mat4&", "mat4"}, "mat4&", [](Interp&, std::vector<Cell>& a) {
    *a[0] = *a[1];
    return Value();
  });
  r.def("translate", {"vec3"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    pine_gpu_mat4_translate(a[0]->f, m);
    return mk_mat(m);
  });
  r.def("translate", {"f32", "f32", "f32"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    const float v[3] = {a[0]->f[0], a[1]->f[0], a[2]->f[0]};
    pine_gpu_mat4_translate(v, m);
    return mk_mat(m);
  });
  r.def("scale", {"vec3"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    pine_gpu_mat4_scale(a[0]->f, m);
    return mk_mat(m);
  });
  r.def("scale", {"f32", "f32", "f32"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    const float v[3] = {a[0]->f[0], a[1]->f[0], a[2]->f[0]};
    pine_gpu_mat4_scale(v, m);
    return mk_mat(m);
  });
  r.def("scale", {"f32"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    const float v[3] = {a[0]->f[0], a[0]->f[0], a[0]->f[0]};
    pine_gpu_mat4_scale(v, m);
    return mk_mat(m);
  });
  r.def("rotate_x", {"f32"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    pine_gpu_mat4_rotate_x(a[0]->f[0], m);
    return mk_mat(m);
  });
  r.def("rotate_y", {"f32"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    pine_gpu_mat4_rotate_y(a[0]->f[0], m);
    return mk_mat(m);
  });
  r.def("rotate_z", {"f32"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    pine_gpu_mat4_rotate_z(a[0]->f[0], m);
    return mk_mat(m);
  });
  r.def("look_at", {"vec3", "vec3"}, "mat4", [mk_mat](Interp&, std::vector<Cell>& a) {
    float m[16];
    pine_gpu_mat4_look_at(a[0]->f, a[1]->f, m);
    return mk_mat(m);
  });

  // ---- shading nodes (node.cpp:29-116) ----
  r.types["Nodef"] = r.types["Node3f"] = true;
  auto nobj = [](const Cell& c) { return obj<NodeObj>(c); };
  auto mk_node = [](NodeP3 n) { return mk_obj(n->is3 ? "Node3f" : "Nodef", n); };
  auto constf = [](float v) {
    auto n = std::make_shared<NodeObj>();
    n->kind = "constf";
    n->f = v;
    return n;
  };
  auto const3 = [](const float* v) {
    auto n = std::make_shared<NodeObj>();
    n->kind = "const3";
    n->is3 = true;
    memcpy(n->v, v, 12);
    return n;
  };
  auto splat = [](NodeP3 x) {
    if (x->is3) return x;
    auto n = std::make_shared<NodeObj>();
    n->kind = "splat";
    n->is3 = true;
    n->a = x;
    return n;
  };
  auto bin = [](char op, NodeP3 x, NodeP3 y) {
    auto n = std::make_shared<NodeObj>();
    n->kind = "bin";
    n->op = op;
    n->is3 = x->is3;
    n->a = x;
    n->b = y;
    return n;
  };
  auto un = [](char op, NodeP3 x) {
    auto n = std::make_shared<NodeObj>();
    n->kind = "un";
    n->op = op;
    n->is3 = x->is3;
    n->a = x;
    return n;
  };
  auto tovec3 = [](NodeP3 x, NodeP3 y, NodeP3 z) {
    auto n = std::make_shared<NodeObj>();
    n->kind = "tovec3";
    n->is3 = true;
    n->a = x;
    n->b = y;
    n->c = z;
    return n;
  };
  r.convert("i32", "Nodef", [=](const Value& v) { return mk_node(constf(float(v.i[0]))); });
  r.convert("f32", "Nodef", [=](const Value& v) { return mk_node(constf(v.f[0])); });
  r.convert("vec3i", "Node3f", [=](const Value& v) {
    const float f[3] = {float(v.i[0]), float(v.i[1]), float(v.i[2])};
    return mk_node(const3(f));
  });
  r.convert("vec3", "Node3f", [=](const Value& v) { return mk_node(const3(v.f)); });
  for (const char* in : {"Position", "Normal", "UV"}) {
    const std::string kind = in == std::string("Position") ? "position" : in == std::string("Normal") ? "normal" : "uv";
    r.def(in, {}, "Node3f", [=](Interp&, std::vector<Cell>&) {
      auto n = std::make_shared<NodeObj>();
      n->kind = kind;
      n->is3 = true;
      return mk_node(n);
    });
  }
  for (const char* opn : {"+", "-", "*", "/", "^"}) {
    const char o = opn[0];
    r.def(opn, {"Nodef", "Nodef"}, "Nodef", [=](Interp&, std::vector<Cell>& a) { return mk_node(bin(o, nobj(a[0]), nobj(a[1]))); });
    r.def(opn, {"Node3f", "Node3f"}, "Node3f", [=](Interp&, std::vector<Cell>& a) { return mk_node(bin(o, nobj(a[0]), nobj(a[1]))); });
  }
  for (const char* opn : {"*", "^", "/"})  // (Node3f, Nodef): node.cpp:82,84,85
    r.def(opn, {"Node3f", "Nodef"}, "Node3f", [=](Interp&, std::vector<Cell>& a) { return mk_node(bin(opn[0], nobj(a[0]), splat(nobj(a[1])))); });
  for (const char* opn : {"*", "/"})  // (Nodef, Node3f): node.cpp:83,86
    r.def(opn, {"Nodef", "Node3f"}, "Node3f", [=](Interp&, std::vector<Cell>& a) { return mk_node(bin(opn[0], splat(nobj(a[0])), nobj(a[1]))); });
  r.def("-x", {"Nodef"}, "Nodef", [=](Interp&, std::vector<Cell>& a) { return mk_node(un('-', nobj(a[0]))); });
  r.def("-x", {"Node3f"}, "Node3f", [=](Interp&, std::vector<Cell>& a) { return mk_node(un('-', nobj(a[0]))); });
  for (auto& f : std::vector<std::pair<const char*, char>>{{"abs", 'a'}, {"sqr", 's'}, {"sqrt", 'r'}, {"fract", 'f'}}) {
    const char o = f.second;
    r.def(f.first, {"Nodef"}, "Nodef", [=](Interp&, std::vector<Cell>& a) { return mk_node(un(o, nobj(a[0]))); });
    r.def(f.first, {"Node3f"}, "Node3f", [=](Interp&, std::vector<Cell>& a) { return mk_node(un(o, nobj(a[0]))); });
  }
  auto comp = [=](std::vector<Cell>& a) {
    if (a[1]->i[0] < 0 || a[1]->i[0] > 2) fail("NodeComponent's second parameter should be 0, 1, or 2, but get " + std::to_string(a[1]->i[0]));
    auto n = std::make_shared<NodeObj>();
    n->kind = "comp";
    n->a = nobj(a[0]);
    n->n = a[1]->i[0];
    return mk_node(n);
  };
  r.def("[]", {"Node3f", "i32"}, "Nodef", [=](Interp&, std::vector<Cell>& a) { return comp(a); });
  r.def("Comp", {"Node3f", "i32"}, "Nodef", [=](Interp&, std::vector<Cell>& a) { return comp(a); });
  r.def("Vec3", {"Nodef"}, "Node3f", [=](Interp&, std::vector<Cell>& a) { return mk_node(tovec3(nobj(a[0]), nullptr, nullptr)); });
  r.def("Vec3", {"Nodef", "Nodef", "Nodef"}, "Node3f", [=](Interp&, std::vector<Cell>& a) { return mk_node(tovec3(nobj(a[0]), nobj(a[1]), nobj(a[2]))); });
  auto checker = [=](NodeP3 p, float ratio) {
    auto n = std::make_shared<NodeObj>();
    n->kind = "checker";
    n->a = p;
    n->f = ratio;
    return mk_node(n);
  };
  r.def("Checkerboard", {"Node3f"}, "Nodef", [=](Interp&, std::vector<Cell>& a) { return checker(nobj(a[0]), 0.5f); });
  r.def("Checkerboard", {"Node3f", "f32"}, "Nodef", [=](Interp&, std::vector<Cell>& a) { return checker(nobj(a[0]), a[1]->f[0]); });
  // lerp: node.cpp:88-102 (the scalar one is psl::lerp, math.cpp:21)
  r.def("lerp", {"f32", "f32", "f32"}, "f32", [](Interp&, std::vector<Cell>& a) { return mk_f32((1 - a[0]->f[0]) * a[1]->f[0] + a[0]->f[0] * a[2]->f[0]); });
  r.def("lerp", {"Nodef", "Nodef", "Nodef"}, "Nodef", [=](Interp&, std::vector<Cell>& a) {
    NodeP3 t = nobj(a[0]);
    return mk_node(bin('+', bin('*', t, nobj(a[2])), bin('*', bin('-', constf(1.0f), t), nobj(a[1]))));
  });
  r.def("lerp", {"Nodef", "Node3f", "Node3f"}, "Node3f", [=](Interp&, std::vector<Cell>& a) {
    NodeP3 t = nobj(a[0]);
    return mk_node(bin('+', bin('*', tovec3(t, nullptr, nullptr), nobj(a[2])),
                       bin('*', tovec3(bin('-', constf(1.0f), t), nullptr, nullptr), nobj(a[1]))));
  });
  r.def("lerp", {"Node3f", "Node3f", "Node3f"}, "Node3f", [=](Interp&, std::vector<Cell>& a) {
    NodeP3 t = nobj(a[0]);
    const float one[3] = {1.0f, 1.0f, 1.0f};
    return mk_node(bin('+', bin('*', t, nobj(a[2])), bin('*', bin('-', const3(one), t), nobj(a[1]))));
  });

  // ---- materials (material.cpp:46-62) ----
  auto material = [](const char* kind) {
    auto m = std::make_shared<MaterialObj>();
    m->kind = kind;
    return m;
  };
  r.def("Emissive", {"Node3f"}, "Emissive", [=](Interp&, std::vector<Cell>& a) {
    auto m = material("Emissive");
    m->albedo = nobj(a[0]);
    if (!m->albedo->constant()) fail("Emissive: only a constant colour is supported by this front-end");
    return mk_obj("Emissive", m);
  });
  r.def("Diffuse", {"Node3f"}, "Diffuse", [=](Interp&, std::vector<Cell>& a) {
    auto m = material("Diffuse");
    m->albedo = nobj(a[0]);
    return mk_obj("Diffuse", m);
  });
  for (int extra = 0; extra <= 3; extra++) {  // Uber(albedo, roughness[, metallic[, transmission[, ior]]])
    std::vector<std::string> pt{"Node3f", "Nodef"};
    for (int k = 0; k < extra; k++) pt.push_back(k < 2 ? "Nodef" : "f32");
    r.def("Uber", pt, "Uber", [=](Interp&, std::vector<Cell>& a) {
      auto m = material("Uber");
      m->albedo = nobj(a[0]);
      m->roughness = nobj(a[1]);
      m->metallic = extra >= 1 ? nobj(a[2]) : constf(0.0f);
      m->transmission = extra >= 2 ? nobj(a[3]) : constf(0.0f);
      if (extra >= 3) m->ior = a[4]->f[0];
      return mk_obj("Uber", m);
    });
  }
  r.def("Subsurface", {"Node3f", "Nodef", "vec3"}, "Subsurface", [=](Interp&, std::vector<Cell>& a) {
    auto m = material("Subsurface");
    m->albedo = nobj(a[0]);
    m->roughness = nobj(a[1]);
    if (!m->albedo->constant() || !m->roughness->constant()) fail("Subsurface: only constant parameters are supported by this front-end");
    memcpy(m->sigma_s, a[2]->f, 12);
    return mk_obj("Subsurface", m);
  });
  r.def("Metal", {"Node3f", "Nodef"}, "Metal", [=](Interp&, std::vector<Cell>& a) {
    auto m = material("Metal");
    m->albedo = nobj(a[0]);
    m->roughness = nobj(a[1]);
    return mk_obj("Metal", m);
  });
  for (const char* kind : {"Glossy", "Glass"})
    for (int with_ior = 0; with_ior < 2; with_ior++) {
      std::vector<std::string> pt{"Node3f", "Nodef"};
      if (with_ior) pt.push_back("Nodef");
      r.def(kind, pt, kind, [=](Interp&, std::vector<Cell>& a) {
        auto m = material(kind);
        m->albedo = nobj(a[0]);
        m->roughness = nobj(a[1]);
        m->ior_node = with_ior ? nobj(a[2]) : constf(1.4f);  // material.h:53,67
        return mk_obj(kind, m);
      });
    }
  for (const char* k : {"Emissive", "Diffuse", "Uber", "Subsurface", "Metal", "Glossy", "Glass"})
    r.convert(k, "Material", [](const Value& v) { return retype(v, "Material"); });

  // ---- shapes (geometry.cpp:901-946) ----
  auto shape = [](const char* kind, std::vector<float> p, bool flag = false) {
    auto s = std::make_shared<ShapeObj>();
    s->kind = kind;
    s->p = std::move(p);
    s->flag = flag;
    return mk_obj(kind, s);
  };
  auto cat = [](std::initializer_list<std::pair<const float*, int>> parts) {
    std::vector<float> p;
    for (auto& q : parts) p.insert(p.end(), q.first, q.first + q.second);
    return p;
  };
  r.def("Rect", {"vec3", "vec3", "vec3"}, "Rect", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Rect", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 3}})); });
  r.def("Rect", {"vec3", "vec3", "vec3", "bool"}, "Rect",
        [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Rect", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 3}}), a[3]->b); });
  r.def("AABB", {"vec3", "vec3"}, "AABB", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("AABB", cat({{a[0]->f, 3}, {a[1]->f, 3}})); });
  r.def("Box", {"vec3", "vec3"}, "AABB", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("AABB", cat({{a[0]->f, 3}, {a[1]->f, 3}})); });
  r.def("OBB", {"AABB", "mat4"}, "OBB", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("OBB", cat({{obj<ShapeObj>(a[0])->p.data(), 6}, {a[1]->f, 16}})); });
  r.def("Box", {"AABB", "mat4"}, "OBB", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("OBB", cat({{obj<ShapeObj>(a[0])->p.data(), 6}, {a[1]->f, 16}})); });
  r.def("Box", {"vec3", "vec3", "mat4"}, "OBB", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("OBB", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 16}})); });
  r.def("Sphere", {"vec3", "f32"}, "Sphere", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Sphere", cat({{a[0]->f, 3}, {a[1]->f, 1}})); });
  r.def("Disk", {"vec3", "vec3", "f32"}, "Disk", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Disk", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 1}})); });
  r.def("Cone", {"vec3", "vec3", "f32", "f32"}, "Cone",
        [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Cone", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 1}, {a[3]->f, 1}})); });
  r.def("Plane", {"vec3", "vec3"}, "Plane", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Plane", cat({{a[0]->f, 3}, {a[1]->f, 3}})); });
  r.def("Line", {"vec3", "vec3", "f32"}, "Line", [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Line", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 1}})); });
  r.def("Cylinder", {"vec3", "vec3", "f32"}, "Cylinder",
        [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Cylinder", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 1}})); });
  r.def("Triangle", {"vec3", "vec3", "vec3"}, "Triangle",
        [shape, cat](Interp&, std::vector<Cell>& a) { return shape("Triangle", cat({{a[0]->f, 3}, {a[1]->f, 3}, {a[2]->f, 3}})); });
  for (const char* k : {"Rect", "AABB", "OBB", "Sphere", "Disk", "Cone", "Plane", "Line", "Cylinder", "Triangle"}) r.convert(k, "Shape", [](const Value& v) { return retype(v, "Shape"); });
  r.def("@ma.AABB.lower", {"AABB"}, "vec3", [](Interp&, std::vector<Cell>& a) { return mk_vecf(3, obj<ShapeObj>(a[0])->p.data()); });
  r.def("@ma.AABB.upper", {"AABB"}, "vec3", [](Interp&, std::vector<Cell>& a) { return mk_vecf(3, obj<ShapeObj>(a[0])->p.data() + 3); });

  // ---- film, tone mappers, camera (film.cpp:97-119, camera.cpp:40-45) ----
  r.def("Uncharted2", {}, "Uncharted2", [](Interp&, std::vector<Cell>&) { return retype(mk_i32(0), "Uncharted2"); });
  r.def("ACES", {}, "ACES", [](Interp&, std::vector<Cell>&) { return retype(mk_i32(1), "ACES"); });
  r.convert("Uncharted2", "ToneMapper", [](const Value& v) { return retype(v, "ToneMapper"); });
  r.convert("ACES", "ToneMapper", [](const Value& v) { return retype(v, "ToneMapper"); });
  auto film = [](const Value& size, int tone) {
    if (size.i[0] <= 0 || size.i[1] <= 0) fail("Film: size must be positive");
    auto f = std::make_shared<FilmObj>();
    f->w = size.i[0];
    f->h = size.i[1];
    f->tone = tone;
    return mk_obj("Film", f);
  };
  r.def("Film", {"vec2i"}, "Film", [film](Interp&, std::vector<Cell>& a) { return film(*a[0], 0); });  // default tone mapper: Uncharted2 (film.h)
  r.def("Film", {"vec2i", "ToneMapper"}, "Film", [film](Interp&, std::vector<Cell>& a) { return film(*a[0], a[1]->i[0]); });
  auto camera = [](std::vector<Cell>& a) {
    auto c = std::make_shared<CameraObj>();
    c->film = obj<FilmObj>(a[0]);
    memcpy(c->from, a[1]->f, 12);
    memcpy(c->to, a[2]->f, 12);
    c->fov = a[3]->f[0];
    if (a.size() == 6) {
      c->len_radius = a[4]->f[0];
      c->focus = a[5]->f[0];
    }
    return mk_obj("ThinLenCamera", c);
  };
  r.def("ThinLenCamera", {"Film", "vec3", "vec3", "f32"}, "ThinLenCamera", [camera](Interp&, std::vector<Cell>& a) { return camera(a); });
  r.def("ThinLenCamera", {"Film", "vec3", "vec3", "f32", "f32", "f32"}, "ThinLenCamera", [camera](Interp&, std::vector<Cell>& a) { return camera(a); });
  r.convert("ThinLenCamera", "Camera", [](const Value& v) { return retype(v, "Camera"); });
  r.def("film", {"Camera&"}, "Film", [](Interp&, std::vector<Cell>& a) { return mk_obj("Film", obj<CameraObj>(a[0])->film); });
  r.def("save", {"Film&", "str_view"}, "void", [](Interp& in, std::vector<Cell>& a) {  // film.cpp:118, fileio.cpp:55-76
    auto f = obj<FilmObj>(a[0]);
    std::string name = a[1]->s;
    const size_t dot = name.find('.');
    const std::string ext = dot == std::string::npos ? "" : name.substr(dot + 1);
    if (ext == "bmp" || ext == "jpg" || ext == "tga") fail("save: only png output is implemented by this front-end (`" + name + "`)");
    if (ext != "png") {
      in.log("[Warning]Unknown format `" + ext + "` during saving `" + name + "`; assuming png\n");
      name += ".png";
    }
    if (in.flags & PINE_PRL_DRY_RUN) {
      in.log("@save " + name + " " + std::to_string(f->w) + "x" + std::to_string(f->h) + "\n");
      return Value();
    }
    if (f->pixels.empty()) f->pixels.assign(size_t(f->w) * f->h * 4, 0.0f);
    std::vector<uint8_t> rgba(size_t(f->w) * f->h * 4);
    gpu_check(pine_gpu_film_finalize_u8(f->pixels.data(), f->w, f->h, f->tone, rgba.data()), "save");
    if (!png_writer::write_rgba8(name, f->w, f->h, rgba.data())) fail("save: cannot write `" + name + "`");
    return Value();
  });

  // ---- scene (scene.cpp:64-79) ----
  r.def("Scene", {}, "Scene", [](Interp&, std::vector<Cell>&) { return mk_obj("Scene", std::make_shared<SceneObj>()); });
  r.def("=", {"Scene&", "Scene"}, "Scene&", [](Interp&, std::vector<Cell>& a) {
    *a[0] = *a[1];
    return Value();
  });
  auto add_material = [](SceneObj& s, const std::string& name, const MaterialObj& m) {
    int id = -1;
    std::map<const NodeObj*, int> memo;
    auto N = [&](const NodeP3& n) { return instantiate_node(s, n, memo); };
    auto all_const = [&](std::initializer_list<NodeP3> ns) {
      for (auto& n : ns)
        if (n && !n->constant()) return false;
      return true;
    };
    if (m.kind == "Emissive") id = pine_gpu_scene_add_material_emissive(s.h, name.c_str(), m.albedo->v);
    else if (m.kind == "Diffuse") {
      if (m.albedo->constant()) id = pine_gpu_scene_add_material_diffuse(s.h, name.c_str(), m.albedo->v);
      else id = pine_gpu_scene_add_material_diffuse_n(s.h, name.c_str(), N(m.albedo));
    } else if (m.kind == "Uber") {
      if (all_const({m.albedo, m.roughness, m.metallic, m.transmission}))
        id = pine_gpu_scene_add_material_uber(s.h, name.c_str(), m.albedo->v, m.roughness->f, m.metallic->f, m.transmission->f, m.ior);
      else {
        const int a = N(m.albedo), ro = N(m.roughness), me = N(m.metallic), tr = N(m.transmission);
        id = pine_gpu_scene_add_material_uber_n(s.h, name.c_str(), a, ro, me, tr, m.ior);
      }
    } else if (m.kind == "Subsurface")
      id = pine_gpu_scene_add_material_subsurface(s.h, name.c_str(), m.albedo->v, m.roughness->f, m.sigma_s);
    else if (m.kind == "Metal") {
      const int a = N(m.albedo), ro = N(m.roughness);
      id = pine_gpu_scene_add_material_metal(s.h, name.c_str(), a, ro);
    } else if (m.kind == "Glossy" || m.kind == "Glass") {
      const int a = N(m.albedo), ro = N(m.roughness), io = N(m.ior_node);
      id = m.kind == "Glossy" ? pine_gpu_scene_add_material_glossy(s.h, name.c_str(), a, ro, io)
                              : pine_gpu_scene_add_material_glass(s.h, name.c_str(), a, ro, io);
    }
    gpu_check(id, "scene.add(material)");
    return id;
  };
  auto add_shape = [](SceneObj& s, const ShapeObj& g, int mat) {
    const float* p = g.p.data();
    int rc = -1;
    if (g.kind == "Rect") rc = pine_gpu_scene_add_rect(s.h, p, p + 3, p + 6, g.flag ? 1 : 0, mat);
    else if (g.kind == "AABB") rc = pine_gpu_scene_add_aabb(s.h, p, p + 3, mat);
    else if (g.kind == "OBB") rc = pine_gpu_scene_add_obb(s.h, p, p + 3, p + 6, mat);
    else if (g.kind == "Sphere") rc = pine_gpu_scene_add_sphere(s.h, p, p[3], mat);
    else if (g.kind == "Disk") rc = pine_gpu_scene_add_disk(s.h, p, p + 3, p[6], mat);
    else if (g.kind == "Cone") rc = pine_gpu_scene_add_cone(s.h, p, p + 3, p[6], p[7], mat);
    else if (g.kind == "Plane") rc = pine_gpu_scene_add_plane(s.h, p, p + 3, mat);
    else if (g.kind == "Line") rc = pine_gpu_scene_add_line(s.h, p, p + 3, p[6], mat);
    else if (g.kind == "Cylinder") rc = pine_gpu_scene_add_cylinder(s.h, p, p + 3, p[6], mat);
    else if (g.kind == "Triangle") rc = pine_gpu_scene_add_triangle(s.h, p, p + 3, p + 6, mat);
    gpu_check(rc, "scene.add(shape)");
  };
  r.def("add", {"Scene&", "str", "Material"}, "void", [add_material](Interp&, std::vector<Cell>& a) {
    add_material(*obj<SceneObj>(a[0]), a[1]->s, *obj<MaterialObj>(a[2]));
    return Value();
  });
  r.def("add", {"Scene&", "Shape", "Material"}, "void", [add_material, add_shape](Interp&, std::vector<Cell>& a) {
    auto s = obj<SceneObj>(a[0]);
    add_shape(*s, *obj<ShapeObj>(a[1]), add_material(*s, "", *obj<MaterialObj>(a[2])));
    return Value();
  });
  r.def("add", {"Scene&", "Shape", "str"}, "void", [add_shape](Interp&, std::vector<Cell>& a) {
    auto s = obj<SceneObj>(a[0]);
    const int m = pine_gpu_scene_find_material(s->h, a[2]->s.c_str());
    gpu_check(m, "scene.add");
    add_shape(*s, *obj<ShapeObj>(a[1]), m);
    return Value();
  });
  r.def("set", {"Scene&", "Camera"}, "void", [](Interp&, std::vector<Cell>& a) {
    auto s = obj<SceneObj>(a[0]);
    auto c = obj<CameraObj>(a[1]);
    s->camera = cell(*a[1]);
    gpu_check(pine_gpu_scene_set_camera_thinlens(s->h, c->film->w, c->film->h, c->film->tone, c->from, c->to, c->fov, c->len_radius, c->focus), "scene.set");
    return Value();
  });
  // load(scene, "file.glb" [, mat4]) (fileio.cpp:584-589): glTF import -- meshes, materials and, when a node carries
  // one, the camera (gltf_import.hpp)
  auto load = [](std::vector<Cell>& a) {
    auto s = obj<SceneObj>(a[0]);
    pine_gltf::ImportedCamera cam;
    try {
      cam = pine_gltf::import_scene(s->h, a[1]->s, a.size() > 2 ? a[2]->f : nullptr);
    } catch (const pine_gltf::Error& e) {
      fail(std::string("load: ") + e.what());
    }
    if (cam.present) {
      auto f = std::make_shared<FilmObj>();
      f->w = cam.film_w;
      f->h = cam.film_h;
      f->tone = 0;
      auto c = std::make_shared<CameraObj>();
      c->film = f;
      memcpy(c->from, cam.from, 12);
      memcpy(c->to, cam.to, 12);
      c->fov = cam.fov;
      s->camera = cell(retype(mk_obj("ThinLenCamera", c), "Camera"));
      gpu_check(pine_gpu_scene_set_camera_thinlens(s->h, f->w, f->h, f->tone, c->from, c->to, c->fov, c->len_radius, c->focus), "load: camera");
    }
    return Value();
  };
  r.def("load", {"Scene&", "str_view"}, "void", [load](Interp&, std::vector<Cell>& a) { return load(a); });
  r.def("load", {"Scene&", "str_view", "mat4"}, "void", [load](Interp&, std::vector<Cell>& a) { return load(a); });
  r.def("@ma.Scene.camera", {"Scene"}, "Camera", [](Interp&, std::vector<Cell>& a) {
    auto s = obj<SceneObj>(a[0]);
    if (!s->camera) fail("scene has no camera");
    return *s->camera;
  });

  // ---- lights (light.cpp:173-186) ----
  auto light = [](const char* kind) {
    auto l = std::make_shared<LightObj>();
    l->kind = kind;
    return l;
  };
  r.def("PointLight", {"vec3", "vec3"}, "PointLight", [=](Interp&, std::vector<Cell>& a) {
    auto l = light("PointLight");
    memcpy(l->a, a[0]->f, 12);
    memcpy(l->c, a[1]->f, 12);
    return mk_obj("PointLight", l);
  });
  for (int with_cutoff = 0; with_cutoff < 2; with_cutoff++) {
    std::vector<std::string> pt{"vec3", "vec3", "vec3", "f32"};
    if (with_cutoff) pt.push_back("f32");
    r.def("SpotLight", pt, "SpotLight", [=](Interp&, std::vector<Cell>& a) {
      auto l = light("SpotLight");
      memcpy(l->a, a[0]->f, 12);
      memcpy(l->b, a[1]->f, 12);
      memcpy(l->c, a[2]->f, 12);
      l->falloff = a[3]->f[0];
      l->extra = with_cutoff ? a[4]->f[0] : 0.0f;
      return mk_obj("SpotLight", l);
    });
  }
  r.def("DirectionalLight", {"vec3", "vec3"}, "DirectionalLight", [=](Interp&, std::vector<Cell>& a) {
    auto l = light("DirectionalLight");
    memcpy(l->b, a[0]->f, 12);
    memcpy(l->c, a[1]->f, 12);
    return mk_obj("DirectionalLight", l);
  });
  for (const char* k : {"PointLight", "SpotLight", "DirectionalLight"}) r.convert(k, "Light", [](const Value& v) { return retype(v, "Light"); });
  r.def("Sky", {"vec3"}, "Sky", [=](Interp&, std::vector<Cell>& a) {
    auto l = light("Sky");
    memcpy(l->c, a[0]->f, 12);
    return mk_obj("Sky", l);
  });
  r.convert("Sky", "EnvironmentLight", [](const Value& v) { return retype(v, "EnvironmentLight"); });
  r.def("add", {"Scene&", "Light"}, "void", [](Interp&, std::vector<Cell>& a) {  // Scene::add_light scene.cpp:29-34
    auto s = obj<SceneObj>(a[0]);
    auto l = obj<LightObj>(a[1]);
    int rc = -1;
    if (l->kind == "PointLight") rc = pine_gpu_scene_add_light_point(s->h, l->a, l->c);
    else if (l->kind == "SpotLight") rc = pine_gpu_scene_add_light_spot(s->h, l->a, l->b, l->c, l->falloff, l->extra);
    else rc = pine_gpu_scene_add_light_directional(s->h, l->b, l->c);
    gpu_check(rc, "scene.add(light)");
    return Value();
  });
  r.def("set", {"Scene&", "EnvironmentLight"}, "void", [](Interp&, std::vector<Cell>& a) {  // Scene::set_env_light scene.cpp:44-46
    gpu_check(pine_gpu_scene_set_env_sky(obj<SceneObj>(a[0])->h, obj<LightObj>(a[1])->c), "scene.set(environment light)");
    return Value();
  });

  // ---- sampler + integrator (sampler.cpp:189-198, program_context.cpp:76-81, path.cpp:7-41) ----
  r.def("BlueSampler", {"i32"}, "BlueSampler", [](Interp&, std::vector<Cell>& a) {
    if (a[0]->i[0] <= 0) fail("`BlueSampler` should have positive samples per pixel");
    return retype(mk_i32(a[0]->i[0]), "BlueSampler");
  });
  r.convert("BlueSampler", "Sampler", [](const Value& v) { return retype(v, "Sampler"); });
  // SobolSampler(i32) (sampler.cpp:182-188): the value carries the kind in i[1] (PINE_GPU_SAMPLER_SOBOL)
  r.def("SobolSampler", {"i32"}, "SobolSampler", [](Interp&, std::vector<Cell>& a) {
    Value v = retype(mk_i32(a[0]->i[0]), "SobolSampler");
    v.i[1] = PINE_GPU_SAMPLER_SOBOL;
    return v;
  });
  r.convert("SobolSampler", "Sampler", [](const Value& v) { return retype(v, "Sampler"); });
  // HaltonSampler(i32) (sampler.cpp:39-62, :176-181)
  r.def("HaltonSampler", {"i32"}, "HaltonSampler", [](Interp&, std::vector<Cell>& a) {
    if (a[0]->i[0] <= 0) fail("`HaltonSampler` should have positive samples per pixel");
    Value v = retype(mk_i32(a[0]->i[0]), "HaltonSampler");
    v.i[1] = PINE_GPU_SAMPLER_HALTON;
    return v;
  });
  r.convert("HaltonSampler", "Sampler", [](const Value& v) { return retype(v, "Sampler"); });
  for (const char* k : {"BlueSampler", "SobolSampler", "HaltonSampler"})
    r.def("spp", {k}, "i32", [](Interp&, std::vector<Cell>& a) {
      int n = a[0]->i[0];
      if (a[0]->i[1] == PINE_GPU_SAMPLER_BLUE) {  // BlueSobolSampler ctor sampler.cpp:115-121
        n = n > 256 ? 256 : n;
        int p = 1;
        while (p < n) p *= 2;
        n = p;
      }
      return mk_i32(n);
    });
  // names a script may reach for that this build deliberately does not provide: say why
  r.def("UniformSampler", {"i32"}, "Sampler", [](Interp&, std::vector<Cell>&) -> Value {
    fail("`UniformSampler` is not provided: its stream depends on the reference's thread scheduling (per-thread RNG clones, "
         "no per-pixel reseed), so there is no result to reproduce -- use BlueSampler, SobolSampler or HaltonSampler");
  });
  // Accel / LightSampler and the four-argument constructor (program_context.cpp:47-52, 76-78).  The reference registers
  // `BVH()` and `Embree()` but lets only Embree convert to Accel (ctor_variant<EmbreeAccel>), so a script on real pine can
  // write PathIntegrator(Embree(), sampler, UniformLightSampler(), n) and nothing else; here BVH() converts as well (an
  // extension: the explicit way to ask for pine-BVH order).  Embree() selects PINE_GPU_FLAG_ORDER_EMBREE -- EmbreeAccel's own
  // order, which reproduces the EmbreeAccel films of the real reference bit for bit (tests/golden/film_embree_*); BVH()
  // selects pine-BVH order, the parity oracle's.
  r.def("BVH", {}, "BVH", [](Interp&, std::vector<Cell>&) { return retype(mk_i32(1), "BVH"); });
  r.def("Embree", {}, "Embree", [](Interp&, std::vector<Cell>&) { return retype(mk_i32(2), "Embree"); });
  r.convert("Embree", "Accel", [](const Value& v) { return retype(v, "Accel"); });
  r.convert("BVH", "Accel", [](const Value& v) { return retype(v, "Accel"); });
  r.def("UniformLightSampler", {}, "UniformLightSampler", [](Interp&, std::vector<Cell>&) { return retype(mk_i32(0), "UniformLightSampler"); });
  r.convert("UniformLightSampler", "LightSampler", [](const Value& v) { return retype(v, "LightSampler"); });
  r.def("PathIntegrator", {"Accel", "Sampler", "LightSampler", "i32"}, "PathIntegrator", [](Interp&, std::vector<Cell>& a) {
    if (a[3]->i[0] <= 0) fail("`PathIntegrator` expect `max_path_length` to be positive, get " + std::to_string(a[3]->i[0]));
    auto p = std::make_shared<IntegratorObj>();
    p->accel = a[0]->i[0];
    p->spp = a[1]->i[0];
    p->sampler = a[1]->i[1];
    p->depth = a[3]->i[0];
    return mk_obj("PathIntegrator", p);
  });
  r.def("PathIntegrator", {"Sampler", "i32"}, "PathIntegrator", [](Interp&, std::vector<Cell>& a) {
    if (a[1]->i[0] <= 0) fail("`PathIntegrator` expect `max_path_length` to be positive, get " + std::to_string(a[1]->i[0]));
    auto p = std::make_shared<IntegratorObj>();
    p->spp = a[0]->i[0];
    p->sampler = a[0]->i[1];
    p->depth = a[1]->i[0];
    return mk_obj("PathIntegrator", p);
  });
  r.def("render", {"PathIntegrator&", "Scene&"}, "void", [](Interp& in, std::vector<Cell>& a) {
    auto p = obj<IntegratorObj>(a[0]);
    auto s = obj<SceneObj>(a[1]);
    if (!s->camera) fail("PathIntegrator.render: scene has no camera");
    auto f = obj<CameraObj>(s->camera)->film;
    if (in.flags & PINE_PRL_DRY_RUN) {
      in.log(std::string("@render PathIntegrator ") + (p->sampler == PINE_GPU_SAMPLER_SOBOL ? "SobolSampler " : p->sampler == PINE_GPU_SAMPLER_HALTON ? "HaltonSampler " : "BlueSampler ") +
             std::to_string(p->spp) + " max_path_length " + std::to_string(p->depth) + (p->accel == 2 ? " accel Embree" : p->accel == 1 ? " accel BVH" : "") + "\n");
      in.log(describe_scene(s->h));
      in.log("@end\n");
      return Value();
    }
    pine_gpu_render_params prm{};
    prm.spp = p->spp;
    prm.max_path_length = p->depth;
    prm.device = in.device;
    prm.shard_rank = 0;
    prm.shard_world = 1;
    prm.sampler = p->sampler;
    // The accel decides WHICH shapes a query asks, in which ORDER and with which arithmetic for triangles -- which a few shapes
    // can see (the scaled Box(AABB, mat4), bbox.cpp:149-171; Plane's finite bounds, Line, Cylinder) and every mesh does (Embree's
    // own triangle test).  Embree(): the reference's EmbreeAccel, restated from the vendored Embree (PINE_GPU_FLAG_ORDER_EMBREE).
    // BVH(): pine's own BVH.  The two-argument constructor is `PathIntegrator(EmbreeAccel(), sampler, UniformLightSampler(), n)`
    // in the reference (program_context.cpp:79-81), so a script that uses it -- scenes/cbox.pine does -- gets what it gets from
    // real pine.  $PINE_PRL_ACCEL=bvh | embree (pine-mi355x --accel) decides for the two-argument form explicitly.
    int accel = p->accel;
    if (accel == 0) {
      const char* e = getenv("PINE_PRL_ACCEL");
      accel = e && std::string(e) == "bvh" ? 1 : 2;
    }
    if (accel == 2) prm.flags |= PINE_GPU_FLAG_ORDER_EMBREE;
    f->pixels.assign(size_t(f->w) * f->h * 4, 0.0f);
    // $PINE_GPU_DEVICES = "0,1,2,...": render on those devices of the node from this one process (pine-mi355x --devices)
    std::vector<int> devices;
    if (const char* e = getenv("PINE_GPU_DEVICES")) {
      for (const char* q = e; *q;) {
        char* end = nullptr;
        const long d = strtol(q, &end, 10);
        if (end == q) break;
        devices.push_back(int(d));
        q = *end == ',' ? end + 1 : end;
      }
    }
    if (devices.size() > 1)
      gpu_check(pine_gpu_path_render_devices(s->h, &prm, devices.data(), int(devices.size()), f->pixels.data()), "PathIntegrator.render");
    else
      gpu_check(pine_gpu_path_render(s->h, &prm, f->pixels.data()), "PathIntegrator.render");
    in.last_film = f;
    return Value();
  });
  // quick_render(scene, from, to, filename): 640x480, ThinLenCamera fov 0.5, BlueSampler(4), depth 4, save
  // (program_context.cpp:120-124) -- composed from the functions above
  r.def("quick_render", {"Scene&", "vec3", "vec3", "str"}, "void", [](Interp& in, std::vector<Cell>& a) {
    Node at;
    at.kind = Node::Call;
    const int size[2] = {640, 480};
    Cell film = in.call(at, "Film", {cell(mk_veci(2, size))});
    Cell cam = in.call(at, "ThinLenCamera", {film, a[1], a[2], cell(mk_f32(0.5f))});
    in.call(at, "set", {a[0], cam});
    Cell integ = in.call(at, "PathIntegrator", {in.call(at, "BlueSampler", {cell(mk_i32(4))}), cell(mk_i32(4))});
    in.call(at, "render", {integ, a[0]});
    in.call(at, "save", {film, a[3]});
    return Value();
  });
}

static std::string value_text(const Value& v) {
  char buf[64];
  std::string s = v.type + " ";
  auto hexf = [&](float x) {
    snprintf(buf, sizeof buf, "%a", double(x));
    return std::string(buf);
  };
  if (v.type == "i32" || v.type == "BlueSampler" || v.type == "SobolSampler" || v.type == "HaltonSampler") s += std::to_string(v.i[0]);
  else if (v.type == "f32") s += hexf(v.f[0]);
  else if (v.type == "Nodef" || v.type == "Node3f") {
    auto n = std::dynamic_pointer_cast<NodeObj>(v.o);
    if (n && n->kind == "constf") s += "const " + hexf(n->f);
    else if (n && n->kind == "const3") s += "const " + hexf(n->v[0]) + " " + hexf(n->v[1]) + " " + hexf(n->v[2]);
    else if (n) s += n->kind + (n->op ? std::string(" ") + n->op : std::string());
  }
  else if (v.type == "bool") s += v.b ? "true" : "false";
  else if (v.type == "str") s += v.s;
  else if (v.type == "mat4") {
    for (int k = 0; k < 16; k++) s += (k ? " " : "") + hexf(v.f[k]);
  } else if (v.type.rfind("vec", 0) == 0) {
    const int n = v.type[3] - '0';
    const bool is_int = v.type.back() == 'i';
    for (int k = 0; k < n; k++) s += (k ? " " : "") + (is_int ? std::to_string(v.i[k]) : hexf(v.f[k]));
  }
  return s;
}

}  // namespace prl

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_prl_error, g_prl_output;
static thread_local std::shared_ptr<prl::FilmObj> g_prl_film;

extern "C" {

int pine_prl_interpret(const char* source, int flags, int device) {
  g_prl_output.clear();
  g_prl_film.reset();
  if (!source) {
    g_prl_error = "null source";
    return -1;
  }
  try {
    prl::Parser parser(source);
    prl::NodeP program = parser.block(true);
    if (!parser.eof()) parser.error("unexpected `}`");
    prl::Interp in;
    in.flags = flags;
    in.device = device;
    if (const char* ms = getenv("PINE_PRL_MAX_STEPS")) in.max_steps = strtoull(ms, nullptr, 10);
    try {
      for (auto& k : program->kids) in.exec(k);  // top level: no new scope (the program IS the outermost block)
    } catch (prl::BreakSignal&) {
      g_prl_output = in.out;
      g_prl_error = "`break` can only be used in a loop";
      return -1;
    } catch (prl::ContinueSignal&) {
      g_prl_output = in.out;
      g_prl_error = "`continue` can only be used in a loop";
      return -1;
    } catch (prl::Error& e) {
      g_prl_output = in.out;
      g_prl_error = e.msg;
      return -1;
    }
    g_prl_output = in.out;
    g_prl_film = in.last_film;
  } catch (prl::Error& e) {
    g_prl_error = e.msg;
    return -1;
  } catch (std::exception& e) {
    g_prl_error = e.what();
    return -1;
  }
  return 0;
}

int64_t pine_prl_last_film(const float** data, int* width, int* height) {
  if (!g_prl_film || g_prl_film->pixels.empty()) return 0;
  if (data) *data = g_prl_film->pixels.data();
  if (width) *width = g_prl_film->w;
  if (height) *height = g_prl_film->h;
  return int64_t(g_prl_film->pixels.size());
}

const char* pine_prl_output(void) { return g_prl_output.c_str(); }
const char* pine_prl_last_error(void) { return g_prl_error.c_str(); }

int64_t pine_prl_eval(const char* expression, char* out, int64_t capacity) {
  if (!expression) {
    g_prl_error = "null expression";
    return -1;
  }
  try {
    prl::Parser parser(expression);
    prl::NodeP e = parser.expr();
    if (!parser.eof()) parser.error("trailing characters after the expression");
    prl::Interp in;
    in.flags = PINE_PRL_DRY_RUN;
    const std::string text = prl::value_text(*in.eval(e));
    if (out && capacity > 0) {
      const size_t n = std::min<size_t>(size_t(capacity) - 1, text.size());
      memcpy(out, text.data(), n);
      out[n] = 0;
    }
    return int64_t(text.size());
  } catch (prl::Error& e) {
    g_prl_error = e.msg;
    return -1;
  } catch (std::exception& e) {
    g_prl_error = e.what();
    return -1;
  }
}

}  // extern "C"
