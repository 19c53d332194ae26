// mjb_mjcf.cpp — MJCF-subset model compiler behind mjb_model_load_xml / mjb_model_load_xml_string (host C++, no GPU).
//
// The reference never parses XML itself: ModelHandle.from_xml_path hands the file to mj.MjModel.from_xml_path (reference
// mujoco_template/model.py:22-37), i.e. to the compiler inside the third-party `mujoco` C library, which is not available.
// This is the replacement for that call: the MJCF subset used by the reference's example models and its test model
// (examples/*/*.xml, tests/test_mujoco_template.py:40-61) compiled into the table of named arrays (field names follow
// mjModel) that mjb_model_create takes.  MuJoCo's compiler semantics (default classes, fromto, inertia from geoms, autolimits,
// invweight0, contact parameter mixing) are restated from its documented behaviour [MJ-KNOWLEDGE, SURVEY.md §8(c)].
//
// Everything outside the supported subset is REJECTED with a message naming it — a model must never simulate silently with
// different physics: every element / attribute is honoured, ignorable (rendering, bookkeeping) or an error.
#include <expat.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mjbatch.h"

namespace mjcf {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// ------------------------------------------------------------------------------------------------------------------
// a small DOM on expat
// ------------------------------------------------------------------------------------------------------------------
struct Elem {
  std::string tag;
  std::vector<std::pair<std::string, std::string>> attr;       // document order
  std::vector<std::unique_ptr<Elem>> kids;
  const std::string* get(const char* k) const {
    for (auto& a : attr) if (a.first == k) return &a.second;
    return nullptr;
  }
  std::string gets(const char* k, const char* dflt) const { auto p = get(k); return p ? *p : std::string(dflt); }
  bool has(const char* k) const { return get(k) != nullptr; }
};

struct ParseState { std::unique_ptr<Elem> root; std::vector<Elem*> stack; };
static void XMLCALL on_start(void* ud, const XML_Char* name, const XML_Char** atts) {
  auto* st = (ParseState*)ud;
  auto e = std::make_unique<Elem>();
  e->tag = name;
  for (int i = 0; atts[i]; i += 2) e->attr.emplace_back(atts[i], atts[i + 1]);
  Elem* raw = e.get();
  if (st->stack.empty()) st->root = std::move(e); else st->stack.back()->kids.push_back(std::move(e));
  st->stack.push_back(raw);
}
static void XMLCALL on_end(void* ud, const XML_Char*) { ((ParseState*)ud)->stack.pop_back(); }

static std::unique_ptr<Elem> parse_xml(const std::string& text) {
  XML_Parser p = XML_ParserCreate(nullptr);
  if (!p) throw Error("cannot create the XML parser");
  ParseState st;
  XML_SetUserData(p, &st);
  XML_SetElementHandler(p, on_start, on_end);
  if (XML_Parse(p, text.data(), (int)text.size(), 1) == XML_STATUS_ERROR) {
    std::ostringstream m;
    m << "XML parse error: " << XML_ErrorString(XML_GetErrorCode(p)) << ": line " << XML_GetCurrentLineNumber(p) << ", column " << XML_GetCurrentColumnNumber(p);
    XML_ParserFree(p);
    throw Error(m.str());
  }
  XML_ParserFree(p);
  if (!st.root) throw Error("XML parse error: no root element");
  return std::move(st.root);
}

static std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Error("XML file not found: " + path);
  std::ostringstream ss; ss << f.rdbuf();
  return ss.str();
}
static std::string dir_of(const std::string& path) {
  auto k = path.find_last_of('/');
  return k == std::string::npos ? std::string(".") : (k == 0 ? std::string("/") : path.substr(0, k));
}

// MuJoCo rejects a file that is included twice; here a repeat (which also covers self-inclusion and cycles) raises MjcfError
// instead of recursing until the stack overflows.  `seen` holds the paths of every file included so far, canonicalised.
static std::string canonical_path(const std::string& path) {
  char buf[PATH_MAX];
  return realpath(path.c_str(), buf) ? std::string(buf) : path;
}
static void expand_includes(Elem& e, const std::string& base_dir, std::set<std::string>& seen, int depth = 0) {
  if (depth > 64) throw Error("<include> nesting deeper than 64 levels");
  std::vector<std::unique_ptr<Elem>> out;
  for (auto& c : e.kids) {
    if (c->tag == "include") {
      const std::string path = base_dir + "/" + c->gets("file", "");
      std::ifstream probe(path);
      if (!probe) throw Error("include file not found: " + path);
      if (!seen.insert(canonical_path(path)).second) throw Error("file included more than once (or an include cycle): " + path);
      auto sub = parse_xml(read_file(path));
      expand_includes(*sub, dir_of(path), seen, depth + 1);
      for (auto& k : sub->kids) out.push_back(std::move(k));
    } else {
      expand_includes(*c, base_dir, seen, depth + 1);
      out.push_back(std::move(c));
    }
  }
  e.kids = std::move(out);
}

// ------------------------------------------------------------------------------------------------------------------
// numbers and small linear algebra
// ------------------------------------------------------------------------------------------------------------------
typedef std::vector<double> Vec;

static Vec floats(const std::string& text, int n = -1) {
  Vec v;
  const char* p = text.c_str();
  for (;;) {
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++;
    if (!*p) break;
    char* end = nullptr;
    double x = std::strtod(p, &end);
    if (end == p) throw Error("not a number in: '" + text + "'");
    v.push_back(x);
    p = end;
  }
  if (n >= 0 && (int)v.size() != n) throw Error("expected " + std::to_string(n) + " numbers, got " + std::to_string(v.size()) + ": '" + text + "'");
  return v;
}
static double to_double(const std::string& s) { Vec v = floats(s, 1); return v[0]; }
static int to_int(const std::string& s) {
  char* end = nullptr;
  long x = std::strtol(s.c_str(), &end, 10);
  while (end && (*end == ' ')) end++;
  if (!end || *end) throw Error("not an integer: '" + s + "'");
  return (int)x;
}
static bool is_float(const std::string& s) {
  char* end = nullptr;
  std::strtod(s.c_str(), &end);
  while (end && *end == ' ') end++;
  return end != s.c_str() && end && !*end;
}

struct Q { double w, x, y, z; };
static Q qmul(const Q& a, const Q& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
static Q qnorm(const Q& q) { double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z); return {q.w / n, q.x / n, q.y / n, q.z / n}; }
struct M3 { double m[3][3]; };
static M3 q2m(const Q& q) {
  const double w = q.w, x = q.x, y = q.y, z = q.z;
  return {{{w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)},
           {2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)},
           {2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z}}};
}
static Q m2q(const M3& a) {
  const double (*m)[3] = a.m;
  const double tr = m[0][0] + m[1][1] + m[2][2];
  Q q;
  if (tr > 0) { double s = std::sqrt(tr + 1.0) * 2; q = {0.25 * s, (m[2][1] - m[1][2]) / s, (m[0][2] - m[2][0]) / s, (m[1][0] - m[0][1]) / s}; }
  else if (m[0][0] > m[1][1] && m[0][0] > m[2][2]) { double s = std::sqrt(1.0 + m[0][0] - m[1][1] - m[2][2]) * 2; q = {(m[2][1] - m[1][2]) / s, 0.25 * s, (m[0][1] + m[1][0]) / s, (m[0][2] + m[2][0]) / s}; }
  else if (m[1][1] > m[2][2]) { double s = std::sqrt(1.0 + m[1][1] - m[0][0] - m[2][2]) * 2; q = {(m[0][2] - m[2][0]) / s, (m[0][1] + m[1][0]) / s, 0.25 * s, (m[1][2] + m[2][1]) / s}; }
  else { double s = std::sqrt(1.0 + m[2][2] - m[0][0] - m[1][1]) * 2; q = {(m[1][0] - m[0][1]) / s, (m[0][2] + m[2][0]) / s, (m[1][2] + m[2][1]) / s, 0.25 * s}; }
  return qnorm(q);
}
struct V3 { double x, y, z; };
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static double norm(V3 a) { return std::sqrt(dot(a, a)); }
static V3 mv(const M3& m, V3 v) { return {m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z, m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z}; }
static V3 col(const M3& m, int k) { return {m.m[0][k], m.m[1][k], m.m[2][k]}; }
static V3 v3(const Vec& v, int o = 0) { return {v[o], v[o + 1], v[o + 2]}; }

static Q z_to_quat(V3 vec) {                                    // shortest arc rotating (0,0,1) onto vec
  const V3 v = (1.0 / norm(vec)) * vec, z{0, 0, 1};
  V3 axis = cross(z, v);
  const double s = norm(axis), c = dot(z, v);
  if (s < 1e-10) return c > 0 ? Q{1, 0, 0, 0} : Q{0, 1, 0, 0};
  axis = (1.0 / s) * axis;
  const double ang = std::atan2(s, c);
  return {std::cos(ang / 2), axis.x * std::sin(ang / 2), axis.y * std::sin(ang / 2), axis.z * std::sin(ang / 2)};
}

// symmetric 3x3 eigen-decomposition (cyclic Jacobi); eigenvalues DESCENDING, eigenvectors in the columns of V
static void eig3(const M3& A, double ev[3], M3& V) {
  double a[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a[i][j] = A.m[i][j]; V.m[i][j] = i == j ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (std::fabs(a[p][q]) < 1e-300) continue;
      const double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
      for (int k = 0; k < 3; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
      for (int k = 0; k < 3; k++) { const double vkp = V.m[k][p], vkq = V.m[k][q]; V.m[k][p] = c * vkp - s * vkq; V.m[k][q] = s * vkp + c * vkq; }
    }
  }
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int i, int j) { return a[i][i] > a[j][j]; });
  M3 W;
  for (int k = 0; k < 3; k++) { ev[k] = a[idx[k]][idx[k]]; for (int r = 0; r < 3; r++) W.m[r][k] = V.m[r][idx[k]]; }
  V = W;
}

// dense symmetric positive definite inverse (Gauss-Jordan with partial pivoting is enough for nv <= 64)
static std::vector<double> inverse(const std::vector<double>& A, int n) {
  std::vector<double> a(A), inv((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++) if (std::fabs(a[(size_t)r * n + c]) > std::fabs(a[(size_t)piv * n + c])) piv = r;
    if (std::fabs(a[(size_t)piv * n + c]) < 1e-300) throw Error("singular mass matrix at qpos0");
    if (piv != c) for (int k = 0; k < n; k++) { std::swap(a[(size_t)piv * n + k], a[(size_t)c * n + k]); std::swap(inv[(size_t)piv * n + k], inv[(size_t)c * n + k]); }
    const double d = 1.0 / a[(size_t)c * n + c];
    for (int k = 0; k < n; k++) { a[(size_t)c * n + k] *= d; inv[(size_t)c * n + k] *= d; }
    for (int r = 0; r < n; r++) if (r != c) {
      const double f = a[(size_t)r * n + c];
      if (f == 0) continue;
      for (int k = 0; k < n; k++) { a[(size_t)r * n + k] -= f * a[(size_t)c * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k]; }
    }
  }
  return inv;
}

// ------------------------------------------------------------------------------------------------------------------
// enums (values follow MuJoCo's mjtJoint / mjtGeom / mjtObj ordering)
// ------------------------------------------------------------------------------------------------------------------
enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { GEOM_PLANE = 0, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH };
enum { TRN_JOINT = 0, TRN_SITE = 4 };
enum { BIAS_NONE = 0, BIAS_AFFINE = 1 };
enum { INT_EULER = 0, INT_RK4 = 1 };
enum { SENS_JOINTPOS = 0, SENS_GYRO, SENS_ACCELEROMETER, SENS_FRAMEQUAT };
enum { OBJ_BODY = 1, OBJ_JOINT = 3, OBJ_GEOM = 5, OBJ_SITE = 6, OBJ_TENDON = 18, OBJ_ACTUATOR = 19, OBJ_SENSOR = 20, OBJ_KEY = 24 };
static const double MINVAL = 1e-15;
static const double DEFAULT_SOLREF[2] = {0.02, 1.0};
static const double DEFAULT_SOLIMP[5] = {0.9, 0.95, 0.001, 0.5, 2.0};

static int geom_type(const std::string& s) {
  static const std::map<std::string, int> t = {{"plane", GEOM_PLANE}, {"hfield", GEOM_HFIELD}, {"sphere", GEOM_SPHERE}, {"capsule", GEOM_CAPSULE},
                                               {"ellipsoid", GEOM_ELLIPSOID}, {"cylinder", GEOM_CYLINDER}, {"box", GEOM_BOX}, {"mesh", GEOM_MESH}};
  auto it = t.find(s);
  if (it == t.end()) throw Error("unknown geom type '" + s + "'");
  return it->second;
}
static int jnt_type(const std::string& s) {
  static const std::map<std::string, int> t = {{"free", JNT_FREE}, {"ball", JNT_BALL}, {"slide", JNT_SLIDE}, {"hinge", JNT_HINGE}};
  auto it = t.find(s);
  if (it == t.end()) throw Error("unknown joint type '" + s + "'");
  return it->second;
}

// ------------------------------------------------------------------------------------------------------------------
// schema: honoured | ignorable | rejected by name
// ------------------------------------------------------------------------------------------------------------------
typedef std::set<std::string> SS;
struct Rule { SS honoured, ignorable; bool any_ignorable; };
static SS uni(SS a, const SS& b) { a.insert(b.begin(), b.end()); return a; }
static const SS ORIENT = {"quat", "axisangle", "euler", "xyaxes", "zaxis"};
static const SS VISUAL = {"rgba", "material", "group"};

static const std::map<std::string, Rule>& schema() {
  static const std::map<std::string, Rule> s = {
      {"mujoco", {{"model"}, {}, false}},
      {"compiler", {{"angle", "autolimits"}, {"meshdir", "texturedir", "assetdir", "strippath", "discardvisual", "balanceinertia", "boundmass", "boundinertia", "fusestatic", "usethread", "alignfree"}, false}},
      {"option", {{"timestep", "gravity", "integrator", "density", "viscosity", "impratio", "tolerance", "iterations", "cone", "solver", "jacobian"},
                  {"ls_iterations", "ls_tolerance", "noslip_tolerance", "ccd_tolerance", "mpr_tolerance", "apirate"}, false}},
      {"body", {uni({"name", "pos", "childclass"}, ORIENT), {"user"}, false}},
      {"joint", {{"name", "class", "type", "pos", "axis", "range", "limited", "damping", "stiffness", "armature", "margin", "ref", "springref", "solreflimit", "solimplimit"}, {"group", "user"}, false}},
      {"freejoint", {{"name"}, {"group"}, false}},
      {"geom", {uni({"name", "class", "type", "size", "pos", "fromto", "contype", "conaffinity", "condim", "friction", "solref", "solimp", "solmix", "margin", "gap", "priority", "density", "mass"}, ORIENT),
                uni(VISUAL, {"user", "mesh", "fitscale"}), false}},
      {"site", {uni({"name", "class", "pos"}, ORIENT), uni(VISUAL, {"size", "type", "fromto", "user"}), false}},
      {"fixed", {{"name", "class", "limited", "range", "margin", "solreflimit", "solimplimit"}, uni(VISUAL, {"user", "width"}), false}},
      {"tendon/joint", {{"joint", "coef"}, {}, false}},
      {"motor", {{"name", "class", "joint", "site", "gear", "ctrllimited", "ctrlrange", "forcelimited", "forcerange", "group"}, {"user"}, false}},
      {"position", {{"name", "class", "joint", "site", "gear", "ctrllimited", "ctrlrange", "forcelimited", "forcerange", "group", "kp", "kv"}, {"user"}, false}},
      {"general", {{"name", "class", "joint", "site", "gear", "ctrllimited", "ctrlrange", "forcelimited", "forcerange", "group", "dyntype", "gaintype", "biastype", "gainprm", "biasprm"}, {"user"}, false}},
      {"jointpos", {{"name", "joint"}, {"noise", "cutoff", "user"}, false}},
      {"gyro", {{"name", "site"}, {"noise", "cutoff", "user"}, false}},
      {"accelerometer", {{"name", "site"}, {"noise", "cutoff", "user"}, false}},
      {"framequat", {{"name", "objtype", "objname"}, {"noise", "cutoff", "user"}, false}},
      {"exclude", {{"name", "body1", "body2"}, {}, false}},
      {"key", {{"name", "qpos", "qvel", "ctrl", "time"}, {}, false}},
      {"include", {{"file"}, {}, false}},
      {"camera", {{}, {}, true}},
      {"light", {{}, {}, true}},
  };
  return s;
}
// attributes whose NON-DEFAULT presence changes the physics and that the engine does not implement: rejected with a name
static const std::map<std::string, std::map<std::string, std::string>>& rejects() {
  static const std::map<std::string, std::map<std::string, std::string>> r = {
      {"joint", {{"frictionloss", "joint frictionloss"}, {"actuatorfrcrange", "actuatorfrcrange"}, {"actuatorfrclimited", "actuatorfrclimited"},
                 {"solreffriction", "joint friction constraints"}, {"solimpfriction", "joint friction constraints"}, {"springdamper", "springdamper"}}},
      {"geom", {{"fluidshape", "ellipsoid fluid model"}, {"fluidcoef", "ellipsoid fluid model"}}},
      {"body", {{"mocap", "mocap bodies"}, {"gravcomp", "gravity compensation"}}},
      {"fixed", {{"frictionloss", "tendon frictionloss"}, {"stiffness", "tendon springs"}, {"damping", "tendon damping"}, {"springlength", "tendon springs"}}},
      {"option", {{"wind", "wind"}, {"magnetic", ""}, {"o_margin", "contact overrides"}, {"o_solref", "contact overrides"}, {"o_solimp", "contact overrides"},
                  {"o_friction", "contact overrides"}, {"noslip_iterations", "the noslip solver"}, {"actuatorgroupdisable", "actuatorgroupdisable (use opt.disableactuator)"}}},
      {"compiler", {{"coordinate", ""}, {"eulerseq", ""}, {"settotalmass", "settotalmass"}, {"inertiafromgeom", ""}, {"inertiagrouprange", "inertiagrouprange"}}},
  };
  return r;
}
// values at which a rejected attribute is harmless (MuJoCo's defaults / what the shipped models state); "*" = any value
static const std::map<std::pair<std::string, std::string>, SS>& reject_ok() {
  static const std::map<std::pair<std::string, std::string>, SS> r = {
      {{"compiler", "coordinate"}, {"local"}}, {{"compiler", "eulerseq"}, {"xyz"}}, {{"compiler", "inertiafromgeom"}, {"true", "auto"}},
      {{"compiler", "settotalmass"}, {"-1"}}, {{"joint", "frictionloss"}, {"0"}}, {{"fixed", "frictionloss"}, {"0"}}, {{"fixed", "stiffness"}, {"0"}},
      {{"fixed", "damping"}, {"0"}}, {{"body", "mocap"}, {"false"}}, {{"body", "gravcomp"}, {"0"}}, {{"option", "wind"}, {"0 0 0"}},
      {{"option", "noslip_iterations"}, {"0"}}, {{"option", "magnetic"}, {"*"}}};
  return r;
}
static std::string squeeze(const std::string& s) {             // " ".join(s.split())
  std::istringstream in(s); std::string w, out;
  while (in >> w) { if (!out.empty()) out += ' '; out += w; }
  return out;
}
static void check_attrs(const std::string& tag, const Elem& e, const std::string& where) {
  const std::string key = (tag == "joint" && where == "tendon") ? "tendon/joint" : tag;
  auto it = schema().find(key);
  if (it == schema().end()) throw Error("<" + tag + "> in <" + where + "> is outside the supported subset");
  const Rule& r = it->second;
  auto rj = rejects().find(key);
  for (auto& a : e.attr) {
    if (r.honoured.count(a.first) || r.any_ignorable || r.ignorable.count(a.first)) continue;
    if (rj != rejects().end() && rj->second.count(a.first)) {
      auto ok = reject_ok().find({key, a.first});
      bool fine = false;
      if (ok != reject_ok().end()) {
        if (ok->second.count("*") || ok->second.count(squeeze(a.second))) fine = true;
        else if (is_float(a.second)) for (auto& o : ok->second) if (is_float(o) && to_double(o) == to_double(a.second)) fine = true;
      }
      if (fine) continue;
      const std::string what = rj->second.at(a.first).empty() ? a.first + "='" + a.second + "'" : rj->second.at(a.first);
      throw Error("<" + tag + " " + a.first + "='" + a.second + "'>: " + what + " is outside the supported subset");
    }
    throw Error("<" + tag + "> attribute '" + a.first + "' is not recognised by this compiler (supported subset; it would be ignored silently otherwise)");
  }
}
static void validate_defaults(const Elem& e) {
  for (auto& c : e.kids) {
    if (c->tag == "default") validate_defaults(*c);
    else if (c->tag == "camera" || c->tag == "light" || c->tag == "material" || c->tag == "mesh") continue;
    else if (c->tag == "joint" || c->tag == "geom" || c->tag == "site" || c->tag == "motor" || c->tag == "position" || c->tag == "general") check_attrs(c->tag, *c, "default");
    else if (c->tag == "tendon") check_attrs("fixed", *c, "default");
    else throw Error("<default><" + c->tag + "> is outside the supported subset");
  }
}
static void validate_body(const Elem& e) {
  for (auto& c : e.kids) {
    if (c->tag == "body") { check_attrs("body", *c, "worldbody"); validate_body(*c); }
    else if (c->tag == "joint" || c->tag == "freejoint" || c->tag == "geom" || c->tag == "site" || c->tag == "camera" || c->tag == "light") check_attrs(c->tag, *c, "body");
    else if (c->tag == "inertial") throw Error("<inertial> is outside the supported subset (inertia comes from geoms)");
    else throw Error("<" + c->tag + "> inside a body is outside the supported subset");
  }
}
static void validate_schema(const Elem& root) {
  static const SS top = {"compiler", "option", "default", "worldbody", "tendon", "actuator", "sensor", "contact", "keyframe", "include", "asset", "visual", "statistic", "size", "custom"};
  check_attrs("mujoco", root, "");
  for (auto& sp : root.kids) {
    const Elem& sec = *sp;
    if (!top.count(sec.tag)) throw Error("<" + sec.tag + "> is outside the supported subset (e.g. <equality> constraints, <deformable>, <extension> are not implemented)");
    if (sec.tag == "asset" || sec.tag == "visual" || sec.tag == "statistic" || sec.tag == "size" || sec.tag == "custom") continue;
    if (sec.tag == "compiler") {
      check_attrs("compiler", sec, "mujoco");
      for (auto& c : sec.kids) throw Error("<compiler><" + c->tag + "> is outside the supported subset");
    } else if (sec.tag == "option") {
      check_attrs("option", sec, "mujoco");
      for (auto& c : sec.kids) {
        if (c->tag != "flag") throw Error("<option><" + c->tag + "> is outside the supported subset");
        static const SS off_by_default = {"override", "energy", "fwdinv", "invdiscrete", "multiccd", "island"};
        for (auto& a : c->attr) {
          const char* dflt = off_by_default.count(a.first) ? "disable" : "enable";
          if (a.second != dflt) throw Error("<option><flag " + a.first + "='" + a.second + "'>: option flags are outside the supported subset (all stay at MuJoCo's defaults)");
        }
      }
    } else if (sec.tag == "default") validate_defaults(sec);
    else if (sec.tag == "worldbody") validate_body(sec);
    else if (sec.tag == "tendon") {
      for (auto& t : sec.kids) {
        if (t->tag != "fixed") throw Error("only fixed tendons are inside the supported subset");
        check_attrs("fixed", *t, "tendon");
        for (auto& w : t->kids) check_attrs(w->tag, *w, "tendon");
      }
    } else if (sec.tag == "actuator") {
      for (auto& e : sec.kids) {
        if (e->tag != "motor" && e->tag != "position" && e->tag != "general") throw Error("actuator <" + e->tag + "> is outside the supported subset");
        check_attrs(e->tag, *e, "actuator");
      }
    } else if (sec.tag == "sensor") {
      for (auto& e : sec.kids) {
        if (e->tag != "jointpos" && e->tag != "gyro" && e->tag != "accelerometer" && e->tag != "framequat") throw Error("sensor <" + e->tag + "> is outside the supported subset");
        check_attrs(e->tag, *e, "sensor");
      }
    } else if (sec.tag == "contact") {
      for (auto& e : sec.kids) {
        if (e->tag != "exclude") throw Error("<contact><pair> is outside the supported subset");
        check_attrs("exclude", *e, "contact");
      }
    } else if (sec.tag == "keyframe") {
      for (auto& e : sec.kids) {
        if (e->tag != "key") throw Error("<keyframe><" + e->tag + "> is outside the supported subset");
        check_attrs("key", *e, "keyframe");
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// default classes
// ------------------------------------------------------------------------------------------------------------------
typedef std::vector<std::pair<std::string, std::string>> Attrs;   // ordered, later entries override earlier ones
static void attrs_update(Attrs& into, const Attrs& from) {
  for (auto& kv : from) {
    bool found = false;
    for (auto& x : into) if (x.first == kv.first) { x.second = kv.second; found = true; break; }
    if (!found) into.push_back(kv);
  }
}
struct AttrMap {
  Attrs a;
  const std::string* get(const char* k) const { for (auto& x : a) if (x.first == k) return &x.second; return nullptr; }
  bool has(const char* k) const { return get(k) != nullptr; }
  std::string gets(const char* k, const char* d) const { auto p = get(k); return p ? *p : std::string(d); }
  double getd(const char* k, double d) const { auto p = get(k); return p ? to_double(*p) : d; }
  int geti(const char* k, int d) const { auto p = get(k); return p ? to_int(*p) : d; }
};
static bool is_act_tag(const std::string& t) { return t == "general" || t == "motor" || t == "position" || t == "velocity"; }

struct Defaults {
  std::map<std::string, std::map<std::string, Attrs>> classes;    // class -> element kind -> attributes
  Defaults() { classes["main"]; }
  void add(const Elem& e, const std::string* parent) {
    std::string cname;
    if (auto c = e.get("class")) cname = *c;
    else if (!parent) cname = "main";
    else throw Error("nested <default> needs a class attribute");
    std::map<std::string, Attrs> base;
    if (parent) base = classes.at(*parent);
    if (cname == "main" && !parent) base = classes["main"];
    for (auto& c : e.kids) {
      if (c->tag == "default") continue;
      const std::string tag = is_act_tag(c->tag) ? "actuator" : c->tag;
      attrs_update(base[tag], c->attr);
    }
    classes[cname] = base;
    for (auto& c : e.kids) if (c->tag == "default") add(*c, &cname);
  }
  AttrMap resolve(const std::string& tag, const Elem& e, const std::string* childclass) const {
    std::string cname = "main";
    if (auto c = e.get("class")) { if (!c->empty()) cname = *c; else if (childclass) cname = *childclass; }
    else if (childclass) cname = *childclass;
    auto it = classes.find(cname);
    if (it == classes.end()) throw Error("unknown default class '" + cname + "'");
    const std::string key = is_act_tag(tag) ? "actuator" : tag;
    AttrMap out;
    auto kt = it->second.find(key);
    if (kt != it->second.end()) out.a = kt->second;
    attrs_update(out.a, e.attr);
    return out;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// output table
// ------------------------------------------------------------------------------------------------------------------
struct Field { std::string name; int dtype; std::vector<double> d; std::vector<int> i; std::string b; };
struct TableOut {
  std::vector<Field> f;
  std::map<std::string, std::vector<int>> shapes;                 // of the model arrays (the Python front's meta_json)
  void D(const std::string& n, const std::vector<double>& v, std::vector<int> shape, bool array = true) { f.push_back({n, 0, v, {}, {}}); if (array) shapes[n] = shape; }
  void I(const std::string& n, const std::vector<int>& v, std::vector<int> shape, bool array = true) { f.push_back({n, 1, {}, v, {}}); if (array) shapes[n] = shape; }
  void Bytes(const std::string& n, const std::string& v) { f.push_back({n, 2, {}, {}, v}); }
};

// ------------------------------------------------------------------------------------------------------------------
// the compiler
// ------------------------------------------------------------------------------------------------------------------
struct Body { std::string name; int parent; V3 pos; Q quat; std::vector<int> jnt, geoms; bool has_cc; std::string cc; };
struct Joint { std::string name; int type, body; V3 pos, axis; bool limited; double range[2], damping, stiffness, armature, margin, solref[2], solimp[5], ref, springref; };
struct Geom { std::string name; int type, body; V3 pos; Q quat; double size[3]; int contype, conaffinity, condim; double friction[3], solref[2], solimp[5], solmix, margin, gap; int priority; double density; bool has_mass; double mass; };
struct Site { std::string name; int body; V3 pos; Q quat; };

struct Kin { std::vector<V3> xpos, xipos, xanchor, xaxis; std::vector<Q> xquat; std::vector<M3> xmat, ximat; };

struct Compiler {
  std::unique_ptr<Elem> root;
  Defaults defaults;
  double angle_scale = M_PI / 180.0;
  bool autolimits = true;
  std::string model_name;
  // options
  double timestep = 0.002, gravity[3] = {0, 0, -9.81}, density = 0, viscosity = 0, impratio = 1, tolerance = 1e-8, meaninertia = 1;
  int integrator = INT_EULER, iterations = 100, ls_iterations = 50, disableactuator = 0;
  std::vector<Body> bodies;
  std::vector<Joint> joints;
  std::vector<Geom> geoms;
  std::vector<Site> sites;
  // derived
  int nq = 0, nv = 0, nu = 0, ntendon = 0, nwrap = 0, nsensor = 0, nsensordata = 0, nkey = 0, npair = 0, nexclude = 0;
  std::vector<int> jnt_qposadr, jnt_dofadr, body_jntadr, body_jntnum, body_dofadr, body_dofnum, dof_body, dof_jnt, dof_parent, weld, rootid, depth;
  std::vector<double> qpos0, qpos_spring, dof_arm, dof_damp, body_mass, body_subtreemass;
  std::vector<V3> body_ipos, body_inertia;
  std::vector<Q> body_iquat;
  std::vector<int> tendon_adr, tendon_num, wrap_obj;
  std::vector<double> wrap_prm;
  std::map<int, std::vector<std::string>> names;
  TableOut T;

  Compiler(std::unique_ptr<Elem> r, const std::string& base_dir) : root(std::move(r)) {
    std::set<std::string> included;
    expand_includes(*root, base_dir, included);
    model_name = root->gets("model", "");
  }

  int name2id(int obj, const std::string& n) const {
    auto it = names.find(obj);
    if (it == names.end()) return -1;
    for (size_t i = 0; i < it->second.size(); i++) if (it->second[i] == n) return (int)i;
    return -1;
  }

  Q orientation(const AttrMap& a) const {
    if (auto p = a.get("quat")) { Vec v = floats(*p, 4); return qnorm({v[0], v[1], v[2], v[3]}); }
    if (auto p = a.get("axisangle")) {
      Vec v = floats(*p, 4);
      const double ang = v[3] * angle_scale, n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), s = std::sin(ang / 2);
      return {std::cos(ang / 2), v[0] / n * s, v[1] / n * s, v[2] / n * s};
    }
    if (auto p = a.get("euler")) {
      Vec e = floats(*p, 3);
      Q q{1, 0, 0, 0};
      for (int i = 0; i < 3; i++) {                             // default eulerseq "xyz", intrinsic
        const double ang = e[i] * angle_scale, s = std::sin(ang / 2);
        q = qmul(q, {std::cos(ang / 2), i == 0 ? s : 0.0, i == 1 ? s : 0.0, i == 2 ? s : 0.0});
      }
      return q;
    }
    if (auto p = a.get("xyaxes")) {
      Vec v = floats(*p, 6);
      V3 x = (1.0 / norm(v3(v))) * v3(v), y0 = v3(v, 3);
      V3 y = y0 - dot(y0, x) * x;
      y = (1.0 / norm(y)) * y;
      V3 z = cross(x, y);
      M3 m = {{{x.x, y.x, z.x}, {x.y, y.y, z.y}, {x.z, y.z, z.z}}};
      return m2q(m);
    }
    if (auto p = a.get("zaxis")) return z_to_quat(v3(floats(*p, 3)));
    return {1, 0, 0, 0};
  }

  bool limited(const AttrMap& a, const char* key, const char* rng_key) const {
    const std::string val = a.gets(key, "auto");
    if (val == "true") return true;
    if (val == "false") return false;
    return autolimits && a.has(rng_key);
  }

  static void fill(double* dst, int n, const Vec& v) { for (int i = 0; i < n && i < (int)v.size(); i++) dst[i] = v[i]; }

  void joint(const Elem& e, int body_id, const std::string* cc) {
    if (body_id == 0) throw Error("joints cannot be attached to the world body");
    AttrMap a;
    int type;
    if (e.tag == "freejoint") { a.a = e.attr; type = JNT_FREE; }
    else { a = defaults.resolve("joint", e, cc); type = jnt_type(a.gets("type", "hinge")); }
    if (type == JNT_BALL) throw Error("ball joints are outside the supported subset");
    Joint j{};
    j.name = a.gets("name", ""); j.type = type; j.body = body_id;
    V3 ax = v3(floats(a.gets("axis", "0 0 1"), 3));
    j.axis = (1.0 / norm(ax)) * ax;
    Vec rng = floats(a.gets("range", "0 0"), 2);
    const double scale = type == JNT_HINGE ? angle_scale : 1.0;
    std::copy(DEFAULT_SOLIMP, DEFAULT_SOLIMP + 5, j.solimp);
    std::copy(DEFAULT_SOLREF, DEFAULT_SOLREF + 2, j.solref);
    if (auto p = a.get("solimplimit")) fill(j.solimp, 5, floats(*p));
    if (auto p = a.get("solreflimit")) fill(j.solref, 2, floats(*p));
    j.limited = limited(a, "limited", "range") && (type == JNT_HINGE || type == JNT_SLIDE);
    if (type != JNT_FREE) { j.damping = a.getd("damping", 0); j.stiffness = a.getd("stiffness", 0); j.armature = a.getd("armature", 0); }
    j.pos = v3(floats(a.gets("pos", "0 0 0"), 3));
    j.range[0] = rng[0] * scale; j.range[1] = rng[1] * scale;
    j.margin = a.getd("margin", 0);
    j.ref = a.getd("ref", 0) * scale; j.springref = a.getd("springref", 0) * scale;
    joints.push_back(j);
    bodies[body_id].jnt.push_back((int)joints.size() - 1);
  }

  void geom(const Elem& e, int body_id, const std::string* cc) {
    AttrMap a = defaults.resolve("geom", e, cc);
    Geom g{};
    g.type = geom_type(a.gets("type", "sphere"));
    if (auto p = a.get("size")) fill(g.size, 3, floats(*p));
    g.pos = v3(floats(a.gets("pos", "0 0 0"), 3));
    g.quat = orientation(a);
    if (auto p = a.get("fromto")) {
      if (g.type != GEOM_CAPSULE && g.type != GEOM_CYLINDER && g.type != GEOM_BOX && g.type != GEOM_ELLIPSOID) throw Error("fromto requires capsule/cylinder/box/ellipsoid");
      Vec ft = floats(*p, 6);
      V3 from = v3(ft), to = v3(ft, 3), vec = from - to;
      const double length = norm(vec);
      g.pos = 0.5 * (from + to);
      g.quat = z_to_quat(vec);
      if (g.type == GEOM_CAPSULE || g.type == GEOM_CYLINDER) g.size[1] = length / 2; else g.size[2] = length / 2;
    }
    g.name = a.gets("name", "");
    if (g.type == GEOM_MESH) {
      g.size[0] = g.size[1] = g.size[2] = 0;
      // meshes are accepted as VISUALS only: a mesh that should carry mass would need its volume (the .obj is never read)
      if (!(a.has("mass") && a.getd("mass", 1) == 0.0)) throw Error("mesh geom '" + g.name + "' needs mass=\"0\": mesh inertia is outside the supported subset");
    }
    std::copy(DEFAULT_SOLREF, DEFAULT_SOLREF + 2, g.solref);
    std::copy(DEFAULT_SOLIMP, DEFAULT_SOLIMP + 5, g.solimp);
    if (auto p = a.get("solref")) fill(g.solref, 2, floats(*p));
    if (auto p = a.get("solimp")) fill(g.solimp, 5, floats(*p));
    g.friction[0] = 1.0; g.friction[1] = 0.005; g.friction[2] = 0.0001;
    if (auto p = a.get("friction")) fill(g.friction, 3, floats(*p));
    g.body = body_id;
    g.contype = a.geti("contype", 1); g.conaffinity = a.geti("conaffinity", 1); g.condim = a.geti("condim", 3);
    g.solmix = a.getd("solmix", 1); g.margin = a.getd("margin", 0); g.gap = a.getd("gap", 0);
    g.priority = a.geti("priority", 0); g.density = a.getd("density", 1000);
    g.has_mass = a.has("mass"); g.mass = g.has_mass ? a.getd("mass", 0) : 0.0;
    if (g.condim != 1 && g.condim != 3) throw Error("only condim 1 and 3 are inside the supported subset");
    geoms.push_back(g);
    bodies[body_id].geoms.push_back((int)geoms.size() - 1);
  }

  void site(const Elem& e, int body_id, const std::string* cc) {
    AttrMap a = defaults.resolve("site", e, cc);
    sites.push_back({a.gets("name", ""), body_id, v3(floats(a.gets("pos", "0 0 0"), 3)), orientation(a)});
  }

  void body_children(const Elem& e, int body_id, const std::string* cc) {
    for (auto& cp : e.kids) {
      const Elem& c = *cp;
      if (c.tag == "body") {
        Body b{};
        const std::string* ncc = cc;
        if (auto p = c.get("childclass")) { b.has_cc = true; b.cc = *p; }
        else if (cc) { b.has_cc = true; b.cc = *cc; }
        AttrMap a; a.a = c.attr;
        b.name = a.gets("name", ""); b.parent = body_id; b.pos = v3(floats(a.gets("pos", "0 0 0"), 3)); b.quat = orientation(a);
        bodies.push_back(b);
        const int id = (int)bodies.size() - 1;
        std::string keep = bodies[id].cc;                       // bodies may reallocate while recursing: copy the class name
        ncc = bodies[id].has_cc ? &keep : nullptr;
        body_children(c, id, ncc);
      } else if (c.tag == "joint" || c.tag == "freejoint") joint(c, body_id, cc);
      else if (c.tag == "geom") geom(c, body_id, cc);
      else if (c.tag == "site") site(c, body_id, cc);
      else if (c.tag == "inertial") throw Error("<inertial> is outside the supported subset (inertia comes from geoms)");
      // camera / light: not part of the physics path
    }
  }

  // volume and unit-density diagonal inertia (about the geom centre, geom frame)
  static double volume_inertia(int type, const double* size, double inr[3]) {
    inr[0] = inr[1] = inr[2] = 0;
    if (type == GEOM_SPHERE) { const double r = size[0], vol = 4.0 / 3.0 * M_PI * r * r * r; inr[0] = inr[1] = inr[2] = 0.4 * vol * r * r; return vol; }
    if (type == GEOM_CAPSULE) {
      const double r = size[0], h = size[1], vc = M_PI * r * r * 2 * h, vs = 4.0 / 3.0 * M_PI * r * r * r;
      const double izz = vc * r * r / 2 + vs * 0.4 * r * r, ixx = vc * (3 * r * r + 4 * h * h) / 12 + vs * (0.4 * r * r + h * h + 0.75 * h * r);
      inr[0] = inr[1] = ixx; inr[2] = izz;
      return vc + vs;
    }
    if (type == GEOM_CYLINDER) { const double r = size[0], h = size[1], vol = M_PI * r * r * 2 * h; inr[0] = inr[1] = vol * (3 * r * r + 4 * h * h) / 12; inr[2] = vol * r * r / 2; return vol; }
    if (type == GEOM_ELLIPSOID) { const double a = size[0], b = size[1], c = size[2], vol = 4.0 / 3.0 * M_PI * a * b * c; inr[0] = vol / 5 * (b * b + c * c); inr[1] = vol / 5 * (a * a + c * c); inr[2] = vol / 5 * (a * a + b * b); return vol; }
    if (type == GEOM_BOX) { const double a = size[0], b = size[1], c = size[2], vol = 8 * a * b * c; inr[0] = vol / 3 * (b * b + c * c); inr[1] = vol / 3 * (a * a + c * c); inr[2] = vol / 3 * (a * a + b * b); return vol; }
    return 0.0;                                                 // plane, mesh (mass-0 visual only), hfield
  }

  void finalize_tree() {
    const int nb = (int)bodies.size(), nj = (int)joints.size();
    // joints / dofs / qpos
    for (int jid = 0; jid < nj; jid++) {
      const Joint& j = joints[jid];
      jnt_qposadr.push_back(nq); jnt_dofadr.push_back(nv);
      int ndof;
      if (j.type == JNT_FREE) {
        const Body& b = bodies[j.body];
        for (int rep = 0; rep < 2; rep++) {
          std::vector<double>& dst = rep == 0 ? qpos0 : qpos_spring;
          dst.insert(dst.end(), {b.pos.x, b.pos.y, b.pos.z, b.quat.w, b.quat.x, b.quat.y, b.quat.z});
        }
        nq += 7; ndof = 6;
      } else { qpos0.push_back(j.ref); qpos_spring.push_back(j.springref); nq += 1; ndof = 1; }
      for (int k = 0; k < ndof; k++) { dof_body.push_back(j.body); dof_jnt.push_back(jid); dof_arm.push_back(j.armature); dof_damp.push_back(j.damping); }
      nv += ndof;
    }
    for (auto& j : joints)
      if (j.type == JNT_FREE && (bodies[j.body].parent != 0 || bodies[j.body].jnt.size() != 1)) throw Error("free joint must be the only joint of a child of the world body");
    body_jntadr.assign(nb, -1); body_jntnum.assign(nb, 0); body_dofadr.assign(nb, -1); body_dofnum.assign(nb, 0);
    for (int b = 0; b < nb; b++) if (!bodies[b].jnt.empty()) {
      body_jntadr[b] = bodies[b].jnt[0]; body_jntnum[b] = (int)bodies[b].jnt.size(); body_dofadr[b] = jnt_dofadr[bodies[b].jnt[0]];
      for (int j : bodies[b].jnt) body_dofnum[b] += joints[j].type == JNT_FREE ? 6 : 1;
    }
    dof_parent.assign(nv, -1);
    for (int d = 0; d < nv; d++) {
      const int bid = dof_body[d];
      if (d > body_dofadr[bid]) dof_parent[d] = d - 1;
      else {
        int p = bodies[bid].parent;
        while (p > 0 && body_dofnum[p] == 0) p = bodies[p].parent;
        if (p > 0) dof_parent[d] = body_dofadr[p] + body_dofnum[p] - 1;
      }
    }
    weld.assign(nb, 0); rootid.assign(nb, 0); depth.assign(nb, 0);
    for (int b = 1; b < nb; b++) {
      const int p = bodies[b].parent;
      weld[b] = body_jntnum[b] > 0 ? b : weld[p];
      rootid[b] = p == 0 ? b : rootid[p];
      depth[b] = depth[p] + 1;
    }
    // body inertial properties from geoms
    body_mass.assign(nb, 0.0); body_ipos.assign(nb, V3{0, 0, 0}); body_iquat.assign(nb, Q{1, 0, 0, 0}); body_inertia.assign(nb, V3{0, 0, 0});
    for (int bid = 1; bid < nb; bid++) {
      struct Part { double mass; V3 pos; M3 R; double inr[3]; };
      std::vector<Part> parts;
      for (int gid : bodies[bid].geoms) {
        const Geom& g = geoms[gid];
        double inr[3];
        const double vol = volume_inertia(g.type, g.size, inr);
        if (vol <= 0) continue;
        const double mass = g.has_mass ? g.mass : g.density * vol;
        if (mass <= 0) continue;
        Part p{mass, g.pos, q2m(g.quat), {inr[0] * (mass / vol), inr[1] * (mass / vol), inr[2] * (mass / vol)}};
        parts.push_back(p);
      }
      double mtot = 0;
      for (auto& p : parts) mtot += p.mass;
      if (mtot <= 0) {
        if (body_jntnum[bid] > 0) throw Error("moving body '" + bodies[bid].name + "' has zero mass");
        continue;
      }
      V3 com{0, 0, 0};
      for (auto& p : parts) com = com + p.mass * p.pos;
      com = (1.0 / mtot) * com;
      M3 I{};
      for (auto& p : parts) {
        const V3 d = p.pos - com;
        const double dd = dot(d, d), dv[3] = {d.x, d.y, d.z};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
          double v = 0;
          for (int k = 0; k < 3; k++) v += p.R.m[r][k] * p.inr[k] * p.R.m[c][k];
          I.m[r][c] += v + p.mass * ((r == c ? dd : 0.0) - dv[r] * dv[c]);
        }
      }
      double amax = 0;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) amax = std::max(amax, std::fabs(I.m[r][c]));
      bool diagonal = true;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) if (r != c && std::fabs(I.m[r][c]) > 1e-14 * std::max(1.0, amax)) diagonal = false;
      double ev[3]; M3 V;
      if (diagonal) { ev[0] = I.m[0][0]; ev[1] = I.m[1][1]; ev[2] = I.m[2][2]; V = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }   // keep the body axes, like MuJoCo for aligned geoms
      else eig3(I, ev, V);
      const V3 c0 = col(V, 0), c1 = col(V, 1), c2 = col(V, 2);
      if (dot(c0, cross(c1, c2)) < 0) for (int r = 0; r < 3; r++) V.m[r][2] = -V.m[r][2];
      body_mass[bid] = mtot; body_ipos[bid] = com; body_iquat[bid] = m2q(V); body_inertia[bid] = {ev[0], ev[1], ev[2]};
    }
    body_subtreemass = body_mass;
    for (int b = nb - 1; b > 0; b--) body_subtreemass[bodies[b].parent] += body_subtreemass[b];
    for (auto& b : bodies) names[OBJ_BODY].push_back(b.name);
    names[OBJ_JOINT]; names[OBJ_GEOM]; names[OBJ_SITE];
    for (auto& j : joints) names[OBJ_JOINT].push_back(j.name);
    for (auto& g : geoms) names[OBJ_GEOM].push_back(g.name);
    for (auto& s : sites) names[OBJ_SITE].push_back(s.name);
    // ---- arrays, in the order of the Python front's table ----
    auto I1 = [&](const char* n, const std::vector<int>& v) { T.I(n, v, {(int)v.size()}); };
    auto D1 = [&](const char* n, const std::vector<double>& v) { T.D(n, v, {(int)v.size()}); };
    std::vector<int> iv; std::vector<double> dv;
    iv.clear(); for (auto& b : bodies) iv.push_back(b.parent); I1("body_parentid", iv);
    dv.clear(); for (auto& b : bodies) dv.insert(dv.end(), {b.pos.x, b.pos.y, b.pos.z}); T.D("body_pos", dv, {nb, 3});
    dv.clear(); for (auto& b : bodies) dv.insert(dv.end(), {b.quat.w, b.quat.x, b.quat.y, b.quat.z}); T.D("body_quat", dv, {nb, 4});
    iv.clear(); for (auto& j : joints) iv.push_back(j.type); I1("jnt_type", iv);
    I1("jnt_qposadr", jnt_qposadr); I1("jnt_dofadr", jnt_dofadr);
    iv.clear(); for (auto& j : joints) iv.push_back(j.body); I1("jnt_bodyid", iv);
    dv.clear(); for (auto& j : joints) dv.insert(dv.end(), {j.pos.x, j.pos.y, j.pos.z}); T.D("jnt_pos", dv, {nj, 3});
    dv.clear(); for (auto& j : joints) dv.insert(dv.end(), {j.axis.x, j.axis.y, j.axis.z}); T.D("jnt_axis", dv, {nj, 3});
    iv.clear(); for (auto& j : joints) iv.push_back(j.limited ? 1 : 0); I1("jnt_limited", iv);
    dv.clear(); for (auto& j : joints) dv.insert(dv.end(), {j.range[0], j.range[1]}); T.D("jnt_range", dv, {nj, 2});
    dv.clear(); for (auto& j : joints) dv.push_back(j.stiffness); D1("jnt_stiffness", dv);
    dv.clear(); for (auto& j : joints) dv.push_back(j.margin); D1("jnt_margin", dv);
    dv.clear(); for (auto& j : joints) dv.insert(dv.end(), j.solref, j.solref + 2); T.D("jnt_solref", dv, {nj, 2});
    dv.clear(); for (auto& j : joints) dv.insert(dv.end(), j.solimp, j.solimp + 5); T.D("jnt_solimp", dv, {nj, 5});
    D1("qpos0", qpos0); D1("qpos_spring", qpos_spring);
    I1("dof_bodyid", dof_body); I1("dof_jntid", dof_jnt); D1("dof_armature", dof_arm); D1("dof_damping", dof_damp);
    I1("body_jntadr", body_jntadr); I1("body_jntnum", body_jntnum); I1("body_dofadr", body_dofadr); I1("body_dofnum", body_dofnum);
    I1("dof_parentid", dof_parent); I1("body_weldid", weld); I1("body_rootid", rootid); I1("body_depth", depth);
    const int ng = (int)geoms.size(), ns = (int)sites.size();
    iv.clear(); for (auto& g : geoms) iv.push_back(g.type); I1("geom_type", iv);
    iv.clear(); for (auto& g : geoms) iv.push_back(g.body); I1("geom_bodyid", iv);
    dv.clear(); for (auto& g : geoms) dv.insert(dv.end(), {g.pos.x, g.pos.y, g.pos.z}); T.D("geom_pos", dv, {ng, 3});
    dv.clear(); for (auto& g : geoms) dv.insert(dv.end(), {g.quat.w, g.quat.x, g.quat.y, g.quat.z}); T.D("geom_quat", dv, {ng, 4});
    dv.clear(); for (auto& g : geoms) dv.insert(dv.end(), g.size, g.size + 3); T.D("geom_size", dv, {ng, 3});
    iv.clear(); for (auto& g : geoms) iv.push_back(g.contype); I1("geom_contype", iv);
    iv.clear(); for (auto& g : geoms) iv.push_back(g.conaffinity); I1("geom_conaffinity", iv);
    iv.clear(); for (auto& g : geoms) iv.push_back(g.condim); I1("geom_condim", iv);
    dv.clear(); for (auto& g : geoms) dv.insert(dv.end(), g.friction, g.friction + 3); T.D("geom_friction", dv, {ng, 3});
    dv.clear(); for (auto& g : geoms) dv.insert(dv.end(), g.solref, g.solref + 2); T.D("geom_solref", dv, {ng, 2});
    dv.clear(); for (auto& g : geoms) dv.insert(dv.end(), g.solimp, g.solimp + 5); T.D("geom_solimp", dv, {ng, 5});
    dv.clear(); for (auto& g : geoms) dv.push_back(g.margin); D1("geom_margin", dv);
    dv.clear(); for (auto& g : geoms) dv.push_back(g.gap); D1("geom_gap", dv);
    iv.clear(); for (auto& s : sites) iv.push_back(s.body); I1("site_bodyid", iv);
    dv.clear(); for (auto& s : sites) dv.insert(dv.end(), {s.pos.x, s.pos.y, s.pos.z}); T.D("site_pos", dv, {ns, 3});
    dv.clear(); for (auto& s : sites) dv.insert(dv.end(), {s.quat.w, s.quat.x, s.quat.y, s.quat.z}); T.D("site_quat", dv, {ns, 4});
    D1("body_mass", body_mass);
    dv.clear(); for (auto& p : body_ipos) dv.insert(dv.end(), {p.x, p.y, p.z}); T.D("body_ipos", dv, {nb, 3});
    dv.clear(); for (auto& q : body_iquat) dv.insert(dv.end(), {q.w, q.x, q.y, q.z}); T.D("body_iquat", dv, {nb, 4});
    dv.clear(); for (auto& p : body_inertia) dv.insert(dv.end(), {p.x, p.y, p.z}); T.D("body_inertia", dv, {nb, 3});
    D1("body_subtreemass", body_subtreemass);
  }

  void tendons() {
    struct Ten { std::string name; int adr, num; bool limited; double range[2], margin, solref[2], solimp[5]; };
    std::vector<Ten> tend;
    for (auto& sp : root->kids) if (sp->tag == "tendon") for (auto& tp : sp->kids) {
      const Elem& t = *tp;
      if (t.tag != "fixed") throw Error("only fixed tendons are inside the supported subset");
      AttrMap a = defaults.resolve("tendon", t, nullptr);
      Ten x{};
      x.adr = (int)wrap_obj.size();
      for (auto& wp : t.kids) if (wp->tag == "joint") {
        const int jid = name2id(OBJ_JOINT, wp->gets("joint", ""));
        if (jid < 0) throw Error("tendon joint not found: " + wp->gets("joint", ""));
        if (joints[jid].type != JNT_HINGE && joints[jid].type != JNT_SLIDE) throw Error("fixed tendon joints must be hinge or slide");
        wrap_obj.push_back(jid);
        wrap_prm.push_back(wp->has("coef") ? to_double(*wp->get("coef")) : 1.0);
      }
      std::copy(DEFAULT_SOLREF, DEFAULT_SOLREF + 2, x.solref);
      std::copy(DEFAULT_SOLIMP, DEFAULT_SOLIMP + 5, x.solimp);
      if (auto p = a.get("solreflimit")) fill(x.solref, 2, floats(*p));
      if (auto p = a.get("solimplimit")) fill(x.solimp, 5, floats(*p));
      x.name = a.gets("name", ""); x.num = (int)wrap_obj.size() - x.adr; x.limited = limited(a, "limited", "range");
      Vec r = floats(a.gets("range", "0 0"), 2);
      x.range[0] = r[0]; x.range[1] = r[1]; x.margin = a.getd("margin", 0);
      tend.push_back(x);
    }
    ntendon = (int)tend.size(); nwrap = (int)wrap_obj.size();
    std::vector<int> iv; std::vector<double> dv;
    for (auto& t : tend) { tendon_adr.push_back(t.adr); tendon_num.push_back(t.num); }
    T.I("tendon_adr", tendon_adr, {ntendon}); T.I("tendon_num", tendon_num, {ntendon});
    iv.clear(); for (auto& t : tend) iv.push_back(t.limited ? 1 : 0); T.I("tendon_limited", iv, {ntendon});
    dv.clear(); for (auto& t : tend) dv.insert(dv.end(), t.range, t.range + 2); T.D("tendon_range", dv, {ntendon, 2});
    dv.clear(); for (auto& t : tend) dv.push_back(t.margin); T.D("tendon_margin", dv, {ntendon});
    dv.clear(); for (auto& t : tend) dv.insert(dv.end(), t.solref, t.solref + 2); T.D("tendon_solref", dv, {ntendon, 2});
    dv.clear(); for (auto& t : tend) dv.insert(dv.end(), t.solimp, t.solimp + 5); T.D("tendon_solimp", dv, {ntendon, 5});
    T.I("wrap_objid", wrap_obj, {nwrap}); T.D("wrap_prm", wrap_prm, {nwrap});
    names[OBJ_TENDON];
    for (auto& t : tend) names[OBJ_TENDON].push_back(t.name);
  }

  void actuators() {
    struct Act { std::string name; int trntype, trnid, biastype, group; double gear[6], gainprm[3], biasprm[3], ctrlrange[2], forcerange[2]; bool ctrllimited, forcelimited; };
    std::vector<Act> acts;
    for (auto& sp : root->kids) if (sp->tag == "actuator") for (auto& ep : sp->kids) {
      const Elem& e = *ep;
      if (e.tag != "motor" && e.tag != "position" && e.tag != "general") throw Error("actuator <" + e.tag + "> is outside the supported subset");
      AttrMap a = defaults.resolve(e.tag, e, nullptr);
      Act x{};
      x.gear[0] = 1.0;
      if (auto p = a.get("gear")) { x.gear[0] = 0; fill(x.gear, 6, floats(*p)); }
      if (auto p = a.get("joint")) {
        x.trntype = TRN_JOINT; x.trnid = name2id(OBJ_JOINT, *p);
        if (x.trnid < 0) throw Error("actuator joint not found: " + *p);
        if (joints[x.trnid].type != JNT_HINGE && joints[x.trnid].type != JNT_SLIDE) throw Error("joint transmission supports hinge/slide only");
      } else if (auto s = a.get("site")) {
        x.trntype = TRN_SITE; x.trnid = name2id(OBJ_SITE, *s);
        if (x.trnid < 0) throw Error("actuator site not found: " + *s);
        if (a.has("refsite")) throw Error("refsite is outside the supported subset");
      } else throw Error("actuator needs joint= or site= transmission");
      x.biastype = BIAS_NONE;
      if (e.tag == "motor") x.gainprm[0] = 1.0;
      else if (e.tag == "position") {
        const double kp = a.getd("kp", 1), kv = a.getd("kv", 0);
        x.gainprm[0] = kp; x.biasprm[0] = 0.0; x.biasprm[1] = -kp; x.biasprm[2] = -kv; x.biastype = BIAS_AFFINE;
      } else {
        if (a.gets("dyntype", "none") != "none" || a.gets("gaintype", "fixed") != "fixed") throw Error("general actuator: only dyntype=none, gaintype=fixed supported");
        x.gainprm[0] = 1.0;
        if (auto p = a.get("gainprm")) fill(x.gainprm, 3, floats(*p));
        const std::string bt = a.gets("biastype", "none");
        if (bt == "affine") { x.biastype = BIAS_AFFINE; fill(x.biasprm, 3, floats(a.gets("biasprm", "0 0 0"))); }
        else if (bt != "none") throw Error("general actuator: biastype must be none/affine");
      }
      x.name = a.gets("name", "");
      x.ctrllimited = limited(a, "ctrllimited", "ctrlrange"); x.forcelimited = limited(a, "forcelimited", "forcerange");
      Vec cr = floats(a.gets("ctrlrange", "0 0"), 2), fr = floats(a.gets("forcerange", "0 0"), 2);
      x.ctrlrange[0] = cr[0]; x.ctrlrange[1] = cr[1]; x.forcerange[0] = fr[0]; x.forcerange[1] = fr[1];
      x.group = a.geti("group", 0);
      acts.push_back(x);
    }
    nu = (int)acts.size();
    std::vector<int> iv; std::vector<double> dv;
    iv.clear(); for (auto& x : acts) iv.push_back(x.trntype); T.I("actuator_trntype", iv, {nu});
    iv.clear(); for (auto& x : acts) { iv.push_back(x.trnid); iv.push_back(-1); } T.I("actuator_trnid", iv, {nu, 2});
    dv.clear(); for (auto& x : acts) dv.insert(dv.end(), x.gear, x.gear + 6); T.D("actuator_gear", dv, {nu, 6});
    dv.clear(); for (auto& x : acts) dv.insert(dv.end(), x.gainprm, x.gainprm + 3); T.D("actuator_gainprm", dv, {nu, 3});
    dv.clear(); for (auto& x : acts) dv.insert(dv.end(), x.biasprm, x.biasprm + 3); T.D("actuator_biasprm", dv, {nu, 3});
    iv.clear(); for (auto& x : acts) iv.push_back(x.biastype); T.I("actuator_biastype", iv, {nu});
    iv.clear(); for (auto& x : acts) iv.push_back(x.ctrllimited ? 1 : 0); T.I("actuator_ctrllimited", iv, {nu});
    dv.clear(); for (auto& x : acts) dv.insert(dv.end(), x.ctrlrange, x.ctrlrange + 2); T.D("actuator_ctrlrange", dv, {nu, 2});
    iv.clear(); for (auto& x : acts) iv.push_back(x.forcelimited ? 1 : 0); T.I("actuator_forcelimited", iv, {nu});
    dv.clear(); for (auto& x : acts) dv.insert(dv.end(), x.forcerange, x.forcerange + 2); T.D("actuator_forcerange", dv, {nu, 2});
    T.I("actuator_actlimited", std::vector<int>(nu, 0), {nu});
    T.D("actuator_actrange", std::vector<double>((size_t)nu * 2, 0.0), {nu, 2});
    iv.clear(); for (auto& x : acts) iv.push_back(x.group); T.I("actuator_group", iv, {nu});
    names[OBJ_ACTUATOR];
    for (auto& x : acts) names[OBJ_ACTUATOR].push_back(x.name);
  }

  void sensors() {
    std::vector<int> type, obj, adr;
    names[OBJ_SENSOR];
    int a = 0;
    for (auto& sp : root->kids) if (sp->tag == "sensor") for (auto& ep : sp->kids) {
      const Elem& e = *ep;
      int st, ob;
      if (e.tag == "jointpos") {
        ob = name2id(OBJ_JOINT, e.gets("joint", ""));
        if (ob < 0) throw Error("sensor joint not found: " + e.gets("joint", ""));
        st = SENS_JOINTPOS;
      } else if (e.tag == "gyro" || e.tag == "accelerometer") {
        ob = name2id(OBJ_SITE, e.gets("site", ""));
        if (ob < 0) throw Error("sensor site not found: " + e.gets("site", ""));
        st = e.tag == "gyro" ? SENS_GYRO : SENS_ACCELEROMETER;
      } else if (e.tag == "framequat") {
        if (e.gets("objtype", "") != "site") throw Error("framequat: only objtype=site supported");
        ob = name2id(OBJ_SITE, e.gets("objname", ""));
        if (ob < 0) throw Error("sensor site not found: " + e.gets("objname", ""));
        st = SENS_FRAMEQUAT;
      } else throw Error("sensor <" + e.tag + "> is outside the supported subset");
      type.push_back(st); obj.push_back(ob); adr.push_back(a);
      names[OBJ_SENSOR].push_back(e.gets("name", ""));
      a += st == SENS_JOINTPOS ? 1 : (st == SENS_FRAMEQUAT ? 4 : 3);
    }
    nsensor = (int)type.size(); nsensordata = a;
    T.I("sensor_type", type, {nsensor}); T.I("sensor_objid", obj, {nsensor}); T.I("sensor_adr", adr, {nsensor});
  }

  // static candidate pair list = MuJoCo's per-step filter applied once; pair parameters follow mj_contactParam for equal priority
  void contacts() {
    std::set<std::pair<int, int>> excl;
    for (auto& sp : root->kids) if (sp->tag == "contact") for (auto& ep : sp->kids) {
      if (ep->tag != "exclude") throw Error("<contact><pair> is outside the supported subset");
      const int b1 = name2id(OBJ_BODY, ep->gets("body1", "")), b2 = name2id(OBJ_BODY, ep->gets("body2", ""));
      if (b1 < 0 || b2 < 0) throw Error("exclude: body not found");
      excl.insert({std::min(b1, b2), std::max(b1, b2)});
    }
    nexclude = (int)excl.size();
    static const std::set<std::pair<int, int>> supported = {{GEOM_PLANE, GEOM_SPHERE}, {GEOM_PLANE, GEOM_CAPSULE}, {GEOM_PLANE, GEOM_BOX}, {GEOM_PLANE, GEOM_ELLIPSOID},
                                                            {GEOM_SPHERE, GEOM_SPHERE}, {GEOM_SPHERE, GEOM_CAPSULE}, {GEOM_CAPSULE, GEOM_CAPSULE}};
    std::vector<int> g1v, g2v, condim;
    std::vector<double> friction, solref, solimp, margin, gap;
    const int ng = (int)geoms.size();
    for (int g1 = 0; g1 < ng; g1++) for (int g2 = g1 + 1; g2 < ng; g2++) {
      const Geom &G1 = geoms[g1], &G2 = geoms[g2];
      const int b1 = G1.body, b2 = G2.body;
      if (b1 == b2) continue;
      if (!((G1.contype & G2.conaffinity) || (G2.contype & G1.conaffinity))) continue;
      const int w1 = weld[b1], w2 = weld[b2];
      if (w1 == w2) continue;
      const int wp1 = weld[bodies[w1].parent], wp2 = weld[bodies[w2].parent];
      if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
      if (excl.count({std::min(b1, b2), std::max(b1, b2)})) continue;
      const int a = G1.type <= G2.type ? g1 : g2, b = G1.type <= G2.type ? g2 : g1;
      const Geom &GA = geoms[a], &GB = geoms[b];
      if (!supported.count({GA.type, GB.type})) throw Error("collision pair types (" + std::to_string(GA.type) + "," + std::to_string(GB.type) + ") are outside the supported subset");
      if (GA.priority != GB.priority) throw Error("geom priority is outside the supported subset");
      const double mix = (GA.solmix + GB.solmix) > MINVAL ? GA.solmix / (GA.solmix + GB.solmix) : 0.5;
      const double fr[3] = {std::max(GA.friction[0], GB.friction[0]), std::max(GA.friction[1], GB.friction[1]), std::max(GA.friction[2], GB.friction[2])};
      double sr[2];
      if (GA.solref[0] > 0 && GB.solref[0] > 0) for (int k = 0; k < 2; k++) sr[k] = mix * GA.solref[k] + (1 - mix) * GB.solref[k];
      else for (int k = 0; k < 2; k++) sr[k] = std::min(GA.solref[k], GB.solref[k]);
      g1v.push_back(a); g2v.push_back(b); condim.push_back(std::max(GA.condim, GB.condim));
      friction.insert(friction.end(), {fr[0], fr[0], fr[1], fr[2], fr[2]});
      solref.insert(solref.end(), sr, sr + 2);
      for (int k = 0; k < 5; k++) solimp.push_back(mix * GA.solimp[k] + (1 - mix) * GB.solimp[k]);
      margin.push_back(std::max(GA.margin, GB.margin)); gap.push_back(std::max(GA.gap, GB.gap));
    }
    npair = (int)g1v.size();
    T.I("pair_geom1", g1v, {npair}); T.I("pair_geom2", g2v, {npair}); T.I("pair_condim", condim, {npair});
    T.D("pair_friction", friction, {npair, 5}); T.D("pair_solref", solref, {npair, 2}); T.D("pair_solimp", solimp, {npair, 5});
    T.D("pair_margin", margin, {npair}); T.D("pair_gap", gap, {npair});
  }

  void keyframes() {
    std::vector<double> kq, kv, kc, kt;
    names[OBJ_KEY];
    for (auto& sp : root->kids) if (sp->tag == "keyframe") for (auto& ep : sp->kids) if (ep->tag == "key") {
      const Elem& e = *ep;
      Vec q = (e.has("qpos") && !e.get("qpos")->empty()) ? floats(*e.get("qpos")) : qpos0;
      Vec v = (e.has("qvel") && !e.get("qvel")->empty()) ? floats(*e.get("qvel")) : Vec(nv, 0.0);
      Vec c = (e.has("ctrl") && !e.get("ctrl")->empty()) ? floats(*e.get("ctrl")) : Vec(nu, 0.0);
      if ((int)q.size() != nq || (int)v.size() != nv || (int)c.size() != nu)
        throw Error("keyframe '" + e.gets("name", "") + "': size mismatch (qpos " + std::to_string(q.size()) + " vs nq " + std::to_string(nq) + ")");
      kq.insert(kq.end(), q.begin(), q.end()); kv.insert(kv.end(), v.begin(), v.end()); kc.insert(kc.end(), c.begin(), c.end());
      kt.push_back(e.has("time") ? to_double(*e.get("time")) : 0.0);
      names[OBJ_KEY].push_back(e.gets("name", ""));
    }
    nkey = (int)kt.size();
    T.D("key_qpos", kq, {nkey, nq}); T.D("key_qvel", kv, {nkey, nv}); T.D("key_ctrl", kc, {nkey, nu}); T.D("key_time", kt, {nkey});
  }

  // ---- kinematics / Jacobians / mass matrix at a configuration (compile time only; Jacobian form, independent of the CRB recursion) ----
  Kin kinematics(const std::vector<double>& qpos) const {
    const int nb = (int)bodies.size(), nj = (int)joints.size();
    Kin k;
    k.xpos.assign(nb, V3{0, 0, 0}); k.xquat.assign(nb, Q{1, 0, 0, 0}); k.xmat.assign(nb, M3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}});
    k.xanchor.assign(nj, V3{0, 0, 0}); k.xaxis.assign(nj, V3{0, 0, 0});
    for (int b = 1; b < nb; b++) {
      const int p = bodies[b].parent, jadr = body_jntadr[b], jnum = body_jntnum[b];
      V3 pos; Q quat;
      if (jnum == 1 && joints[jadr].type == JNT_FREE) {
        const int qa = jnt_qposadr[jadr];
        pos = {qpos[qa], qpos[qa + 1], qpos[qa + 2]};
        quat = qnorm({qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]});
        k.xanchor[jadr] = pos; k.xaxis[jadr] = {0, 0, 1};
      } else {
        pos = k.xpos[p] + mv(k.xmat[p], bodies[b].pos);
        quat = qmul(k.xquat[p], bodies[b].quat);
        for (int j = jadr; j < jadr + jnum; j++) {
          const int qa = jnt_qposadr[j];
          const M3 R = q2m(quat);
          const V3 anchor = mv(R, joints[j].pos) + pos, axis = mv(R, joints[j].axis);
          k.xanchor[j] = anchor; k.xaxis[j] = axis;
          const double val = qpos[qa] - qpos0[qa];
          if (joints[j].type == JNT_SLIDE) pos = pos + val * axis;
          else {
            const double s = std::sin(val / 2);
            quat = qmul(quat, {std::cos(val / 2), joints[j].axis.x * s, joints[j].axis.y * s, joints[j].axis.z * s});
            pos = anchor - mv(q2m(quat), joints[j].pos);
          }
        }
      }
      quat = qnorm(quat);
      k.xpos[b] = pos; k.xquat[b] = quat; k.xmat[b] = q2m(quat);
    }
    k.xipos.resize(nb); k.ximat.resize(nb);
    for (int b = 0; b < nb; b++) {
      k.xipos[b] = k.xpos[b] + mv(k.xmat[b], body_ipos[b]);
      const M3 iq = q2m(body_iquat[b]);
      M3 r{};
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int t = 0; t < 3; t++) r.m[i][j] += k.xmat[b].m[i][t] * iq.m[t][j];
      k.ximat[b] = r;
    }
    return k;
  }
  // jp, jr: 3 x nv row-major
  void jac_point(const Kin& k, int body, V3 point, std::vector<double>& jp, std::vector<double>& jr) const {
    jp.assign((size_t)3 * nv, 0.0); jr.assign((size_t)3 * nv, 0.0);
    auto setc = [&](std::vector<double>& J, int d, V3 v) { J[d] = v.x; J[nv + d] = v.y; J[2 * nv + d] = v.z; };
    for (int b = body; b > 0; b = bodies[b].parent)
      for (int j = body_jntadr[b]; j < body_jntadr[b] + body_jntnum[b]; j++) {
        const int d = jnt_dofadr[j], t = joints[j].type;
        if (t == JNT_FREE) {
          setc(jp, d, {1, 0, 0}); setc(jp, d + 1, {0, 1, 0}); setc(jp, d + 2, {0, 0, 1});
          for (int c = 0; c < 3; c++) { const V3 ax = col(k.xmat[b], c); setc(jr, d + 3 + c, ax); setc(jp, d + 3 + c, cross(ax, point - k.xpos[b])); }
        } else if (t == JNT_SLIDE) setc(jp, d, k.xaxis[j]);
        else { setc(jr, d, k.xaxis[j]); setc(jp, d, cross(k.xaxis[j], point - k.xanchor[j])); }
      }
  }
  std::vector<double> mass_matrix(const Kin& k) const {
    std::vector<double> M((size_t)nv * nv, 0.0);
    for (int i = 0; i < nv; i++) M[(size_t)i * nv + i] = dof_arm[i];
    std::vector<double> jp, jr;
    for (int b = 1; b < (int)bodies.size(); b++) {
      if (body_mass[b] <= 0) continue;
      jac_point(k, b, k.xipos[b], jp, jr);
      double Iw[3][3] = {};
      const double inr[3] = {body_inertia[b].x, body_inertia[b].y, body_inertia[b].z};
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) for (int t = 0; t < 3; t++) Iw[r][c] += k.ximat[b].m[r][t] * inr[t] * k.ximat[b].m[c][t];
      for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) {
        double s = 0;
        for (int r = 0; r < 3; r++) {
          s += body_mass[b] * jp[r * nv + i] * jp[r * nv + j];
          for (int c = 0; c < 3; c++) s += jr[r * nv + i] * Iw[r][c] * jr[c * nv + j];
        }
        M[(size_t)i * nv + j] += s;
      }
    }
    return M;
  }

  // constants that need physics at qpos0 (mj_setConst)
  void set_const() {
    const int nb = (int)bodies.size();
    const Kin k = kinematics(qpos0);
    const std::vector<double> M = mass_matrix(k);
    T.D("qM0", M, {nv, nv});
    std::vector<double> Minv;
    if (nv > 0) {
      Minv = inverse(M, nv);
      double tr = 0;
      for (int i = 0; i < nv; i++) tr += M[(size_t)i * nv + i];
      meaninertia = tr / nv;
    } else meaninertia = 1.0;
    std::vector<double> dinv(nv);
    for (int i = 0; i < nv; i++) dinv[i] = Minv[(size_t)i * nv + i];
    for (int j = 0; j < (int)joints.size(); j++) if (joints[j].type == JNT_FREE) {
      const int d0 = jnt_dofadr[j];
      const double a = (dinv[d0] + dinv[d0 + 1] + dinv[d0 + 2]) / 3, b = (dinv[d0 + 3] + dinv[d0 + 4] + dinv[d0 + 5]) / 3;
      for (int t = 0; t < 3; t++) { dinv[d0 + t] = a; dinv[d0 + 3 + t] = b; }
    }
    T.D("dof_invweight0", dinv, {nv});
    auto trace_JMJ = [&](const std::vector<double>& J) {      // trace(J Minv J^T), J 3 x nv
      double tr = 0;
      for (int r = 0; r < 3; r++) for (int i = 0; i < nv; i++) {
        if (J[r * nv + i] == 0) continue;
        double s = 0;
        for (int j = 0; j < nv; j++) s += Minv[(size_t)i * nv + j] * J[r * nv + j];
        tr += J[r * nv + i] * s;
      }
      return tr;
    };
    std::vector<double> binv((size_t)nb * 2, 0.0), jp, jr;
    for (int b = 1; b < nb; b++) {
      if (weld[b] == 0) continue;
      jac_point(k, b, k.xipos[b], jp, jr);
      binv[2 * b] = std::max(MINVAL, trace_JMJ(jp) / 3);
      binv[2 * b + 1] = std::max(MINVAL, trace_JMJ(jr) / 3);
    }
    T.D("body_invweight0", binv, {nb, 2});
    std::vector<double> tinv(ntendon, 0.0);
    for (int t = 0; t < ntendon; t++) {
      std::vector<double> J(nv, 0.0);
      for (int w = tendon_adr[t]; w < tendon_adr[t] + tendon_num[t]; w++) J[jnt_dofadr[wrap_obj[w]]] = wrap_prm[w];
      double s = 0;
      for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) s += J[i] * Minv[(size_t)i * nv + j] * J[j];
      tinv[t] = std::max(MINVAL, s);
    }
    T.D("tendon_invweight0", tinv, {ntendon});
    std::vector<int> anc((size_t)nb * nv, 0);
    for (int b = 1; b < nb; b++) for (int p = b; p > 0; p = bodies[p].parent)
      for (int d = 0; d < body_dofnum[p]; d++) anc[(size_t)b * nv + body_dofadr[p] + d] = 1;
    T.I("body_dofmask", anc, {nb, nv});
  }

  void options() {
    for (auto& sp : root->kids) {
      const Elem& e = *sp;
      if (e.tag == "compiler") {
        if (auto p = e.get("angle")) { if (*p == "radian") angle_scale = 1.0; else if (*p == "degree") angle_scale = M_PI / 180.0; }
        if (auto p = e.get("autolimits")) autolimits = *p == "true";
      } else if (e.tag == "option") {
        auto nonempty = [&](const char* k) { auto p = e.get(k); return p && !p->empty(); };
        if (nonempty("timestep")) timestep = to_double(*e.get("timestep"));
        if (nonempty("gravity")) { Vec g = floats(*e.get("gravity"), 3); gravity[0] = g[0]; gravity[1] = g[1]; gravity[2] = g[2]; }
        if (auto p = e.get("integrator")) {
          if (*p == "Euler") integrator = INT_EULER; else if (*p == "RK4") integrator = INT_RK4;
          else throw Error("integrator '" + *p + "' is outside the supported subset (Euler, RK4)");
        }
        if (nonempty("density")) density = to_double(*e.get("density"));
        if (nonempty("viscosity")) viscosity = to_double(*e.get("viscosity"));
        if (nonempty("impratio")) impratio = to_double(*e.get("impratio"));
        if (nonempty("tolerance")) tolerance = to_double(*e.get("tolerance"));
        if (nonempty("iterations")) iterations = to_int(*e.get("iterations"));
        for (const char* k : {"cone", "solver", "jacobian"})
          if (auto p = e.get(k)) if (*p != "pyramidal" && *p != "Newton" && *p != "dense" && *p != "auto") throw Error(std::string("option ") + k + "='" + *p + "' is outside the supported subset");
      }
    }
  }

  static std::string json_str(const std::string& s) {
    std::string o = "\"";
    for (char c : s) { if (c == '"' || c == '\\') { o += '\\'; o += c; } else if ((unsigned char)c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); o += b; } else o += c; }
    return o + "\"";
  }

  void compile() {
    if (root->tag != "mujoco") throw Error("root element must be <mujoco>");
    validate_schema(*root);
    options();
    for (auto& sp : root->kids) if (sp->tag == "default") defaults.add(*sp, nullptr);
    bodies.push_back(Body{"world", 0, {0, 0, 0}, {1, 0, 0, 0}, {}, {}, false, ""});
    for (auto& sp : root->kids) if (sp->tag == "worldbody") body_children(*sp, 0, nullptr);
    finalize_tree();
    tendons();
    actuators();
    sensors();
    contacts();
    keyframes();
    set_const();
  }

  // scalars first (the order of the Python front's table), then the arrays, then the Python front's meta field and the names
  std::vector<Field> table() const {
    std::vector<Field> out;
    auto IS = [&](const char* n, int v) { out.push_back({n, 1, {}, {v}, {}}); };
    auto DS = [&](const char* n, double v) { out.push_back({n, 0, {v}, {}, {}}); };
    IS("nq", nq); IS("nv", nv); IS("nu", nu); IS("nbody", (int)bodies.size()); IS("njnt", (int)joints.size()); IS("ngeom", (int)geoms.size());
    IS("nsite", (int)sites.size()); IS("ntendon", ntendon); IS("nwrap", nwrap); IS("nsensor", nsensor); IS("nsensordata", nsensordata); IS("nkey", nkey);
    IS("npair", npair); IS("integrator", integrator); IS("disableactuator", disableactuator); IS("iterations", iterations);
    DS("timestep", timestep); DS("density", density); DS("viscosity", viscosity); DS("impratio", impratio); DS("tolerance", tolerance); DS("meaninertia", meaninertia);
    out.push_back({"gravity", 0, {gravity[0], gravity[1], gravity[2]}, {}, {}});
    for (auto& f : T.f) out.push_back(f);
    std::string meta = "{\"ls_iterations\": " + std::to_string(ls_iterations) + ", \"na\": 0, \"name\": " + json_str(model_name) + ", \"nexclude\": " + std::to_string(nexclude) + ", \"shapes\": {";
    bool first = true;
    for (auto& f : T.f) {
      auto it = T.shapes.find(f.name);
      if (it == T.shapes.end()) continue;
      if (!first) meta += ", ";
      first = false;
      meta += json_str(f.name) + ": [";
      for (size_t k = 0; k < it->second.size(); k++) { if (k) meta += ", "; meta += std::to_string(it->second[k]); }
      meta += "]";
    }
    meta += "}}";
    out.push_back({"meta_json", 2, {}, {}, meta});
    for (auto& kv : names) {
      std::string blob;
      for (auto& n : kv.second) { blob += n; blob.push_back('\0'); }
      out.push_back({"names_" + std::to_string(kv.first), 2, {}, {}, blob});
    }
    return out;
  }
};

static int create_from(Compiler& c, mjbModel** out) {
  const std::vector<Field> t = c.table();
  std::vector<const char*> names; std::vector<const void*> ptrs; std::vector<int> dts; std::vector<long> cnts;
  for (auto& f : t) {
    names.push_back(f.name.c_str()); dts.push_back(f.dtype);
    if (f.dtype == 0) { ptrs.push_back(f.d.data()); cnts.push_back((long)f.d.size()); }
    else if (f.dtype == 1) { ptrs.push_back(f.i.data()); cnts.push_back((long)f.i.size()); }
    else { ptrs.push_back(f.b.data()); cnts.push_back((long)f.b.size()); }
  }
  return mjb_model_create((int)t.size(), names.data(), ptrs.data(), dts.data(), cnts.data(), out);
}

}  // namespace mjcf

extern "C" {
// defined in mjb_api.hip: sets the thread-local error string of mjb_last_error()
int mjb_set_error_(int code, const char* msg);

int mjb_model_load_xml_string(const char* xml_text, const char* base_dir, mjbModel** out) {
  if (!xml_text || !out) return mjb_set_error_(MJB_ERR_ARG, "xml/out is NULL");
  try {
    mjcf::Compiler c(mjcf::parse_xml(xml_text), base_dir && *base_dir ? base_dir : ".");
    c.compile();
    return mjcf::create_from(c, out);
  } catch (const std::exception& e) {
    return mjb_set_error_(MJB_ERR_MODEL, e.what());
  }
}

int mjb_model_load_xml(const char* path, mjbModel** out) {
  if (!path || !out) return mjb_set_error_(MJB_ERR_ARG, "path/out is NULL");
  try {
    const std::string text = mjcf::read_file(path);
    std::string dir = mjcf::dir_of(path);
    if (dir.empty() || dir[0] != '/') {                         // like the Python front: includes resolve against the absolute directory
      char buf[4096];
      if (realpath(dir.c_str(), buf)) dir = buf;
    }
    return mjb_model_load_xml_string(text.c_str(), dir.c_str(), out);
  } catch (const std::exception& e) {
    return mjb_set_error_(MJB_ERR_MODEL, e.what());
  }
}
}
