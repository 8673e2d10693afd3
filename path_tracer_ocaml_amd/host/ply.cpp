// ply.cpp -- PLY reader with the behaviour of the reference's ply_format/src/ply.ml,
// and the ganesha scene built from a real PLY file (ganesha/bin/main.ml:50-85,121-131,141-203).
//
// Same acceptance rules as the reference parser:
//  * the file starts with "ply\n" (ply.ml:322-330); header lines up to "end_header" (ply.ml:288-299);
//  * `format binary_little_endian 1.0` only -- ascii / big endian are refused (ply.ml:340-350);
//  * type names char uchar short ushort int uint float double, plus int8 / uint8 (ply.ml:78-88);
//  * an element is either all scalar properties (read into columns by name, ply.ml:208-217) or exactly ONE
//    list property (read into rows and keyed by the PROPERTY name, ply.ml:219-235,238-249);
//    a mix is refused ("TO DO: parse mixed list/non-list element");
//  * the list reader does not advance the input (ply.ml:219-235), so a list element must be the last one --
//    here that is an explicit error instead of silently mis-parsing what follows.
// One deliberate difference: 16-bit integer properties are read as 16 bits (the reference reads ONE byte for
// short / ushort, ply.ml:104-105, which can only be a bug); no scene in scope has such a property.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../csrc/pt_vec.h"
#include "host.h"

namespace {
thread_local std::string g_err;

enum class Ty { Char, Uchar, Short, Ushort, Int, Uint, Float, Double };

bool parse_type(const std::string& s, Ty* out) {
  static const std::map<std::string, Ty> names = {
      {"uint8", Ty::Uchar}, {"int8", Ty::Char},   {"char", Ty::Char},   {"uchar", Ty::Uchar}, {"short", Ty::Short},
      {"ushort", Ty::Ushort}, {"int", Ty::Int},   {"uint", Ty::Uint},   {"float", Ty::Float}, {"double", Ty::Double},
      // [%of_sexp: t] also accepts the capitalised constructor names
      {"Char", Ty::Char},   {"Uchar", Ty::Uchar}, {"Short", Ty::Short}, {"Ushort", Ty::Ushort}, {"Int", Ty::Int},
      {"Uint", Ty::Uint},   {"Float", Ty::Float}, {"Double", Ty::Double}};
  auto it = names.find(s);
  if (it == names.end()) return false;
  *out = it->second;
  return true;
}
size_t type_size(Ty t) {
  switch (t) {
    case Ty::Char: case Ty::Uchar: return 1;
    case Ty::Short: case Ty::Ushort: return 2;
    case Ty::Int: case Ty::Uint: case Ty::Float: return 4;
    default: return 8;
  }
}
bool is_float(Ty t) { return t == Ty::Float || t == Ty::Double; }

double read_float(const uint8_t* p, Ty t) {
  if (t == Ty::Float) {
    float f;
    std::memcpy(&f, p, 4);
    return (double)f; // Int32.float_of_bits
  }
  double d;
  std::memcpy(&d, p, 8);
  return d;
}
int64_t read_int(const uint8_t* p, Ty t) {
  switch (t) {
    case Ty::Char: return (int8_t)p[0];
    case Ty::Uchar: return p[0];
    case Ty::Short: { int16_t v; std::memcpy(&v, p, 2); return v; }
    case Ty::Ushort: { uint16_t v; std::memcpy(&v, p, 2); return v; }
    case Ty::Int: { int32_t v; std::memcpy(&v, p, 4); return v; }
    case Ty::Uint: { uint32_t v; std::memcpy(&v, p, 4); return v; }
    default: return 0;
  }
}

struct Property {
  bool is_list = false;
  Ty type = Ty::Float, length_type = Ty::Uchar;
  std::string name;
};
struct Element {
  std::string name;
  long long count = 0;
  std::vector<Property> props;
};

std::vector<std::string> split(const std::string& s) { // String.split ~on:' '
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) {
    if (c == ' ') {
      out.push_back(cur);
      cur.clear();
    } else cur.push_back(c);
  }
  out.push_back(cur);
  return out;
}
bool starts_with(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
}  // namespace

struct pth_ply {
  std::map<std::string, std::map<std::string, std::vector<double>>> floats; // element -> property -> column
  std::map<std::string, std::map<std::string, std::vector<int64_t>>> ints;
  std::map<std::string, std::vector<std::vector<int64_t>>> rows;          // list PROPERTY name -> rows (OCaml ints: a uint above 2^31 stays positive)
  std::map<std::string, long long> counts;
  std::vector<int64_t> flat_rows;   // last accessed list, flattened for the C interface
  std::vector<int32_t> row_lengths;
};

extern "C" {

const char* pth_last_error(void) { return g_err.c_str(); }

static pth_ply* ply_load_impl(const char* path);
pth_ply* pth_ply_load(const char* path) {
  try {
    return ply_load_impl(path);
  } catch (const std::exception& e) { // bad_alloc / length_error on a hostile header: report, never terminate the host
    g_err = std::string("PLY load failed: ") + e.what();
    return nullptr;
  }
}
static pth_ply* ply_load_impl(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    g_err = std::string("cannot open ") + path;
    return nullptr;
  }
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  size_t pos = 0;
  if (buf.size() < 4) {
    g_err = "Could not read ply header (not enough bytes)";
    return nullptr;
  }
  if (std::memcmp(buf.data(), "ply\n", 4) != 0) {
    g_err = "expected file to start with \"ply\\n\"";
    return nullptr;
  }
  pos = 4;
  std::vector<std::string> lines;
  bool ended = false;
  while (pos <= buf.size()) {
    size_t nl = pos;
    while (nl < buf.size() && buf[nl] != '\n') ++nl;
    if (nl >= buf.size()) break;
    std::string line((const char*)&buf[pos], nl - pos);
    pos = nl + 1;
    if (line == "end_header") {
      ended = true;
      break;
    }
    lines.push_back(line);
  }
  if (!ended) {
    g_err = "missing \"end_header\" line";
    return nullptr;
  }
  std::string format;
  for (const std::string& l : lines)
    if (starts_with(l, "format ")) {
      auto w = split(l);
      if (w.size() != 3 || w[2] != "1.0") {
        g_err = "cannot parse format line: " + l;
        return nullptr;
      }
      format = w[1];
      break;
    }
  if (format.empty()) {
    g_err = "header has no format line";
    return nullptr;
  }
  if (format != "binary_little_endian" && format != "Binary_little_endian") {
    if (format == "ascii" || format == "binary_big_endian" || format == "Ascii" || format == "Binary_big_endian") g_err = "to do: handle message format " + format;
    else g_err = "unrecognized format " + format;
    return nullptr;
  }
  std::vector<Element> elements;
  for (const std::string& l : lines) {
    const bool is_elt = starts_with(l, "element "), is_prop = starts_with(l, "property ");
    if (!is_elt && !is_prop) continue; // comments, obj_info, format
    auto w = split(l);
    if (is_elt) {
      if (w.size() != 3) {
        g_err = "expected element: " + l;
        return nullptr;
      }
      Element e;
      e.name = w[1];
      // Int.of_string (ply.ml:278) raises on anything that is not an integer literal; a negative count then
      // fails in Array.create.  Both are refused here (digits and '_' separators only, as Base accepts).
      {
        const std::string& c = w[2];
        size_t k = (!c.empty() && c[0] == '+') ? 1 : 0;
        bool ok = k < c.size() && c[k] != '_';
        long long v = 0;
        for (; ok && k < c.size(); ++k) {
          if (c[k] == '_') continue;
          if (c[k] < '0' || c[k] > '9' || v > (1ll << 52)) ok = false;
          else v = v * 10 + (c[k] - '0');
        }
        if (!ok) {
          g_err = "Int.of_string: " + c + " (element count must be a non-negative integer)";
          return nullptr;
        }
        e.count = v;
      }
      elements.push_back(e);
    } else {
      if (elements.empty()) {
        g_err = "expected element: " + l;
        return nullptr;
      }
      Property p;
      if (w.size() == 5 && w[1] == "list") {
        p.is_list = true;
        p.name = w[4];
        if (!parse_type(w[2], &p.length_type) || !parse_type(w[3], &p.type)) {
          g_err = "unrecognized type in: " + l;
          return nullptr;
        }
      } else if (w.size() == 3) {
        p.name = w[2];
        if (!parse_type(w[1], &p.type)) {
          g_err = "unrecognized type " + w[1];
          return nullptr;
        }
      } else {
        g_err = "cannot parse property: " + l;
        return nullptr;
      }
      elements.back().props.push_back(p);
    }
  }
  pth_ply* ply = new pth_ply();
  auto fail = [&](const std::string& m) {
    g_err = m;
    delete ply;
    return (pth_ply*)nullptr;
  };
  for (size_t ei = 0; ei < elements.size(); ++ei) {
    const Element& e = elements[ei];
    size_t n_list = 0;
    for (const Property& p : e.props) n_list += p.is_list ? 1 : 0;
    if (n_list == 1 && e.props.size() == 1) {
      const Property& p = e.props[0];
      if (is_float(p.length_type) || is_float(p.type)) return fail("expected integer type in list property " + p.name);
      if (ply->rows.count(p.name) || ply->counts.count(p.name)) return fail("duplicate key " + p.name); // Map.of_alist_exn
      if (ei + 1 != elements.size()) return fail("a list element must be the last element (the reference's list reader does not advance its input, ply.ml:219-235)");
      if ((size_t)e.count > (buf.size() - pos) / (type_size(p.length_type) ? type_size(p.length_type) : 1)) return fail("truncated list element " + e.name);
      std::vector<std::vector<int64_t>> rows((size_t)e.count);
      const size_t ls = type_size(p.length_type), es = type_size(p.type);
      for (long long i = 0; i < e.count; ++i) {
        if (pos + ls > buf.size()) return fail("truncated list element " + e.name);
        const int64_t len = read_int(&buf[pos], p.length_type);
        pos += ls;
        if (len < 0 || pos + (size_t)len * es > buf.size()) return fail("truncated list element " + e.name);
        rows[(size_t)i].resize((size_t)len);
        for (int64_t k = 0; k < len; ++k) rows[(size_t)i][(size_t)k] = read_int(&buf[pos + (size_t)k * es], p.type);
        pos += (size_t)len * es;
      }
      ply->rows[p.name] = std::move(rows);
      ply->counts[p.name] = e.count;
    } else if (n_list == 0) {
      if (ply->counts.count(e.name)) return fail("duplicate key " + e.name);
      size_t width = 0;
      for (const Property& p : e.props) width += type_size(p.type);
      if (width > 0 && (size_t)e.count > (buf.size() - pos) / width) return fail("truncated element " + e.name);
      if (width == 0 && e.count > (1ll << 28)) return fail("element " + e.name + " has no properties and an absurd count");
      size_t off = 0;
      for (const Property& p : e.props) {
        if (is_float(p.type)) {
          if (ply->floats[e.name].count(p.name) || ply->ints[e.name].count(p.name)) return fail("duplicate key " + p.name);
          std::vector<double> col((size_t)e.count);
          for (long long i = 0; i < e.count; ++i) col[(size_t)i] = read_float(&buf[pos + width * (size_t)i + off], p.type);
          ply->floats[e.name][p.name] = std::move(col);
        } else {
          if (ply->floats[e.name].count(p.name) || ply->ints[e.name].count(p.name)) return fail("duplicate key " + p.name);
          std::vector<int64_t> col((size_t)e.count);
          for (long long i = 0; i < e.count; ++i) col[(size_t)i] = read_int(&buf[pos + width * (size_t)i + off], p.type);
          ply->ints[e.name][p.name] = std::move(col);
        }
        off += type_size(p.type);
      }
      pos += width * (size_t)e.count;
      ply->counts[e.name] = e.count;
    } else {
      return fail("TO DO: parse mixed list/non-list element");
    }
  }
  return ply;
}

void pth_ply_free(pth_ply* p) { delete p; }

int64_t pth_ply_count(const pth_ply* p, const char* key) {
  if (!p) return -1;
  auto it = p->counts.find(key);
  return it == p->counts.end() ? -1 : it->second;
}

const double* pth_ply_floats(const pth_ply* p, const char* element, const char* property) {
  if (!p) return nullptr;
  auto e = p->floats.find(element);
  if (e == p->floats.end()) return nullptr;
  auto c = e->second.find(property);
  return c == e->second.end() ? nullptr : c->second.data();
}

const int64_t* pth_ply_ints(const pth_ply* p, const char* element, const char* property) {
  if (!p) return nullptr;
  auto e = p->ints.find(element);
  if (e == p->ints.end()) return nullptr;
  auto c = e->second.find(property);
  return c == e->second.end() ? nullptr : c->second.data();
}

// rows of list property `name`: lengths_out[i] = row length, returns the flattened values (or NULL)
const int64_t* pth_ply_rows(pth_ply* p, const char* name, const int32_t** lengths_out) {
  if (!p) return nullptr;
  auto it = p->rows.find(name);
  if (it == p->rows.end()) return nullptr;
  p->flat_rows.clear();
  p->row_lengths.clear();
  for (const auto& r : it->second) {
    p->row_lengths.push_back((int32_t)r.size());
    p->flat_rows.insert(p->flat_rows.end(), r.begin(), r.end());
  }
  if (lengths_out) *lengths_out = p->row_lengths.data();
  return p->flat_rows.data();
}

} // extern "C"

// ---------------------------------------------------------------- ganesha from a PLY (ganesha/bin/main.ml)
pth_scene* pth_scene_ganesha_from_mesh(int32_t width, int32_t height, const std::vector<double>& x, const std::vector<double>& y,
                                       const std::vector<double>& z, const std::vector<int32_t>& tri, bool sky);

extern "C" pth_scene* pth_scene_ganesha_ply(const char* path, int32_t width, int32_t height) {
  pth_ply* ply = pth_ply_load(path); // load_ply_exn
  if (!ply) return nullptr;
  auto bail = [&](const std::string& m) {
    g_err = m;
    pth_ply_free(ply);
    return (pth_scene*)nullptr;
  };
  // Mesh.create (main.ml:50-85): Map.find_exn d "vertex", "vertex_indices" -> "rows", x/y/z Floats
  auto v = ply->floats.find("vertex");
  if (v == ply->floats.end()) return bail("key not found: vertex");
  auto rows = ply->rows.find("vertex_indices");
  if (rows == ply->rows.end()) return bail("key not found: vertex_indices");
  for (const char* ax : {"x", "y", "z"})
    if (!v->second.count(ax)) return bail(ply->ints["vertex"].count(ax) ? "floats_exn: expected Floats" : std::string("key not found: ") + ax);
  const std::vector<double>&x = v->second["x"], &y = v->second["y"], &z = v->second["z"];
  std::vector<int32_t> tri;
  tri.reserve(rows->second.size() * 3);
  const int32_t nv = (int32_t)x.size();
  for (const auto& r : rows->second) {
    for (int64_t a : r)
      if (a < 0 || a >= nv) return bail("face index out of bounds"); // assert (Array.for_all faces ~f:in_bounds)
    if (r.size() != 3) return bail("expected triangular face");      // main.ml:182-185
    for (int64_t a : r) tri.push_back((int32_t)a);
  }
  if (tri.empty()) return bail("Shape_tree.create: expected non-empty list of shapes");
  pth_scene* s = pth_scene_ganesha_from_mesh(width, height, x, y, z, tri, true);
  pth_ply_free(ply);
  return s;
}
