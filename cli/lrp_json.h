// lrp_json.h — a small JSON document type for the CLI's config-file mode (row f4): parse,
// edit, dump.  Objects keep their keys sorted (what the reference's nlohmann::json does),
// unknown keys survive a read-modify-write, numbers keep their integer / real nature and
// reals print in the shortest form that reads back to the same double.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lrp_json {

class Value {
public:
  enum Kind { Null, Bool, Int, Real, String, Array, Object };
  Kind kind = Null;
  bool b = false;
  long long i = 0;
  double d = 0.0;
  std::string s;
  std::vector<Value> arr;
  std::map<std::string, Value> obj;

  Value() = default;
  static Value boolean(bool v) { Value x; x.kind = Bool; x.b = v; return x; }
  static Value integer(long long v) { Value x; x.kind = Int; x.i = v; return x; }
  static Value real(double v) { Value x; x.kind = Real; x.d = v; return x; }
  static Value string(const std::string &v) { Value x; x.kind = String; x.s = v; return x; }
  static Value array() { Value x; x.kind = Array; return x; }
  static Value object() { Value x; x.kind = Object; return x; }

  bool contains(const std::string &k) const { return kind == Object && obj.count(k); }
  // object access; a null value silently becomes an object / array on write access,
  // missing keys throw on read access (like nlohmann's at()).
  Value &operator[](const std::string &k) {
    if (kind == Null) kind = Object;
    if (kind != Object) throw std::invalid_argument("JSON value is not an object (key '" + k + "')");
    return obj[k];
  }
  const Value &at(const std::string &k) const {
    if (kind != Object || !obj.count(k)) throw std::invalid_argument("JSON key '" + k + "' not found");
    return obj.at(k);
  }
  Value &operator[](size_t n) {
    if (kind == Null) kind = Array;
    if (kind != Array) throw std::invalid_argument("JSON value is not an array");
    if (arr.size() <= n) arr.resize(n + 1);
    return arr[n];
  }
  const Value &at(size_t n) const {
    if (kind != Array || n >= arr.size()) throw std::invalid_argument("JSON array index out of range");
    return arr[n];
  }
  double number() const {
    if (kind == Int) return (double)i;
    if (kind == Real) return d;
    throw std::invalid_argument("JSON value is not a number");
  }
  float as_float() const { return (float)number(); }
  int as_int() const { return kind == Int ? (int)i : (int)number(); }
  const std::string &str() const {
    if (kind != String) throw std::invalid_argument("JSON value is not a string");
    return s;
  }

  std::string dump(int indent) const {
    std::string out;
    write(out, indent, 0);
    return out;
  }

private:
  static void write_string(std::string &out, const std::string &v) {
    out.push_back('"');
    for (unsigned char c : v) {
      switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      case '\b': out += "\\b"; break;
      case '\f': out += "\\f"; break;
      default:
        if (c < 0x20) {
          char buf[8];
          std::snprintf(buf, sizeof(buf), "\\u%04x", c);
          out += buf;
        } else {
          out.push_back((char)c);
        }
      }
    }
    out.push_back('"');
  }
  static std::string real_text(double v) {
    if (!std::isfinite(v)) return "null";
    char buf[40];
    for (int prec = 1; prec <= 17; ++prec) {
      std::snprintf(buf, sizeof(buf), "%.*g", prec, v);
      if (std::strtod(buf, nullptr) == v) break;
    }
    std::string t = buf;
    if (t.find_first_of(".eEn") == std::string::npos) t += ".0";
    return t;
  }
  void write(std::string &out, int indent, int depth) const {
    const std::string pad((size_t)(indent * (depth + 1)), ' '), pad_end((size_t)(indent * depth), ' ');
    switch (kind) {
    case Null: out += "null"; break;
    case Bool: out += b ? "true" : "false"; break;
    case Int: out += std::to_string(i); break;
    case Real: out += real_text(d); break;
    case String: write_string(out, s); break;
    case Array:
      if (arr.empty()) {
        out += "[]";
        break;
      }
      out += "[\n";
      for (size_t k = 0; k < arr.size(); ++k) {
        out += pad;
        arr[k].write(out, indent, depth + 1);
        out += k + 1 < arr.size() ? ",\n" : "\n";
      }
      out += pad_end + "]";
      break;
    case Object: {
      if (obj.empty()) {
        out += "{}";
        break;
      }
      out += "{\n";
      size_t k = 0;
      for (const auto &kv : obj) {
        out += pad;
        write_string(out, kv.first);
        out += ": ";
        kv.second.write(out, indent, depth + 1);
        out += ++k < obj.size() ? ",\n" : "\n";
      }
      out += pad_end + "}";
      break;
    }
    }
  }
};

class Parser {
public:
  explicit Parser(const std::string &text) : t(text) {}
  Value parse_document() {
    Value v = parse_value();
    skip();
    if (p != t.size()) fail("trailing characters");
    return v;
  }

private:
  const std::string &t;
  size_t p = 0;
  [[noreturn]] void fail(const char *what) const {
    throw std::invalid_argument(std::string("JSON parse error at byte ") + std::to_string(p) + ": " + what);
  }
  void skip() {
    while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\r' || t[p] == '\t')) ++p;
  }
  bool eat(const char *lit) {
    const size_t n = std::char_traits<char>::length(lit);
    if (t.compare(p, n, lit) == 0) {
      p += n;
      return true;
    }
    return false;
  }
  Value parse_value() {
    skip();
    if (p >= t.size()) fail("unexpected end");
    const char c = t[p];
    if (c == '{') return parse_object();
    if (c == '[') return parse_array();
    if (c == '"') return Value::string(parse_string());
    if (eat("true")) return Value::boolean(true);
    if (eat("false")) return Value::boolean(false);
    if (eat("null")) return Value();
    return parse_number();
  }
  Value parse_number() {
    const size_t start = p;
    if (p < t.size() && t[p] == '-') ++p;
    bool real = false;
    while (p < t.size() && (std::isdigit((unsigned char)t[p]) || t[p] == '.' || t[p] == 'e' || t[p] == 'E' || t[p] == '+' || t[p] == '-')) {
      if (t[p] == '.' || t[p] == 'e' || t[p] == 'E') real = true;
      ++p;
    }
    if (p == start) fail("unexpected character");
    const std::string tok = t.substr(start, p - start);
    if (!real) {
      errno = 0;
      char *end = nullptr;
      const long long v = std::strtoll(tok.c_str(), &end, 10);
      if (errno == 0 && end && *end == 0) return Value::integer(v);
    }
    char *end = nullptr;
    const double v = std::strtod(tok.c_str(), &end);
    if (!end || *end) fail("bad number");
    return Value::real(v);
  }
  static void append_utf8(std::string &s, unsigned cp) {
    if (cp < 0x80) {
      s.push_back((char)cp);
    } else if (cp < 0x800) {
      s.push_back((char)(0xC0 | (cp >> 6)));
      s.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
      s.push_back((char)(0xE0 | (cp >> 12)));
      s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      s.push_back((char)(0xF0 | (cp >> 18)));
      s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      s.push_back((char)(0x80 | (cp & 0x3F)));
    }
  }
  unsigned hex4() {
    if (p + 4 > t.size()) fail("bad \\u escape");
    unsigned v = 0;
    for (int k = 0; k < 4; ++k) {
      const char c = t[p++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string parse_string() {
    ++p; // opening quote
    std::string s;
    for (;;) {
      if (p >= t.size()) fail("unterminated string");
      const char c = t[p++];
      if (c == '"') break;
      if (c != '\\') {
        s.push_back(c);
        continue;
      }
      if (p >= t.size()) fail("unterminated escape");
      const char e = t[p++];
      switch (e) {
      case '"': s.push_back('"'); break;
      case '\\': s.push_back('\\'); break;
      case '/': s.push_back('/'); break;
      case 'b': s.push_back('\b'); break;
      case 'f': s.push_back('\f'); break;
      case 'n': s.push_back('\n'); break;
      case 'r': s.push_back('\r'); break;
      case 't': s.push_back('\t'); break;
      case 'u': {
        unsigned cp = hex4();
        if (cp >= 0xD800 && cp < 0xDC00 && t.compare(p, 2, "\\u") == 0) {
          p += 2;
          const unsigned lo = hex4();
          cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
        }
        append_utf8(s, cp);
        break;
      }
      default: fail("bad escape");
      }
    }
    return s;
  }
  Value parse_array() {
    ++p;
    Value v = Value::array();
    skip();
    if (p < t.size() && t[p] == ']') {
      ++p;
      return v;
    }
    for (;;) {
      v.arr.push_back(parse_value());
      skip();
      if (p < t.size() && t[p] == ',') {
        ++p;
        continue;
      }
      if (p < t.size() && t[p] == ']') {
        ++p;
        return v;
      }
      fail("expected ',' or ']'");
    }
  }
  Value parse_object() {
    ++p;
    Value v = Value::object();
    skip();
    if (p < t.size() && t[p] == '}') {
      ++p;
      return v;
    }
    for (;;) {
      skip();
      if (p >= t.size() || t[p] != '"') fail("expected a key");
      const std::string key = parse_string();
      skip();
      if (p >= t.size() || t[p] != ':') fail("expected ':'");
      ++p;
      v.obj[key] = parse_value();
      skip();
      if (p < t.size() && t[p] == ',') {
        ++p;
        continue;
      }
      if (p < t.size() && t[p] == '}') {
        ++p;
        return v;
      }
      fail("expected ',' or '}'");
    }
  }
};

inline Value parse(const std::string &text) { return Parser(text).parse_document(); }

} // namespace lrp_json
