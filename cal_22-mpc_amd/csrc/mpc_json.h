// mpc_json.h -- a small JSON reader (RFC 8259 subset: no \u surrogate pairing
// beyond the BMP) used to read VPC configuration files.  The reference reads
// them with jsoncpp (VPC.cpp:84-95), which is not vendored; the accessors below
// follow the jsoncpp conversions VPC::parseConfig relies on: a missing member
// is null, null converts to 0 / 0.0 / false / "", a real converts to int by
// truncation, numbers convert to bool as != 0.
#pragma once

#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace mpcjson {

struct Value {
  enum Type { Null, Bool, Int, Real, String, Array, Object } type = Null;
  bool b = false;
  int64_t i = 0;
  double d = 0.0;
  std::string s;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  static const Value &null_value()
  {
    static const Value v;
    return v;
  }
  bool isNull() const { return type == Null; }
  size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
  const Value &operator[](const std::string &key) const
  {
    if (type == Object)
      for (const auto &kv : obj)
        if (kv.first == key) return kv.second;
    return null_value();
  }
  const Value &operator[](const char *key) const { return (*this)[std::string(key)]; }
  const Value &operator[](int idx) const
  {
    if (type == Array && idx >= 0 && (size_t)idx < arr.size()) return arr[(size_t)idx];
    return null_value();
  }
  // conversions; ok is cleared when jsoncpp would have thrown
  int asInt(bool &ok) const
  {
    switch (type) {
    case Null: return 0;
    case Bool: return b ? 1 : 0;
    case Int:
      if (i < INT32_MIN || i > INT32_MAX) ok = false;
      return (int)i;
    case Real:
      if (!(d >= -2147483648.0 && d <= 2147483647.0)) { ok = false; return 0; }
      return (int)d;
    default: ok = false; return 0;
    }
  }
  float asFloat(bool &ok) const
  {
    switch (type) {
    case Null: return 0.0f;
    case Bool: return b ? 1.0f : 0.0f;
    case Int: return (float)(double)i;
    case Real: return (float)d;
    default: ok = false; return 0.0f;
    }
  }
  bool asBool(bool &ok) const
  {
    switch (type) {
    case Null: return false;
    case Bool: return b;
    case Int: return i != 0;
    case Real: return d != 0.0;
    default: ok = false; return false;
    }
  }
  std::string asString(bool &ok) const
  {
    switch (type) {
    case Null: return "";
    case String: return s;
    case Bool: return b ? "true" : "false";
    default: ok = false; return "";
    }
  }
};

class Parser {
public:
  Parser(const char *p, size_t n) : p_(p), e_(p + n) {}
  bool parse(Value &out, std::string &err)
  {
    skip();
    if (!value(out, 0)) { err = err_; return false; }
    skip();
    if (p_ != e_) { err = "trailing characters after JSON value"; return false; }
    return true;
  }

private:
  const char *p_, *e_;
  std::string err_;
  bool fail(const char *m) { if (err_.empty()) err_ = m; return false; }
  void skip()
  {
    for (;;) {
      while (p_ < e_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++;
      // jsoncpp accepts comments by default
      if (p_ + 1 < e_ && p_[0] == '/' && p_[1] == '/') {
        while (p_ < e_ && *p_ != '\n') p_++;
        continue;
      }
      if (p_ + 1 < e_ && p_[0] == '/' && p_[1] == '*') {
        p_ += 2;
        while (p_ + 1 < e_ && !(p_[0] == '*' && p_[1] == '/')) p_++;
        p_ = (p_ + 1 < e_) ? p_ + 2 : e_;
        continue;
      }
      break;
    }
  }
  bool lit(const char *w)
  {
    const char *q = p_;
    while (*w) {
      if (q >= e_ || *q != *w) return false;
      q++; w++;
    }
    p_ = q;
    return true;
  }
  bool string(std::string &out)
  {
    if (p_ >= e_ || *p_ != '"') return fail("expected string");
    p_++;
    while (p_ < e_ && *p_ != '"') {
      char c = *p_++;
      if (c != '\\') { out.push_back(c); continue; }
      if (p_ >= e_) return fail("bad escape");
      char x = *p_++;
      switch (x) {
      case '"': out.push_back('"'); break;
      case '\\': out.push_back('\\'); break;
      case '/': out.push_back('/'); break;
      case 'b': out.push_back('\b'); break;
      case 'f': out.push_back('\f'); break;
      case 'n': out.push_back('\n'); break;
      case 'r': out.push_back('\r'); break;
      case 't': out.push_back('\t'); break;
      case 'u': {
        if (e_ - p_ < 4) return fail("bad \\u escape");
        unsigned cp = 0;
        for (int k = 0; k < 4; k++) {
          char h = *p_++;
          cp <<= 4;
          if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
          else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
          else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
          else return fail("bad \\u escape");
        }
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        break;
      }
      default: return fail("bad escape");
      }
    }
    if (p_ >= e_) return fail("unterminated string");
    p_++;
    return true;
  }
  bool number(Value &out)
  {
    const char *s = p_;
    bool real = false;
    if (p_ < e_ && (*p_ == '-' || *p_ == '+')) p_++;
    while (p_ < e_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) {
      if (*p_ == '.' || *p_ == 'e' || *p_ == 'E') real = true;
      p_++;
    }
    if (p_ == s) return fail("expected value");
    std::string t(s, p_);
    char *end = nullptr;
    if (!real) {
      errno = 0;
      long long v = std::strtoll(t.c_str(), &end, 10);
      if (end && *end == 0 && errno == 0) {
        out.type = Value::Int;
        out.i = v;
        out.d = (double)v;
        return true;
      }
    }
    double dv = std::strtod(t.c_str(), &end);
    if (!end || *end != 0) return fail("bad number");
    out.type = Value::Real;
    out.d = dv;
    return true;
  }
  bool value(Value &out, int depth)
  {
    if (depth > 64) return fail("nesting too deep");
    skip();
    if (p_ >= e_) return fail("unexpected end of input");
    char c = *p_;
    if (c == '{') {
      p_++;
      out.type = Value::Object;
      skip();
      if (p_ < e_ && *p_ == '}') { p_++; return true; }
      for (;;) {
        skip();
        std::string k;
        if (!string(k)) return false;
        skip();
        if (p_ >= e_ || *p_ != ':') return fail("expected ':'");
        p_++;
        Value v;
        if (!value(v, depth + 1)) return false;
        out.obj.emplace_back(std::move(k), std::move(v));
        skip();
        if (p_ < e_ && *p_ == ',') { p_++; continue; }
        if (p_ < e_ && *p_ == '}') { p_++; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      p_++;
      out.type = Value::Array;
      skip();
      if (p_ < e_ && *p_ == ']') { p_++; return true; }
      for (;;) {
        Value v;
        if (!value(v, depth + 1)) return false;
        out.arr.push_back(std::move(v));
        skip();
        if (p_ < e_ && *p_ == ',') { p_++; continue; }
        if (p_ < e_ && *p_ == ']') { p_++; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      out.type = Value::String;
      return string(out.s);
    }
    if (lit("true")) { out.type = Value::Bool; out.b = true; return true; }
    if (lit("false")) { out.type = Value::Bool; out.b = false; return true; }
    if (lit("null")) { out.type = Value::Null; return true; }
    return number(out);
  }
};

inline bool parse(const std::string &text, Value &out, std::string &err)
{
  Parser p(text.data(), text.size());
  return p.parse(out, err);
}

}  // namespace mpcjson
