#include "json.h"

#include <cctype>
#include <cstdlib>

namespace host {
namespace {

struct Reader {
  const std::string &s;
  size_t i = 0;
  int depth = 0;  // open arrays and objects: bounded, a scene nests a few dozen levels at most
  static constexpr int kMaxDepth = 512;
  explicit Reader(const std::string &t) : s(t) {}

  [[noreturn]] void fail(const char *what) const { throw JsonError(std::string("JSON: ") + what + " at offset " + std::to_string(i)); }

  void ws() {
    for (;;) {
      while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
      if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') {  // line comments are tolerated
        while (i < s.size() && s[i] != '\n') i++;
        continue;
      }
      if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') {
        i += 2;
        while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) i++;
        i = i + 2 <= s.size() ? i + 2 : s.size();
        continue;
      }
      return;
    }
  }

  Json value() {
    ws();
    if (i >= s.size()) fail("unexpected end");
    char c = s[i];
    if (c == '{' || c == '[') {
      if (depth >= kMaxDepth) fail("nested too deeply");
      depth++;
      Json j = c == '{' ? object() : array();
      depth--;
      return j;
    }
    if (c == '"') {
      Json j;
      j.type = Json::String;
      j.str = string();
      return j;
    }
    if (s.compare(i, 4, "null") == 0) { i += 4; return Json(); }
    if (s.compare(i, 4, "true") == 0) { i += 4; Json j; j.type = Json::Bool; j.b = true; return j; }
    if (s.compare(i, 5, "false") == 0) { i += 5; Json j; j.type = Json::Bool; j.b = false; return j; }
    if (c == '-' || std::isdigit((unsigned char)c)) {
      const char *start = s.c_str() + i;
      char *end = nullptr;
      double v = std::strtod(start, &end);
      if (end == start) fail("bad number");
      i += (size_t)(end - start);
      Json j;
      j.type = Json::Number;
      j.num = v;
      return j;
    }
    fail("unexpected character");
  }

  std::string string() {
    std::string out;
    i++;  // opening quote
    while (i < s.size() && s[i] != '"') {
      char c = s[i++];
      if (c == '\\') {
        if (i >= s.size()) fail("bad escape");
        char e = s[i++];
        switch (e) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            if (i + 4 > s.size()) fail("bad \\u escape");
            unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16);
            i += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += e;
        }
      } else {
        out += c;
      }
    }
    if (i >= s.size()) fail("unterminated string");
    i++;
    return out;
  }

  Json array() {
    Json j;
    j.type = Json::Array;
    i++;
    ws();
    if (i < s.size() && s[i] == ']') { i++; return j; }
    for (;;) {
      j.arr.push_back(value());
      ws();
      if (i >= s.size()) fail("unterminated array");
      if (s[i] == ',') { i++; continue; }
      if (s[i] == ']') { i++; return j; }
      fail("expected , or ]");
    }
  }

  Json object() {
    Json j;
    j.type = Json::Object;
    i++;
    ws();
    if (i < s.size() && s[i] == '}') { i++; return j; }
    for (;;) {
      ws();
      if (i >= s.size() || s[i] != '"') fail("expected a key");
      std::string k = string();
      ws();
      if (i >= s.size() || s[i] != ':') fail("expected :");
      i++;
      j.obj[k] = value();
      ws();
      if (i >= s.size()) fail("unterminated object");
      if (s[i] == ',') { i++; continue; }
      if (s[i] == '}') { i++; return j; }
      fail("expected , or }");
    }
  }
};

}  // namespace

Json Json::parse(const std::string &text) {
  Reader r(text);
  Json j = r.value();
  r.ws();
  if (r.i != text.size()) r.fail("trailing characters");
  return j;
}

const Json &Json::at(const std::string &k) const {
  if (type != Object) throw JsonError("JSON: not an object");
  auto it = obj.find(k);
  if (it == obj.end()) throw JsonError("JSON: key '" + k + "' not found");
  return it->second;
}
const Json &Json::at(size_t i) const {
  if (type != Array || i >= arr.size()) throw JsonError("JSON: array index out of range");
  return arr[i];
}
double Json::number() const {
  if (type != Number) throw JsonError("JSON: not a number");
  return num;
}
const std::string &Json::string() const {
  if (type != String) throw JsonError("JSON: not a string");
  return str;
}
bool Json::boolean() const {
  if (type != Bool) throw JsonError("JSON: not a boolean");
  return b;
}
const std::string &Json::first_key() const {
  if (type != Object || obj.empty()) throw JsonError("JSON: expected a non-empty object");
  return obj.begin()->first;
}

}  // namespace host
