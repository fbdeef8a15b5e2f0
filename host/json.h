// json.h — a small JSON reader for the host's scene loader (objects keep their
// members sorted by key, which is what the reference's JSON library iterates in:
// `object.begin().key()` in parser/JsonParser.cpp:459 is the alphabetically first key).
#ifndef MIPT_HOST_JSON_H_
#define MIPT_HOST_JSON_H_

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace host {

struct JsonError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

class Json {
 public:
  enum Type { Null, Bool, Number, String, Array, Object };
  Type type = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::map<std::string, Json> obj;

  static Json parse(const std::string &text);

  bool is_null() const { return type == Null; }
  bool is_number() const { return type == Number; }
  bool is_array() const { return type == Array; }
  bool is_object() const { return type == Object; }
  bool has(const std::string &k) const { return type == Object && obj.count(k) != 0; }
  // Throwing accessors (the loader turns a throw into "leave the default", like
  // the IGNORE_MISSING macro of parser/JsonParser.cpp:26-36).
  const Json &at(const std::string &k) const;
  const Json &at(size_t i) const;
  double number() const;
  const std::string &string() const;
  bool boolean() const;
  size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
  const std::string &first_key() const;
};

}  // namespace host
#endif
