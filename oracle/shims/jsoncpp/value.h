// Oracle shim (test infrastructure, NOT product code).
// A small recording Json::Value: the reference only uses it for the debug dump
// (task_dispatcher.cc:538-614). It keeps what the reference stores (objects with sorted keys,
// like jsoncpp writes them; arrays; scalars) and can write itself out as JSON text, so that
// the dump of the MI355X host class can be compared with the reference's own dump
// (oracle/ref_driver.cc: ref_dump_internals).
#ifndef ORACLE_SHIM_JSONCPP_VALUE_H_
#define ORACLE_SHIM_JSONCPP_VALUE_H_
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <type_traits>
#include <vector>
namespace Json {
using UInt64 = unsigned long long;
using Int64 = long long;
class Value {
 public:
  Value() = default;
  Value(const std::string& s) : kind_(kString), str_(s) {}
  Value(const char* s) : kind_(kString), str_(s) {}
  Value(bool b) : kind_(kBool), int_(b) {}
  template <class T, class = std::enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool>>>
  Value(T v) : kind_(std::is_signed_v<T> ? kInt : kUInt), int_((long long)v), uint_((unsigned long long)v) {}
  Value& operator[](int i) {
    kind_ = kArray;
    if ((std::size_t)i >= arr_.size()) arr_.resize(i + 1);
    return arr_[i];
  }
  Value& operator[](const char* k) { return (*this)[std::string(k)]; }
  Value& operator[](const std::string& k) {
    kind_ = kObject;
    return obj_[k];
  }
  Value& append(const Value& v) {
    kind_ = kArray;
    arr_.push_back(v);
    return arr_.back();
  }
  std::string Dump() const {
    std::string out;
    Write(&out);
    return out;
  }

 private:
  enum Kind { kNull, kInt, kUInt, kBool, kString, kArray, kObject };
  static void Escape(const std::string& s, std::string* out) {
    out->push_back('"');
    for (unsigned char c : s) {
      if (c == '"' || c == '\\') {
        out->push_back('\\');
        out->push_back((char)c);
      } else if (c < 0x20) {
        char buf[8];
        std::snprintf(buf, sizeof(buf), "\\u%04x", c);
        *out += buf;
      } else {
        out->push_back((char)c);
      }
    }
    out->push_back('"');
  }
  void Write(std::string* out) const {
    switch (kind_) {
      case kNull: *out += "null"; break;
      case kInt: *out += std::to_string(int_); break;
      case kUInt: *out += std::to_string(uint_); break;
      case kBool: *out += int_ ? "true" : "false"; break;
      case kString: Escape(str_, out); break;
      case kArray: {
        out->push_back('[');
        for (std::size_t i = 0; i != arr_.size(); ++i) {
          if (i) out->push_back(',');
          arr_[i].Write(out);
        }
        out->push_back(']');
        break;
      }
      case kObject: {
        out->push_back('{');
        bool first = true;
        for (auto&& [k, v] : obj_) {
          if (!first) out->push_back(',');
          first = false;
          Escape(k, out);
          out->push_back(':');
          v.Write(out);
        }
        out->push_back('}');
        break;
      }
    }
  }
  Kind kind_ = kNull;
  long long int_ = 0;
  unsigned long long uint_ = 0;
  std::string str_;
  std::vector<Value> arr_;
  std::map<std::string, Value> obj_;
};
}  // namespace Json
#endif
