// Oracle shim (test infrastructure, NOT product code).
// Plain-C++ stand-in for the protoc output of yadcc/api/env_desc.proto:20-28.
#ifndef ORACLE_SHIM_ENV_DESC_PB_H_
#define ORACLE_SHIM_ENV_DESC_PB_H_
#include <string>
namespace yadcc {
class EnvironmentDesc {
 public:
  const std::string& compiler_digest() const { return compiler_digest_; }
  void set_compiler_digest(std::string v) { compiler_digest_ = std::move(v); }

 private:
  std::string compiler_digest_;
};
}  // namespace yadcc
#endif
