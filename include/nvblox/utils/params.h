// nvblox/utils/params.h -- Param<T> with its constexpr Description {name, default_value, help_string}, as nvblox_ros declares
// them (node_params.hpp:37-300: `constexpr Param<float>::Description kVoxelSizeParamDesc{"voxel_size", .05F, "..."}`,
// `Param<float> voxel_size{kVoxelSizeParamDesc}`, `.get()` nvblox_node.cpp:90-95, `StringParam`), and consumes the core's
// own descriptions (utils.hpp:60-82 declareParameter(desc.name, desc.default_value, desc.help_string)).
#pragma once
#include <string>

namespace nvblox {

template <typename T>
class Param {
 public:
  struct Description { const char* name; T default_value; const char* help_string; };
  Param() = default;
  Param(const Description& d) : value_(d.default_value), name_(d.name), help_(d.help_string) {}   // NOLINT
  const T& get() const { return value_; }
  void set(const T& v) { value_ = v; }
  operator T() const { return value_; }                                                               // NOLINT
  Param& operator=(const T& v) { value_ = v; return *this; }
  const char* name() const { return name_; }
  const char* help() const { return help_; }
 private:
  T value_{};
  const char* name_ = "";
  const char* help_ = "";
};

// string parameters keep a constexpr-friendly description (const char*) and own a std::string value
class StringParam {
 public:
  struct Description { const char* name; const char* default_value; const char* help_string; };
  StringParam() = default;
  StringParam(const Description& d) : value_(d.default_value), name_(d.name), help_(d.help_string) {}   // NOLINT
  const std::string& get() const { return value_; }
  void set(const std::string& v) { value_ = v; }
  StringParam& operator=(const std::string& v) { value_ = v; return *this; }
  const char* name() const { return name_; }
 private:
  std::string value_;
  const char* name_ = "";
  const char* help_ = "";
};

}  // namespace nvblox
