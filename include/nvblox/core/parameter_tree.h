// nvblox/core/parameter_tree.h -- parameters::ParameterTreeNode: the name/value tree the node prints at start-up.
// Use in the reference: member `ParameterTreeNode parameter_tree_{"nvblox_node", {}}` (nvblox_node.hpp:556, fuser_node.hpp:101);
// leaves `ParameterTreeNode(desc.name, value)` pushed onto `tree->children().value()` (node_params.cpp:36-43,56-63);
// `parameter_tree_.children().value().push_back(multi_mapper_->getParameterTree())` and `parameterTreeToString(tree)`
// (nvblox_node.cpp:119-124).
#pragma once
#include <optional>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace nvblox {
namespace parameters {

class ParameterTreeNode {
 public:
  // non-leaf: a name and its children
  ParameterTreeNode(const std::string& name, const std::vector<ParameterTreeNode>& children) : name_(name), children_(children) {}
  // leaves: a name and a value (kept as text).  Templates, so that `ParameterTreeNode{"nvblox_node", {}}` (nvblox_node.hpp:556)
  // can only mean "no children yet".
  template <typename T, std::enable_if_t<std::is_convertible<T, std::string>::value && !std::is_arithmetic<std::decay_t<T>>::value &&
                                         !std::is_null_pointer<std::decay_t<T>>::value, int> = 0>
  ParameterTreeNode(const std::string& name, const T& value) : name_(name), value_(std::string(value)) {}
  template <typename T, std::enable_if_t<std::is_arithmetic<T>::value || std::is_enum<T>::value, int> = 0>
  ParameterTreeNode(const std::string& name, T value) : name_(name) {
    std::ostringstream o;
    if constexpr (std::is_same<T, bool>::value) o << (value ? "true" : "false");
    else if constexpr (std::is_enum<T>::value) o << static_cast<long long>(value);
    else o << value;
    value_ = o.str();
  }
  const std::string& name() const { return name_; }
  const std::optional<std::string>& value() const { return value_; }       // leaves
  std::optional<std::vector<ParameterTreeNode>>& children() { return children_; }
  const std::optional<std::vector<ParameterTreeNode>>& children() const { return children_; }
  bool isLeaf() const { return !children_.has_value(); }
 private:
  std::string name_;
  std::optional<std::string> value_;
  std::optional<std::vector<ParameterTreeNode>> children_;
};

namespace detail {
inline void print(const ParameterTreeNode& n, int depth, std::ostringstream* o) {
  for (int i = 0; i < depth; i++) *o << "  ";
  if (n.isLeaf()) { *o << n.name() << ": " << n.value().value_or("") << "\n"; return; }
  *o << n.name() << ":\n";
  for (const ParameterTreeNode& c : n.children().value()) print(c, depth + 1, o);
}
}  // namespace detail

inline std::string parameterTreeToString(const ParameterTreeNode& root) {
  std::ostringstream o;
  detail::print(root, 0, &o);
  return o.str();
}

}  // namespace parameters
}  // namespace nvblox
