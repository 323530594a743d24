// Test-infrastructure shim (NOT boost): minimal boost::exception surface used by common/Exceptions.hpp.
#pragma once
#include <exception>
#include <string>
#include <typeinfo>
namespace boost {
class exception {
 public:
  virtual ~exception() noexcept {}
 protected:
  exception() {}
};
template <class Tag, class T> struct error_info {
  typedef T value_type;
  explicit error_info(const T& v) : value(v) {}
  T value;
};
template <class E, class Tag, class T> const E& operator<<(const E& e, const error_info<Tag, T>&) { return e; }
inline std::string diagnostic_information(const exception& e) {
  const std::exception* se = dynamic_cast<const std::exception*>(&e);
  return se ? std::string(se->what()) : std::string("boost::exception (shim)");
}
}  // namespace boost
