// Test-infrastructure shim (NOT boost): boost::any as an opaque holder (the candidate VCF writer only passes it through)
#pragma once
#include <memory>
#include <typeinfo>
namespace boost {
class any {
public:
  any() {}
  template <typename T>
  any(const T& v) : _p(std::make_shared<holder<T>>(v))
  {
  }
  bool empty() const { return !_p; }
  struct base {
    virtual ~base() {}
    virtual const std::type_info& type() const = 0;
  };
  template <typename T>
  struct holder : base {
    explicit holder(const T& v) : value(v) {}
    const std::type_info& type() const override { return typeid(T); }
    T value;
  };
  std::shared_ptr<base> _p;
};
template <typename T>
T any_cast(const any& a)
{
  typedef typename std::remove_cv<typename std::remove_reference<T>::type>::type V;
  return static_cast<any::holder<V>*>(a._p.get())->value;
}
}  // namespace boost
