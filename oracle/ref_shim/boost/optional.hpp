// Test-infrastructure shim (NOT boost): minimal optional
#pragma once
namespace boost {
template <typename T> class optional {
 public:
  optional() : _has(false), _v() {}
  optional(const T& v) : _has(true), _v(v) {}
  explicit operator bool() const { return _has; }
  bool operator!() const { return !_has; }
  const T& operator*() const { return _v; }
  T& operator*() { return _v; }
  const T* operator->() const { return &_v; }
  const T& get() const { return _v; }
  void reset() { _has = false; }
  optional& operator=(const T& v) { _has = true; _v = v; return *this; }
 private:
  bool _has;
  T _v;
};
}  // namespace boost
