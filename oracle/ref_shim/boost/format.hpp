// Test-infrastructure shim (NOT boost): the subset of boost::format the reference's VCF writer and id generator use --
// positional "%N%" directives and printf-style "%i" / "%s" / "%d", arguments fed with operator%, str(), and re-use of a
// format object (feeding an argument to a completed format starts over, as boost does).
#pragma once
#include <sstream>
#include <string>
#include <vector>
namespace boost {
class format {
public:
  format(const char* f) : _fmt(f) { scan(); }
  format(const std::string& f) : _fmt(f) { scan(); }
  template <typename T>
  format& operator%(const T& v)
  {
    if (_args.size() >= _expected) _args.clear();
    std::ostringstream os;
    os << v;
    _args.push_back(os.str());
    return *this;
  }
  std::string str() const
  {
    std::string out;
    size_t      seq = 0;
    for (size_t i = 0; i < _fmt.size(); ++i) {
      if (_fmt[i] != '%') {
        out.push_back(_fmt[i]);
        continue;
      }
      if (i + 1 < _fmt.size() && _fmt[i + 1] == '%') {
        out.push_back('%');
        ++i;
        continue;
      }
      size_t j = i + 1, n = 0;
      while (j < _fmt.size() && _fmt[j] >= '0' && _fmt[j] <= '9') n = n * 10 + size_t(_fmt[j++] - '0');
      if (j < _fmt.size() && _fmt[j] == '%' && j > i + 1) {  // positional
        if (n >= 1 && n <= _args.size()) out += _args[n - 1];
        i = j;
      } else {  // printf style: one conversion character
        if (seq < _args.size()) out += _args[seq];
        ++seq;
        i = j;
      }
    }
    return out;
  }

private:
  void scan()
  {
    size_t maxPos = 0, seq = 0;
    for (size_t i = 0; i < _fmt.size(); ++i) {
      if (_fmt[i] != '%') continue;
      if (i + 1 < _fmt.size() && _fmt[i + 1] == '%') {
        ++i;
        continue;
      }
      size_t j = i + 1, n = 0;
      while (j < _fmt.size() && _fmt[j] >= '0' && _fmt[j] <= '9') n = n * 10 + size_t(_fmt[j++] - '0');
      if (j < _fmt.size() && _fmt[j] == '%' && j > i + 1) {
        if (n > maxPos) maxPos = n;
        i = j;
      } else {
        ++seq;
        i = j;
      }
    }
    _expected = (maxPos > seq) ? maxPos : seq;
  }
  std::string              _fmt;
  std::vector<std::string> _args;
  size_t                   _expected = 0;
};
inline std::string str(const format& f) { return f.str(); }
inline std::ostream& operator<<(std::ostream& os, const format& f) { return os << f.str(); }
}  // namespace boost
