// Test-infrastructure shim (NOT boost): cpu_timer surface used by blt_util/time_util.hpp (wall clock only)
#pragma once
#include <chrono>
#include <cstdint>
namespace boost { namespace timer {
typedef std::int_least64_t nanosecond_type;
struct cpu_times {
  nanosecond_type wall = 0, user = 0, system = 0;
  void clear() { wall = user = system = 0; }
};
class cpu_timer {
 public:
  cpu_timer() { start(); }
  void start() { _stopped = false; _acc = cpu_times(); _t0 = std::chrono::steady_clock::now(); }
  void stop() { if (!_stopped) { _acc.wall += since(); _stopped = true; } }
  void resume() { if (_stopped) { _stopped = false; _t0 = std::chrono::steady_clock::now(); } }
  bool is_stopped() const { return _stopped; }
  cpu_times elapsed() const { cpu_times t(_acc); if (!_stopped) t.wall += since(); return t; }
 private:
  nanosecond_type since() const { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - _t0).count(); }
  std::chrono::steady_clock::time_point _t0;
  cpu_times _acc;
  bool _stopped = false;
};
}}  // namespace boost::timer
