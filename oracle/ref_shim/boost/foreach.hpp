// Test-infrastructure shim (NOT boost): the one loop macro the reference hot path uses
// (assembly/IterativeAssembler.cpp:817, blt_util/align_path.cpp) expressed with C++14 range-for.
#pragma once
#include <iterator>
namespace manta_ref_shim {
template <typename C> struct reversed_view {
  C& c;
  auto begin() const -> decltype(c.rbegin()) { return c.rbegin(); }
  auto end() const -> decltype(c.rend()) { return c.rend(); }
};
template <typename C> reversed_view<C> reversed(C& c) { return reversed_view<C>{c}; }
template <typename C> reversed_view<const C> reversed(const C& c) { return reversed_view<const C>{c}; }
}  // namespace manta_ref_shim
#define BOOST_FOREACH(decl, container) for (decl : container)
#define BOOST_REVERSE_FOREACH(decl, container) for (decl : ::manta_ref_shim::reversed(container))
