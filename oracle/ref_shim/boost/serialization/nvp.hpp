// Test-infrastructure shim (NOT boost): serialization is never exercised by the oracle; only the names used inside
// (never instantiated) serialize() templates and the class-level macros must parse.
#pragma once
#include "boost/serialization/level.hpp"
#include <cassert>
#ifndef MANTA_REF_SHIM_SERIALIZATION
#define MANTA_REF_SHIM_SERIALIZATION
#define BOOST_SERIALIZATION_NVP(x) x
#define BOOST_SERIALIZATION_SPLIT_MEMBER()
namespace boost { namespace serialization {
class access {};
template <class T> const T& make_nvp(const char*, const T& t) { return t; }
}}  // namespace boost::serialization
#endif
