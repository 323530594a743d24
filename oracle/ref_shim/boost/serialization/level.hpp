// Test-infrastructure shim (NOT boost): serialization traits are irrelevant to the oracle build.
#pragma once
#define BOOST_CLASS_IMPLEMENTATION(T, L)
