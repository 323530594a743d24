// Test-infrastructure shim (NOT boost)
#pragma once
#define BOOST_THROW_EXCEPTION(x) throw(x)
