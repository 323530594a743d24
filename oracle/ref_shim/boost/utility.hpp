// Test-infrastructure shim (NOT boost)
#pragma once
#include "boost/noncopyable.hpp"
