// Test-infrastructure shim (NOT boost)
#pragma once
#include <cerrno>
