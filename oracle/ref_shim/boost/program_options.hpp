// Test-infrastructure shim (NOT boost): only the type names that appear in option-parser DECLARATIONS
#pragma once
#include <string>
namespace boost { namespace program_options {
class options_description { public: options_description() {} explicit options_description(const std::string&) {} };
class variables_map {};
}}  // namespace boost::program_options
