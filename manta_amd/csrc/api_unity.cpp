// The three host sources as ONE translation unit: the wave-emulator build of tests/emu (kernels are ordinary functions there, defined in
// the headers: a second translation unit would define them again) and the single-unit developer variants of manta_amd/build.py.
#include "api.cpp"
#include "api_batch.cpp"
#include "api_reads.cpp"
