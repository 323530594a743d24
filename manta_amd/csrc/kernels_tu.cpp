// One translation unit per kernel family of the product library: manta_amd/build.py compiles this file once per family with
// -DMANTA_TU=<id> (wave.hpp lists the ids).  The family's kernels are defined here -- device code and host stubs --, everything else
// is only declared; the host sources (api.cpp, MANTA_TU_HOST) launch the kernels through their ordinary external handles.
#include "wave.hpp"
#if MANTA_TU == MANTA_TU_ALL || MANTA_TU == MANTA_TU_HOST
#error "kernels_tu.cpp is compiled once per kernel family: -DMANTA_TU=<MANTA_TU_ASM .. MANTA_TU_GLUE>"
#endif

#if MANTA_TU == MANTA_TU_ASM
#include "assemble_kernels.hpp"
#include "small_asm.hpp"
#elif MANTA_TU == MANTA_TU_ASM_GENERIC
#include "assemble_kernels.hpp"
#elif MANTA_TU == MANTA_TU_GRAPH || MANTA_TU == MANTA_TU_GRAPH_BIG || MANTA_TU == MANTA_TU_CONTIG || MANTA_TU == MANTA_TU_REPEAT
#include "asm_lds.hpp"
#elif MANTA_TU == MANTA_TU_ALIGN0 || MANTA_TU == MANTA_TU_ALIGN1 || MANTA_TU == MANTA_TU_ALIGN2
#include "align_kernels.hpp"
#elif MANTA_TU == MANTA_TU_ALIGN_PAIR
#include "align_pair.hpp"
#elif MANTA_TU == MANTA_TU_JUMP_PAIR
#include "align_jump_pair.hpp"
#elif MANTA_TU == MANTA_TU_GLUE
#include "pipeline_kernels.hpp"
#include "split_kernels.hpp"
#include "read_class_kernels.hpp"
#else
#error "unknown MANTA_TU"
#endif
