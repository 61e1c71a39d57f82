// hs_inst.hip -- one group of the engine's kernel instantiations (hs_kernels.hpp, HS_INST_GROUP_k) per object file:
//     hipcc ... -DHS_INST=k -c hs_inst.hip -o hs_inst_k.o          (k = 0 .. HS_INST_GROUPS - 1; _native.py builds them in parallel)
// The big kernels (hs_net_async / hs_net_window: ~20 000 instructions each, one per concurrency bound) dominate the build;
// as one translation unit the library took four minutes.
#include "hs_kernels.hpp"

#define HS_CAT2(a, b) a##b
#define HS_CAT(a, b) HS_CAT2(a, b)
HS_CAT(HS_INST_GROUP_, HS_INST)(HS_DEFINE_INST)
