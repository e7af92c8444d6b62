// cloudsc2_kern_nl.hip -- one kernel family of the library as a translation unit of its own (cloudsc2_sweep_kernels.hpp says why):
// nl_kernel<F>: SATUR + CLOUDSC2 (and the adjoint's trajectory pass as a kernel of its own), every valid flag combination, reached through one accessor.
#include "cloudsc2_sweep_kernels.hpp"

namespace cloudsc2 {
namespace {
// (the .NOT.LPHYLIN form exists for the plain sweep only: no shipped main uses it, the Taylor test's perturbed runs and the
//  adjoint's trajectory pass belong to CLOUDSC2TL / CLOUDSC2AD, which have the LPHYLIN form alone; the trajectory pass differs from
//  the plain NL sweep only with the evaporation branch: the cover checkpoint)
C2_VARIANT_TABLE(g_nl_kernels, nl_kernel, NlArgs, 128,
                 (F & C2F_CKPT) ? ((F & C2F_EVAP) && !(F & (C2F_PERT | C2F_NOLIN))) : !((F & C2F_NOLIN) && (F & C2F_PERT)))
}  // namespace
KernelFn<NlArgs> nl_variant(unsigned f) { return f < g_nl_kernels.size() ? g_nl_kernels[f] : nullptr; }
}  // namespace cloudsc2
