// cloudsc2_kern_ad.hip -- one kernel family of the library as a translation unit of its own (cloudsc2_sweep_kernels.hpp says why):
// ad_kernel<F> (both sweeps of CLOUDSC2AD) and ad_reverse_kernel<F> (the reverse sweep alone), every valid flag combination, reached through one accessor.
#include "cloudsc2_sweep_kernels.hpp"

namespace cloudsc2 {
namespace {
C2_VARIANT_TABLE(g_ad_reverse_kernels, ad_reverse_kernel, AdArgs, 64, !(F & C2F_ADNORM) || ((F & C2F_ASSIGN) && !(F & C2F_EVAP)))
C2_VARIANT_TABLE(g_ad_kernels, ad_kernel, AdArgs, 64, C2_AD_FUSED != 0 && !(F & C2F_ADNORM))
}  // namespace
KernelFn<AdArgs> ad_reverse_variant(unsigned f) { return f < g_ad_reverse_kernels.size() ? g_ad_reverse_kernels[f] : nullptr; }
KernelFn<AdArgs> ad_variant(unsigned f) { return f < g_ad_kernels.size() ? g_ad_kernels[f] : nullptr; }
}  // namespace cloudsc2
