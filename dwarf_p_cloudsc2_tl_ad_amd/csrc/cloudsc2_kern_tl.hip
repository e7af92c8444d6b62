// cloudsc2_kern_tl.hip -- one kernel family of the library as a translation unit of its own (cloudsc2_sweep_kernels.hpp says why):
// tl_kernel<F>: CLOUDSC2TL, every valid flag combination, reached through one accessor.
#include "cloudsc2_sweep_kernels.hpp"

namespace cloudsc2 {
namespace {
C2_VARIANT_TABLE(g_tl_kernels, tl_kernel, TlArgs, 64, true)
}  // namespace
KernelFn<TlArgs> tl_variant(unsigned f) { return f < g_tl_kernels.size() ? g_tl_kernels[f] : nullptr; }
}  // namespace cloudsc2
