// cloudsc2_kern_taylor.hip -- one kernel family of the library as a translation unit of its own (cloudsc2_sweep_kernels.hpp says why):
// taylor_kernel<F>: the ten perturbed NL runs of the Taylor test on the lanes of a wave, every valid flag combination, reached through one accessor.
#include "cloudsc2_sweep_kernels.hpp"

namespace cloudsc2 {
namespace {
C2_VARIANT_TABLE(g_taylor_kernels, taylor_kernel, TaylorArgs, 64, !(F & (C2F_PERT | C2F_CKPT)))
}  // namespace
KernelFn<TaylorArgs> taylor_variant(unsigned f) { return f < g_taylor_kernels.size() ? g_taylor_kernels[f] : nullptr; }
}  // namespace cloudsc2
