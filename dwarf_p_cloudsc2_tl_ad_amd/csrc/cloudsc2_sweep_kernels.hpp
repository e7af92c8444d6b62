// cloudsc2_sweep_kernels.hpp -- the __global__ wrappers of the column sweeps (NL, TL, AD, the Taylor test's lambda sweep), their
// compile-time variant tables and what they share.  The library is built from five translation units so that an edit to one sweep
// does not rebuild every variant table (384 slots, ~65 s as one unit): cloudsc2_kernels.hip (host code, launchers, the data-format
// and norm kernels) and one unit per kernel family -- cloudsc2_kern_{nl,tl,ad,taylor}.hip --, each of which instantiates its table and
// exports it through one accessor (nl_variant(F) ...).  -DC2_SINGLE_TU compiles everything as ONE unit again (cloudsc2_kernels.hip
// then includes the family files): the experiment builds of `make variant`, `make asm`, `make resources` and the -DC2_WAVE_TIMES
// diagnostic, whose log pointer is a __device__ global that separate code objects cannot share.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <utility>

#include "cloudsc2_column.hpp"

namespace cloudsc2 {

// ---------------------------------------------------------------------------------------------------------
// kernels: thin wrappers around the per-column functions of cloudsc2_column.hpp.  Every kernel has ONE by-value
// argument block; the device code reads it in place from the kernel-argument segment (scalar cache).
// ---------------------------------------------------------------------------------------------------------
#ifndef C2_BLOCK
#define C2_BLOCK 128
#endif
constexpr int kBlock = C2_BLOCK;
// minimum waves per SIMD requested from the register allocator (0 = let the compiler decide)
#ifndef C2_NL_WAVES
#define C2_NL_WAVES 0
#endif
#ifndef C2_TL_WAVES
#define C2_TL_WAVES 0
#endif
#ifndef C2_AD_WAVES
#define C2_AD_WAVES 0
#endif
#define C2_BOUNDS(w) __launch_bounds__(kBlock, (w) > 0 ? (w) : 1)

__device__ __forceinline__ long long global_column() { return (long long)blockIdx.x * blockDim.x + threadIdx.x; }

#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
__device__ __forceinline__ const C2_CONST_AS T* kernarg() {
  return (const C2_CONST_AS T*)__builtin_amdgcn_kernarg_segment_ptr();
}
#define C2_KERNEL_BODY(call) call
#else
#define C2_KERNEL_BODY(call)
#endif

// The NL variants without the evaporation branch fit 168 VGPRs (3 waves per SIMD) even with the two-level-deep
// prefetch; asking for it keeps the allocator from spending a few registers too many.  The others take what they need.
// -DC2_WAVE_TIMES (diagnostic build, tools/wave_times.py): every wave of the NL kernel logs when it started and ended (the 100 MHz
// constant clock) and where it ran (HW_ID, XCC_ID) -- how evenly a launch's waves start, progress and finish.
#ifdef C2_WAVE_TIMES
#ifndef C2_SINGLE_TU
#error "-DC2_WAVE_TIMES needs -DC2_SINGLE_TU (one code object: the log pointer is a __device__ global)"
#endif
__device__ unsigned long long* g_wave_log = nullptr;  // [wave][4]: start, end, HW_ID, XCC_ID
#define C2_WAVE_LOG_BEGIN const unsigned long long c2_t0 = __builtin_amdgcn_s_memrealtime();
#define C2_WAVE_LOG_END                                                                                         \
  if (g_wave_log && (threadIdx.x & 63) == 0) {                                                                  \
    unsigned long long* e = g_wave_log + 4 * (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);         \
    e[0] = c2_t0; e[1] = __builtin_amdgcn_s_memrealtime();                                                     \
    e[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); e[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);        \
  }
#else
#define C2_WAVE_LOG_BEGIN
#define C2_WAVE_LOG_END
#endif
template <unsigned F>
__global__ void __launch_bounds__(kBlock, (C2_NL_WAVES > 0) ? C2_NL_WAVES : ((F & C2F_EVAP) ? 1 : 3)) nl_kernel(NlArgs args) {
  C2_KERNEL_BODY(C2_WAVE_LOG_BEGIN);
  C2_KERNEL_BODY((nl_column<F>(global_column(), kernarg<NlArgs>())));
  C2_KERNEL_BODY(C2_WAVE_LOG_END);
}

// fp32 only: the TL variants with 32-bit offsets and without the evaporation branch need 173 VGPRs; held to 168 (3 waves
// per SIMD) they spill at most 7 dwords, and all 2500 waves of a 160 000-column launch are resident at once instead of
// 2048 + 452 (0.94 -> 0.88 ms).  Every other TL variant, and the fp64 ones, spill heavily below what they ask for.
template <unsigned F>
__global__ void __launch_bounds__(kBlock, (C2_TL_WAVES > 0) ? C2_TL_WAVES
                                          : (sizeof(real_t) == 4 && (F & C2F_OFF32) && !(F & C2F_EVAP)) ? 3 : 1)
tl_kernel(TlArgs args) {
  C2_KERNEL_BODY(C2_WAVE_LOG_BEGIN);
#if C2_TL_DMA  // experiment build (profiles/EXPERIMENTS.md section 6): the look-ahead through LDS-DMA; NPROMA 128, fp64 only
  static_assert(sizeof(real_t) == 8 && kBlock == 128, "C2_TL_DMA: fp64, workgroups of 128 threads");
  C2_KERNEL_BODY((tl_column_dma<F>(kernarg<TlArgs>())));
#else
  C2_KERNEL_BODY((tl_column<F>(global_column(), kernarg<TlArgs>())));
#endif
  C2_KERNEL_BODY(C2_WAVE_LOG_END);
}

// C2_AD_FUSED=1: one kernel runs a column's trajectory pass and then its reverse pass (waves in the bandwidth-heavy
// forward phase and waves in the arithmetic-heavy reverse phase share the CUs); 0: two kernels in stream order;
// 2: both are built and launches of at most kAdSplitBelow columns take the two-kernel form.  fp64: the two forms measure
// the same at every size (both passes need one wave per SIMD's worth of registers in the fused kernel anyway).  fp32: the
// trajectory pass alone runs six waves per SIMD instead of the fused kernel's two, which is worth 7 % when the whole
// launch is one round of waves (160 000 columns: 1.71 -> 1.59 ms) and nothing at 1 M columns (9.09 vs 9.17 ms).
#ifndef C2_AD_FUSED
#if defined(CLOUDSC2_SINGLE)
#define C2_AD_FUSED 2
#else
#define C2_AD_FUSED 1
#endif
#endif
constexpr long long kAdSplitBelow = 400000;
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ void atomic_max_pos(double* addr, double v) {
  // v >= 0: the IEEE bit pattern of non-negative doubles orders like unsigned integers
  atomicMax((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
}
template <unsigned F>
__global__ void C2_BOUNDS(C2_AD_WAVES) ad_reverse_kernel(AdArgs args) {
  C2_KERNEL_BODY(C2_WAVE_LOG_BEGIN);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((F & C2F_ADNORM) != 0) {  // the adjoint test's norms formed in the sweep: the wave's largest |norm3| joins the global one
    const double m = wave_max(ad_reverse_column<F>(global_column(), kernarg<AdArgs>()));
    if ((threadIdx.x & 63) == 0) atomic_max_pos(kernarg<AdArgs>()->gmax, m);
  } else {
    ad_reverse_column<F>(global_column(), kernarg<AdArgs>());
  }
#endif
  C2_KERNEL_BODY(C2_WAVE_LOG_END);
}
template <unsigned F>
__global__ void C2_BOUNDS(C2_AD_WAVES) ad_kernel(AdArgs args) {
  C2_KERNEL_BODY(C2_WAVE_LOG_BEGIN);
  C2_KERNEL_BODY((nl_column<(F & ~C2F_ASSIGN) | C2F_CKPT>(global_column(), &kernarg<AdArgs>()->nl)));
  C2_KERNEL_BODY((ad_reverse_column<F>(global_column(), kernarg<AdArgs>())));
  C2_KERNEL_BODY(C2_WAVE_LOG_END);
}

// The ten perturbed NL runs of the Taylor test in one sweep, the lambdas on the lanes (taylor_column): the grid is over THREADS,
// 64 per kTaylorCols columns.  A wave reads 6 columns = 48 bytes of every 128-byte line it touches, so two or three consecutive
// waves share each line -- the one sweep whose workgroups share data.  Blocks are dealt round-robin over the 8 XCDs (b and b + 8
// share one, each XCD with its own L2): consecutive LOGICAL blocks are therefore mapped to physical blocks 8 apart, so that the
// waves sharing a line sit on one XCD and its L2 fetches the line once (the grid is a multiple of 8 blocks; logical blocks past the
// end find no column and leave).  Measured (rocprofv3 --pmc FETCH_SIZE, 160 000 columns): see profiles/r03_taylor_sweep_ab.txt.
template <unsigned F>
__global__ void __launch_bounds__(kBlock) taylor_kernel(TaylorArgs args) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned per_xcd = gridDim.x >> 3;
  const long long block = (long long)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  taylor_column<F>(block * blockDim.x + threadIdx.x, kernarg<TaylorArgs>());
#endif
}

// Variant tables: kernel<F> for every valid flag combination F, indexed by F (see C2F_* in cloudsc2_column.hpp).
template <class Args> using KernelFn = void (*)(Args);
#define C2_VARIANT_TABLE(table, kern, Args, NF, valid_expr)                                                        \
  template <unsigned F> constexpr KernelFn<Args> table##_entry() {                                                 \
    if constexpr (valid_expr) return kern<F>; else return nullptr;                                                 \
  }                                                                                                                \
  template <unsigned... F> constexpr std::array<KernelFn<Args>, sizeof...(F)> table##_make(                        \
      std::integer_sequence<unsigned, F...>) { return {{table##_entry<F>()...}}; }                                 \
  [[maybe_unused]] const std::array<KernelFn<Args>, NF> table = table##_make(std::make_integer_sequence<unsigned, NF>{});
// (the trajectory pass differs from the plain NL sweep only with the evaporation branch: the cover checkpoint)

// the variant tables live in the family units; F past a table's size or a combination that is not built: nullptr
KernelFn<NlArgs> nl_variant(unsigned f);
KernelFn<TlArgs> tl_variant(unsigned f);
KernelFn<AdArgs> ad_variant(unsigned f);
KernelFn<AdArgs> ad_reverse_variant(unsigned f);
KernelFn<TaylorArgs> taylor_variant(unsigned f);

}  // namespace cloudsc2
