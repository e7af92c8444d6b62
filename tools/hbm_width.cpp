// hbm_width.cpp -- what bounds the column sweeps' memory pattern at 160 000 columns: request width, bytes in flight or waves?
// A skeleton of the NL sweep (16 input planes read, 11 output planes written per level, one lane per column, 137 dependent
// levels, non-temporal accesses, one level prefetched ahead) in three forms on the SAME placed allocation:
//   k8   : 8 bytes per lane and request (the reference's (NPROMA,NLEV,NBLOCKS) layout, what the kernels do today)
//   k8d2 : the same with two levels in flight
//   k16  : a level-pair-interleaved layout (NPROMA,2,NLEV/2,NBLOCKS): 16 bytes per lane and request, one pair ahead
// each at 2, 3 and "as many as fit" waves per SIMD (occupancy limited with dynamic LDS).
// usage: hbm_width NGPTOT        the three forms above
//        hbm_width NGPTOT 1      + the plane counts of TL (32 read / 20 written) and of the AD reverse sweep (44 / 26), one placed buffer each
//        hbm_width NGPTOT 2      + arithmetic between a level's loads and stores (0 / 280 / 560 / 1120 FMAs), look-ahead 1 and 2
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/bin/hbm_width tools/hbm_width.cpp \
//          -L dwarf_p_cloudsc2_tl_ad_amd/csrc -lcloudsc2_hip -Wl,-rpath,$PWD/dwarf_p_cloudsc2_tl_ad_amd/csrc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cloudsc2_hip.h"

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

constexpr int kIn = 16, kOut = 11, kLev = 137, kLevPad = 138, kNproma = 128;
typedef double dbl2 __attribute__((ext_vector_type(2)));

// plane p of the inputs starts at p * plane_elems, plane q of the outputs at (kIn + q) * plane_elems
template <int DEPTH>
__global__ void __launch_bounds__(kNproma) k8(const double* __restrict__ base, double* __restrict__ wbase, long long plane_elems) {
  extern __shared__ double lds[];
  const long long off = (long long)blockIdx.x * kLevPad * kNproma + threadIdx.x;
  const double* in = base + off;
  double* out = wbase + off;
  double buf[DEPTH + 1][kIn];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int p = 0; p < kIn; ++p) buf[d][p] = __builtin_nontemporal_load(in + p * plane_elems + d * kNproma);
  double carry = 0.0;
#pragma unroll 1
  for (int jk0 = 0; jk0 < kLev; jk0 += DEPTH + 1) {
#pragma unroll
    for (int s = 0; s <= DEPTH; ++s) {
      const int jk = jk0 + s;
      if (jk < kLev) {
        const int slot_new = (s + DEPTH) % (DEPTH + 1);
        if (jk + DEPTH < kLev) {
#pragma unroll
          for (int p = 0; p < kIn; ++p) buf[slot_new][p] = __builtin_nontemporal_load(in + p * plane_elems + (jk + DEPTH) * kNproma);
        }
        double acc = carry;
#pragma unroll
        for (int p = 0; p < kIn; ++p) acc = fma(buf[s][p], 1.0 + 0.125 * p, acc);
        carry = acc * 0.5;
#pragma unroll
        for (int q = 0; q < kOut; ++q) __builtin_nontemporal_store(acc + q, out + q * plane_elems + jk * kNproma);
      }
    }
  }
}

// k8 with arithmetic between the loads and the stores of a level, as much as the NL physics has (NFMA dependent-chain FMAs on 8
// accumulators per level ~ NFMA vector instructions): what does a level's compute cost on top of the memory pattern, and does a
// second level of look-ahead (DEPTH 2, registers) buy it back at the same occupancy?
template <int DEPTH, int NFMA>
__global__ void __launch_bounds__(kNproma) kcomp(const double* __restrict__ base, double* __restrict__ wbase, long long plane_elems) {
  extern __shared__ double lds[];
  const long long off = (long long)blockIdx.x * kLevPad * kNproma + threadIdx.x;
  const double* in = base + off;
  double* out = wbase + off;
  double buf[DEPTH + 1][kIn];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int p = 0; p < kIn; ++p) buf[d][p] = __builtin_nontemporal_load(in + p * plane_elems + d * kNproma);
  double carry = 0.0;
#pragma unroll 1
  for (int jk0 = 0; jk0 < kLev; jk0 += DEPTH + 1) {
#pragma unroll
    for (int s = 0; s <= DEPTH; ++s) {
      const int jk = jk0 + s;
      if (jk < kLev) {
        const int slot_new = (s + DEPTH) % (DEPTH + 1);
        if (jk + DEPTH < kLev) {
#pragma unroll
          for (int p = 0; p < kIn; ++p) buf[slot_new][p] = __builtin_nontemporal_load(in + p * plane_elems + (jk + DEPTH) * kNproma);
        }
        double a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = buf[s][i] + carry;
#pragma unroll 1
        for (int it = 0; it < NFMA / 8; ++it) {
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = fma(a[i], buf[s][8 + i], 1.0e-3);
        }
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += a[i];
        carry = acc * 1.0e-9;
#pragma unroll
        for (int q = 0; q < kOut; ++q) __builtin_nontemporal_store(acc + q, out + q * plane_elems + jk * kNproma);
      }
    }
  }
}

// the same sweep with the plane counts of the other kernels (TL: 32 read / 20 written per level; AD reverse sweep: 44 / 26)
template <int NIN, int NOUT>
__global__ void __launch_bounds__(kNproma) kshape(const double* __restrict__ base, double* __restrict__ wbase, long long plane_elems) {
  extern __shared__ double lds[];
  const long long off = (long long)blockIdx.x * kLevPad * kNproma + threadIdx.x;
  const double* in = base + off;
  double* out = wbase + off;
  double cur[NIN], nxt[NIN];
#pragma unroll
  for (int p = 0; p < NIN; ++p) cur[p] = __builtin_nontemporal_load(in + p * plane_elems);
  double carry = 0.0;
#pragma unroll 1
  for (int jk = 0; jk < kLev; ++jk) {
    if (jk + 1 < kLev) {
#pragma unroll
      for (int p = 0; p < NIN; ++p) nxt[p] = __builtin_nontemporal_load(in + p * plane_elems + (jk + 1) * kNproma);
    }
    double acc = carry;
#pragma unroll
    for (int p = 0; p < NIN; ++p) acc = fma(cur[p], 1.0 + 0.125 * p, acc);
    carry = acc * 0.5;
#pragma unroll
    for (int q = 0; q < NOUT; ++q) __builtin_nontemporal_store(acc + q, out + q * plane_elems + jk * kNproma);
#pragma unroll
    for (int p = 0; p < NIN; ++p) cur[p] = nxt[p];
  }
}

__global__ void __launch_bounds__(kNproma) k16(const dbl2* __restrict__ base, dbl2* __restrict__ wbase, long long plane_pairs) {
  extern __shared__ double lds[];
  constexpr int kPairs = kLevPad / 2;
  const long long off = (long long)blockIdx.x * kPairs * kNproma + threadIdx.x;
  const dbl2* in = base + off;
  dbl2* out = wbase + off;
  dbl2 cur[kIn], nxt[kIn];
#pragma unroll
  for (int p = 0; p < kIn; ++p) cur[p] = __builtin_nontemporal_load(in + p * plane_pairs);
  double carry = 0.0;
#pragma unroll 1
  for (int jp = 0; jp < kPairs; ++jp) {
    if (jp + 1 < kPairs) {
#pragma unroll
      for (int p = 0; p < kIn; ++p) nxt[p] = __builtin_nontemporal_load(in + p * plane_pairs + (long long)(jp + 1) * kNproma);
    }
    double a0 = carry;
#pragma unroll
    for (int p = 0; p < kIn; ++p) a0 = fma(cur[p].x, 1.0 + 0.125 * p, a0);
    double a1 = a0 * 0.5;
#pragma unroll
    for (int p = 0; p < kIn; ++p) a1 = fma(cur[p].y, 1.0 + 0.125 * p, a1);
    carry = a1 * 0.5;
#pragma unroll
    for (int q = 0; q < kOut; ++q) {
      dbl2 v = {a0 + q, a1 + q};
      __builtin_nontemporal_store(v, out + q * plane_pairs + (long long)jp * kNproma);
    }
#pragma unroll
    for (int p = 0; p < kIn; ++p) cur[p] = nxt[p];
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
};

template <class F>
double median_ms(F&& launch, int reps) {
  static Timer t;
  for (int i = 0; i < 5; ++i) launch();
  CHECK(hipDeviceSynchronize());
  std::vector<float> v;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(t.a));
    launch();
    CHECK(hipEventRecord(t.b));
    CHECK(hipEventSynchronize(t.b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, t.a, t.b));
    v.push_back(ms);
  }
  CHECK(hipGetLastError());
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  const long long ncol = argc > 1 ? atoll(argv[1]) : 160000;
  if (ncol % kNproma) { fprintf(stderr, "ncol must be a multiple of %d\n", kNproma); return 2; }
  const long long nblk = ncol / kNproma;
  const long long plane_elems = nblk * kLevPad * kNproma;
  const bool shapes = argc > 2 && atoi(argv[2]) != 0;  // second argument 1: also the TL- and AD-shaped sweeps (a 70-plane arena)
  const int planes = kIn + kOut;
  const size_t bytes = (size_t)planes * plane_elems * sizeof(double);
  void* arena = nullptr;
  if (cloudsc2_device_malloc(&arena, bytes) != 0) { fprintf(stderr, "allocation of %zu bytes failed\n", bytes); return 2; }
  int cand = 0; double best = 0, med = 0, worst = 0;
  cloudsc2_device_malloc_info(&cand, &best, &med, &worst);
  printf("# %lld columns, arena %.2f GB, placement candidates %d probe best/median/worst %.3f/%.3f/%.3f ms\n", ncol, bytes / 1e9, cand, best, med, worst);
  CHECK(hipMemset(arena, 0, bytes));
  double* base = (double*)arena;
  double* wbase = base + (long long)kIn * plane_elems;
  // every thread touches levels 0..137 of its column in 27 planes: the last address is inside the arena by construction
  // (block b, level l, lane t -> (b*138 + l)*128 + t < plane_elems).
  const double useful = (double)(kIn + kOut) * ncol * kLev * 8.0;
  CHECK(hipFuncSetAttribute((const void*)k8<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k8<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  struct Occ { const char* name; size_t lds; } occ[] = {{"1 wave/SIMD", 80 * 1024}, {"2 waves/SIMD", 40 * 1024}, {"3 waves/SIMD", 26 * 1024}, {"4 waves/SIMD", 20 * 1024}, {"unlimited", 0}};
  for (int round = 0; round < 2; ++round)
    for (auto& o : occ) {
      double t8 = median_ms([&] { hipLaunchKernelGGL(k8<1>, dim3((unsigned)nblk), dim3(kNproma), o.lds, 0, base, wbase, plane_elems); }, 15);
      double t8d2 = median_ms([&] { hipLaunchKernelGGL(k8<2>, dim3((unsigned)nblk), dim3(kNproma), o.lds, 0, base, wbase, plane_elems); }, 15);
      double t16 = median_ms([&] { hipLaunchKernelGGL(k16, dim3((unsigned)nblk), dim3(kNproma), o.lds, 0, (const dbl2*)base, (dbl2*)wbase, plane_elems / 2); }, 15);
      printf("%-13s k8 %.4f ms %.2f TB/s | k8d2 %.4f ms %.2f TB/s | k16 %.4f ms %.2f TB/s\n", o.name, t8, useful / t8 / 1e9, t8d2, useful / t8d2 / 1e9,
             t16, useful / t16 / 1e9);
      fflush(stdout);
    }
  if (argc > 2 && atoi(argv[2]) == 2) {  // arithmetic between loads and stores: 0 / 280 / 560 / 1120 FMAs per level, look-ahead 1 and 2
    CHECK(hipFuncSetAttribute((const void*)kcomp<1, 280>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void*)kcomp<1, 560>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void*)kcomp<1, 1120>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void*)kcomp<2, 280>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void*)kcomp<2, 560>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void*)kcomp<2, 1120>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const size_t lds3 = 26 * 1024;  // three waves per SIMD, like the NL kernel
#define RUN(K) median_ms([&] { hipLaunchKernelGGL((K), dim3((unsigned)nblk), dim3(kNproma), lds3, 0, base, wbase, plane_elems); }, 15)
    for (int round = 0; round < 3; ++round) {
      double t0 = RUN(k8<1>), u0 = RUN(k8<2>);
      double t1 = RUN((kcomp<1, 280>)), u1 = RUN((kcomp<2, 280>));
      double t2 = RUN((kcomp<1, 560>)), u2 = RUN((kcomp<2, 560>));
      double t3 = RUN((kcomp<1, 1120>)), u3 = RUN((kcomp<2, 1120>));
      printf("3 waves/SIMD, FMAs per level 0 / 280 / 560 / 1120: look-ahead 1: %.4f %.4f %.4f %.4f ms | look-ahead 2: %.4f %.4f %.4f %.4f ms\n", t0, t1, t2, t3, u0, u1,
             u2, u3);
      fflush(stdout);
    }
#undef RUN
  }
  if (shapes && atoi(argv[2]) == 1) {
    // each shape in an allocation of its own size, placed by the library: where in a buffer the WRITTEN planes lie decides its speed
    // (a sweep over the first 27 planes of a 70-plane buffer placed as a whole ran at 4.95 TB/s where the whole ran at 5.75)
    CHECK(hipFuncSetAttribute((const void*)kshape<16, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void*)kshape<32, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void*)kshape<44, 26>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    cloudsc2_device_free(arena);
    arena = nullptr;
    void *an = nullptr, *at = nullptr, *aa = nullptr;
    if (cloudsc2_device_malloc(&an, (size_t)27 * plane_elems * 8) || cloudsc2_device_malloc(&at, (size_t)52 * plane_elems * 8) ||
        cloudsc2_device_malloc(&aa, (size_t)70 * plane_elems * 8)) { fprintf(stderr, "allocation failed\n"); return 2; }
    double *bn = (double*)an, *bt = (double*)at, *ba = (double*)aa;
    for (int round = 0; round < 2; ++round)
      for (auto& o : {occ[0], occ[4]}) {
        const double per_plane = (double)ncol * kLev * 8.0;
        double tn = median_ms([&] { hipLaunchKernelGGL((kshape<16, 11>), dim3((unsigned)nblk), dim3(kNproma), o.lds, 0, bn, bn + 16 * plane_elems, plane_elems); }, 15);
        double tt = median_ms([&] { hipLaunchKernelGGL((kshape<32, 20>), dim3((unsigned)nblk), dim3(kNproma), o.lds, 0, bt, bt + 32 * plane_elems, plane_elems); }, 15);
        double ta = median_ms([&] { hipLaunchKernelGGL((kshape<44, 26>), dim3((unsigned)nblk), dim3(kNproma), o.lds, 0, ba, ba + 44 * plane_elems, plane_elems); }, 15);
        printf("%-13s NL-shaped 16r/11w %.4f ms %.2f TB/s | TL-shaped 32r/20w %.4f ms %.2f TB/s | AD-reverse-shaped 44r/26w %.4f ms %.2f TB/s\n", o.name, tn,
               27 * per_plane / tn / 1e9, tt, 52 * per_plane / tt / 1e9, ta, 70 * per_plane / ta / 1e9);
        fflush(stdout);
      }
    cloudsc2_device_free(an); cloudsc2_device_free(at); cloudsc2_device_free(aa);
  }
  if (arena) cloudsc2_device_free(arena);
  return 0;
}
