// hbm_vmm.cpp -- does the fast / slow write class belong to the PHYSICAL memory or to the virtual mapping?
//
// With the virtual-memory API (hipMemCreate = physical chunk, hipMemMap = where it appears) the two can be separated:
//   1. N physical chunks of C MiB, chunk i mapped at slot i of one reserved range: fill GB/s per chunk;
//   2. the same chunks mapped in REVERSE order into the same slots: does the class follow the chunk or the slot?
//   3. the fastest K and the slowest K chunks mapped contiguously as two arenas: the NL sweep's strided plane writes and a
//      contiguous fill on each -- can a write-fast arena be composed from probed chunks?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hbm_vmm tools/placement/hbm_vmm.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) fill16(v2d* base, long long n2) {
  const v2d val = {1.0, 2.0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) __builtin_nontemporal_store(val, base + i);
}

__global__ void __launch_bounds__(128) nl_writes(double* base, long long nblocks) {
  const long long b = blockIdx.x;
  if (b >= nblocks) return;
  double* blk = base + b * (8LL * 137 * 128) + threadIdx.x;
  for (int jk = 0; jk < 137; ++jk)
    for (int pl : {0, 2, 3, 4, 7}) __builtin_nontemporal_store((double)jk, blk + (long long)pl * 137 * 128 + jk * 128);
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
};

template <class F>
double median_ms(F launch, int warm, int reps) {
  static Timer t;
  for (int i = 0; i < warm; ++i) launch();
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
  const int n = argc > 1 ? atoi(argv[1]) : 400;
  const size_t chunk = (size_t)(argc > 2 ? atoll(argv[2]) : 512) << 20;
  const int K = argc > 3 ? atoi(argv[3]) : 6;
  int dev = 0;
  CHECK(hipGetDevice(&dev));
  size_t free_b = 0, total_b = 0;
  CHECK(hipMemGetInfo(&free_b, &total_b));
  if (chunk * n + (4ull << 30) > free_b || K * 2 > n) { fprintf(stderr, "too much\n"); return 2; }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  printf("%d chunks of %zu MiB (granularity %zu KiB)\n", n, chunk >> 20, gran >> 10);

  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (auto& x : h) CHECK(hipMemCreate(&x, chunk, &prop, 0));
  char* va = nullptr;
  CHECK(hipMemAddressReserve((void**)&va, chunk * n, 1ull << 30, nullptr, 0));
  auto map_all = [&](const std::vector<int>& slot_of_chunk) {
    for (int i = 0; i < n; ++i) CHECK(hipMemMap(va + chunk * slot_of_chunk[i], chunk, 0, h[i], 0));
    CHECK(hipMemSetAccess(va, chunk * n, &acc, 1));
  };
  auto unmap_all = [&] {
    CHECK(hipDeviceSynchronize());
    for (int s = 0; s < n; ++s) CHECK(hipMemUnmap(va + chunk * s, chunk));
  };
  const long long n2 = (long long)(chunk / 16);
  auto measure_slots = [&](std::vector<double>& gbs) {
    for (int s = 0; s < n; ++s) hipLaunchKernelGGL(fill16, dim3(2048), dim3(256), 0, 0, (v2d*)(va + chunk * s), n2);
    CHECK(hipDeviceSynchronize());
    gbs.resize(n);
    for (int s = 0; s < n; ++s)
      gbs[s] = chunk / (median_ms([&] { hipLaunchKernelGGL(fill16, dim3(2048), dim3(256), 0, 0, (v2d*)(va + chunk * s), n2); }, 2, 7) * 1e-3) / 1e9;
  };
  auto print_map = [&](const char* title, const std::vector<double>& v) {
    printf("%s\n", title);
    for (int i = 0; i < n; ++i) printf("%5.0f%s", v[i], (i % 20 == 19 || i == n - 1) ? "\n" : " ");
    fflush(stdout);
  };

  std::vector<int> ident(n), rev(n);
  std::iota(ident.begin(), ident.end(), 0);
  for (int i = 0; i < n; ++i) rev[i] = n - 1 - i;
  std::vector<double> by_slot1, by_slot2;
  map_all(ident);
  measure_slots(by_slot1);
  print_map("pass 1, chunk i in slot i: fill GB/s by CHUNK", by_slot1);
  unmap_all();
  map_all(rev);
  measure_slots(by_slot2);
  std::vector<double> by_chunk2(n);
  for (int i = 0; i < n; ++i) by_chunk2[i] = by_slot2[rev[i]];
  print_map("pass 2, chunk i in slot n-1-i: fill GB/s by CHUNK", by_chunk2);
  // correlation of the chunk's rate between the two mappings, and of the slot's rate
  auto corr = [&](const std::vector<double>& a, const std::vector<double>& b) {
    double ma = 0, mb = 0;
    for (int i = 0; i < n; ++i) { ma += a[i]; mb += b[i]; }
    ma /= n; mb /= n;
    double sab = 0, saa = 0, sbb = 0;
    for (int i = 0; i < n; ++i) { sab += (a[i] - ma) * (b[i] - mb); saa += (a[i] - ma) * (a[i] - ma); sbb += (b[i] - mb) * (b[i] - mb); }
    return sab / std::sqrt(saa * sbb + 1e-300);
  };
  printf("correlation of the fill rate across the two mappings: by CHUNK %.3f, by SLOT %.3f\n", corr(by_slot1, by_chunk2), corr(by_slot1, by_slot2));
  unmap_all();

  // arenas composed of the K fastest / K slowest chunks (mean of the two passes)
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::vector<double> mean(n);
  for (int i = 0; i < n; ++i) mean[i] = 0.5 * (by_slot1[i] + by_chunk2[i]);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return mean[a] > mean[b]; });
  const size_t arena_bytes = chunk * K;
  const long long nblocks = (long long)(arena_bytes / (8ull * 137 * 128 * 8));
  for (int which = 0; which < 2; ++which) {
    char* ar = nullptr;
    CHECK(hipMemAddressReserve((void**)&ar, arena_bytes, 1ull << 30, nullptr, 0));
    printf("%s arena of %d chunks:", which == 0 ? "FAST" : "SLOW", K);
    for (int k = 0; k < K; ++k) {
      const int c = which == 0 ? order[k] : order[n - 1 - k];
      printf(" %d(%.0f)", c, mean[c]);
      CHECK(hipMemMap(ar + chunk * k, chunk, 0, h[c], 0));
    }
    CHECK(hipMemSetAccess(ar, arena_bytes, &acc, 1));
    const double tf = median_ms([&] { hipLaunchKernelGGL(fill16, dim3(8192), dim3(256), 0, 0, (v2d*)ar, (long long)(arena_bytes / 16)); }, 5, 9);
    const double tn = median_ms([&] { hipLaunchKernelGGL(nl_writes, dim3((unsigned)nblocks), dim3(128), 0, 0, (double*)ar, nblocks); }, 5, 9);
    printf("\n   contiguous fill %.0f GB/s, NL-shaped writes (%lld blocks) %.0f GB/s\n", arena_bytes / (tf * 1e-3) / 1e9, nblocks,
           nblocks * 5.0 * 137 * 1024 / (tn * 1e-3) / 1e9);
    CHECK(hipDeviceSynchronize());
    for (int k = 0; k < K; ++k) CHECK(hipMemUnmap(ar + chunk * k, chunk));
    CHECK(hipMemAddressFree(ar, arena_bytes));
  }
  for (auto x : h) CHECK(hipMemRelease(x));
  CHECK(hipMemAddressFree(va, chunk * n));
  return 0;
}
