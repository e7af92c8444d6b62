// hbm_vmm2.cpp -- which physical chunks go well together?  N chunks of C MiB are created in one go (hipMemCreate; index =
// allocation order), then 1.5 GiB arenas are composed from them in different ways and the NL sweep's plane writes (1250
// blocks) are timed on each arena:
//   consecutive : chunks i .. i+m-1                     (what one large allocation gets)
//   spread      : chunks i, i+N/m, i+2N/m, ...          (taken from all over the memory)
//   pairs       : half of the arena from region a, half from region b, alternating, for all pairs of R regions
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hbm_vmm2 tools/placement/hbm_vmm2.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

constexpr long long kBlocks = 1250;
constexpr size_t kNeed = (size_t)kBlocks * 8 * 137 * 128 * 8;

__global__ void __launch_bounds__(128) nl_writes(double* base, long long nblocks) {
  const long long b = blockIdx.x;
  if (b >= nblocks) return;
  double* blk = base + b * (8LL * 137 * 128) + threadIdx.x;
  for (int jk = 0; jk < 137; ++jk)
    for (int pl : {0, 2, 3, 4, 7}) __builtin_nontemporal_store((double)jk, blk + (long long)pl * 137 * 128 + jk * 128);
}

struct Timer { hipEvent_t a, b; Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); } };
template <class F>
double median_ms(F launch, int warm, int reps) {
  static Timer t;
  for (int i = 0; i < warm; ++i) launch();
  std::vector<float> v;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(t.a)); launch(); CHECK(hipEventRecord(t.b)); CHECK(hipEventSynchronize(t.b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, t.a, t.b)); v.push_back(ms);
  }
  CHECK(hipGetLastError());
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000;
  const size_t chunk = (size_t)(argc > 2 ? atoll(argv[2]) : 256) << 20;
  const int R = argc > 3 ? atoi(argv[3]) : 10;
  const int m = (int)((kNeed + chunk - 1) / chunk);  // chunks per arena
  int dev = 0;
  CHECK(hipGetDevice(&dev));
  size_t free_b = 0, total_b = 0;
  CHECK(hipMemGetInfo(&free_b, &total_b));
  if (chunk * n + (2ull << 30) > free_b) { fprintf(stderr, "too much\n"); return 2; }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (auto& x : h) CHECK(hipMemCreate(&x, chunk, &prop, 0));
  printf("%d chunks of %zu MiB; an arena = %d chunks\n", n, chunk >> 20, m);
  char* va = nullptr;
  const size_t arena = chunk * m;
  CHECK(hipMemAddressReserve((void**)&va, arena, 1ull << 30, nullptr, 0));
  auto time_arena = [&](const std::vector<int>& ids) {
    for (int k = 0; k < m; ++k) CHECK(hipMemMap(va + chunk * k, chunk, 0, h[ids[k]], 0));
    CHECK(hipMemSetAccess(va, arena, &acc, 1));
    const double t = median_ms([&] { hipLaunchKernelGGL(nl_writes, dim3((unsigned)kBlocks), dim3(128), 0, 0, (double*)va, kBlocks); }, 6, 9);
    CHECK(hipDeviceSynchronize());
    for (int k = 0; k < m; ++k) CHECK(hipMemUnmap(va + chunk * k, chunk));
    return t;
  };
  printf("consecutive chunks (arena i = chunks i*%d ..), ms:\n", m);
  {
    int col = 0;
    for (int i = 0; i + m <= n; i += m * std::max(1, n / (m * 80))) {
      std::vector<int> ids(m);
      for (int k = 0; k < m; ++k) ids[k] = i + k;
      printf("%.3f%s", time_arena(ids), (++col % 20 == 0) ? "\n" : " ");
    }
    printf("\n");
  }
  printf("spread chunks (i, i+N/m, ...), ms:\n");
  {
    int col = 0;
    for (int i = 0; i < n / m; i += std::max(1, n / m / 40)) {
      std::vector<int> ids(m);
      for (int k = 0; k < m; ++k) ids[k] = i + k * (n / m);
      printf("%.3f%s", time_arena(ids), (++col % 20 == 0) ? "\n" : " ");
    }
    printf("\n");
  }
  printf("pairs of regions (region r = chunks r*%d ..; arena alternates chunks of region a and region b), ms:\n", n / R);
  for (int a = 0; a < R; ++a) {
    for (int b = 0; b < R; ++b) {
      std::vector<int> ids(m);
      for (int k = 0; k < m; ++k) ids[k] = ((k & 1) ? b : a) * (n / R) + (n / R) / 2 + k;  // from the middle of each region; distinct when a == b
      printf("%.3f ", time_arena(ids));
    }
    printf("\n");
    fflush(stdout);
  }
  for (auto x : h) CHECK(hipMemRelease(x));
  return 0;
}
