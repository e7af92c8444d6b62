// hbm_alloc.cpp -- is the fast / slow write class of tools/placement/hbm_probe.cpp a property of the ALLOCATION?
//
// tools/placement/hbm_map.cpp found no slow place anywhere inside one 256 GiB allocation, while 150 separate 1.4 GB allocations
// fall into two classes.  This program allocates series of buffers in different ways and times the same work on each
// (a 16-byte-per-lane fill and the NL-shaped strided write over the first GiB), printing the exact addresses:
//   series A: N x hipMalloc(1 402 880 000 + 4 MiB)   (the odd size of a 160 000-column B_LOC)
//   series B: N x hipMalloc(1.5 GiB)                 (a multiple of 512 MiB)
//   series C: N windows of 1.5 GiB inside ONE hipMalloc
//   series D: N x (hipMemAddressReserve aligned to 1 GiB + hipMemCreate + hipMemMap), 1.5 GiB each
//   series E: N x hipMalloc(6.4 GB)                  (a whole state arena)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hbm_alloc tools/placement/hbm_alloc.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
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
constexpr size_t kWork = 1ull << 30;  // every buffer is at least this large; every measurement touches exactly this much

__global__ void __launch_bounds__(256) fill16(v2d* base, long long n2) {
  const v2d val = {1.0, 2.0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) __builtin_nontemporal_store(val, base + i);
}

// NL-shaped: one workgroup per block of 8 planes x 137 rows x 1 KiB; planes 0,2,3,4,7 written row by row
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

void measure(const char* series, const std::vector<char*>& bufs) {
  const long long nblocks = (long long)(kWork / (8ull * 137 * 128 * 8));  // 956 blocks fit the first GiB
  for (auto b : bufs) hipLaunchKernelGGL(fill16, dim3(4096), dim3(256), 0, 0, (v2d*)b, (long long)(kWork / 16));
  CHECK(hipDeviceSynchronize());
  printf("series %s: address, 2MiB-offset, fill16 GB/s, NL-shaped GB/s\n", series);
  for (auto b : bufs) {
    const double tf = median_ms([&] { hipLaunchKernelGGL(fill16, dim3(4096), dim3(256), 0, 0, (v2d*)b, (long long)(kWork / 16)); }, 3, 7);
    const double tn = median_ms([&] { hipLaunchKernelGGL(nl_writes, dim3((unsigned)nblocks), dim3(128), 0, 0, (double*)b, nblocks); }, 3, 7);
    printf("  %p  %7llu KiB  %6.0f  %6.0f\n", (void*)b, (unsigned long long)(((uintptr_t)b & ((1u << 21) - 1)) >> 10), kWork / (tf * 1e-3) / 1e9,
           nblocks * 5.0 * 137 * 1024 / (tn * 1e-3) / 1e9);
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 24;
  const std::string which = argc > 2 ? argv[2] : "ABCDE";
  size_t free_b = 0, total_b = 0;
  CHECK(hipMemGetInfo(&free_b, &total_b));
  printf("%.1f GiB free\n", free_b / 1073741824.0);
  const size_t odd = 1402880000ull + (4u << 20), even = 3ull << 29;
  if (which.find('A') != std::string::npos) {
    std::vector<char*> bufs(n);
    for (auto& b : bufs) CHECK(hipMalloc((void**)&b, odd));
    measure("A (separate hipMalloc, 1 402 880 000 + 4 MiB bytes)", bufs);
    for (auto b : bufs) CHECK(hipFree(b));
  }
  if (which.find('B') != std::string::npos) {
    std::vector<char*> bufs(n);
    for (auto& b : bufs) CHECK(hipMalloc((void**)&b, even));
    measure("B (separate hipMalloc, 1.5 GiB)", bufs);
    for (auto b : bufs) CHECK(hipFree(b));
  }
  if (which.find('C') != std::string::npos) {
    char* big = nullptr;
    CHECK(hipMalloc((void**)&big, even * n));
    std::vector<char*> bufs(n);
    for (int i = 0; i < n; ++i) bufs[i] = big + even * i;
    measure("C (windows of ONE hipMalloc, 1.5 GiB apart)", bufs);
    CHECK(hipFree(big));
  }
  if (which.find('D') != std::string::npos) {
    int dev = 0;
    CHECK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM recommended granularity %zu KiB\n", gran >> 10);
    std::vector<char*> bufs(n);
    std::vector<hipMemGenericAllocationHandle_t> hs(n);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < n; ++i) {
      void* va = nullptr;
      CHECK(hipMemAddressReserve(&va, even, 1ull << 30, nullptr, 0));
      CHECK(hipMemCreate(&hs[i], even, &prop, 0));
      CHECK(hipMemMap(va, even, 0, hs[i], 0));
      CHECK(hipMemSetAccess(va, even, &acc, 1));
      bufs[i] = (char*)va;
    }
    measure("D (VMM: reserve aligned to 1 GiB, create, map; 1.5 GiB)", bufs);
    for (int i = 0; i < n; ++i) {
      CHECK(hipMemUnmap(bufs[i], even));
      CHECK(hipMemRelease(hs[i]));
      CHECK(hipMemAddressFree(bufs[i], even));
    }
  }
  if (which.find('E') != std::string::npos) {
    const int ne = std::min(n, 16);
    std::vector<char*> bufs(ne);
    for (auto& b : bufs) CHECK(hipMalloc((void**)&b, 6400000000ull));
    measure("E (separate hipMalloc, 6.4 GB)", bufs);
    for (auto b : bufs) CHECK(hipFree(b));
  }
  return 0;
}
