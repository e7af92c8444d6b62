// hbm_alloc2.cpp -- same write stream (the NL sweep's plane writes, 1250 blocks = 1.4 GB), four ways of obtaining the memory,
// allocated alternately so that all four kinds are spread over the same parts of the HBM:
//   a: hipMalloc(1.4 GB)                       b: reserve + 3 x hipMemCreate(512 MiB)
//   c: reserve + 1 x hipMemCreate(1.4 GB)      d: reserve + 22 x hipMemCreate(64 MiB)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hbm_alloc2 tools/placement/hbm_alloc2.cpp
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

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr long long kBlocks = 1250;
constexpr size_t kNeed = (size_t)kBlocks * 8 * 137 * 128 * 8;  // 1 402 880 000 bytes

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

hipMemAllocationProp g_prop = {};
char* vmm_buffer(size_t chunk) {
  const size_t unit = 2u << 20;
  const size_t total = (kNeed + unit - 1) / unit * unit;
  char* va = nullptr;
  CHECK(hipMemAddressReserve((void**)&va, total, 1ull << 30, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    const size_t sz = std::min(chunk, total - off);
    hipMemGenericAllocationHandle_t h;
    CHECK(hipMemCreate(&h, sz, &g_prop, 0));
    CHECK(hipMemMap(va + off, sz, 0, h, 0));
    CHECK(hipMemRelease(h));  // the mapping keeps the memory alive
  }
  hipMemAccessDesc acc = {};
  acc.location = g_prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CHECK(hipMemSetAccess(va, total, &acc, 1));
  return va;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 36;  // buffers per kind
  int dev = 0;
  CHECK(hipGetDevice(&dev));
  g_prop.type = hipMemAllocationTypePinned;
  g_prop.location.type = hipMemLocationTypeDevice;
  g_prop.location.id = dev;
  const char* names[4] = {"a hipMalloc 1.4 GB", "b 3 x hipMemCreate 512 MiB", "c 1 x hipMemCreate 1.4 GB", "d 22 x hipMemCreate 64 MiB"};
  std::vector<char*> bufs[4];
  for (int i = 0; i < n; ++i) {
    char* p = nullptr;
    CHECK(hipMalloc((void**)&p, kNeed));
    bufs[0].push_back(p);
    bufs[1].push_back(vmm_buffer(512u << 20));
    bufs[2].push_back(vmm_buffer((size_t)4 << 30));
    bufs[3].push_back(vmm_buffer(64u << 20));
  }
  for (int k = 0; k < 4; ++k)
    for (auto b : bufs[k]) hipLaunchKernelGGL(fill16, dim3(4096), dim3(256), 0, 0, (v2d*)b, (long long)(kNeed / 16));
  CHECK(hipDeviceSynchronize());
  for (int pass = 0; pass < 2; ++pass)
    for (int k = 0; k < 4; ++k) {
      printf("pass %d  %-28s NL-shaped write ms:", pass, names[k]);
      std::vector<double> v;
      for (auto b : bufs[k]) {
        v.push_back(median_ms([&] { hipLaunchKernelGGL(nl_writes, dim3((unsigned)kBlocks), dim3(128), 0, 0, (double*)b, kBlocks); }, 4, 9));
        printf(" %.3f", v.back());
      }
      std::sort(v.begin(), v.end());
      printf("   | min %.4f median %.4f max %.4f\n", v.front(), v[v.size() / 2], v.back());
      fflush(stdout);
    }
  return 0;
}
