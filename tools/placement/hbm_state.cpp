// hbm_state.cpp -- is a buffer's write class intrinsic, or does it depend on what else is allocated?
// Buffer 0 (1.4 GB, hipMalloc) is timed alone, then again after 20, 60 and 150 further buffers exist, then after all but the
// fastest of them have been freed.  Same kernel (the NL sweep's plane writes, 1250 blocks) every time.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hbm_state tools/placement/hbm_state.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
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
double time_buf(double* p) {
  static Timer t;
  auto launch = [&] { hipLaunchKernelGGL(nl_writes, dim3((unsigned)kBlocks), dim3(128), 0, 0, p, kBlocks); };
  for (int i = 0; i < 8; ++i) launch();
  std::vector<float> v;
  for (int i = 0; i < 11; ++i) {
    CHECK(hipEventRecord(t.a)); launch(); CHECK(hipEventRecord(t.b)); CHECK(hipEventSynchronize(t.b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, t.a, t.b)); v.push_back(ms);
  }
  CHECK(hipGetLastError());
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  std::vector<double*> bufs;
  auto grow = [&](size_t n) {
    while (bufs.size() < n) { double* p = nullptr; CHECK(hipMalloc((void**)&p, kNeed)); bufs.push_back(p); }
  };
  auto report = [&](const char* what) {
    printf("%-44s buffer0 %.4f ms |", what, time_buf(bufs[0]));
    std::vector<double> v;
    for (auto b : bufs) v.push_back(time_buf(b));
    int nf = 0;
    for (double x : v) nf += x < 0.165;
    std::vector<double> s = v;
    std::sort(s.begin(), s.end());
    printf(" %zu buffers: min %.4f median %.4f max %.4f, %d below 0.165 ms\n   ", v.size(), s.front(), s[s.size() / 2], s.back(), nf);
    for (double x : v) printf("%c", x < 0.145 ? 'V' : x < 0.165 ? 'F' : 'S');
    printf("\n");
    fflush(stdout);
    return v;
  };
  grow(1);   report("buffer 0 alone");
  grow(21);  report("21 buffers (29 GB)");
  grow(61);  report("61 buffers (86 GB)");
  grow(151); auto v = report("151 buffers (212 GB)");
  const size_t best = std::min_element(v.begin() + 1, v.end()) - v.begin();
  const size_t worst = std::max_element(v.begin() + 1, v.end()) - v.begin();
  double* pb = bufs[best]; double* pw = bufs[worst];
  printf("fastest %zu (%.4f), slowest %zu (%.4f); freeing everything else\n", best, v[best], worst, v[worst]);
  for (size_t i = 1; i < bufs.size(); ++i) if (i != best && i != worst) CHECK(hipFree(bufs[i]));
  printf("after the free: buffer0 %.4f  fastest %.4f  slowest %.4f\n", time_buf(bufs[0]), time_buf(pb), time_buf(pw));
  // and fresh allocations now
  std::vector<double*> again;
  for (int i = 0; i < 20; ++i) { double* p = nullptr; CHECK(hipMalloc((void**)&p, kNeed)); again.push_back(p); }
  printf("20 fresh buffers after the free:");
  for (auto p : again) printf(" %.3f", time_buf(p));
  printf("\n");
  return 0;
}
