// hbm_map.cpp -- map of the write (and read) bandwidth of one large allocation, window by window, on an MI355X.
//
// tools/placement/hbm_probe.cpp showed that EVERY write pattern -- the NL sweep's strided plane writes as well as a plain
// contiguous 16-byte-per-lane fill -- runs 10-20 % slower on some allocations than on others, while reads do not care.
// This program asks where those places are and who pays:
//   1. one allocation of G GiB, filled window by window (W MiB each): GB/s per window  -> the map, and its granularity;
//   2. on the fastest and the slowest window: fills issued from ONE XCD at a time (workgroup id mod 8), streaming reads,
//      and fills of the two windows concurrently;
//   3. mode "pmc": ten fills of the fastest window (kernel fill_tag<2>) and ten of the slowest (fill_tag<1>) for rocprofv3.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hbm_map tools/placement/hbm_map.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
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

// fill n2 16-byte elements; xcd >= 0: only workgroups with (blockIdx.x & 7) == xcd take part (they cover everything)
template <int TAG>
__global__ void __launch_bounds__(256) fill_tag(v2d* base, long long n2, int xcd) {
  long long first = blockIdx.x, step = gridDim.x;
  if (xcd >= 0) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    first = blockIdx.x >> 3;
    step = gridDim.x >> 3;
  }
  const v2d val = {1.0, 2.0};
  for (long long i = first * 256 + threadIdx.x; i < n2; i += step * 256) __builtin_nontemporal_store(val, base + i);
}

__global__ void __launch_bounds__(256) read_sum(const v2d* base, long long n2, double* sink) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
    const v2d v = __builtin_nontemporal_load(base + i);
    acc += v.x + v.y;
  }
  if (acc == 12345.678) sink[0] = acc;
}

__global__ void __launch_bounds__(256) copy16(const v2d* src, v2d* dst, long long n2) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
  ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
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
  const std::string mode = argc > 1 ? argv[1] : "map";
  const double gib = argc > 2 ? atof(argv[2]) : 240.0;
  const long long win_mib = argc > 3 ? atoll(argv[3]) : 512;
  const size_t win = (size_t)win_mib << 20;
  const long long nwin = (long long)(gib * 1024 / win_mib);
  const size_t bytes = win * (size_t)nwin;
  size_t free_b = 0, total_b = 0;
  CHECK(hipMemGetInfo(&free_b, &total_b));
  if (bytes + (1ull << 30) > free_b) { fprintf(stderr, "not enough free memory\n"); return 2; }
  char* base = nullptr;
  CHECK(hipMalloc((void**)&base, bytes));
  double* sink = nullptr;
  CHECK(hipMalloc((void**)&sink, 64));
  printf("one allocation of %.1f GiB at %p, %lld windows of %lld MiB\n", bytes / 1073741824.0, (void*)base, nwin, win_mib);
  const long long n2 = (long long)(win / 16);
  const dim3 grid(4096), block(256);
  auto fill = [&](long long w, int xcd) { hipLaunchKernelGGL(fill_tag<0>, grid, block, 0, 0, (v2d*)(base + w * win), n2, xcd); };
  // first touch
  for (long long w = 0; w < nwin; ++w) fill(w, -1);
  CHECK(hipDeviceSynchronize());
  std::vector<double> gbs(nwin);
  for (int pass = 0; pass < 2; ++pass) {
    for (long long w = 0; w < nwin; ++w) gbs[w] = win / (median_ms([&] { fill(w, -1); }, 2, 5) * 1e-3) / 1e9;
    printf("fill GB/s per window, pass %d:\n", pass);
    for (long long w = 0; w < nwin; ++w) printf("%5.0f%s", gbs[w], (w % 16 == 15 || w == nwin - 1) ? "\n" : " ");
  }
  const long long wF = std::max_element(gbs.begin(), gbs.end()) - gbs.begin();
  const long long wS = std::min_element(gbs.begin(), gbs.end()) - gbs.begin();
  printf("fastest window %lld (%.0f GB/s), slowest window %lld (%.0f GB/s)\n", wF, gbs[wF], wS, gbs[wS]);
  fflush(stdout);

  if (mode == "map" || mode == "all") {
    for (long long w : {wF, wS}) {
      printf("window %lld (%s): fills from one XCD at a time, GB/s:", w, w == wF ? "fast" : "slow");
      for (int x = 0; x < 8; ++x) printf(" %5.0f", win / (median_ms([&] { fill(w, x); }, 2, 5) * 1e-3) / 1e9);
      const double r = win / (median_ms([&] { hipLaunchKernelGGL(read_sum, grid, block, 0, 0, (const v2d*)(base + w * win), n2, sink); }, 2, 5) * 1e-3) / 1e9;
      printf("   streaming read %5.0f GB/s\n", r);
    }
    // copies between and within the classes (read window a, write window b); neighbours of the windows serve as partners
    auto copy = [&](long long a, long long b) {
      return 2.0 * win / (median_ms([&] { hipLaunchKernelGGL(copy16, grid, block, 0, 0, (const v2d*)(base + a * win), (v2d*)(base + b * win), n2); }, 2, 5) * 1e-3) / 1e9;
    };
    printf("copy (read+write GB/s): fast->fast' %.0f  fast->slow %.0f  slow->fast %.0f\n", copy(wF, wF == 0 ? 1 : wF - 1), copy(wF, wS), copy(wS, wF));
    // a finer map around the slowest window: 32 MiB steps across [wS-1, wS+2)
    const size_t fine = 32u << 20;
    const long long lo = std::max<long long>(0, wS - 1) * (long long)(win / fine), hi = std::min<long long>(nwin, wS + 2) * (long long)(win / fine);
    printf("fine map (32 MiB windows) around the slowest window, GB/s:\n");
    int col = 0;
    for (long long f = lo; f < hi; ++f) {
      const double t = median_ms([&] { hipLaunchKernelGGL(fill_tag<0>, dim3(2048), block, 0, 0, (v2d*)(base + f * fine), (long long)(fine / 16), -1); }, 2, 7);
      printf("%5.0f%s", fine / (t * 1e-3) / 1e9, (++col % 16 == 0) ? "\n" : " ");
    }
    printf("\n");
    fflush(stdout);
  }
  if (mode == "pmc" || mode == "all") {
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill_tag<2>, grid, block, 0, 0, (v2d*)(base + wF * win), n2, -1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill_tag<1>, grid, block, 0, 0, (v2d*)(base + wS * win), n2, -1);
    CHECK(hipDeviceSynchronize());
    printf("pmc launches done: fill_tag<2> on the fastest, fill_tag<1> on the slowest window\n");
  }
  CHECK(hipFree(base));
  CHECK(hipFree(sink));
  return 0;
}
