// hbm_probe.cpp -- what makes the write stream into B_LOC 15 % slower in some regions of the MI355X's HBM?
//
// B_LOC is the reference's AoSoA tendency buffer (cloudsc2_array_state_mod.F90:129-151): per NPROMA block 8 planes of
// NLEV*NPROMA reals, of which the NL sweep writes 5 (T,Q,QL,QI,QV): 137 KiB written, stride 1096 KiB.  This stand-alone
// program reproduces that write stream without any physics and measures it on many allocations in one process:
//   mode "map"     : N buffers of the B_LOC size; time the NL-shaped strided write on each (fast / slow classes)
//   mode "sweep"   : on the fastest and the slowest buffer: which property of the pattern matters (plane set, chunk,
//                    stride, order, store width, reads instead of writes, page-touch probes of 4 KiB..2 MiB strides)
//   mode "slices"  : the slowest buffer cut in 8 slices of blocks: is the slowness local to a part of it?
//   mode "pmc"     : ten launches on the fastest (kernel nl_writes<2>) and ten on the slowest buffer (nl_writes<1>),
//                    for rocprofv3 --pmc passes (per-instance TCC counters)
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_probe tools/placement/hbm_probe.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

constexpr int kNlev = 137, kNproma = 128, kPlanes = 8;

struct Pattern {
  long long block_stride;  // doubles between blocks
  long long plane;         // doubles per plane
  int nlev;                // rows of nproma doubles written per plane
  unsigned plane_mask;     // which of the 8 planes are touched
  long long nblocks;
};

// One workgroup = one NPROMA block, walking the levels like the NL sweep: per level one 1 KiB row of every touched plane.
template <int TAG, bool NT, bool READ>
__global__ void __launch_bounds__(kNproma) nl_writes(double* base, Pattern p, double* sink) {
  const long long b = blockIdx.x;
  if (b >= p.nblocks) return;
  double* blk = base + b * p.block_stride + threadIdx.x;
  double acc = 0.0;
  for (int jk = 0; jk < p.nlev; ++jk) {
#pragma unroll
    for (int pl = 0; pl < kPlanes; ++pl) {
      if (!(p.plane_mask >> pl & 1u)) continue;
      double* q = blk + pl * p.plane + (long long)jk * kNproma;
      if (READ) acc += NT ? __builtin_nontemporal_load(q) : *q;
      else if (NT) __builtin_nontemporal_store((double)jk, q);
      else *q = (double)jk;
    }
  }
  if (READ && acc == 12345.678) sink[0] = acc;
}

// NL order, but a workgroup writes R consecutive levels of one plane back to back (R KiB contiguous) before it turns to the
// next plane: what a sweep that stages R levels of its outputs (LDS or registers) would issue.
template <int R>
__global__ void __launch_bounds__(kNproma) burst_writes(double* base, Pattern p) {
  const long long b = blockIdx.x;
  if (b >= p.nblocks) return;
  double* blk = base + b * p.block_stride + threadIdx.x;
  for (int j0 = 0; j0 < p.nlev; j0 += R) {
#pragma unroll
    for (int pl = 0; pl < kPlanes; ++pl) {
      if (!(p.plane_mask >> pl & 1u)) continue;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (j0 + r < p.nlev) __builtin_nontemporal_store((double)j0, blk + pl * p.plane + (long long)(j0 + r) * kNproma);
    }
  }
}

// NL order with 16 bytes per lane: 64 lanes cover one 1 KiB row (one wave per NPROMA block of 128 columns)
__global__ void __launch_bounds__(64) wide_writes(double* base, Pattern p) {
  const long long b = blockIdx.x;
  if (b >= p.nblocks) return;
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d* blk = (v2d*)(base + b * p.block_stride) + threadIdx.x;
  const v2d val = {1.0, 2.0};
  for (int jk = 0; jk < p.nlev; ++jk) {
#pragma unroll
    for (int pl = 0; pl < kPlanes; ++pl) {
      if (!(p.plane_mask >> pl & 1u)) continue;
      __builtin_nontemporal_store(val, blk + (pl * p.plane + (long long)jk * kNproma) / 2);
    }
  }
}

// Persistent form: `gridDim.x` workgroups share the blocks (workgroup w takes blocks w, w+G, ...): fewer concurrent streams.
__global__ void __launch_bounds__(kNproma) persistent_writes(double* base, Pattern p) {
  for (long long b = blockIdx.x; b < p.nblocks; b += gridDim.x) {
    double* blk = base + b * p.block_stride + threadIdx.x;
    for (int jk = 0; jk < p.nlev; ++jk) {
#pragma unroll
      for (int pl = 0; pl < kPlanes; ++pl) {
        if (!(p.plane_mask >> pl & 1u)) continue;
        __builtin_nontemporal_store((double)jk, blk + pl * p.plane + (long long)jk * kNproma);
      }
    }
  }
}

// The same set of addresses in "flat" order: consecutive workgroups write consecutive 1 KiB rows of one plane of one block.
__global__ void __launch_bounds__(kNproma) flat_writes(double* base, Pattern p, int nsel, const int* sel) {
  const long long row = blockIdx.x;  // over nblocks * nsel * nlev
  const long long per_block = (long long)nsel * p.nlev;
  const long long b = row / per_block;
  if (b >= p.nblocks) return;
  const long long r = row - b * per_block;
  const int pl = sel[r / p.nlev];
  const int jk = (int)(r % p.nlev);
  base[b * p.block_stride + pl * p.plane + (long long)jk * kNproma + threadIdx.x] = (double)jk;
}

// contiguous fill of n doubles, 16 B per lane
__global__ void __launch_bounds__(256) fill16(double2* base, long long n2) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x)
    base[i] = make_double2(1.0, 2.0);
}

// page-touch probe: one 64-byte line (8 lanes x 8 B) per `stride` bytes, pages visited in a scrambled order
__global__ void __launch_bounds__(256) touch_pages(double* base, long long npages, long long stride_doubles, long long mul) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long page = t >> 3;
  if (page >= npages) return;
  const long long pg = (page * mul) % npages;  // mul coprime to npages: a permutation
  base[pg * stride_doubles + (t & 7)] = 3.0;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
  ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};

template <class F>
double median_ms(F launch, int warm = 5, int reps = 9) {
  Timer t;
  for (int i = 0; i < warm; ++i) launch();
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

static int popcount(unsigned m) { return __builtin_popcount(m); }

Pattern nl_pattern(long long nblocks) {
  Pattern p;
  p.plane = (long long)kNlev * kNproma;
  p.block_stride = kPlanes * p.plane;
  p.nlev = kNlev;
  p.plane_mask = 0x9d;  // planes 0,2,3,4,7 = T,Q,QL,QI,QV
  p.nblocks = nblocks;
  return p;
}

double gbps(const Pattern& p, double ms) { return popcount(p.plane_mask) * (double)p.nlev * kNproma * 8 * p.nblocks / (ms * 1e-3) / 1e9; }

size_t g_bytes = 0;  // size of every buffer; every launch is checked against it on the host
void check_fits(const Pattern& p, long long extra_doubles = 0) {
  const long long last = (p.nblocks - 1) * p.block_stride + (kPlanes - 1) * p.plane + (long long)p.nlev * kNproma + extra_doubles;
  if (p.nblocks < 1 || (size_t)last * 8 > g_bytes) { fprintf(stderr, "pattern does not fit the buffer\n"); exit(3); }
}

template <int TAG>
void launch_nl(double* buf, const Pattern& p, bool nt = true, bool read = false, double* sink = nullptr) {
  check_fits(p);
  dim3 g((unsigned)p.nblocks), b(kNproma);
  if (read) { if (nt) hipLaunchKernelGGL((nl_writes<TAG, true, true>), g, b, 0, 0, buf, p, sink); else hipLaunchKernelGGL((nl_writes<TAG, false, true>), g, b, 0, 0, buf, p, sink); }
  else if (nt) hipLaunchKernelGGL((nl_writes<TAG, true, false>), g, b, 0, 0, buf, p, sink);
  else hipLaunchKernelGGL((nl_writes<TAG, false, false>), g, b, 0, 0, buf, p, sink);
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "map";
  const int nbuf = argc > 2 ? atoi(argv[2]) : 40;
  const long long nblocks = argc > 3 ? atoll(argv[3]) : 1250;
  const Pattern p0 = nl_pattern(nblocks);
  const size_t bytes = (size_t)(p0.block_stride + 8192) * nblocks * 8 + (4 << 20);  // room for the padded-stride variants
  g_bytes = bytes;
  size_t free_b = 0, total_b = 0;
  CHECK(hipMemGetInfo(&free_b, &total_b));
  printf("device memory: %.1f GiB free of %.1f GiB; %d buffers of %.3f GB\n", free_b / 1073741824.0, total_b / 1073741824.0, nbuf, bytes / 1e9);
  if ((double)bytes * nbuf > 0.9 * free_b) { fprintf(stderr, "too many buffers\n"); return 2; }
  std::vector<double*> buf(nbuf, nullptr);
  for (auto& b : buf) { CHECK(hipMalloc((void**)&b, bytes)); }
  double* sink = nullptr;
  CHECK(hipMalloc((void**)&sink, 64));
  // first touch of everything (fresh allocations need a few launches to settle)
  for (auto b : buf) hipLaunchKernelGGL(fill16, dim3(4096), dim3(256), 0, 0, (double2*)b, (long long)(bytes / 16));
  CHECK(hipDeviceSynchronize());

  std::vector<double> ms(nbuf);
  for (int rnd = 0; rnd < 2; ++rnd) {
    for (int i = 0; i < nbuf; ++i) ms[i] = median_ms([&] { launch_nl<0>(buf[i], p0); });
    printf("map round %d (NL-shaped writes, ms):", rnd);
    for (int i = 0; i < nbuf; ++i) printf(" %.4f", ms[i]);
    printf("\n");
  }
  printf("addresses (GiB):");
  for (int i = 0; i < nbuf; ++i) printf(" %.2f", (double)(uintptr_t)buf[i] / 1073741824.0);
  printf("\n");
  const int iF = (int)(std::min_element(ms.begin(), ms.end()) - ms.begin());
  const int iS = (int)(std::max_element(ms.begin(), ms.end()) - ms.begin());
  printf("fastest %d: %.4f ms %.0f GB/s   slowest %d: %.4f ms %.0f GB/s   ratio %.3f\n", iF, ms[iF], gbps(p0, ms[iF]), iS, ms[iS],
         gbps(p0, ms[iS]), ms[iS] / ms[iF]);
  fflush(stdout);
  double* F = buf[iF];
  double* S = buf[iS];

  if (mode == "sweep" || mode == "all") {
    auto both = [&](const char* name, const Pattern& p, bool nt, bool read) {
      const double a = median_ms([&] { launch_nl<0>(F, p, nt, read, sink); });
      const double b = median_ms([&] { launch_nl<0>(S, p, nt, read, sink); });
      printf("  %-58s F %.4f ms %6.0f GB/s | S %.4f ms %6.0f GB/s | S/F %.3f\n", name, a, gbps(p, a), b, gbps(p, b), b / a);
      fflush(stdout);
    };
    printf("sweep on fastest (F) and slowest (S) buffer:\n");
    both("NL planes {0,2,3,4,7}, nt stores", p0, true, false);
    both("NL planes {0,2,3,4,7}, plain stores", p0, false, false);
    both("NL planes {0,2,3,4,7}, nt LOADS", p0, true, true);
    Pattern p = p0;
    p.plane_mask = 0x1f; both("planes {0,1,2,3,4} (685 KiB contiguous per block)", p, true, false);
    p.plane_mask = 0xff; both("all 8 planes (whole buffer, NL order)", p, true, false);
    p.plane_mask = 0x01; both("plane {0} only", p, true, false);
    p.plane_mask = 0x11; both("planes {0,4}", p, true, false);
    p.plane_mask = 0x55; both("planes {0,2,4,6}", p, true, false);
    p.plane_mask = 0xaa; both("planes {1,3,5,7}", p, true, false);
    p = p0; p.nlev = 128; p.plane = 128LL * kNproma; p.block_stride = 8 * p.plane; both("128 levels: plane 128 KiB, stride 1 MiB", p, true, false);
    p = p0; p.block_stride = p0.block_stride + 512; both("stride + 4 KiB", p, true, false);
    p = p0; p.block_stride = p0.block_stride + 8192; both("stride + 64 KiB", p, true, false);
    p = p0; p.nlev = 136; both("136 of 137 levels written (1 KiB hole per plane)", p, true, false);
    {
      auto both_k = [&](const char* name, auto launch) {
        check_fits(p0);
        const double a = median_ms([&] { launch(F); });
        const double b = median_ms([&] { launch(S); });
        printf("  %-58s F %.4f ms %6.0f GB/s | S %.4f ms %6.0f GB/s | S/F %.3f\n", name, a, gbps(p0, a), b, gbps(p0, b), b / a);
        fflush(stdout);
      };
      dim3 g((unsigned)nblocks), blk(kNproma);
      both_k("bursts of 2 levels per plane", [&](double* x) { hipLaunchKernelGGL(burst_writes<2>, g, blk, 0, 0, x, p0); });
      both_k("bursts of 4 levels per plane", [&](double* x) { hipLaunchKernelGGL(burst_writes<4>, g, blk, 0, 0, x, p0); });
      both_k("bursts of 8 levels per plane", [&](double* x) { hipLaunchKernelGGL(burst_writes<8>, g, blk, 0, 0, x, p0); });
      both_k("bursts of 16 levels per plane", [&](double* x) { hipLaunchKernelGGL(burst_writes<16>, g, blk, 0, 0, x, p0); });
      both_k("16 B per lane, one wave per block", [&](double* x) { hipLaunchKernelGGL(wide_writes, g, dim3(64), 0, 0, x, p0); });
      both_k("persistent, 256 workgroups", [&](double* x) { hipLaunchKernelGGL(persistent_writes, dim3(256), blk, 0, 0, x, p0); });
      both_k("persistent, 512 workgroups", [&](double* x) { hipLaunchKernelGGL(persistent_writes, dim3(512), blk, 0, 0, x, p0); });
      both_k("persistent, 1024 workgroups", [&](double* x) { hipLaunchKernelGGL(persistent_writes, dim3(1024), blk, 0, 0, x, p0); });
    }
    // flat order
    {
      int hsel[5] = {0, 2, 3, 4, 7};
      int* dsel;
      CHECK(hipMalloc((void**)&dsel, sizeof hsel));
      CHECK(hipMemcpy(dsel, hsel, sizeof hsel, hipMemcpyHostToDevice));
      const long long rows = nblocks * 5 * kNlev;
      check_fits(p0);
      const double a = median_ms([&] { hipLaunchKernelGGL(flat_writes, dim3((unsigned)rows), dim3(kNproma), 0, 0, F, p0, 5, dsel); });
      const double b = median_ms([&] { hipLaunchKernelGGL(flat_writes, dim3((unsigned)rows), dim3(kNproma), 0, 0, S, p0, 5, dsel); });
      printf("  %-58s F %.4f ms %6.0f GB/s | S %.4f ms %6.0f GB/s | S/F %.3f\n", "same addresses, flat order (row per workgroup)", a, gbps(p0, a), b, gbps(p0, b), b / a);
      CHECK(hipFree(dsel));
    }
    {
      const long long n2 = p0.block_stride * nblocks / 2;
      const double a = median_ms([&] { hipLaunchKernelGGL(fill16, dim3(8192), dim3(256), 0, 0, (double2*)F, n2); });
      const double b = median_ms([&] { hipLaunchKernelGGL(fill16, dim3(8192), dim3(256), 0, 0, (double2*)S, n2); });
      printf("  %-58s F %.4f ms %6.0f GB/s | S %.4f ms %6.0f GB/s | S/F %.3f\n", "contiguous fill, 16 B per lane", a, n2 * 16 / a / 1e6, b, n2 * 16 / b / 1e6, b / a);
    }
    // page-touch probes: translation reach.  One 64-B line per page-stride, scrambled order.
    for (long long stride : {4096LL, 65536LL, 2097152LL}) {
      const long long npages = p0.block_stride * nblocks * 8 / stride;
      const long long mul = 1000003;  // prime, coprime to npages unless npages is a multiple of it
      const unsigned grid = (unsigned)((npages * 8 + 255) / 256);
      const double a = median_ms([&] { hipLaunchKernelGGL(touch_pages, dim3(grid), dim3(256), 0, 0, F, npages, stride / 8, mul); });
      const double b = median_ms([&] { hipLaunchKernelGGL(touch_pages, dim3(grid), dim3(256), 0, 0, S, npages, stride / 8, mul); });
      printf("  touch one line per %7lld B, %8lld pages, scrambled        F %.4f ms %6.1f Mpages/s | S %.4f ms %6.1f Mpages/s | S/F %.3f\n", stride, npages,
             a, npages / a / 1e3, b, npages / b / 1e3, b / a);
    }
    fflush(stdout);
  }

  if (mode == "slices" || mode == "all") {
    printf("slices of 1/8 of the blocks (NL-shaped writes, GB/s):\n");
    for (double* base : {F, S}) {
      printf("  %s:", base == F ? "F" : "S");
      for (int s = 0; s < 8; ++s) {
        Pattern p = p0;
        p.nblocks = nblocks / 8;
        double* b = base + (long long)s * p.nblocks * p0.block_stride;
        // a slice alone is a short launch: repeat it over the slice 1x, report GB/s
        const double t = median_ms([&] { launch_nl<0>(b, p); });
        printf(" %6.0f", gbps(p, t));
      }
      printf("\n");
    }
    fflush(stdout);
  }

  if (mode == "pmc" || mode == "all") {
    for (int i = 0; i < 10; ++i) launch_nl<2>(F, p0);
    for (int i = 0; i < 10; ++i) launch_nl<1>(S, p0);
    CHECK(hipDeviceSynchronize());
    printf("pmc launches done: nl_writes<2> on the fastest, nl_writes<1> on the slowest buffer\n");
  }
  for (auto b : buf) CHECK(hipFree(b));
  CHECK(hipFree(sink));
  return 0;
}
