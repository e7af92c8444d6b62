// cloudsc2_resident.cpp -- the dwarf's NL run with the state resident on the GPU, from plain C++ on the C ABI
// (include/cloudsc2_hip.h, include/cloudsc2_io.h): read input.h5, tile the KLON-column tables into NPROMA blocks on the
// device, run SATUR + CLOUDSC2, validate against reference.h5 on the device and print the reference's report
// (flow of src/cloudsc2_nl/dwarf_cloudsc.F90:79-124).  No array of NGPTOT columns ever exists on the host.
//
//   hipcc -O2 -std=c++17 -I include examples/cloudsc2_resident.cpp -L dwarf_p_cloudsc2_tl_ad_amd/csrc \
//         -lcloudsc2_hip -lcloudsc2_io -Wl,-rpath,$PWD/dwarf_p_cloudsc2_tl_ad_amd/csrc -o cloudsc2_resident
//   ./cloudsc2_resident NGPTOT NPROMA [input.h5 [reference.h5]]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "cloudsc2_hip.h"
#include "cloudsc2_io.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define C2_OK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s: rc=%d %s | %s\n", #x, rc_, cloudsc2_last_error(), cloudsc2_io_last_error()); exit(3); } } while (0)

static double* dev_alloc(size_t n) {
  double* p = nullptr;
  HIP_OK(hipMalloc((void**)&p, n * sizeof(double)));
  HIP_OK(hipMemset(p, 0, n * sizeof(double)));
  return p;
}

// one dataset of the file -> device table (KLON, nlevx, ndim)
static double* upload_table(cloudsc2_file* f, const char* name, size_t count) {
  std::vector<double> h(count);
  C2_OK(cloudsc2_file_read_f64(f, name, h.data(), (long long)count));
  double* d = dev_alloc(count);
  HIP_OK(hipMemcpy(d, h.data(), count * sizeof(double), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const long long ngptot = argc > 1 ? atoll(argv[1]) : 16384;
  const int nproma = argc > 2 ? atoi(argv[2]) : 128;
  const std::string in_path = argc > 3 ? argv[3] : "input.h5";
  const std::string ref_path = argc > 4 ? argv[4] : "reference.h5";
  if (!cloudsc2_device_available()) { fprintf(stderr, "no HIP device: this engine has no CPU path\n"); return 1; }

  cloudsc2_file* fin = nullptr;
  C2_OK(cloudsc2_file_open(in_path.c_str(), 0, &fin));
  cloudsc2_params prm;
  cloudsc2_params_default(&prm);  // sets the flags; every constant below comes from the file
  double ptsphy = 0.0;
  C2_OK(cloudsc2_file_read_params(fin, &prm, &ptsphy));
  int klon = 0;
  C2_OK(cloudsc2_file_read_i32(fin, "KLON", &klon, 1));
  const int nlev = prm.nlev;
  const long long nblocks = (ngptot + nproma - 1) / nproma;
  const long long S = (long long)nproma * nlev, H = (long long)nproma * (nlev + 1);
  long long start = 0; int period = 0;
  cloudsc2_expand_offsets(klon, ngptot, 0, 0, 1, &start, &period);

  // GLOBAL_STATE on the device (cloudsc2_array_state_mod.F90:26-79): ONE allocation from the library's allocator, which places it
  // (cloudsc2_device_malloc_state, include/cloudsc2_hip.h: candidates are judged by the NL sweep on a state in the layout below --
  // read-only arrays first, then what the sweeps write, each 256-byte aligned)
  auto r256 = [](size_t n) { return (n + 31) & ~(size_t)31; };  // doubles
  const size_t nfull = r256((size_t)nblocks * S), nhalf = r256((size_t)nblocks * H);
  const size_t total = 11 * nfull + 5 * nhalf + 2 * r256((size_t)nblocks * 8 * S) + r256((size_t)nblocks * 5 * S);
  double* arena = nullptr;
  C2_OK(cloudsc2_device_malloc_state((void**)&arena, total * sizeof(double), nproma, nlev, (int)ngptot));
  HIP_OK(hipMemset(arena, 0, total * sizeof(double)));
  size_t used = 0;
  auto take = [&](size_t n) { double* p = arena + used; used += r256(n); return p; };
  auto full = [&] { return take((size_t)nblocks * S); };
  auto half = [&] { return take((size_t)nblocks * H); };
  double *pt = full(), *pq = full(), *pap = full(), *plu = full(), *plude = full(), *pmfu = full(), *pmfd = full(), *psupsat = full(),
         *paph = half(), *b_cml = take((size_t)nblocks * 8 * S), *pclv = take((size_t)nblocks * 5 * S);
  double *pa = full(), *pcovptot = full(), *qsat_unused = full(), *pfplsl = half(), *pfplsn = half(), *pfhpsl = half(), *pfhpsn = half(),
         *b_loc = take((size_t)nblocks * 8 * S);
  (void)qsat_unused;  // SATUR is fused into the NL sweep; the slot keeps the layout the library's own state has
  auto expand = [&](const char* name, int nlevx, int ndim, double* dst, long long stride) {
    double* tab = upload_table(fin, name, (size_t)klon * nlevx * ndim);
    C2_OK(cloudsc2_expand_launch(tab, klon, period, start, nlevx, ndim, nproma, ngptot, cloudsc2_field{dst, stride}, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipFree(tab));
  };
  expand("PT", nlev, 1, pt, S); expand("PQ", nlev, 1, pq, S); expand("PAP", nlev, 1, pap, S); expand("PAPH", nlev + 1, 1, paph, H);
  expand("PLU", nlev, 1, plu, S); expand("PLUDE", nlev, 1, plude, S); expand("PMFU", nlev, 1, pmfu, S);
  expand("PMFD", nlev, 1, pmfd, S); expand("PA", nlev, 1, pa, S); expand("PSUPSAT", nlev, 1, psupsat, S);
  expand("PCLV", nlev, 5, pclv, 5 * S);
  expand("TENDENCY_CML_T", nlev, 1, b_cml + 0 * S, 8 * S); expand("TENDENCY_CML_A", nlev, 1, b_cml + 1 * S, 8 * S);
  expand("TENDENCY_CML_Q", nlev, 1, b_cml + 2 * S, 8 * S); expand("TENDENCY_CML_CLD", nlev, 5, b_cml + 3 * S, 8 * S);
  C2_OK(cloudsc2_file_close(fin));

  // driver-array -> kernel-dummy mapping of cloudsc_driver_mod.F90:94-107
  cloudsc2_inputs in = {};
  in.paph = {paph, H}; in.pap = {pap, S}; in.q = {pq, S}; in.qsat = {nullptr, S}; in.t = {pt, S};
  in.l = {pclv + 0 * S, 5 * S}; in.i = {pclv + 1 * S, 5 * S}; in.lude = {plude, S}; in.lu = {plu, S};
  in.mfu = {pmfu, S}; in.mfd = {pmfd, S};
  in.gtent = {b_cml + 0 * S, 8 * S}; in.gtenq = {b_cml + 2 * S, 8 * S}; in.gtenl = {b_cml + 3 * S, 8 * S};
  in.gteni = {b_cml + 4 * S, 8 * S}; in.supsat = {psupsat, S};
  cloudsc2_outputs out = {};
  out.tent = {b_loc + 0 * S, 8 * S}; out.tenq = {b_loc + 2 * S, 8 * S}; out.tenl = {b_loc + 3 * S, 8 * S};
  out.teni = {b_loc + 4 * S, 8 * S}; out.clc = {pa, S}; out.covptot = {pcovptot, S};
  out.fplsl = {pfplsl, H}; out.fplsn = {pfplsn, H}; out.fhpsl = {pfhpsl, H}; out.fhpsn = {pfhpsn, H};

  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int w = 0; w < 10; ++w)  // warm-up (a fresh allocation needs a few launches to reach its steady time)
    C2_OK(cloudsc2_nl_launch(&prm, ptsphy, nproma, nlev, (int)ngptot, &in, &out, cloudsc2_field{b_loc + 7 * S, 8 * S}, 0.0, nullptr));
  HIP_OK(hipEventRecord(e0, nullptr));
  C2_OK(cloudsc2_nl_launch(&prm, ptsphy, nproma, nlev, (int)ngptot, &in, &out, cloudsc2_field{b_loc + 7 * S, 8 * S}, 0.0, nullptr));
  HIP_OK(hipEventRecord(e1, nullptr));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  fprintf(stderr, " NGPTOT=%lld NPROMA=%d NGPBLKS=%lld NLEV=%d: SATUR+CLOUDSC2 %.3f ms = %.3e columns/s\n", ngptot, nproma, nblocks, nlev,
          ms, ngptot / (ms * 1e-3));

  // validation on the device against the KLON-column reference (cloudsc2_array_state_mod.F90:205-258)
  cloudsc2_file* fref = nullptr;
  if (cloudsc2_file_open(ref_path.c_str(), 0, &fref) != 0) { fprintf(stderr, " no %s: validation skipped\n", ref_path.c_str()); return 0; }
  double* ws = dev_alloc((size_t)cloudsc2_validate_workspace_doubles());
  double* dstats = dev_alloc(5);
  char line[200];
  C2_OK(cloudsc2_validate_header(line, sizeof line));
  puts(line);
  struct Item { const char* dataset; const char* label; double* field; long long stride; int nlevx, ndim; };
  const Item items[] = {
      {"PLUDE", "PLUDE", plude, S, nlev, 1}, {"PCOVPTOT", "PCOVPTOT", pcovptot, S, nlev, 1},
      {"PFPLSL", "PFPLSL", pfplsl, H, nlev + 1, 1}, {"PFPLSN", "PFPLSN", pfplsn, H, nlev + 1, 1},
      {"PFHPSL", "PFHPSL", pfhpsl, H, nlev + 1, 1}, {"PFHPSN", "PFHPSN", pfhpsn, H, nlev + 1, 1},
      {"TENDENCY_LOC_A", "TENDENCY_LOC%A", b_loc + 1 * S, 8 * S, nlev, 1}, {"TENDENCY_LOC_Q", "TENDENCY_LOC%Q", b_loc + 2 * S, 8 * S, nlev, 1},
      {"TENDENCY_LOC_T", "TENDENCY_LOC%T", b_loc + 0 * S, 8 * S, nlev, 1}, {"TENDENCY_LOC_CLD", "TENDENCY_LOC%CLD", b_loc + 3 * S, 8 * S, nlev, 5}};
  int flagged = 0;
  for (const Item& it : items) {
    double* tab = upload_table(fref, it.dataset, (size_t)klon * it.nlevx * it.ndim);
    C2_OK(cloudsc2_validate_launch(tab, klon, period, start, it.nlevx, it.ndim, nproma, ngptot, cloudsc2_field{it.field, it.stride}, ws,
                                   dstats, nullptr));
    double st[5];
    HIP_OK(hipMemcpy(st, dstats, sizeof st, hipMemcpyDeviceToHost));
    HIP_OK(hipFree(tab));
    C2_OK(cloudsc2_validate_format(it.label, it.ndim == 1 ? 2 : 3, st, ngptot, line, sizeof line));
    puts(line);
    int iopt, warn;
    cloudsc2_validate_relerr(st[3], st[4], &iopt, &warn);
    flagged += warn;
  }
  C2_OK(cloudsc2_file_close(fref));
  C2_OK(cloudsc2_device_free(arena));
  return flagged ? 4 : 0;
}
