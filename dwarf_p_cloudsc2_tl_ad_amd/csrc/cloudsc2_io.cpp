// cloudsc2_io.cpp -- libcloudsc2_io.so: input.h5 / reference.h5 through the HDF5 C API (include/cloudsc2_io.h).
// Host code only.  The reference does this in Fortran on top of the HDF5 Fortran modules (hdf5_file_mod.F90,
// file_io_mod.F90, the *_LOAD_PARAMETERS routines); this is a from-scratch C++ equivalent for toolchains where those
// modules are not usable.
#include <hdf5.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/cloudsc2_io.h"

struct cloudsc2_file {
  hid_t id;
  bool writable;
};

namespace {

thread_local std::string g_io_err;

int io_fail(int code, const std::string& msg) {
  g_io_err = msg;
  return code;
}

// HDF5 prints its own error stack to stderr by default; the callers get the message through last_error instead
struct QuietHdf5 {
  QuietHdf5() { H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr); }
};

long long count_of(hid_t space) {
  const hssize_t n = H5Sget_simple_extent_npoints(space);
  return n < 0 ? -1 : (long long)n;
}

int read_any(cloudsc2_file* f, const char* name, hid_t memtype, void* buf, long long count) {
  if (!f || !name || !buf) return io_fail(CLOUDSC2_EINVAL, "NULL argument");
  hid_t d = H5Dopen2(f->id, name, H5P_DEFAULT);
  if (d < 0) return io_fail(CLOUDSC2_EIO, std::string("dataset not found: ") + name);
  hid_t sp = H5Dget_space(d);
  const long long n = count_of(sp);
  H5Sclose(sp);
  int rc = 0;
  if (n != count) {
    rc = io_fail(CLOUDSC2_EINVAL, std::string("dataset ") + name + " holds " + std::to_string(n) + " elements, caller expects " +
                                      std::to_string(count));
  } else if (H5Dread(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) {
    rc = io_fail(CLOUDSC2_EIO, std::string("H5Dread failed for ") + name);
  }
  H5Dclose(d);
  return rc;
}

// `filetype` is what the dataset is stored as: reals are always IEEE doubles in the file, whatever the caller holds
int write_any(cloudsc2_file* f, const char* name, hid_t memtype, hid_t filetype, int ndims, const long long* dims, const void* buf) {
  if (!f || !name || !buf || !dims) return io_fail(CLOUDSC2_EINVAL, "NULL argument");
  if (!f->writable) return io_fail(CLOUDSC2_EINVAL, "file was opened read-only");
  if (ndims < 1 || ndims > 4) return io_fail(CLOUDSC2_EINVAL, "1..4 dimensions supported");
  hsize_t h[4];
  for (int i = 0; i < ndims; ++i) {
    if (dims[i] < 1) return io_fail(CLOUDSC2_EINVAL, "non-positive dimension");
    h[i] = (hsize_t)dims[i];
  }
  hid_t sp = H5Screate_simple(ndims, h, nullptr);
  hid_t d = H5Dcreate2(f->id, name, filetype, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  int rc = 0;
  if (d < 0) rc = io_fail(CLOUDSC2_EIO, std::string("cannot create dataset ") + name);
  else if (H5Dwrite(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) rc = io_fail(CLOUDSC2_EIO, std::string("H5Dwrite failed for ") + name);
  if (d >= 0) H5Dclose(d);
  H5Sclose(sp);
  return rc;
}

struct Scalar { const char* name; double cloudsc2_params::*field; };
// dataset name -> parameter (yomcst.F90:168-176, yoethf.F90:80-96, yoecldp.F90:247-259, yoephli.F90:86)
const Scalar kScalars[] = {
    {"RG", &cloudsc2_params::rg}, {"RD", &cloudsc2_params::rd}, {"RCPD", &cloudsc2_params::rcpd},
    {"RETV", &cloudsc2_params::retv}, {"RLVTT", &cloudsc2_params::rlvtt}, {"RLSTT", &cloudsc2_params::rlstt},
    {"RLMLT", &cloudsc2_params::rlmlt}, {"RTT", &cloudsc2_params::rtt},
    {"R2ES", &cloudsc2_params::r2es}, {"R3LES", &cloudsc2_params::r3les}, {"R3IES", &cloudsc2_params::r3ies},
    {"R4LES", &cloudsc2_params::r4les}, {"R4IES", &cloudsc2_params::r4ies}, {"R5LES", &cloudsc2_params::r5les},
    {"R5IES", &cloudsc2_params::r5ies}, {"R5ALVCP", &cloudsc2_params::r5alvcp}, {"R5ALSCP", &cloudsc2_params::r5alscp},
    {"RALVDCP", &cloudsc2_params::ralvdcp}, {"RALSDCP", &cloudsc2_params::ralsdcp}, {"RTWAT", &cloudsc2_params::rtwat},
    {"RTICE", &cloudsc2_params::rtice}, {"RTICECU", &cloudsc2_params::rticecu},
    {"RTWAT_RTICE_R", &cloudsc2_params::rtwat_rtice_r}, {"RTWAT_RTICECU_R", &cloudsc2_params::rtwat_rticecu_r},
    {"YRECLDP_RCLCRIT", &cloudsc2_params::rclcrit}, {"YRECLDP_RKCONV", &cloudsc2_params::rkconv},
    {"YRECLDP_RPECONS", &cloudsc2_params::rpecons}, {"YRECLDP_RLMIN", &cloudsc2_params::rlmin},
    {"YREPHLI_RLPTRC", &cloudsc2_params::rlptrc},
};

}  // namespace

extern "C" {

const char* cloudsc2_io_last_error(void) { return g_io_err.c_str(); }

int cloudsc2_file_open(const char* path, int mode, cloudsc2_file** out) {
  static QuietHdf5 quiet;
  if (!path || !out) return io_fail(CLOUDSC2_EINVAL, "NULL argument");
  *out = nullptr;
  if (H5open() < 0) return io_fail(CLOUDSC2_EIO, "H5open failed");
  H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
  hid_t id = mode == 1 ? H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT) : H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (id < 0) return io_fail(CLOUDSC2_EIO, std::string(mode == 1 ? "cannot create " : "cannot open ") + path);
  *out = new cloudsc2_file{id, mode == 1};
  return 0;
}

int cloudsc2_file_close(cloudsc2_file* f) {
  if (!f) return 0;
  const herr_t e = H5Fclose(f->id);
  delete f;
  return e < 0 ? io_fail(CLOUDSC2_EIO, "H5Fclose failed") : 0;
}

int cloudsc2_file_has(cloudsc2_file* f, const char* name) {
  if (!f || !name) return 0;
  return H5Lexists(f->id, name, H5P_DEFAULT) > 0 ? 1 : 0;
}

int cloudsc2_file_shape(cloudsc2_file* f, const char* name, int* ndims, long long dims[4]) {
  if (!f || !name || !ndims || !dims) return io_fail(CLOUDSC2_EINVAL, "NULL argument");
  hid_t d = H5Dopen2(f->id, name, H5P_DEFAULT);
  if (d < 0) return io_fail(CLOUDSC2_EIO, std::string("dataset not found: ") + name);
  hid_t sp = H5Dget_space(d);
  int nd = H5Sget_simple_extent_ndims(sp);
  int rc = 0;
  if (nd < 0 || nd > 4) rc = io_fail(CLOUDSC2_EIO, std::string("unsupported rank for ") + name);
  else {
    hsize_t h[4] = {1, 1, 1, 1};
    if (nd > 0) H5Sget_simple_extent_dims(sp, h, nullptr);
    if (nd == 0) { nd = 1; h[0] = 1; }  // true HDF5 scalars read like the dwarf's 1-element datasets
    *ndims = nd;
    for (int i = 0; i < 4; ++i) dims[i] = i < nd ? (long long)h[i] : 1;
  }
  H5Sclose(sp);
  H5Dclose(d);
  return rc;
}

int cloudsc2_file_read_f64(cloudsc2_file* f, const char* name, double* buf, long long count) {
  return read_any(f, name, H5T_NATIVE_DOUBLE, buf, count);
}
int cloudsc2_file_read_i32(cloudsc2_file* f, const char* name, int* buf, long long count) {
  return read_any(f, name, H5T_NATIVE_INT, buf, count);
}
int cloudsc2_file_write_f64(cloudsc2_file* f, const char* name, int ndims, const long long* dims, const double* buf) {
  return write_any(f, name, H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, ndims, dims, buf);
}
int cloudsc2_file_write_i32(cloudsc2_file* f, const char* name, int ndims, const long long* dims, const int* buf) {
  return write_any(f, name, H5T_NATIVE_INT, H5T_NATIVE_INT, ndims, dims, buf);
}
// fp32 callers (JPRB = fp32 builds): HDF5 converts between the fp64 file data and the fp32 buffer
int cloudsc2_file_read_f32(cloudsc2_file* f, const char* name, float* buf, long long count) {
  return read_any(f, name, H5T_NATIVE_FLOAT, buf, count);
}
int cloudsc2_file_write_f32(cloudsc2_file* f, const char* name, int ndims, const long long* dims, const float* buf) {
  return write_any(f, name, H5T_NATIVE_FLOAT, H5T_NATIVE_DOUBLE, ndims, dims, buf);
}

int cloudsc2_file_read_params(cloudsc2_file* f, cloudsc2_params* prm, double* ptsphy) {
  if (!f || !prm) return io_fail(CLOUDSC2_EINVAL, "NULL argument");
  int rc;
  for (const Scalar& s : kScalars)
    if ((rc = cloudsc2_file_read_f64(f, s.name, &(prm->*(s.field)), 1))) return rc;
  prm->rvtmp2 = 0.0;   // never loaded by the reference (yoethf.F90:30)
  prm->lphylin = 1;    // src/cloudsc2_nl/dwarf_cloudsc.F90:107
  prm->levapls2 = 0;   // :105
  int klev = 0, klon = 0;
  if ((rc = cloudsc2_file_read_i32(f, "KLEV", &klev, 1))) return rc;
  if ((rc = cloudsc2_file_read_i32(f, "KLON", &klon, 1))) return rc;
  if (klev < 1 || klev > 200) return io_fail(CLOUDSC2_EINVAL, "KLEV outside 1..200 (dwarf_cloudsc.F90:87-90)");
  prm->nlev = klev;
  if (ptsphy && (rc = cloudsc2_file_read_f64(f, "PTSPHY", ptsphy, 1))) return rc;
  if (cloudsc2_file_has(f, "PAP") && cloudsc2_file_has(f, "PAPH")) {
    // CETA(JK) = PAP(1,JK,1) / PAPH(1,KLEV+1,1)  (src/cloudsc2_nl/dwarf_cloudsc.F90:100-102)
    std::vector<double> pap((size_t)klon * klev), paph((size_t)klon * (klev + 1));
    if ((rc = cloudsc2_file_read_f64(f, "PAP", pap.data(), (long long)pap.size()))) return rc;
    if ((rc = cloudsc2_file_read_f64(f, "PAPH", paph.data(), (long long)paph.size()))) return rc;
    const double psurf = paph[(size_t)klon * klev];
    for (int jk = 0; jk < klev; ++jk) prm->ceta[jk] = pap[(size_t)klon * jk] / psurf;
  }
  return 0;
}

int cloudsc2_file_write_params(cloudsc2_file* f, const cloudsc2_params* prm, double ptsphy, int klon) {
  if (!f || !prm) return io_fail(CLOUDSC2_EINVAL, "NULL argument");
  const long long one = 1;
  int rc;
  for (const Scalar& s : kScalars)
    if ((rc = cloudsc2_file_write_f64(f, s.name, 1, &one, &(prm->*(s.field))))) return rc;
  if ((rc = cloudsc2_file_write_f64(f, "PTSPHY", 1, &one, &ptsphy))) return rc;
  if ((rc = cloudsc2_file_write_i32(f, "KLEV", 1, &one, &prm->nlev))) return rc;
  if ((rc = cloudsc2_file_write_i32(f, "KLON", 1, &one, &klon))) return rc;
  return 0;
}

}  // extern "C"
