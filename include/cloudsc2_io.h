/* cloudsc2_io.h -- C ABI of libcloudsc2_io.so: the HDF5 files of the dwarf (input.h5, reference.h5) through the HDF5
 * C API.  Replaces, for builds that cannot use the HDF5 Fortran modules (flang cannot read the gfortran-format
 * hdf5.mod of this image), what the reference does in
 *   src/common/module/hdf5_file_mod.F90      (open/create, load_{l,i,r}0, load_r1..r3, write_i0, write_r1..r3)
 *   src/common/module/file_io_mod.F90        (LOAD_SCALAR, LOAD_ARRAY, WRITE_SCALAR, WRITE_ARRAY)
 *   src/common/module/yomcst.F90:165-177, yoethf.F90:77-99, yoecldp.F90:241-333, yoephli.F90:78-97
 *                                             (the *_LOAD_PARAMETERS routines: which scalars the kernels' constants come from)
 *   src/common/module/cloudsc2_array_state_mod.F90:153-203 (LOAD), :260-287 (WRITE_REFERENCE).
 * Datasets are stored (…, KLON) in C order, i.e. Fortran (KLON, …); scalars are 1-element datasets; logicals are
 * integers.  Everything here is host code and needs no GPU; tiling a table into NPROMA blocks and validating
 * against a reference table are device work and live in include/cloudsc2_hip.h (cloudsc2_expand_launch,
 * cloudsc2_validate_launch).
 * Every function returns 0 on success or a negative CLOUDSC2_E* code; cloudsc2_io_last_error() has the text. */
#ifndef CLOUDSC2_IO_H
#define CLOUDSC2_IO_H
#include "cloudsc2_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define CLOUDSC2_EIO (-4)      /* HDF5 call failed / file or dataset missing */

typedef struct cloudsc2_file cloudsc2_file;

const char* cloudsc2_io_last_error(void);

/* hdf5_file_open / hdf5_file_create (hdf5_file_mod.F90).  mode: 0 = read only, 1 = create/truncate. */
int cloudsc2_file_open(const char* path, int mode, cloudsc2_file** f);
int cloudsc2_file_close(cloudsc2_file* f);
/* 1 if the dataset exists, 0 if not */
int cloudsc2_file_has(cloudsc2_file* f, const char* name);
/* rank and C-order dimensions of a dataset as stored (scalars: ndims 1, dims {1}) */
int cloudsc2_file_shape(cloudsc2_file* f, const char* name, int* ndims, long long dims[4]);
/* LOAD_SCALAR / LOAD_ARRAY: the whole dataset, converted to double / int32 whatever its stored type; `count` must
 * equal the number of elements */
int cloudsc2_file_read_f64(cloudsc2_file* f, const char* name, double* buf, long long count);
int cloudsc2_file_read_i32(cloudsc2_file* f, const char* name, int* buf, long long count);
/* WRITE_SCALAR / WRITE_ARRAY: dims in C order, native double / int32 */
int cloudsc2_file_write_f64(cloudsc2_file* f, const char* name, int ndims, const long long* dims, const double* buf);
int cloudsc2_file_write_i32(cloudsc2_file* f, const char* name, int ndims, const long long* dims, const int* buf);
/* The same for fp32 callers (JPRB = fp32, the reference's -DSINGLE build): the file data stay IEEE doubles, HDF5
 * converts to/from the fp32 buffer -- what LOAD_ARRAY does when JPRB is single (file_io_mod.F90) */
int cloudsc2_file_read_f32(cloudsc2_file* f, const char* name, float* buf, long long count);
int cloudsc2_file_write_f32(cloudsc2_file* f, const char* name, int ndims, const long long* dims, const float* buf);

/* The constants CLOUDSC2 / TL / AD read (YOMCST, YOETHF, YRECLDP, YREPHLI) + PTSPHY + KLEV from an input file, as
 * the four *_LOAD_PARAMETERS routines + CLOUDSC2_ARRAY_STATE_LOAD do.  What the mains set afterwards is applied
 * too: LPHYLIN=.TRUE., LEVAPLS2=.FALSE. (src/cloudsc2_nl/dwarf_cloudsc.F90:105-107), RVTMP2 = 0 (never loaded: yoethf.F90:30);
 * LREGCL and LDRAIN1D stay as passed in (the TL/AD mains set YRNCL%LREGCL, src/cloudsc2_ad/dwarf_cloudsc.F90:105).  prm->ceta is
 * filled from PAP(1,:)/PAPH(1,KLEV+1) (src/cloudsc2_nl/dwarf_cloudsc.F90:100-102) when the file holds PAP and PAPH. */
int cloudsc2_file_read_params(cloudsc2_file* f, cloudsc2_params* prm, double* ptsphy);
/* writes the same scalars (used to build test inputs in the input.h5 format) */
int cloudsc2_file_write_params(cloudsc2_file* f, const cloudsc2_params* prm, double ptsphy, int klon);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDSC2_IO_H */
