/*
 * cloudsc2_hip.h -- C ABI of the MI355X-native CLOUDSC2 NL / TL / AD column-physics engine.
 *
 * This is the drop-in boundary for the hot path of ecmwf-ifs/dwarf-p-cloudsc2-tl-ad.  The reference has
 * no FFI layer; its de-facto boundary is the three driver procedures and the F77-style kernels they call
 * (SURVEY.md 8b).  Every entry point below cites the reference interface it replaces (paths relative to
 * the reference checkout).  All arrays are `cloudsc2_real` = JPRB (src/common/module/parkind1.F90:40-44): fp64 in
 * libcloudsc2_hip.so, fp32 in libcloudsc2_hip_sp.so (this header compiled with -DCLOUDSC2_SINGLE, the reference's
 * -DSINGLE; same entry points, cloudsc2_real_bytes() tells the two apart).  Constants, time step, test statistics and
 * norms are `double` in both.  Layout: the reference's NPROMA-blocked one, column index fastest:
 *
 *     field(NPROMA, NLEV or NLEV+1, NBLOCKS)         f[jl + NPROMA*(jk + NLEVx*ibl)]
 *     B_CML / B_LOC (NPROMA, NLEV, 8, NBLOCKS)       planes T=0, A=1, Q=2, CLD(QL,QI,QR,QS,QV)=3..7
 *                                                    (src/common/module/cloudsc2_array_state_mod.F90:129-151)
 *     PCLV (NPROMA, NLEV, 5, NBLOCKS)                planes QL=0, QI=1 used (yoecldp.F90:86-91)
 *
 * No torch types, no C++ types: plain pointers and sizes.  Return value 0 = OK, otherwise a negative
 * CLOUDSC2_E* code or a positive hipError_t; cloudsc2_last_error() gives the text.  The library has NO CPU
 * fallback: without a HIP device every launch/run entry point fails with CLOUDSC2_ENODEVICE.
 */
#ifndef CLOUDSC2_HIP_H
#define CLOUDSC2_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#ifdef CLOUDSC2_SINGLE
typedef float cloudsc2_real;
#else
typedef double cloudsc2_real;
#endif

#define CLOUDSC2_MAX_NLEV 200 /* the reference's own limit, src/cloudsc2_nl/dwarf_cloudsc.F90:87 */

#define CLOUDSC2_EINVAL    (-1)
#define CLOUDSC2_ENODEVICE (-2)
#define CLOUDSC2_ETLWRONG  (-3) /* "TL is totally wrong" STOP of cloudsc_driver_tl_mod.F90:247-249 */

/* Constants and switches the reference keeps in modules and loads from input.h5:
 * YOMCST (yomcst.F90:167-177), YOETHF (yoethf.F90:79-99), YOECLDP (yoecldp.F90:247-250,...),
 * YOEPHLI (yoephli.F90:79-97), YOECLD%CETA (dwarf_cloudsc.F90:100-102), YRPHNC%LEVAPLS2 (:105),
 * YRNCL%LREGCL (src/cloudsc2_tl/dwarf_cloudsc.F90:105), LDRAIN1D (cloudsc_driver_mod.F90:61).
 * The first 30 doubles are also the order oracle/ref_harness.F90 takes them in. */
typedef struct cloudsc2_params {
  double rg, rd, rcpd, retv, rlvtt, rlstt, rlmlt, rtt;                         /* YOMCST  */
  double r2es, r3les, r3ies, r4les, r4ies, r5les, r5ies, r5alvcp, r5alscp,     /* YOETHF  */
         ralvdcp, ralsdcp, rtwat, rtice, rtwat_rtice_r, rvtmp2;
  double rclcrit, rkconv, rlmin, rpecons;                                      /* YOECLDP */
  double rlptrc;                                                               /* YOEPHLI */
  double rticecu, rtwat_rticecu_r;                                             /* YOETHF, dead branches only */
  int lphylin;   /* YREPHLI%LPHYLIN: 1 in every reference main (dwarf_cloudsc.F90:107).  0 selects the FOEALFA / FOEEWM form of the NL
                  * sweep's saturation pressure (cloudsc2.F90:349,365-369; unless ldrain1d); CLOUDSC2TL / CLOUDSC2AD and the driver's
                  * SATUR call (cloudsc_driver_mod.F90:91: LDPHYLIN=.TRUE.) do not depend on it: cloudsc2_tl_launch* and
                  * cloudsc2_ad_launch* IGNORE this field and run the LPHYLIN form, as the reference's TL / AD do
                  * (cloudsc2tl.F90 / cloudsc2ad.F90 have no other); cloudsc2_nl_launch with pert_lambda != 0 (a perturbed run of
                  * the Taylor test) refuses lphylin = 0 with CLOUDSC2_EINVAL */
  int levapls2;  /* precipitation evaporation on/off (LEVAPLS2 .OR. LDRAIN1D, cloudsc2.F90:557) */
  int lregcl;    /* TL/AD regularisation (cloudsc2tl.F90:575,657,754,794,998) */
  int ldrain1d;
  int nlev;
  int math_mode; /* arithmetic of THIS call: 0 = the process default (cloudsc2_set_math_mode), 1 = fast, 2 = precise */
  double ceta[CLOUDSC2_MAX_NLEV];
} cloudsc2_params;

/* Fill *p with the standard IFS constants and the reference mains' switches; nlev and ceta
 * are left for the caller (ceta(jk) = PAP(1,jk,1)/PAPH(1,nlev+1,1), dwarf_cloudsc.F90:100-102). */
void cloudsc2_params_default(cloudsc2_params* p);

const char* cloudsc2_last_error(void);
/* 1 if a HIP device is usable by this process, else 0. */
int cloudsc2_device_available(void);
/* ordinal of the calling thread's current HIP device (what the Fortran timing table prints where the reference's prints the
 * core a thread ran on, timer_mod.F90:97,159-162); 0 when there is no device */
int cloudsc2_current_device(void);
/* sizeof(cloudsc2_real) of THIS build of the library: 8, or 4 for the -DCLOUDSC2_SINGLE build. */
int cloudsc2_real_bytes(void);

/* Arithmetic of the kernels.  Every call carries its own mode in cloudsc2_params.math_mode (1 fast, 2 precise); 0 there
 * means the process default set here (initial value from the environment variable CLOUDSC2_MATH=fast|precise).  The
 * default is only READ by the launchers, never changed by the library, so concurrent calls do not influence each other.
 *   fast (default): quotients on shared / batch-inverted v_rcp_f64 reciprocals refined by one third-order step, a
 *     branch-free exp, tanh from one exp -- a few ulp from the correctly rounded values;
 *   precise: the reference's own operation order with IEEE division and libm exp/tanh/cosh.
 * cloudsc2_tl_taylor_run evaluates in precise mode unless params.math_mode asks for fast explicitly: its verdict is
 * decided by round-off noise (the fast-mode verdict is recorded by tests/test_gpu_parity.py). */
void cloudsc2_set_math_mode(int precise);
int cloudsc2_get_math_mode(void);

/* Device memory for callers that keep the state resident (and what the library uses for its own buffers).
 * On MI355X write streams run 10-20 % slower into some parts of the HBM than into others, whatever the access pattern
 * (profiles/r02_hbm_placement.md): 0.81 vs 0.96 ms for the NL kernel at 160 000 columns, decided by where the OUTPUT arrays
 * happen to lie.  This allocator places what it hands out: for a request of 256 MiB or more it makes candidate hipMalloc
 * allocations of the full size -- enough of them to span 96 GiB together (12 to 64, never more than fit 60 % of the free memory;
 * 85 % for requests above 12 GiB) -- times two probe streams over each (the sweeps' write stream and the NL sweep's whole
 * read/write pattern), keeps the best and frees the others (about 10 ms per candidate, once per allocation).
 * The mixed stream reads the first 59 % of the buffer and writes the last 41 %: put what your sweeps WRITE at the end of the
 * buffer (a sweep that wrote at 23-39 % of a buffer placed this way ran at 5.0 instead of 5.8 TB/s).
 * Allocate a state -- inputs AND outputs -- as ONE request (an arena): that is what was measured to land in
 * the fast class 8 times of 8, whereas placing only the written arrays and leaving the inputs elsewhere does not (0.88-0.93 ms;
 * profiles/r02_placement/z_one_arena_vs_split.txt).  The pointer is ordinary hipMalloc memory of the current device; free it with
 * cloudsc2_device_free.  cloudsc2_device_malloc_info reports the last placement: number of
 * candidates and the probe time of the best, the median and the worst of them (0 when there was no choice).
 * There is no reference counterpart: the reference's arrays are host ALLOCATABLEs (cloudsc2_array_state_mod.F90:97-151). */
#include <stddef.h>
int cloudsc2_device_malloc(void** ptr, size_t bytes);
void cloudsc2_device_malloc_info(int* candidates, double* best_ms, double* median_ms, double* worst_ms);
/* Who is searched.  cloudsc2_device_malloc, cloudsc2_device_malloc_state and the resident states (cloudsc2_state_create and the
 * scratch of their self-tests): yes.  The workspace of the HOST-ARRAY drivers cloudsc2_nl_run / cloudsc2_tl_taylor_run /
 * cloudsc2_ad_symmetry_run: one plain hipMalloc -- those calls are PCIe-bound (55-71 ms per 160 000 columns against a 0.1 ms
 * difference of the kernel), and a host model's free HBM is not taken transiently on their behalf; CLOUDSC2_PLACE=1 in the
 * environment asks for the search there too.  Staging buffers of re-blocked transfers and the validator's workspace: never.
 * CLOUDSC2_PLACE=0 switches every search off.  The counters say how many allocations of this process were searched (more than
 * one candidate) and how many were plain. */
void cloudsc2_device_malloc_counts(long long* searched, long long* plain);
int cloudsc2_device_free(void* ptr);
/* The same for a buffer that will hold a state of (nproma, nlev, ngptot) at its start (and whatever the caller keeps behind it:
 * perturbation sets, scratch): the placement search times the NL kernel itself on a zero-filled state laid out in every candidate
 * -- read-only arrays first (PT PQ PAP PLU PLUDE PMFU PMFD PSUPSAT PAPH B_CML PCLV), then what the sweeps write (PA PCOVPTOT QSAT
 * PFPLSL PFPLSN PFHPSL PFHPSN B_LOC), each 256-byte aligned, the layout cloudsc2_state_* uses -- instead of the two generic probe
 * streams, which stop predicting the kernel at 1 M columns (profiles/r02_placement/r_1m_probe_vs_kernel.txt).  When more than
 * 256 MiB follow the state the two generic streams over the whole buffer judge instead (a buffer's speed is a property of the
 * whole, and the NL sweep sees only its first part).  `bytes` must cover the state; the buffer comes back zero-filled only by
 * accident, clear what you need.  Free with cloudsc2_device_free. */
int cloudsc2_device_malloc_state(void** ptr, size_t bytes, int nproma, int nlev, int ngptot);
/* Diagnostic: times one of the allocator's probe streams over [ptr, ptr+bytes) and OVERWRITES it.  kind 0: the sweeps' write
 * stream (what the placement search uses); kind 1: the NL sweep's whole pattern -- 16 planes read, 11 written per level, the
 * buffer taken as 27 planes.  *ms = median of `rounds` launches after one warm-up (tools/probe_vs_kernel.py). */
int cloudsc2_device_probe(void* ptr, size_t bytes, int kind, int rounds, double* ms);

/* ------------------------------------------------------------------------------------------------
 * Kernel level: DEVICE pointers, asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * One field = base pointer + stride between NPROMA blocks (in elements); level stride is NPROMA,
 * column stride 1.  A (NPROMA,NLEV,NBLOCKS) array has block_stride NPROMA*NLEV; plane p of B_CML has
 * ptr = b_cml + p*NPROMA*NLEV and block_stride 8*NPROMA*NLEV.
 * ------------------------------------------------------------------------------------------------ */
typedef struct cloudsc2_field {
  cloudsc2_real* ptr;
  long long block_stride;
} cloudsc2_field;

/* The 16 inputs of CLOUDSC2 in the dummy-argument order of src/cloudsc2_nl/cloudsc2.F90:13-16:
 * PAPHP1(NLEV+1) PAPP1 PQM1 PQS PTM1 PL PI PLUDE PLU PMFU PMFD PGTENT PGTENQ PGTENL PGTENI PSUPSAT. */
typedef struct cloudsc2_inputs {
  cloudsc2_field paph, pap, q, qsat, t, l, i, lude, lu, mfu, mfd, gtent, gtenq, gtenl, gteni, supsat;
} cloudsc2_inputs;

/* The 10 outputs (cloudsc2.F90:15-18): PTENT PTENQ PTENL PTENI PCLC PFPLSL PFPLSN PFHPSL PFHPSN
 * (fluxes NLEV+1) PCOVPTOT. */
typedef struct cloudsc2_outputs {
  cloudsc2_field tent, tenq, tenl, teni, clc, fplsl, fplsn, fhpsl, fhpsn, covptot;
} cloudsc2_outputs;

/* SATUR + CLOUDSC2 for all blocks: replaces the body of the block loop of
 * src/cloudsc2_nl/cloudsc_driver_mod.F90:82-111 (SATUR call :91, CLOUDSC2 call :94-107).
 * in->qsat.ptr == NULL  => PQS is computed in-kernel from PAP,PT (SATUR fused, satur.F90:106-123);
 * in->qsat.ptr != NULL  => PQS is read (the TL/AD test drivers perturb it independently).
 * zero_plane (optional, may have ptr NULL): an extra (NPROMA,NLEV) plane per block that is zero-filled,
 * i.e. TENDENCY_LOC(IBL)%cld(:,:,NCLV)=0 of cloudsc_driver_mod.F90:88.
 * pert_lambda != 0: every input x is replaced on load by x + pert_lambda*(0.01*x), the perturbed state
 * of the Taylor test (cloudsc_driver_tl_mod.F90:156-171,200-215); with fused SATUR, PQS is perturbed the
 * same way after SATUR on the unperturbed PAP,PT (:159,:203).
 * Asynchronous on `stream`, always: no launcher of this library allocates, copies or synchronises on behalf of the launch
 * heuristics below -- they only read what cloudsc2_device_prepare cached.  (The one thing a launcher may create is the device copy
 * of the per-level table of a CETA it sees for the first time: a 3 KB allocation and a copy through a private non-blocking stream,
 * with the thread's capture mode relaxed; a launch on a stream that is ITSELF being captured needs that table to exist already --
 * any earlier launch or driver call with the same CETA made it.) */
int cloudsc2_nl_launch(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                       const cloudsc2_inputs* in, const cloudsc2_outputs* out, cloudsc2_field zero_plane,
                       double pert_lambda, void* stream);

/* SATUR only (satur.F90:106-123, LDPHYLIN branch): qsat(NPROMA,NLEV,NBLOCKS) from pap, t. */
int cloudsc2_satur_launch(const cloudsc2_params* prm, int nproma, int nlev, int ngptot,
                          cloudsc2_field pap, cloudsc2_field t, cloudsc2_field qsat, void* stream);

/* CLOUDSC2TL (src/cloudsc2_tl/cloudsc2tl.F90:10-24): trajectory in -> trajectory out (give all ten traj_out
 * fields, or all NULL to skip the trajectory stores), perturbation in -> perturbation out.  traj_in->qsat NULL => fused
 * SATUR for PQS5.  pert_in must give all 16 fields.
 * Launch pacing (TL and AD sweeps; results unaffected): these kernels run one wave per SIMD, so a launch is a number of rounds of
 * workgroups on the device's slots; when it is two to eight whole rounds plus a partial one that fills at most half of the slots
 * (160 000 columns on MI355X: 1250 workgroups on 512 slots), the workgroups whose slot has one workgroup less to run nap at every
 * level for 1/k of the level's measured time, leaving their share of the bandwidth to the slots on the critical path: TL -5 %, AD
 * -7 % at 160 000 columns (profiles/r04_pacing_ab.txt) -- on a device whose dispatcher cloudsc2_device_prepare
 * (below) has seen to be in-order and slot-stable, which is what the rule rests on.  CLOUDSC2_PACE=0 in the environment switches it off,
 * CLOUDSC2_PACE_VERBOSE=1 reports paced launches on stderr.  cloudsc2_pace_plan is the rule itself (pure arithmetic, no device):
 * for `workgroups` (of 128 threads) on `slots` workgroup slots it returns 1 when the launch would be paced, with the number of whole
 * rounds, the number of leading positions of every round that form the fast class, and 65536 / k (the nap as a share of a level). */
int cloudsc2_pace_plan(long long workgroups, long long slots, int* whole_rounds, int* fast_first, int* nap_recip_q16);
/* The NL sweep's counterpart for launches that are ONE round of waves (all resident at once; 160 000 columns on MI355X): inside the
 * CUs that carry the most workgroups, the waves on SIMDs with fewer waves than the CU's fullest SIMD nap 15 % of every level
 * (CLOUDSC2_NL_LIGHT=percent, 0 = off): -2 % at 160 000 columns, results unaffected.  How many waves share a SIMD follows from the
 * block index alone; cloudsc2_simd_population is that rule (pure arithmetic, no device): for wave `wave_in_block` (0 or 1) of block
 * `block` of a launch of `workgroups` 128-thread blocks on `cus` CUs, *mine = waves of the launch on its SIMD, *most = on the fullest
 * SIMD of its CU.  tools/wave_times.py checks it against the hardware's own record (HW_ID) wave by wave. */
int cloudsc2_simd_population(long long workgroups, int cus, long long block, int wave_in_block, int* mine, int* most);
/* The library takes neither rule on trust, and checks neither from a launch.  cloudsc2_device_prepare is the one SYNCHRONOUS moment:
 * on the calling thread's current device, once per device and process (later calls return at once), it runs
 *   - the dispatch probe of the NL rule: a 40 us launch of the NL kernel's shape whose waves record where they run (HW_ID); the
 *     rule is checked wave by wave, one miss -- another dispatcher, other work on the device at that moment -- and the lighter
 *     SIMDs' nap stays off for the device;
 *   - the pace probe of the TL / AD rule, for every occupancy (workgroups per CU) a TL / AD kernel of this build has: a 130 us launch
 *     of 2.44 rounds of workgroups that do nothing but stay as long as their class would (fast 40 us; napping 60 us) and record
 *     where and when they ran; per CU the workgroups it ran must be (k+1) of the fast class per fast workgroup of its first round
 *     and k of the slow class per slow one, and the whole first round must have been resident at once.  One miss -- a dispatcher
 *     that is not in-order and slot-stable, CU masking, another partition mode's queueing -- and TL / AD launches are not paced.
 * Each probe: one small allocation, two launches on a private non-blocking stream, one copy back, with the thread's stream-capture
 * mode relaxed meanwhile (a graph capture going on elsewhere in the process is not disturbed); a probe that ends in a HIP error
 * leaves no verdict and is repeated by the next call.  The library calls it from every entry point that allocates device memory
 * (cloudsc2_device_malloc*, cloudsc2_state_create, the host-array drivers' workspace), so callers of those never need to.  A caller
 * that brings its OWN device memory to the kernel-level entry points and wants the naps calls cloudsc2_device_prepare once at
 * start-up; without it every launch simply runs unpaced (results are the same bits either way).  CLOUDSC2_PACE_VERBOSE=1 prints
 * the verdicts.  cloudsc2_dispatch_probe / cloudsc2_pace_probe run one probe and return its counts without caching anything (GPU
 * tests); cloudsc2_device_rules returns the cached verdicts of the current device (1 on, 0 off, -1 never probed);
 * cloudsc2_kernel_occupancy the workgroups per CU of one kernel variant (kernel 0 NL, 1 TL, 2 AD both sweeps, 3 AD reverse sweep;
 * flags = its C2F_* variant bits, cloudsc2_column.hpp) as the runtime reports them.
 *
 * What a caller with its own hipMalloc should expect.  The same kernel on the same data runs 0.78 or 0.92 ms (NL, 160 000 columns:
 * 0.73 vs 0.63 of the HBM peak) depending on WHERE in the HBM the state lies (profiles/r02_hbm_placement.md); cloudsc2_device_malloc*
 * search for a good place, a plain first hipMalloc of a process typically lands on a slow one (bench line:
 * roofline.unplaced_first_allocation).  A host model that allocates its own state gets the kernels' full rate by taking that one
 * allocation from cloudsc2_device_malloc_state (ordinary hipMalloc memory, freed with cloudsc2_device_free), or by allocating
 * candidates itself and keeping the one cloudsc2_device_probe times fastest. */
int cloudsc2_device_prepare(void);
int cloudsc2_dispatch_probe(long long* waves_checked, long long* waves_wrong);
int cloudsc2_pace_probe(int workgroups_per_cu, long long* workgroups_checked, long long* workgroups_wrong);
int cloudsc2_device_rules(int workgroups_per_cu, int* nl_nap, int* pacing);
int cloudsc2_kernel_occupancy(int kernel, int flags, int* workgroups_per_cu);
int cloudsc2_tl_launch(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                       const cloudsc2_inputs* traj_in, const cloudsc2_outputs* traj_out,
                       const cloudsc2_inputs* pert_in, const cloudsc2_outputs* pert_out, void* stream);

/* CLOUDSC2TL with the increments of the reference's two test drivers, dx = 0.01*x for every input
 * (cloudsc_driver_tl_mod.F90:156-171; cloudsc_driver_ad_mod.F90:124-139, where ZSUPSAT = 0): they are formed from the
 * trajectory inputs the sweep reads anyway, so no increment arrays exist (17.5 KB per column less to read, and nothing to
 * fill first).  supsat_increment: the factor of the PSUPSAT increment (0.01 Taylor test, 0 adjoint test).
 * yy: NULL, or NBLOCKS*NPROMA device doubles receiving <y,y> of each active column's TL outputs -- the adjoint test's norm1
 * (cloudsc_driver_ad_mod.F90:184-195), formed while the outputs are in registers. */
int cloudsc2_tl_launch_self(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                            const cloudsc2_inputs* traj_in, const cloudsc2_outputs* traj_out, double supsat_increment,
                            const cloudsc2_outputs* pert_out, double* yy, void* stream);

/* CLOUDSC2AD (src/cloudsc2_ad/cloudsc2ad.F90:10-24): trajectory in -> trajectory out; adj_out holds the
 * output adjoints on entry and is zeroed on return (:917-919,955-966,1173,1572,1678-1691); adj_in is
 * accumulated (+=, :1723-1738) except PSUPSAT which is assigned PTSPHY*zqp1 exactly as the reference does
 * (:1733).  `scratch` must hold (ngptot rounded up to NPROMA blocks) * nlev elements: the precipitation-cover
 * carry checkpoints, written and re-read only when LEVAPLS2 .OR. LDRAIN1D (the evaporation branch is the carried
 * cover's one reader); otherwise it is not touched and may be NULL. */
int cloudsc2_ad_launch(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                       const cloudsc2_inputs* traj_in, const cloudsc2_outputs* traj_out,
                       const cloudsc2_inputs* adj_in, const cloudsc2_outputs* adj_out,
                       cloudsc2_real* scratch, void* stream);

/* The same with the input adjoints ASSIGNED instead of accumulated: adj_in = A^T adj_out, the old contents of adj_in are
 * neither read nor required to be zero (16 planes of reads less per level).  This is what "zero the increments, then call
 * CLOUDSC2AD" (cloudsc_driver_ad_mod.F90:198-237) amounts to; cloudsc2_ad_symmetry_run uses it.  Padded tail columns of the
 * last block are not written. */
int cloudsc2_ad_launch_assign(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                              const cloudsc2_inputs* traj_in, const cloudsc2_outputs* traj_out,
                              const cloudsc2_inputs* adj_in, const cloudsc2_outputs* adj_out,
                              cloudsc2_real* scratch, void* stream);

/* CLOUDSC2AD as its two sweeps.  _forward is the forward sweep (cloudsc2ad.F90:366-866): trajectory in -> trajectory out
 * (+ the cover checkpoints in `scratch` when LEVAPLS2 .OR. LDRAIN1D).  _reverse is the reverse sweep (:877-1740): it re-reads
 * the trajectory inputs, takes the rain / snow flux carried into every level from traj_out->fplsl / fplsn (PFPLSL5, PFPLSN5;
 * the other traj_out fields are not read and may be NULL) and, with the evaporation branch only, the cover from `scratch`;
 * adj_out / adj_in as for cloudsc2_ad_launch (assign != 0: as for cloudsc2_ad_launch_assign).  _forward + _reverse on one
 * stream equals cloudsc2_ad_launch bit for bit.  A caller whose PFPLSL5 / PFPLSN5 are already on the device -- any NL or TL
 * sweep over the same state wrote them, e.g. the TL leg of the adjoint test (cloudsc_driver_ad_mod.F90:160-181) -- runs
 * _reverse alone and saves the trajectory pass (28.5 KB per column of traffic). */
int cloudsc2_ad_launch_forward(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                               const cloudsc2_inputs* traj_in, const cloudsc2_outputs* traj_out,
                               cloudsc2_real* scratch, void* stream);
int cloudsc2_ad_launch_reverse(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                               const cloudsc2_inputs* traj_in, const cloudsc2_outputs* traj_out,
                               const cloudsc2_inputs* adj_in, const cloudsc2_outputs* adj_out,
                               const cloudsc2_real* scratch, int assign, void* stream);

/* The AD leg of the adjoint test with its norms formed in the sweep (cloudsc_driver_ad_mod.F90:198-267): the reverse sweep alone
 * in the assign form (zeroed input adjoints + CLOUDSC2AD), and for every active column norm2 = <x0, x_adj> with x0 = 0.01 * the
 * trajectory inputs (ZSUPSAT0 = 0, :139,240-256) and norm3 = |norm1 - norm2| / EPSILON(1._8) [/ norm2] (:258-264), taken while
 * x_adj is in registers instead of re-reading 32 planes.  norms(3, NBLOCKS*NPROMA) device doubles: row 0 = norm1 on entry
 * (cloudsc2_tl_launch_self's yy, or cloudsc2_adjoint_norms_launch's first half), rows 1-2 written; *blockmax (device double,
 * zero or an earlier maximum on entry) is raised to the largest |norm3| (NaN counts as +inf).  Not with LEVAPLS2 / LDRAIN1D
 * (the reverse sweep alone lacks the cover checkpoints: cloudsc2_ad_launch_assign + cloudsc2_adjoint_norms_launch then). */
int cloudsc2_ad_launch_reverse_norms(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot,
                                     const cloudsc2_inputs* traj_in, const cloudsc2_outputs* traj_out,
                                     const cloudsc2_inputs* adj_in, const cloudsc2_outputs* adj_out,
                                     double* norms, double* blockmax, void* stream);

/* Taylor-test statistics for one lambda (ERROR_NORM, cloudsc_driver_tl_mod.F90:21-31, calls :233-244):
 * for each NPROMA block and each of the 10 output fields, sums over the block's active columns and all
 * levels of (F(x)-F(x+lambda dx)) and of (TL dx * lambda).  sums(NBLOCKS,10,2) device doubles. */
int cloudsc2_taylor_sums_launch(int nproma, int nlev, int ngptot, const cloudsc2_outputs* f,
                                const cloudsc2_outputs* f_pert, const cloudsc2_outputs* tl, double lambda,
                                double* sums, void* stream);

/* The lambda loop of the Taylor test in one sweep (cloudsc_driver_tl_mod.F90:197-244: the ten perturbed CLOUDSC2 calls,
 * ZLAMBDA = 10^-1 .. 10^-10, each followed by the ten ERROR_NORM calls).  The lambdas lie on the lanes of a wave -- 6 columns x
 * 10 lambdas per wave64 --, every lane runs the NL sweep of its column on x + lambda*(0.01*x) (:200-215), compares each level's
 * outputs with the BASE run's (`out`, read) in registers and sums TL output field k of its column; a second small kernel sums
 * the per-column results over the active columns of each block of the statistic (`nproma_stat`: the caller's NPROMA, which may
 * differ from the arrays' blocking `nproma`).  The state is read once for all ten runs and no perturbed outputs are written.
 *   in   the unperturbed state (qsat: the SATUR result, or NULL = SATUR in the sweep)
 *   out  the outputs of the base run (cloudsc2_nl_launch on the same state), tl the TL outputs (cloudsc2_tl_launch)
 *   work device scratch of cloudsc2_taylor_sweep_work_doubles(nproma, ngptot) doubles
 *   sums(2, 10, nblocks_stat, 10) device doubles, the layout of ten cloudsc2_taylor_sums_launch calls one after the other:
 *        sums[((il*nblocks_stat + ibl)*10 + f)*2 + {0,1}] = { sum(F - F5(lambda_il)), sum(TL)*lambda_il }. */
int cloudsc2_taylor_sweep_work_doubles(int nproma, int ngptot, long long* n);
int cloudsc2_taylor_sweep_launch(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot, int nproma_stat,
                                 const cloudsc2_inputs* in, const cloudsc2_outputs* out, const cloudsc2_outputs* tl,
                                 double* work, double* sums, void* stream);

/* Adjoint-test norms per column (cloudsc_driver_ad_mod.F90:184-195,240-264):
 * norm1 = sum_lev sum_10 y*y, norm2 = sum_lev sum_16 (0.01*x_traj)*x_adj (PSUPSAT term uses x0=0, :139),
 * norm3 = |n1-n2|/eps [/n2].  norms(3, ncols_padded) device doubles; *blockmax (device double) receives
 * max(norm3) over all columns via atomic max (must be zeroed by the caller). */
int cloudsc2_adjoint_norms_launch(int nproma, int nlev, int ngptot, const cloudsc2_inputs* traj_in,
                                  const cloudsc2_field* qsat, const cloudsc2_outputs* y,
                                  const cloudsc2_inputs* x_adj, double* norms, double* blockmax, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Driver level: HOST pointers in the GLOBAL_STATE layout, synchronous.  These are what the Fortran
 * drivers with the reference signatures bind through ISO_C_BINDING.  Device buffers
 * are owned by the library and cached between calls (cloudsc2_release_workspace frees them).
 * Argument order follows CLOUDSC_DRIVER (src/cloudsc2_nl/cloudsc_driver_mod.F90:22-30):
 *   PT PQ TENDENCY_CML TENDENCY_LOC PAP PAPH PLU PLUDE PMFU PMFD PA PCLV PSUPSAT PCOVPTOT PFPLSL PFPLSN
 *   PFHPSL PFHPSN, with TENDENCY_* passed as the base address of B_CML / B_LOC.
 * kernel_ms (optional): device time of the kernels only (hipEvent), excluding H2D/D2H.
 * Only what the kernels read is uploaded and only what they write is downloaded (17.5 + 12.1 KB per column);
 * everything else keeps its host value, as under the reference.  cloudsc2_nl_run processes the blocks in slabs of
 * ~16384 columns, uploading slab i+1 from a helper thread while slab i is computed and downloaded (kernel_ms is then
 * the sum over the slabs' launches).
 * ------------------------------------------------------------------------------------------------ */
int cloudsc2_nl_run(const cloudsc2_params* prm, int nproma, int nlev, int ngptot, double ptsphy,
                    const cloudsc2_real* pt, const cloudsc2_real* pq, const cloudsc2_real* b_cml, cloudsc2_real* b_loc,
                    const cloudsc2_real* pap, const cloudsc2_real* paph, const cloudsc2_real* plu, const cloudsc2_real* plude,
                    const cloudsc2_real* pmfu, const cloudsc2_real* pmfd, cloudsc2_real* pa, const cloudsc2_real* pclv,
                    const cloudsc2_real* psupsat, cloudsc2_real* pcovptot, cloudsc2_real* pfplsl, cloudsc2_real* pfplsn,
                    cloudsc2_real* pfhpsl, cloudsc2_real* pfhpsn, double* kernel_ms);

/* CLOUDSC_DRIVER_TL (src/cloudsc2_tl/cloudsc_driver_tl_mod.F90:33-314): NL, 1 % increments, TL, ten
 * perturbed NL runs, ERROR_NORM per block, max over blocks.  znormg[10] receives the raw ratios
 * (the values printed at :275).  Returns CLOUDSC2_ETLWRONG where the reference STOPs (:247-249). */
int cloudsc2_tl_taylor_run(const cloudsc2_params* prm, int nproma, int nlev, int ngptot, double ptsphy,
                           const cloudsc2_real* pt, const cloudsc2_real* pq, const cloudsc2_real* b_cml, cloudsc2_real* b_loc,
                           const cloudsc2_real* pap, const cloudsc2_real* paph, const cloudsc2_real* plu, const cloudsc2_real* plude,
                           const cloudsc2_real* pmfu, const cloudsc2_real* pmfd, cloudsc2_real* pa, const cloudsc2_real* pclv,
                           const cloudsc2_real* psupsat, cloudsc2_real* pcovptot, cloudsc2_real* pfplsl, cloudsc2_real* pfplsn,
                           cloudsc2_real* pfhpsl, cloudsc2_real* pfhpsn, double znormg[10], double* kernel_ms);

/* CLOUDSC_DRIVER_AD (src/cloudsc2_ad/cloudsc_driver_ad_mod.F90:22-297): TL, norm1, zero, AD, norm2,
 * norm3; *znormg = max over columns of norm3 (the value printed at :287). */
int cloudsc2_ad_symmetry_run(const cloudsc2_params* prm, int nproma, int nlev, int ngptot, double ptsphy,
                             const cloudsc2_real* pt, const cloudsc2_real* pq, const cloudsc2_real* b_cml, cloudsc2_real* b_loc,
                             const cloudsc2_real* pap, const cloudsc2_real* paph, const cloudsc2_real* plu, const cloudsc2_real* plude,
                             const cloudsc2_real* pmfu, const cloudsc2_real* pmfd, cloudsc2_real* pa, const cloudsc2_real* pclv,
                             const cloudsc2_real* psupsat, cloudsc2_real* pcovptot, cloudsc2_real* pfplsl, cloudsc2_real* pfplsn,
                             cloudsc2_real* pfhpsl, cloudsc2_real* pfhpsn, double* znormg, double* kernel_ms);

void cloudsc2_release_workspace(void);

/* ------------------------------------------------------------------------------------------------
 * Resident state: a GLOBAL_STATE (cloudsc2_array_state_mod.F90:26-79) that lives on the device and is owned by the
 * library -- for callers whose model keeps its physics state in HBM, and for the Fortran mains' CLOUDSC2_RESIDENT=1 mode.
 * The host-pointer entry points above spend 98 % of their time on PCIe; with a handle the flow of the reference main
 * (LOAD -> driver -> VALIDATE) never builds the NPROMA-blocked arrays on the host:
 *   cloudsc2_state_create     allocates all arrays as ONE placed allocation (cloudsc2_device_malloc), zero-filled
 *                             (FIELD_INIT, :186-190); the layout is the reference's (include file header).
 *   cloudsc2_state_expand     EXPAND_R2 (expand_mod.F90:270-302) of one field from a KLON-column HOST table (KLON, NLEVx),
 *                             `period`/`start` as cloudsc2_expand_offsets gives them; repeated calls fill the state.
 *   cloudsc2_state_upload / _download   the same transfers as the host-pointer drivers (what the kernels read / write).
 *   cloudsc2_state_nl         SATUR + CLOUDSC2 over all blocks, `repeats` times back to back; *kernel_ms = mean device time.
 *   cloudsc2_state_tl_taylor, cloudsc2_state_ad_symmetry   the two self-tests of cloudsc2_tl_taylor_run / _ad_symmetry_run
 *                             on the resident state (their scratch arrays are a second allocation owned by the handle).
 *   cloudsc2_state_validate   VALIDATE_R2/R3 (validate_mod.F90:165-261) of one field against a KLON-column HOST reference
 *                             table (KLON, NLEVx, ndim): stats[5] as cloudsc2_validate_launch.
 *   cloudsc2_state_field      the device pointer + block stride of one field, for the kernel-level entry points -- in the
 *                             DEVICE blocking (cloudsc2_state_blocking), which is the library's choice, see below.
 * Blocking.  `nproma` of cloudsc2_state_create is the CALLER's NPROMA: the blocking of the host arrays of _upload / _download, the
 * block ERROR_NORM of the Taylor test sums over (cloudsc_driver_tl_mod.F90:21-31) and the blocks whose padding MINVAL / MAXVAL of
 * the validator see (validate_mod.F90:186-187).  The device arrays themselves are blocked for the kernels: the caller's NPROMA
 * when it is a multiple of 64, else 128 (CLOUDSC2_STATE_NPROMA=0: always the caller's) -- every column is independent, so the
 * results are the same bits whatever the blocking, and NPROMA 32 (the reference README's) or 100 cost the TL / AD sweeps 8 %.
 * A handle belongs to the HIP device that was current when it was created; calls are synchronous.
 * ------------------------------------------------------------------------------------------------ */
typedef struct cloudsc2_state cloudsc2_state;
enum {
  CLOUDSC2_F_PT = 0, CLOUDSC2_F_PQ, CLOUDSC2_F_PAP, CLOUDSC2_F_PAPH, CLOUDSC2_F_PLU, CLOUDSC2_F_PLUDE, CLOUDSC2_F_PMFU,
  CLOUDSC2_F_PMFD, CLOUDSC2_F_PA, CLOUDSC2_F_PSUPSAT, CLOUDSC2_F_PCOVPTOT,                  /* 0..10  */
  CLOUDSC2_F_PFPLSL, CLOUDSC2_F_PFPLSN, CLOUDSC2_F_PFHPSL, CLOUDSC2_F_PFHPSN, CLOUDSC2_F_QSAT, /* 11..15 */
  CLOUDSC2_F_CML_T = 16, /* + plane: T A Q QL QI QR QS QV = 16..23 (TENDENCY_CML) */
  CLOUDSC2_F_LOC_T = 24, /* 24..31 (TENDENCY_LOC) */
  CLOUDSC2_F_PCLV_QL = 32 /* 32..36 (PCLV: QL QI QR QS QV) */
};
int cloudsc2_state_create(int nproma, int nlev, int ngptot, cloudsc2_state** state);
void cloudsc2_state_destroy(cloudsc2_state* state);
int cloudsc2_state_field(const cloudsc2_state* state, int field, cloudsc2_field* f);
int cloudsc2_state_blocking(const cloudsc2_state* state, int* nproma_device, int* nproma_caller);
int cloudsc2_state_expand(cloudsc2_state* state, int field, const cloudsc2_real* table, int klon, int period, long long start);
int cloudsc2_state_upload(cloudsc2_state* state,
                          const cloudsc2_real* pt, const cloudsc2_real* pq, const cloudsc2_real* b_cml, cloudsc2_real* b_loc,
                          const cloudsc2_real* pap, const cloudsc2_real* paph, const cloudsc2_real* plu, const cloudsc2_real* plude,
                          const cloudsc2_real* pmfu, const cloudsc2_real* pmfd, cloudsc2_real* pa, const cloudsc2_real* pclv,
                          const cloudsc2_real* psupsat, cloudsc2_real* pcovptot, cloudsc2_real* pfplsl, cloudsc2_real* pfplsn,
                          cloudsc2_real* pfhpsl, cloudsc2_real* pfhpsn);
int cloudsc2_state_download(cloudsc2_state* state, cloudsc2_real* b_loc, cloudsc2_real* pa, cloudsc2_real* pcovptot,
                            cloudsc2_real* pfplsl, cloudsc2_real* pfplsn, cloudsc2_real* pfhpsl, cloudsc2_real* pfhpsn);
int cloudsc2_state_nl(cloudsc2_state* state, const cloudsc2_params* prm, double ptsphy, int repeats, double* kernel_ms);
int cloudsc2_state_tl_taylor(cloudsc2_state* state, const cloudsc2_params* prm, double ptsphy, double znormg[10], double* kernel_ms);
int cloudsc2_state_ad_symmetry(cloudsc2_state* state, const cloudsc2_params* prm, double ptsphy, double* znormg, double* kernel_ms);
int cloudsc2_state_validate(cloudsc2_state* state, int field, int ndim, const cloudsc2_real* ref_table, int klon, int period,
                            long long start, double stats[5]);

/* The synthetic KLON-column atmosphere every front end of this repository loads when config-files/input.h5 is absent (it is not
 * distributed with the reference: .MISSING_LARGE_BLOBS) -- what CLOUDSC2_ARRAY_STATE_LOAD (cloudsc2_array_state_mod.F90:153-204) would
 * read from the file: PT PQ PAP PAPH(nlev+1) PLU PLUDE PMFU PMFD PCLV(QL) PCLV(QI) TENDENCY_CML%T TENDENCY_CML%Q, each (nlev[+1], klon)
 * row-major = Fortran (KLON, KLEV[+1]), always double; everything else of the state is zero.  Pure host code, one implementation for
 * the Fortran mains and the Python harness: the Taylor test's verdict is decided by round-off, so tables that differ
 * in the last place (numpy's, flang's and glibc's exp / pow do) give different verdicts for the same library and size.  rd, rv, rtt:
 * YOMCST's RD, RV, RTT (287.0597, 461.5250, 273.16). */
int cloudsc2_synthetic_table(int klon, int nlev, double rd, double rv, double rtt, double* pt, double* pq, double* pap, double* paph,
                             double* plu, double* plude, double* pmfu, double* pmfd, double* pql, double* pqi, double* tend_t,
                             double* tend_q);

/* Verdict logic of the two self-tests, pure host code (no device needed).
 * cloudsc2_taylor_verdict: cloudsc_driver_tl_mod.F90:272-311; znormg = raw ratios; returns 1 = PASSED;
 *   *itest = penalty / error code (13 when no lambda <= 1e-4 reaches |1-ratio| < 0.5).
 * cloudsc2_adjoint_verdict: cloudsc_driver_ad_mod.F90:289; returns 1 = TEST OK. */
int cloudsc2_taylor_verdict(const double znormg[10], int* itest);
int cloudsc2_adjoint_verdict(double znormg);

/* --------------------------------------------------------------------------------------------------
 * Data formats either side of the path, device side.  The HDF5 files hold
 * KLON-column tables, stored (…, KLON) in C order = Fortran (KLON, …) (hdf5_file_mod.F90); the file
 * reader/writer itself is include/cloudsc2_io.h (libcloudsc2_io.so, needs libhdf5).
 *
 * cloudsc2_expand_launch  replaces EXPAND_R2/R3 (src/common/module/expand_mod.F90:270-335) for data that
 *   stays on the GPU: field(jl,jk,jm,ibl) = table((start + (ibl*NPROMA + jl) mod period) mod KLON, jk, jm), zero in
 *   the padded tail of the last block.  `table` (device) is (KLON, nlevx, ndim) column-fastest; `period` and
 *   `start` are GET_OFFSETS' size and start-1 (expand_mod.F90:30-46; see cloudsc2_expand_offsets): the rank's
 *   table slice START..END tiled with period SIZE, as LOAD_AND_EXPAND does (:101-116).  (For those pairs start + period
 *   <= KLON and the outer mod never acts; with period = KLON any start >= 0 continues the periodic tiling at global
 *   column `start`.)  The
 *   reference indexes out of bounds when a block starts at a multiple of KLON other than KLON itself
 *   (MOD(gidx,nlon) = 0, :289); this implements the periodic tiling it intends.
 * cloudsc2_validate_launch  replaces VALIDATE_R2/R3 (src/common/module/validate_mod.F90:165-261) without
 *   expanding the reference: stats[5] (device) = { min FIELD, max FIELD (whole blocks, padding included),
 *   max |FIELD-REF|, sum |FIELD-REF|, sum |REF| (active columns) }.  `workspace` (device) needs
 *   cloudsc2_validate_workspace_doubles() doubles.  Sums are folded in a fixed order (deterministic).
 * ------------------------------------------------------------------------------------------------ */
int cloudsc2_expand_launch(const cloudsc2_real* table, int klon, int period, long long start, int nlevx, int ndim,
                           int nproma, long long ngptot, cloudsc2_field field, void* stream);
int cloudsc2_validate_workspace_doubles(void);
int cloudsc2_validate_launch(const cloudsc2_real* table, int klon, int period, long long start, int nlevx, int ndim,
                             int nproma, long long ngptot, cloudsc2_field field, double* workspace,
                             double* stats, void* stream);
/* GET_OFFSETS (expand_mod.F90:30-46): which table columns rank `irank` of `numproc` tiles from.
 * ngptotg <= 0 means "not given".  *start is 0-based. */
void cloudsc2_expand_offsets(int klon, long long ngptot, long long ngptotg, int irank, int numproc,
                             long long* start, int* period);
/* ERROR_PRINT (validate_mod.F90:263-296): relative error in percent, option code 1..3, and whether the
 * line carries the "!!!!" warning (relative error > 10 eps).  Pure host code.  zavgpgp = esum / ngptotg. */
double cloudsc2_validate_relerr(double esum, double rsum, int* iopt, int* warn);
/* The line ERROR_PRINT writes, FORMAT(1X,A20,1X,I1,'D',I1,5(1X,E20.13),A); buf needs >= 160 bytes. */
int cloudsc2_validate_format(const char* name, int ndim, const double stats[5], long long ngptotg,
                             char* buf, int buflen);
/* The header line CLOUDSC2_ARRAY_STATE_VALIDATE prints first (cloudsc2_array_state_mod.F90:226-229). */
int cloudsc2_validate_header(char* buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDSC2_HIP_H */
