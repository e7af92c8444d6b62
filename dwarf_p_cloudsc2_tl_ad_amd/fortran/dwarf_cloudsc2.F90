! dwarf-cloudsc2-{nl,tl,ad}: the reference's three command lines (NUMOMP NGPTOTG NPROMA, defaults 1 / 16384 / 32,
! src/cloudsc2_nl/dwarf_cloudsc.F90:27-29,48-75) on top of the HIP drivers, with the flow of the reference's mains
! (:79-124): GLOBAL_STATE%LOAD -> module set-up -> CLOUDSC_DRIVER* -> GLOBAL_STATE%VALIDATE [-> WRITE_REFERENCE].
! Select the program with -DDWARF_NL, -DDWARF_TL or -DDWARF_AD.  input.h5 / reference.h5 are read from the working
! directory when present (HDF5 C API, cloudsc2_hip_state_mod.F90); config-files/input.h5 is not distributed with the
! reference, so without it the state is the synthetic 100-column atmosphere of SURVEY.md 8d.
!
! Where the state lives (CLOUDSC2_RESIDENT in the environment):
!   unset  dwarf-cloudsc2-nl runs BOTH: first the driver's work on a state RESIDENT on the GPU (cloudsc2_state_*: the tables go up,
!          are tiled there, the kernel's rate is what a model with its physics state in HBM gets), then the reference's own flow --
!          host arrays through CLOUDSC_DRIVER, which moves 29.6 KB per column over PCIe per call and is bound by that -- and prints
!          the two rates side by side (the validation is the reference flow's).  States above 400 000 columns (16 GB of host arrays)
!          run resident only.  dwarf-cloudsc2-tl / -ad (the self-tests) run resident: their verdicts are the same bits either way
!          (tests/test_gpu_parity.py::test_taylor_verdict_over_the_full_block_set).
!   1      resident only; nothing but tables and statistics crosses PCIe.
!   0      the reference-identical flow only: GLOBAL_STATE%LOAD builds the NPROMA-blocked arrays on the host, CLOUDSC_DRIVER* takes
!          them (src/cloudsc2_nl/dwarf_cloudsc.F90:79-122).
PROGRAM DWARF_CLOUDSC
  USE PARKIND1,  ONLY : JPIM, JPRB
  USE YOECLD,  ONLY : YRECLD
  USE YOEPHLI, ONLY : YREPHLI
  USE YOPHNC,  ONLY : YRPHNC
  USE YOMNCL,  ONLY : YRNCL
  USE CLOUDSC2_HIP_STATE_MOD, ONLY : CLOUDSC2_HIP_STATE
  USE CLOUDSC_MPI_MOD, ONLY : CLOUDSC_MPI_INIT, CLOUDSC_MPI_END, NUMPROC, IRANK
#if defined(DWARF_TL)
  USE CLOUDSC_DRIVER_TL_MOD, ONLY : CLOUDSC_DRIVER_TL
#elif defined(DWARF_AD)
  USE CLOUDSC_DRIVER_AD_MOD, ONLY : CLOUDSC_DRIVER_AD
#else
  USE CLOUDSC_DRIVER_MOD, ONLY : CLOUDSC_DRIVER
#endif
  USE CLOUDSC_DRIVER_MOD, ONLY : CLOUDSC2_LAST_KERNEL_MS, CLOUDSC2_LAST_WALL_S
  USE, INTRINSIC :: ISO_C_BINDING, ONLY : C_DOUBLE
  IMPLICIT NONE

  CHARACTER(LEN=20) :: CLARG
  CHARACTER(LEN=1) :: WRITE_REFERENCE
  CHARACTER(LEN=8) :: CLRES
  INTEGER(KIND=JPIM) :: IARGS, NUMOMP, NGPTOT, NGPTOTG, NPROMA, IWHICH
  INTEGER(KIND=JPIM) :: IMODE   ! 0 host arrays only, 1 resident only, 2 both (resident first, then the reference flow)
  INTEGER(KIND=JPIM), PARAMETER :: NGPTOT_BOTH_MAX = 400000
  REAL(C_DOUBLE) :: ZRES_MS
  TYPE(CLOUDSC2_HIP_STATE), TARGET :: GLOBAL_STATE, RESIDENT_STATE

  ! --- command line (dwarf_cloudsc.F90:48-75)
  NUMOMP = 1; NGPTOTG = 16384; NPROMA = 32
  IARGS = COMMAND_ARGUMENT_COUNT()
  IF (IARGS >= 1) THEN
    CALL GET_COMMAND_ARGUMENT(1, CLARG); READ(CLARG,*) NUMOMP
  ENDIF
  ! one process per GPU: ranks from the launcher's environment, HIP device LOCAL_RANK, RCCL communicator (dwarf_cloudsc.F90:56)
  CALL CLOUDSC_MPI_INIT(NUMOMP)
  IF (IARGS >= 2) THEN
    CALL GET_COMMAND_ARGUMENT(2, CLARG); READ(CLARG,*) NGPTOTG
  ENDIF
  IF (IARGS >= 3) THEN
    CALL GET_COMMAND_ARGUMENT(3, CLARG); READ(CLARG,*) NPROMA
  ENDIF
  ! the local number of grid points (dwarf_cloudsc.F90:64-69)
  NGPTOT = (NGPTOTG - 1) / NUMPROC + 1
  IF (IRANK == NUMPROC - 1) NGPTOT = NGPTOTG - (NUMPROC - 1) * NGPTOT
  WRITE_REFERENCE = '0'
  CALL GET_ENVIRONMENT_VARIABLE('CLOUDSC2_WRITE_REFERENCE', WRITE_REFERENCE)

#if defined(DWARF_TL)
  IWHICH = 1
#elif defined(DWARF_AD)
  IWHICH = 2
#else
  IWHICH = 0
#endif
  CLRES = ' '
  CALL GET_ENVIRONMENT_VARIABLE('CLOUDSC2_RESIDENT', CLRES)
  IF (CLRES(1:1) == '1') THEN
    IMODE = 1
  ELSEIF (CLRES(1:1) == '0') THEN
    IMODE = 0
  ELSEIF (IWHICH /= 0 .OR. NGPTOT > NGPTOT_BOTH_MAX .OR. WRITE_REFERENCE == '1') THEN
    IMODE = 1
    IF (WRITE_REFERENCE == '1') IMODE = 0   ! (WRITE_REFERENCE needs the outputs on the host)
  ELSE
    IMODE = 2
  ENDIF

  ZRES_MS = 0.0_C_DOUBLE
  IF (IMODE == 2) THEN
    ! the driver's work on a device-resident state first: what the kernel delivers when nothing crosses PCIe
    CALL RESIDENT_STATE%LOAD(NPROMA, NGPTOT, NGPTOTG, LDRESIDENT=.TRUE.)
    CALL SETUP_MODULES(RESIDENT_STATE)
    CALL RESIDENT_STATE%RUN_RESIDENT(IWHICH, NUMOMP, NPROMA, NGPTOT, NGPTOTG)
    ZRES_MS = CLOUDSC2_LAST_KERNEL_MS
    CALL RESIDENT_STATE%RELEASE()
  ENDIF

  CALL GLOBAL_STATE%LOAD(NPROMA, NGPTOT, NGPTOTG, LDRESIDENT=(IMODE == 1))
  CALL SETUP_MODULES(GLOBAL_STATE)
  IF (GLOBAL_STATE%RESIDENT) THEN
    ! the state never exists on the host; the driver's work runs on the device-resident state
    CALL GLOBAL_STATE%RUN_RESIDENT(IWHICH, NUMOMP, NPROMA, NGPTOT, NGPTOTG)
  ELSE
#if defined(DWARF_TL)
  CALL CLOUDSC_DRIVER_TL(NUMOMP, NPROMA, GLOBAL_STATE%KLEV, NGPTOT, NGPTOTG, GLOBAL_STATE%PTSPHY, &
#elif defined(DWARF_AD)
  CALL CLOUDSC_DRIVER_AD(NUMOMP, NPROMA, GLOBAL_STATE%KLEV, NGPTOT, NGPTOTG, GLOBAL_STATE%PTSPHY, &
#else
  CALL CLOUDSC_DRIVER(NUMOMP, NPROMA, GLOBAL_STATE%KLEV, NGPTOT, NGPTOTG, GLOBAL_STATE%PTSPHY, &
#endif
   & GLOBAL_STATE%PT, GLOBAL_STATE%PQ, GLOBAL_STATE%TENDENCY_CML, GLOBAL_STATE%TENDENCY_LOC, &
   & GLOBAL_STATE%PAP, GLOBAL_STATE%PAPH, GLOBAL_STATE%PLU, GLOBAL_STATE%PLUDE, GLOBAL_STATE%PMFU, GLOBAL_STATE%PMFD, &
   & GLOBAL_STATE%PA, GLOBAL_STATE%PCLV, GLOBAL_STATE%PSUPSAT, GLOBAL_STATE%PCOVPTOT, &
   & GLOBAL_STATE%PFPLSL, GLOBAL_STATE%PFPLSN, GLOBAL_STATE%PFHPSL, GLOBAL_STATE%PFHPSN)
  ENDIF
  IF (IMODE == 2 .AND. IRANK == 0 .AND. ZRES_MS > 0.0_C_DOUBLE .AND. CLOUDSC2_LAST_WALL_S > 0.0_C_DOUBLE) THEN
    ! the two ways of running the same sweep, side by side: the kernel on a resident state, and the reference-signature call that
    ! carries the state over PCIe both ways (never the kernel's rate: the gap is the link, not the library)
    WRITE(0,'(1X,A,F9.3,A,ES10.3,A)') 'state resident on the GPU (cloudsc2_state_*, CLOUDSC2_RESIDENT=1): ', ZRES_MS, ' ms = ', &
     & REAL(NGPTOT,C_DOUBLE)/(ZRES_MS*1.0E-3_C_DOUBLE), ' columns/s per sweep'
    WRITE(0,'(1X,A,F9.3,A,ES10.3,A,F7.1,A)') 'CLOUDSC_DRIVER on host arrays (the reference flow, PCIe-bound):   ', &
     & CLOUDSC2_LAST_WALL_S*1.0E3_C_DOUBLE, ' ms = ', REAL(NGPTOT,C_DOUBLE)/CLOUDSC2_LAST_WALL_S, ' columns/s, first call  (', &
     & CLOUDSC2_LAST_WALL_S*1.0E3_C_DOUBLE/ZRES_MS, ' x the resident sweep; one-off workspace allocation and first touch included: repeated calls take 53-87 ms per 160 000 columns)'
  ENDIF

#if !defined(DWARF_TL) && !defined(DWARF_AD)
  ! Validate the output against reference.h5 (dwarf_cloudsc.F90:117-122)
  CALL GLOBAL_STATE%VALIDATE(NPROMA, NGPTOT, NGPTOTG)
  IF (WRITE_REFERENCE == '1') CALL GLOBAL_STATE%WRITE_REFERENCE(NPROMA, NGPTOT)
#endif
  CALL CLOUDSC_MPI_END()   ! dwarf_cloudsc.F90:125

CONTAINS
  ! --- the set-up of the reference mains (dwarf_cloudsc.F90:83-107), from whichever state was loaded first (CETA needs block 1 of PAP / PAPH)
  SUBROUTINE SETUP_MODULES(ST)
    TYPE(CLOUDSC2_HIP_STATE), INTENT(IN) :: ST
    INTEGER(KIND=JPIM) :: JK
    IF (ASSOCIATED(YRECLD)) RETURN
    ALLOCATE(YRECLD)
    ALLOCATE(YRECLD%CETA(ST%KLEV))
    IF (ST%KLEV > 200) THEN
      PRINT *, ' Dimension of ZPRES/ZPRESF is too short. '
      STOP
    ENDIF
    DO JK = 1, ST%KLEV
      YRECLD%CETA(JK) = ST%PAP(1,JK,1)/ST%PAPH(1,ST%KLEV+1,1)
    ENDDO
    ALLOCATE(YRPHNC)
    YRPHNC%LEVAPLS2 = .FALSE.
    YREPHLI%LPHYLIN = .TRUE.
    ALLOCATE(YRNCL)
#if defined(DWARF_AD)
    YRNCL%LREGCL = .TRUE.
#else
    YRNCL%LREGCL = .FALSE.
#endif
  END SUBROUTINE SETUP_MODULES
END PROGRAM DWARF_CLOUDSC
