! CLOUDSC_DRIVER_TL with the reference's signature (src/cloudsc2_tl/cloudsc_driver_tl_mod.F90:33-64): the Taylor
! test.  NL trajectory, 1 % increments, TL and the ten perturbed NL runs with their ERROR_NORM sums run on the GPU
! (cloudsc2_tl_taylor_run); the verdict logic and print-out below are the reference's (:272-311).
MODULE CLOUDSC_DRIVER_TL_MOD
  USE, INTRINSIC :: ISO_C_BINDING
  USE PARKIND1,  ONLY : JPIM, JPRB
  USE YOMPHYDER, ONLY : STATE_TYPE
  USE CLOUDSC2_HIP_MOD
  USE CLOUDSC_DRIVER_MOD, ONLY : CLOUDSC2_PRINT_PERFORMANCE
  USE CLOUDSC_MPI_MOD, ONLY : NUMPROC, IRANK, CLOUDSC_MPI_REDUCE_MAX

  IMPLICIT NONE

CONTAINS

  SUBROUTINE CLOUDSC_DRIVER_TL( &
     & NUMOMP, NPROMA, NLEV, NGPTOT, NGPTOTG, PTSPHY, &
     & PT, PQ, TENDENCY_CML, TENDENCY_LOC, &
     & PAP,      PAPH, &
     & PLU,      PLUDE,    PMFU,     PMFD, &
     & PA,       PCLV,     PSUPSAT,&
     & PCOVPTOT, &
     & PFPLSL,   PFPLSN,   PFHPSL,   PFHPSN &
     & )
    INTEGER(KIND=JPIM), INTENT(IN)    :: NUMOMP, NPROMA, NLEV, NGPTOT, NGPTOTG
    REAL(KIND=JPRB),    INTENT(IN)    :: PTSPHY
    REAL(KIND=JPRB),    INTENT(IN), TARGET, CONTIGUOUS :: PT(:,:,:), PQ(:,:,:)
    TYPE(STATE_TYPE),   INTENT(IN), TARGET    :: TENDENCY_CML(:)
    TYPE(STATE_TYPE),   INTENT(OUT), TARGET   :: TENDENCY_LOC(:)
    REAL(KIND=JPRB),    INTENT(IN), TARGET, CONTIGUOUS :: PAP(:,:,:), PAPH(:,:,:), PLU(:,:,:)
    REAL(KIND=JPRB),    INTENT(INOUT), TARGET, CONTIGUOUS :: PLUDE(:,:,:)
    REAL(KIND=JPRB),    INTENT(IN), TARGET, CONTIGUOUS :: PMFU(:,:,:), PMFD(:,:,:)
    REAL(KIND=JPRB),    INTENT(IN), TARGET, CONTIGUOUS :: PA(:,:,:)
    REAL(KIND=JPRB),    INTENT(IN), TARGET, CONTIGUOUS :: PCLV(:,:,:,:), PSUPSAT(:,:,:)
    REAL(KIND=JPRB),    INTENT(INOUT), TARGET, CONTIGUOUS :: PCOVPTOT(:,:,:)
    REAL(KIND=JPRB),    INTENT(OUT), TARGET, CONTIGUOUS :: PFPLSL(:,:,:), PFPLSN(:,:,:), PFHPSL(:,:,:), PFHPSN(:,:,:)

    TYPE(CLOUDSC2_PARAMS_T) :: PRM
    INTEGER(KIND=JPIM) :: NGPBLKS, IRC
    REAL(C_DOUBLE) :: ZKERNEL_MS, ZNORMG(10)
    INTEGER(KIND=8) :: ICLK0, ICLK1, IRATE
    LOGICAL, PARAMETER :: LDRAIN1D = .FALSE.

    NGPBLKS = (NGPTOT / NPROMA) + MIN(MOD(NGPTOT,NPROMA), 1)
1003 format(5x,'NUMPROC=',i0,', NUMOMP=',i0,', NGPTOTG=',i0,', NPROMA=',i0,', NGPBLKS=',i0)
    IF (IRANK == 0) WRITE(0,1003) NUMPROC, NUMOMP, NGPTOTG, NPROMA, NGPBLKS

    CALL CLOUDSC2_FILL_PARAMS(PRM, NLEV, LDRAIN1D)
    CALL SYSTEM_CLOCK(ICLK0, IRATE)
    IRC = CLOUDSC2_TL_TAYLOR_RUN(PRM, NPROMA, NLEV, NGPTOT, REAL(PTSPHY,C_DOUBLE), &
     & C_LOC(PT), C_LOC(PQ), CLOUDSC2_STATE_BASE(TENDENCY_CML, NPROMA, NLEV, 'TENDENCY_CML'), &
     & CLOUDSC2_STATE_BASE(TENDENCY_LOC, NPROMA, NLEV, 'TENDENCY_LOC'), &
     & C_LOC(PAP), C_LOC(PAPH), C_LOC(PLU), C_LOC(PLUDE), C_LOC(PMFU), C_LOC(PMFD), C_LOC(PA), C_LOC(PCLV), &
     & C_LOC(PSUPSAT), C_LOC(PCOVPTOT), C_LOC(PFPLSL), C_LOC(PFPLSN), C_LOC(PFHPSL), C_LOC(PFHPSN), ZNORMG, ZKERNEL_MS)
    CALL SYSTEM_CLOCK(ICLK1)
    IF (IRC /= 0 .AND. IRC /= -3) CALL CLOUDSC2_FAIL('cloudsc2_tl_taylor_run failed', IRC)
    IF (IRC == -3) ZNORMG(:) = HUGE(1.0_C_DOUBLE)   ! "TL is totally wrong" on this rank: carried through the reduction
    CALL CLOUDSC2_PRINT_PERFORMANCE(NUMOMP, NPROMA, NGPBLKS, NGPTOT, ZKERNEL_MS, REAL(ICLK1-ICLK0,C_DOUBLE)/REAL(IRATE,C_DOUBLE))
    CALL CLOUDSC2_REPORT_TAYLOR(ZNORMG)
  END SUBROUTINE CLOUDSC_DRIVER_TL

  ! The verdict of the Taylor test.  The reference max-reduces ZNORMG over the OpenMP threads of one process
  ! (cloudsc_driver_tl_mod.F90:125); with the columns split over GPUs the same MAX is taken over the ranks (RCCL), then rank 0
  ! evaluates the test and prints the output exactly like the reference (:247-249,272-311).
  SUBROUTINE CLOUDSC2_REPORT_TAYLOR(ZNORMG)
    REAL(C_DOUBLE), INTENT(INOUT) :: ZNORMG(10)
    INTEGER(KIND=JPIM) :: ILAM, ITEST, IOK
    CALL CLOUDSC_MPI_REDUCE_MAX(ZNORMG, 10, 0)
    IF (ANY(ZNORMG >= HUGE(1.0_C_DOUBLE))) THEN
      IF (IRANK == 0) print *, ' TL is totally wrong !!! '
      STOP
    ENDIF
    IF (IRANK /= 0) RETURN
    print *, ' TL Taylor test '
    print *, '                Lambda   Result'
    DO ILAM=1,10
      print *, ILAM, ZNORMG(ILAM)
    ENDDO
    IOK = CLOUDSC2_TAYLOR_VERDICT(ZNORMG, ITEST)
    print *, '   ==============================================   '
    IF (IOK == 0) THEN
      print *, '       TEST FAILLED, err ',ITEST
    ELSE
      print *, '       TEST PASSED, penalty ',ITEST
    ENDIF
    print *, '   ==============================================   '
  END SUBROUTINE CLOUDSC2_REPORT_TAYLOR

END MODULE CLOUDSC_DRIVER_TL_MOD
