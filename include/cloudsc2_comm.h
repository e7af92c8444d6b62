/*
 * cloudsc2_comm.h -- the few collectives of the dwarf, over RCCL / xGMI: C ABI of libcloudsc2_comm.so.
 *
 * Replaces src/common/module/cloudsc_mpi_mod.F90 of the reference (CLOUDSC_MPI_INIT :58-88, CLOUDSC_MPI_END :90-100,
 * CLOUDSC_MPI_REDUCE_{SUM,MIN,MAX} :102-262, CLOUDSC_MPI_GATHER :264-327): one process per GPU, columns split over the
 * ranks (dwarf_cloudsc.F90:66-69), no collective on the data path; what is exchanged are a handful of numbers -- the
 * validation statistics (validate_mod.F90:197-199), the timer table (timer_mod.F90:155) and, new with this build, the
 * verdict norms of the Taylor and adjoint tests (cloudsc_driver_tl_mod.F90:125, cloudsc_driver_ad_mod.F90:107), which the
 * reference max-reduces over OpenMP threads only.  fortran/cloudsc_mpi_mod.F90 is the same-named Fortran module on top of it.
 *
 * Ranks come from the launcher's environment: RANK / WORLD_SIZE / LOCAL_RANK (torchrun, bench.py --gpus N,
 * tools/launch_ranks.sh), else OMPI_COMM_WORLD_{RANK,SIZE,LOCAL_RANK}, else SLURM_{PROCID,NTASKS,LOCALID}; none set = one
 * rank.  cloudsc2_comm_init selects HIP device LOCAL_RANK and builds the RCCL communicator; the ncclUniqueId travels through
 * a file in CLOUDSC2_COMM_DIR (default /tmp) named after MASTER_PORT and the launcher's pid.  A caller that already has a
 * way to broadcast 128 bytes (bench.py: torch.distributed) uses cloudsc2_comm_unique_id + cloudsc2_comm_init_rank instead.
 * Transport "shm" (CLOUDSC2_COMM=shm): the same calls through a POSIX shared-memory segment, for rehearsing more ranks than
 * the node has GPUs (RCCL refuses two ranks on one device) -- the counterpart of CLOUDSC2_DIST_BACKEND=gloo on the Python
 * side; it carries only these few doubles, never field data.
 * All functions return 0 or a negative CLOUDSC2_COMM_E* / positive hipError_t / ncclResult_t code; buffers are HOST memory.
 */
#ifndef CLOUDSC2_COMM_H
#define CLOUDSC2_COMM_H
#ifdef __cplusplus
extern "C" {
#endif

#define CLOUDSC2_COMM_EINVAL   (-1)
#define CLOUDSC2_COMM_ENODEV   (-2)
#define CLOUDSC2_COMM_ETIMEOUT (-3)
#define CLOUDSC2_COMM_UNIQUE_ID_BYTES 128

enum { CLOUDSC2_COMM_SUM = 0, CLOUDSC2_COMM_MIN = 1, CLOUDSC2_COMM_MAX = 2 };

int cloudsc2_comm_init(void);                                   /* CLOUDSC_MPI_INIT */
int cloudsc2_comm_unique_id(char id[CLOUDSC2_COMM_UNIQUE_ID_BYTES]);
int cloudsc2_comm_init_rank(const char id[CLOUDSC2_COMM_UNIQUE_ID_BYTES], int rank, int world, int local_rank);
int cloudsc2_comm_finalize(void);                               /* CLOUDSC_MPI_END */
int cloudsc2_comm_rank(void);                                   /* IRANK   */
int cloudsc2_comm_size(void);                                   /* NUMPROC */
const char* cloudsc2_comm_transport(void);                      /* "single", "rccl" or "shm" */
const char* cloudsc2_comm_last_error(void);
/* element-wise reduction of buf[n] over all ranks, result on EVERY rank (a superset of the reference's reduce-to-root) */
int cloudsc2_comm_allreduce_f64(double* buf, int n, int op);
int cloudsc2_comm_allreduce_i32(int* buf, int n, int op);
/* CLOUDSC_MPI_GATHER_INT: recv[count * size] on every rank, rank r's block at recv + r*count */
int cloudsc2_comm_allgather_i32(const int* send, int count, int* recv);
int cloudsc2_comm_barrier(void);

#ifdef __cplusplus
}
#endif
#endif
