// cloudsc2_comm.cpp -- libcloudsc2_comm.so: the dwarf's few collectives over RCCL (include/cloudsc2_comm.h).
//
// One process per GPU.  The payloads are tiny (1..64 numbers: verdict norms, validation statistics, the timer table), so a
// call is latency-bound whatever the algorithm: the host buffer is staged through a 4 KB device buffer, one ncclAllReduce /
// ncclAllGather on the communicator's own stream, copied back.  Ring or tree, bucket sizes and the per-link bound of xGMI do
// not matter at this size and are left to RCCL.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <string>

#include "../../include/cloudsc2_comm.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

enum Transport { T_NONE, T_SINGLE, T_RCCL, T_SHM };
Transport g_transport = T_NONE;
int g_rank = 0, g_world = 1, g_local = 0;
ncclComm_t g_comm = nullptr;
hipStream_t g_stream = nullptr;
void* g_dev = nullptr;  // staging buffer
constexpr size_t kStageBytes = 4096;
std::string g_id_file;

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail((int)e_, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)
#define NCCL_TRY(expr)                                                                                 \
  do {                                                                                                 \
    ncclResult_t e_ = (expr);                                                                          \
    if (e_ != ncclSuccess) return fail((int)e_, std::string(#expr) + ": " + ncclGetErrorString(e_));   \
  } while (0)

bool env_int(const char* name, int* v) {
  const char* e = getenv(name);
  if (!e || !*e) return false;
  *v = atoi(e);
  return true;
}

void ranks_from_env(int* rank, int* world, int* local) {
  *rank = 0; *world = 1; *local = 0;
  if (env_int("WORLD_SIZE", world)) { env_int("RANK", rank); if (!env_int("LOCAL_RANK", local)) *local = *rank; return; }
  if (env_int("OMPI_COMM_WORLD_SIZE", world)) {
    env_int("OMPI_COMM_WORLD_RANK", rank);
    if (!env_int("OMPI_COMM_WORLD_LOCAL_RANK", local)) *local = *rank;
    return;
  }
  if (env_int("SLURM_NTASKS", world)) { env_int("SLURM_PROCID", rank); if (!env_int("SLURM_LOCALID", local)) *local = *rank; return; }
}

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

std::string rendezvous_name(const char* suffix) {
  const char* dir = getenv("CLOUDSC2_COMM_DIR");
  const char* port = getenv("MASTER_PORT");
  const char* token = getenv("CLOUDSC2_COMM_TOKEN");  // a launcher may name the job itself
  char buf[512];
  if (token && *token) snprintf(buf, sizeof buf, "%s/cloudsc2_comm_%s%s", (dir && *dir) ? dir : "/tmp", token, suffix);
  else snprintf(buf, sizeof buf, "%s/cloudsc2_comm_%s_%ld%s", (dir && *dir) ? dir : "/tmp", (port && *port) ? port : "0", (long)getppid(), suffix);
  return buf;
}

// What tells this job's rendezvous file / segment from a leftover of a crashed one with the same name (MASTER_PORT "0" under
// mpirun / srun, a reused launcher pid): a nonce every rank of ONE launch derives alike -- the launcher's own name for the job if it
// gave one (CLOUDSC2_COMM_TOKEN, TORCHELASTIC_RUN_ID), else MASTER_PORT + the launcher's pid + the launcher's START TIME
// (/proc/<ppid>/stat field 22), which a later process with the same pid does not share.  Readers wait until they see it.
// Single node only: pid, /proc, /tmp and POSIX shared memory are per node (one node of 8 GPUs is the dwarf's scope);
// across nodes hand the id in through cloudsc2_comm_init_rank.
unsigned long long job_nonce() {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](const char* s) { for (; s && *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ull; } h ^= 0xff; h *= 1099511628211ull; };
  const char* token = getenv("CLOUDSC2_COMM_TOKEN");
  const char* run = getenv("TORCHELASTIC_RUN_ID");
  if (token && *token) { mix("token"); mix(token); }
  else if (run && *run && strcmp(run, "none") != 0) { mix("run"); mix(run); mix(getenv("MASTER_PORT")); }
  else {
    char buf[1024], path[64];
    snprintf(buf, sizeof buf, "%ld", (long)getppid());
    mix("ppid"); mix(buf); mix(getenv("MASTER_PORT"));
    snprintf(path, sizeof path, "/proc/%ld/stat", (long)getppid());
    if (FILE* f = fopen(path, "r")) {
      const size_t n = fread(buf, 1, sizeof buf - 1, f);
      fclose(f);
      buf[n] = 0;
      const char* q = strrchr(buf, ')');  // the command name may contain blanks and parentheses
      int field = 2;
      for (q = q ? q + 1 : buf; *q && field < 22; ++q) if (*q == ' ' && q[1] != ' ') ++field;
      char start[32] = {0};
      for (int i = 0; i < 31 && q[i] && q[i] != ' '; ++i) start[i] = q[i];
      mix(start);
    }
  }
  return h ? h : 1;  // 0 means "not published yet" in a fresh (zero-filled) segment
}
constexpr unsigned long long kIdMagic = 0x32435344554f4c43ull;  // "CLOUDSC2"
struct IdFile { unsigned long long magic, nonce; ncclUniqueId id; };

// ---- shared-memory transport (rehearsal: more ranks than GPUs) -------------------------------------------------------
struct ShmSeg {
  std::atomic<unsigned long long> nonce;  // job_nonce(), stored by rank 0 once the segment is ready
  std::atomic<int> arrived;   // barrier counter
  std::atomic<int> sense;     // barrier generation
  std::atomic<int> attached;
  int world;
  double slots[64][64];       // [rank][element]
};
ShmSeg* g_shm = nullptr;
std::string g_shm_name;
int g_shm_sense = 0;

int shm_barrier() {
  const int my = g_shm_sense ^= 1;
  if (g_shm->arrived.fetch_add(1) + 1 == g_world) {
    g_shm->arrived.store(0);
    g_shm->sense.store(my);
  } else {
    const double t0 = now_s();
    while (g_shm->sense.load() != my) {
      if (now_s() - t0 > 120.0) return fail(CLOUDSC2_COMM_ETIMEOUT, "shm barrier: a rank did not arrive within 120 s");
      usleep(50);
    }
  }
  return 0;
}

int shm_open_segment() {
  if (g_world > 64) return fail(CLOUDSC2_COMM_EINVAL, "shm transport: at most 64 ranks");
  std::string n = rendezvous_name(".shm");
  for (auto& c : n) if (c == '/') c = '_';
  g_shm_name = "/" + n;
  const unsigned long long nonce = job_nonce();
  if (g_rank == 0) {
    shm_unlink(g_shm_name.c_str());  // a leftover of a crashed job: ranks that already opened it see the wrong nonce and retry
    const int fd = shm_open(g_shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(ShmSeg)) != 0) { if (fd >= 0) close(fd); return fail(CLOUDSC2_COMM_EINVAL, "shm_open/ftruncate failed for " + g_shm_name); }
    void* p = mmap(nullptr, sizeof(ShmSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(CLOUDSC2_COMM_EINVAL, "mmap of the shm segment failed");
    g_shm = (ShmSeg*)p;
    g_shm->world = g_world;      // (a fresh segment is zero-filled: counters start at 0)
    g_shm->nonce.store(nonce);   // ready
  } else {
    const double t0 = now_s();
    for (;;) {  // until THIS job's segment is there: same name, this launch's nonce, owned by this user
      struct stat sb;
      const int fd = shm_open(g_shm_name.c_str(), O_RDWR, 0600);
      if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(ShmSeg) && sb.st_uid == geteuid()) {
        void* p = mmap(nullptr, sizeof(ShmSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p != MAP_FAILED) {
          if (((ShmSeg*)p)->nonce.load() == nonce) { g_shm = (ShmSeg*)p; break; }
          munmap(p, sizeof(ShmSeg));  // a stale segment (or rank 0 has not finished): look the name up again
        }
      } else if (fd >= 0) {
        close(fd);
      }
      if (now_s() - t0 > 120.0) return fail(CLOUDSC2_COMM_ETIMEOUT, "rank 0 did not create " + g_shm_name + " (for this launch) within 120 s");
      usleep(1000);
    }
  }
  g_shm->attached.fetch_add(1);
  const double t0 = now_s();
  while (g_shm->attached.load() < g_world) {
    if (now_s() - t0 > 120.0) return fail(CLOUDSC2_COMM_ETIMEOUT, "not all ranks attached to the shm segment within 120 s");
    usleep(200);
  }
  return 0;
}

template <class T>
T combine(T a, T b, int op) { return op == CLOUDSC2_COMM_SUM ? a + b : op == CLOUDSC2_COMM_MIN ? (b < a ? b : a) : (b > a ? b : a); }

int shm_allreduce(double* buf, int n, int op) {
  if (n > 64) return fail(CLOUDSC2_COMM_EINVAL, "shm transport: at most 64 elements per call");
  int rc;
  for (int i = 0; i < n; ++i) g_shm->slots[g_rank][i] = buf[i];
  if ((rc = shm_barrier())) return rc;
  for (int i = 0; i < n; ++i) {
    double v = g_shm->slots[0][i];
    for (int r = 1; r < g_world; ++r) v = combine(v, g_shm->slots[r][i], op);  // rank order: every rank gets the same bits
    buf[i] = v;
  }
  return shm_barrier();  // nobody overwrites a slot before everyone has read it
}

// ---- RCCL transport ------------------------------------------------------------------------------------------------------
int rccl_setup(const ncclUniqueId& id) {
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(CLOUDSC2_COMM_ENODEV, "no HIP device");
  if (g_local >= ndev)
    return fail(CLOUDSC2_COMM_ENODEV, "LOCAL_RANK " + std::to_string(g_local) + " but the node shows " + std::to_string(ndev) +
                                      " GPU(s): RCCL needs one GPU per rank (CLOUDSC2_COMM=shm rehearses on fewer)");
  HIP_TRY(hipSetDevice(g_local));
  HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  HIP_TRY(hipMalloc(&g_dev, kStageBytes));
  NCCL_TRY(ncclCommInitRank(&g_comm, g_world, id, g_rank));
  g_transport = T_RCCL;
  return 0;
}

int publish_or_fetch_id(ncclUniqueId* id) {
  g_id_file = rendezvous_name(".id");
  IdFile rec;
  rec.magic = kIdMagic;
  rec.nonce = job_nonce();
  if (g_rank == 0) {
    NCCL_TRY(ncclGetUniqueId(id));
    rec.id = *id;
    // written under a private name (created exclusively, never through a symlink, readable by this user only), then renamed over
    // whatever a crashed job may have left under the public one: readers see the old record (wrong nonce: ignored) or the new one
    const std::string tmp = g_id_file + ".tmp." + std::to_string((long)getpid());
    unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_NOFOLLOW | O_WRONLY, 0600);
    if (fd < 0 || write(fd, &rec, sizeof rec) != (ssize_t)sizeof rec) { if (fd >= 0) close(fd); unlink(tmp.c_str()); return fail(CLOUDSC2_COMM_EINVAL, "cannot write " + tmp); }
    close(fd);
    if (rename(tmp.c_str(), g_id_file.c_str()) != 0) { unlink(tmp.c_str()); return fail(CLOUDSC2_COMM_EINVAL, "cannot publish " + g_id_file); }
    return 0;
  }
  const double t0 = now_s();
  for (;;) {
    const int fd = open(g_id_file.c_str(), O_RDONLY | O_NOFOLLOW);
    if (fd >= 0) {
      struct stat sb;
      IdFile got;
      const bool ok = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_uid == geteuid() && read(fd, &got, sizeof got) == (ssize_t)sizeof got &&
                      got.magic == kIdMagic && got.nonce == rec.nonce;
      close(fd);
      if (ok) { *id = got.id; return 0; }
    }
    if (now_s() - t0 > 120.0) return fail(CLOUDSC2_COMM_ETIMEOUT, "rank 0 did not publish " + g_id_file + " (for this launch) within 120 s");
    usleep(2000);
  }
}

template <class T>
int rccl_allreduce(T* buf, int n, int op, ncclDataType_t dt) {
  if ((size_t)n * sizeof(T) > kStageBytes) return fail(CLOUDSC2_COMM_EINVAL, "allreduce: at most 4096 bytes per call");
  const ncclRedOp_t rop = op == CLOUDSC2_COMM_SUM ? ncclSum : op == CLOUDSC2_COMM_MIN ? ncclMin : ncclMax;
  HIP_TRY(hipMemcpyAsync(g_dev, buf, n * sizeof(T), hipMemcpyHostToDevice, g_stream));
  NCCL_TRY(ncclAllReduce(g_dev, g_dev, (size_t)n, dt, rop, g_comm, g_stream));
  HIP_TRY(hipMemcpyAsync(buf, g_dev, n * sizeof(T), hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return 0;
}

}  // namespace

extern "C" {

const char* cloudsc2_comm_last_error(void) { return g_err.c_str(); }
int cloudsc2_comm_rank(void) { return g_rank; }
int cloudsc2_comm_size(void) { return g_world; }
const char* cloudsc2_comm_transport(void) {
  return g_transport == T_RCCL ? "rccl" : g_transport == T_SHM ? "shm" : g_transport == T_SINGLE ? "single" : "none";
}

int cloudsc2_comm_unique_id(char id[CLOUDSC2_COMM_UNIQUE_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == CLOUDSC2_COMM_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) return fail(CLOUDSC2_COMM_EINVAL, "NULL id");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return 0;
}

int cloudsc2_comm_init_rank(const char id[CLOUDSC2_COMM_UNIQUE_ID_BYTES], int rank, int world, int local_rank) {
  if (g_transport != T_NONE) return fail(CLOUDSC2_COMM_EINVAL, "communicator already initialised");
  if (world < 1 || rank < 0 || rank >= world || local_rank < 0) return fail(CLOUDSC2_COMM_EINVAL, "bad rank / world size");
  g_rank = rank; g_world = world; g_local = local_rank;
  if (world == 1) { g_transport = T_SINGLE; return 0; }
  if (!id) return fail(CLOUDSC2_COMM_EINVAL, "NULL id");
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  return rccl_setup(u);
}

int cloudsc2_comm_init(void) {
  if (g_transport != T_NONE) return 0;  // idempotent, like a second USE of an initialised MPI
  ranks_from_env(&g_rank, &g_world, &g_local);
  if (g_world < 1 || g_rank < 0 || g_rank >= g_world) return fail(CLOUDSC2_COMM_EINVAL, "inconsistent RANK / WORLD_SIZE in the environment");
  const char* t = getenv("CLOUDSC2_COMM");
  // one rank needs no communicator; CLOUDSC2_COMM=rccl builds one all the same (a 1-rank RCCL communicator: the GPU tests' way
  // to run the RCCL code path on a one-GPU box)
  if (g_world == 1 && !(t && !strcmp(t, "rccl"))) { g_transport = T_SINGLE; return 0; }
  if (t && !strcmp(t, "shm")) {
    int rc = shm_open_segment();
    if (rc) return rc;
    g_transport = T_SHM;
    return 0;
  }
  ncclUniqueId id;
  int rc = publish_or_fetch_id(&id);
  if (rc) return rc;
  return rccl_setup(id);
}

int cloudsc2_comm_finalize(void) {
  if (g_transport == T_RCCL) {
    if (g_comm) (void)ncclCommDestroy(g_comm);
    if (g_dev) (void)hipFree(g_dev);
    if (g_stream) (void)hipStreamDestroy(g_stream);
    g_comm = nullptr; g_dev = nullptr; g_stream = nullptr;
    if (g_rank == 0 && !g_id_file.empty()) unlink(g_id_file.c_str());
  } else if (g_transport == T_SHM && g_shm) {
    (void)shm_barrier();
    munmap(g_shm, sizeof(ShmSeg));
    g_shm = nullptr;
    if (g_rank == 0) shm_unlink(g_shm_name.c_str());
  }
  g_transport = T_NONE; g_rank = 0; g_world = 1; g_local = 0;
  return 0;
}

int cloudsc2_comm_allreduce_f64(double* buf, int n, int op) {
  if (!buf || n < 0 || op < 0 || op > 2) return fail(CLOUDSC2_COMM_EINVAL, "allreduce: bad argument");
  if (g_transport == T_NONE) return fail(CLOUDSC2_COMM_EINVAL, "cloudsc2_comm_init has not been called");
  if (n == 0 || g_transport == T_SINGLE) return 0;
  if (g_transport == T_SHM) return shm_allreduce(buf, n, op);
  return rccl_allreduce(buf, n, op, ncclFloat64);
}

int cloudsc2_comm_allreduce_i32(int* buf, int n, int op) {
  if (!buf || n < 0 || op < 0 || op > 2) return fail(CLOUDSC2_COMM_EINVAL, "allreduce: bad argument");
  if (g_transport == T_NONE) return fail(CLOUDSC2_COMM_EINVAL, "cloudsc2_comm_init has not been called");
  if (n == 0 || g_transport == T_SINGLE) return 0;
  if (g_transport == T_SHM) {
    if (n > 64) return fail(CLOUDSC2_COMM_EINVAL, "shm transport: at most 64 elements per call");
    double tmp[64];
    for (int i = 0; i < n; ++i) tmp[i] = buf[i];  // exact: |int32| < 2^53, sums of <= 64 of them too
    int rc = shm_allreduce(tmp, n, op);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) buf[i] = (int)tmp[i];
    return 0;
  }
  return rccl_allreduce(buf, n, op, ncclInt32);
}

int cloudsc2_comm_allgather_i32(const int* send, int count, int* recv) {
  if (!send || !recv || count < 0) return fail(CLOUDSC2_COMM_EINVAL, "allgather: bad argument");
  if (g_transport == T_NONE) return fail(CLOUDSC2_COMM_EINVAL, "cloudsc2_comm_init has not been called");
  if (g_transport == T_SINGLE) { memcpy(recv, send, (size_t)count * sizeof(int)); return 0; }
  if (g_transport == T_SHM) {
    if (count > 64) return fail(CLOUDSC2_COMM_EINVAL, "shm transport: at most 64 elements per rank");
    int rc;
    for (int i = 0; i < count; ++i) g_shm->slots[g_rank][i] = send[i];
    if ((rc = shm_barrier())) return rc;
    for (int r = 0; r < g_world; ++r)
      for (int i = 0; i < count; ++i) recv[(size_t)r * count + i] = (int)g_shm->slots[r][i];
    return shm_barrier();
  }
  if ((size_t)count * sizeof(int) * g_world > kStageBytes) return fail(CLOUDSC2_COMM_EINVAL, "allgather: at most 4096 bytes in total");
  HIP_TRY(hipMemcpyAsync((char*)g_dev + (size_t)g_rank * count * sizeof(int), send, count * sizeof(int), hipMemcpyHostToDevice, g_stream));
  NCCL_TRY(ncclAllGather((char*)g_dev + (size_t)g_rank * count * sizeof(int), g_dev, (size_t)count, ncclInt32, g_comm, g_stream));
  HIP_TRY(hipMemcpyAsync(recv, g_dev, (size_t)count * sizeof(int) * g_world, hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return 0;
}

int cloudsc2_comm_barrier(void) {
  if (g_transport == T_NONE) return fail(CLOUDSC2_COMM_EINVAL, "cloudsc2_comm_init has not been called");
  if (g_transport == T_SINGLE) return 0;
  if (g_transport == T_SHM) return shm_barrier();
  int one = 1;
  return rccl_allreduce(&one, 1, CLOUDSC2_COMM_SUM, ncclInt32);
}

}  // extern "C"
