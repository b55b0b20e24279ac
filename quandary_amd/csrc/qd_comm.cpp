// RCCL communicator behind the C ABI (qd_comm_*): the reference's comm_init split and its MPI_Allreduce calls
// (src/main.cpp:133-177, src/optimproblem.cpp:292-298, :454-460, :527) as ncclAllReduce over xGMI, one process
// per GPU, on the handle's HIP stream.  Bootstrap needs no MPI: the ncclUniqueId of rank 0 travels as 128 plain
// bytes through whatever the caller has (a file on a shared file system here, torch.distributed/gloo in bench.py).
#include <dlfcn.h>

#include <chrono>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <cstdio>
#include <cstring>
#include <thread>

#include "qd_handle.h"

using namespace qd;

static int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

// RCCL is loaded on first use of a communicator, not linked: the single-GPU paths of the library load and run on a ROCm installation
// without it, and a non-default prefix is found through ROCM_PATH.  Search order: $ROCM_PATH/lib, the prefix of the build, the loader's
// own path.  Only the seven entry points below are used (types and constants from <rccl/rccl.h> at build time).
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

#ifndef QD_ROCM_PATH
#define QD_ROCM_PATH "/opt/rocm"
#endif

Rccl& rccl_slot() {
  static Rccl r;
  return r;
}

// nullptr + an error text in rccl_slot().error when the library or one of its symbols is missing
Rccl* rccl() {
  static std::once_flag once;
  Rccl& r = rccl_slot();
  std::call_once(once, [&r] {
    std::vector<std::string> names;
    if (const char* e = getenv("ROCM_PATH")) names.push_back(std::string(e) + "/lib/librccl.so.1");
    names.push_back(std::string(QD_ROCM_PATH) + "/lib/librccl.so.1");
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string& n : names) {
      r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
      if (const char* e = dlerror()) r.error += std::string(r.error.empty() ? "" : "; ") + e;
    }
    if (!r.lib) return;
    r.error.clear();
    auto sym = [&r](const char* name) -> void* {
      void* p = dlsym(r.lib, name);
      if (!p) r.error += std::string(r.error.empty() ? "missing symbol " : ", ") + name;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return (r.lib && r.error.empty()) ? &r : nullptr;
}
}  // namespace

#define QD_RCCL_OR_FAIL(R)                                                                                   \
  Rccl* R = rccl();                                                                                          \
  if (!R) return fail(QD_ERR_DEVICE, "RCCL is not available (librccl.so.1 under $ROCM_PATH/lib, " QD_ROCM_PATH "/lib or the loader path): " + rccl_slot().error)

#define QD_NCCL(R, expr)                                                                     \
  do {                                                                                      \
    ncclResult_t _r = (expr);                                                               \
    if (_r != ncclSuccess) return fail(QD_ERR_DEVICE, std::string(#expr) + ": " + (R)->GetErrorString(_r)); \
  } while (0)

static_assert(QD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "QD_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");

extern "C" int qd_comm_unique_id(unsigned char* id) {
  if (!id) return fail(QD_ERR_INVALID, "qd_comm_unique_id: null argument");
  QD_RCCL_OR_FAIL(R);
  ncclUniqueId u;
  QD_NCCL(R, R->GetUniqueId(&u));
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return QD_OK;
}

extern "C" int qd_comm_create(const unsigned char* id, int rank, int nranks, int device_ordinal, qd_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(QD_ERR_INVALID, "qd_comm_create: bad argument");
  *out = nullptr;
  QD_RCCL_OR_FAIL(R);
  QD_HIP(qd::use_device(device_ordinal));
  qd_comm* c = new qd_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = device_ordinal;
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = R->CommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(QD_ERR_DEVICE, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    R->CommDestroy(c->comm);
    delete c;
    return fail(QD_ERR_DEVICE, "qd_comm_create: stream creation failed");
  }
  *out = c;
  return QD_OK;
}

// Rank 0 writes the id to `path` (atomically: temporary name + rename), the others wait for the file.  The file carries a header
// {magic, job nonce}: a leftover of a crashed run or the file of another job must not be mistaken for this job's id (ranks > 0 would
// then block in ncclCommInitRank forever - the timeout only covers the wait for the file).  The nonce is a hash of the environment
// variable QD_JOB_ID when the launcher sets one (the same for all ranks of a job); without it, files last modified more than two
// minutes before this rank entered the call are treated as leftovers.  Rank 0 removes whatever is there before it publishes.
static unsigned long long job_nonce() {
  const char* j = getenv("QD_JOB_ID");
  if (!j) return 0ull;
  unsigned long long h = 1469598103934665603ull;  // FNV-1a
  for (const char* c = j; *c; c++) h = (h ^ (unsigned char)*c) * 1099511628211ull;
  return h ? h : 1ull;
}

extern "C" int qd_comm_create_from_file(const char* path, int rank, int nranks, int device_ordinal, double timeout_s, qd_comm** out) {
  if (!path || !out) return fail(QD_ERR_INVALID, "qd_comm_create_from_file: null argument");
  static const char magic[8] = {'Q', 'D', 'C', 'O', 'M', 'M', '0', '2'};
  const unsigned long long nonce = job_nonce();
  const time_t entered = time(nullptr);
  unsigned char id[QD_COMM_ID_BYTES];
  if (rank == 0) {
    (void)remove(path);  // a leftover of an earlier run
    int r = qd_comm_unique_id(id);
    if (r) return r;
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(magic, 1, sizeof magic, f) != sizeof magic || fwrite(&nonce, 1, sizeof nonce, f) != sizeof nonce ||
        fwrite(id, 1, sizeof id, f) != sizeof id) {
      if (f) fclose(f);
      return fail(QD_ERR_INVALID, "qd_comm_create_from_file: cannot write the id file");
    }
    fclose(f);
    if (rename(tmp.c_str(), path) != 0) return fail(QD_ERR_INVALID, "qd_comm_create_from_file: cannot publish the id file");
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      FILE* f = fopen(path, "rb");
      if (f) {
        char m[8];
        unsigned long long n2 = 0;
        const bool ok = fread(m, 1, sizeof m, f) == sizeof m && fread(&n2, 1, sizeof n2, f) == sizeof n2 && fread(id, 1, sizeof id, f) == sizeof id;
        struct stat sb;
        const bool have_stat = fstat(fileno(f), &sb) == 0;
        fclose(f);
        const bool fresh = nonce ? n2 == nonce : (have_stat && difftime(entered, sb.st_mtime) <= 120.0);
        if (ok && std::memcmp(m, magic, sizeof m) == 0 && fresh) break;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        return fail(QD_ERR_STATE, "qd_comm_create_from_file: timed out waiting for rank 0's id file (a file that is there but stale - other "
                                  "QD_JOB_ID, or older than two minutes - does not count)");
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  return qd_comm_create(id, rank, nranks, device_ordinal, out);
}

extern "C" void qd_comm_destroy(qd_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  struct Quiet { ~Quiet() { (void)hipGetLastError(); } } quiet;  // teardown never leaves a sticky error behind
  c->dbuf.release();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm)
    if (Rccl* R = rccl()) R->CommDestroy(c->comm);
  delete c;
}

extern "C" int qd_comm_size(const qd_comm* c) {
  if (!c) return QD_ERR_INVALID;
  int n = 0;
  Rccl* R = rccl();
  if (!R || R->CommCount(c->comm, &n) != ncclSuccess) return QD_ERR_DEVICE;
  return n;
}
extern "C" int qd_comm_rank(const qd_comm* c) { return c ? c->rank : QD_ERR_INVALID; }

int qd_comm_allreduce_dev(qd_comm* c, double* dbuf, size_t n, int op, hipStream_t st) {
  QD_RCCL_OR_FAIL(R);
  QD_NCCL(R, R->AllReduce(dbuf, dbuf, n, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, st));
  return QD_OK;
}

// host convenience (timings, test hooks): staged through a device buffer, blocking
extern "C" int qd_comm_allreduce(qd_comm* c, double* buf, int n, int op) {
  if (!c || !buf || n < 0 || (op != 0 && op != 1)) return fail(QD_ERR_INVALID, "qd_comm_allreduce: bad argument");
  if (n == 0) return QD_OK;
  QD_HIP(qd::use_device(c->device));
  int r;
  if ((r = c->dbuf.ensure(n))) return r;
  QD_HIP(hipMemcpyAsync(c->dbuf.p, buf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  if ((r = qd_comm_allreduce_dev(c, c->dbuf.p, n, op, c->stream))) return r;
  QD_HIP(hipMemcpyAsync(buf, c->dbuf.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  QD_HIP(hipStreamSynchronize(c->stream));
  return QD_OK;
}

extern "C" int qd_comm_barrier(qd_comm* c) {
  double z = 0.0;
  return qd_comm_allreduce(c, &z, 1, 0);
}
