// RCCL communicator behind the C ABI (qd_comm_*): the reference's comm_init split and its MPI_Allreduce calls
// (src/main.cpp:133-177, src/optimproblem.cpp:292-298, :454-460, :527) as ncclAllReduce over xGMI, one process
// per GPU, on the handle's HIP stream.  Bootstrap needs no MPI: the ncclUniqueId of rank 0 travels as 128 plain
// bytes through whatever the caller has (a file on a shared file system here, torch.distributed/gloo in bench.py).
//
// Second backend, HOST: the ranks of one node reduce through a POSIX shared-memory segment - the same call sites
// (qd_comm_allreduce_dev on the handle's stream, qd_comm_allreduce), the same [7 | ndesign] buffers, a fixed summation order that is
// the same on every rank.  It exists for ranks that SHARE a GPU (mpirun -np 4 on a one-GPU box: RCCL refuses two ranks on one device) and
// for nodes without RCCL; the reduction buffer then makes one round trip HBM -> pinned host -> HBM per collective.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <random>

#include <chrono>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <signal.h>
#include <cerrno>
#include <functional>
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


// ---------------------------------------------------------------------------------------------------------------
// HOST backend: all-reduce through a shared-memory segment
// ---------------------------------------------------------------------------------------------------------------
namespace qd {
constexpr int kHostMaxRanks = 64;
constexpr size_t kHostSlot = 1u << 16;  // doubles per rank and round (512 KiB); longer buffers go in rounds
struct alignas(64) HostFlag {
  std::atomic<unsigned long long> v;
};
struct HostSeg {
  std::atomic<unsigned long long> magic;  // published last by rank 0
  unsigned long long token;               // random, rank 0
  int nranks;
  int pad_;
  HostFlag hello[kHostMaxRanks], ack[kHostMaxRanks];    // bootstrap handshake
  HostFlag arrive[kHostMaxRanks], done[kHostMaxRanks];  // per-round sequence numbers
  HostFlag pid[kHostMaxRanks];                          // process ids (liveness of a peer a collective waits for)
  // followed by nranks slots of kHostSlot doubles
  double* slot(int r) { return reinterpret_cast<double*>(reinterpret_cast<char*>(this) + sizeof(HostSeg)) + (size_t)r * kHostSlot; }
};
struct HostRing {
  HostSeg* seg = nullptr;
  size_t bytes = 0;
  std::string name;
  unsigned long long seq = 0;
  int rank = 0, nranks = 1;
  double timeout_s = 600.0;
};
}  // namespace qd

static constexpr unsigned long long kHostMagic = 0x51444853484d3032ull;  // "QDHSHM02"

static size_t host_seg_bytes(int nranks) { return sizeof(qd::HostSeg) + sizeof(double) * qd::kHostSlot * (size_t)nranks; }

static std::string host_shm_name(const char* name) {
  // shm names: one leading slash, no others
  std::string s = "/qdcomm_";
  for (const char* c = name; *c; c++) s += (isalnum((unsigned char)*c) || *c == '-' || *c == '.') ? *c : '_';
  if (s.size() > 200) {
    unsigned long long h = 1469598103934665603ull;
    for (const char* c = name; *c; c++) h = (h ^ (unsigned char)*c) * 1099511628211ull;
    s = "/qdcomm_" + std::to_string(h);
  }
  return s;
}

static unsigned long long random_token() {
  std::random_device rd;
  unsigned long long t = ((unsigned long long)rd() << 32) ^ rd() ^ ((unsigned long long)getpid() << 17) ^
                         (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  return t ? t : 1ull;
}

// wait until pred() holds; false on timeout
template <class P>
static bool spin_until(P pred, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (pred()) return true;
    if (spins < 2000) continue;
    if ((spins & 63) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    if (spins < 20000) sched_yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// Collectives have no time limit of their own: the ranks of the host backend share a GPU, their sweeps serialise, and the arrival skew of a
// large or chunked gradient evaluation is legitimately long (RCCL has no such limit either).  A wait ends when pred() holds, when a peer
// PROCESS is gone (checked once a second: kill(pid, 0)), or after QD_COMM_COLLECTIVE_TIMEOUT_S (default one day).
// (a process that has exited but has not been reaped by its launcher yet still answers kill(pid, 0): state Z in /proc/<pid>/stat)
// The check assumes that the ranks share ONE PID namespace (ranks of one node started by one launcher: the only way they can share a
// POSIX segment and a GPU here).  Ranks in separate containers with a common /dev/shm see each other's pids as foreign numbers: set
// QD_COMM_NO_LIVENESS=1 there (collectives then end on pred() or on the time limit only).
static bool process_alive(long pid) {
  static const bool off = [] { const char* e = getenv("QD_COMM_NO_LIVENESS"); return e && atoi(e) != 0; }();
  if (off) return true;
  if (kill((pid_t)pid, 0) != 0 && errno == ESRCH) return false;
  char path[64], buf[512];
  snprintf(path, sizeof path, "/proc/%ld/stat", pid);
  FILE* f = fopen(path, "r");
  if (!f) return true;  // (no procfs: kill's answer stands)
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  const char* rp = strrchr(buf, ')');  // the command name may contain spaces and parentheses
  return !(rp && rp[1] == ' ' && (rp[2] == 'Z' || rp[2] == 'X'));
}
static int host_wait(qd::HostRing* g, const std::function<bool()>& pred) {
  static const double limit = [] {
    const char* e = getenv("QD_COMM_COLLECTIVE_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 86400.0;
  }();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (spin_until(pred, 1.0)) return 0;
    for (int r = 0; r < g->nranks; r++) {
      const long p = (long)g->seg->pid[r].v.load(std::memory_order_acquire);
      // (a peer that has arrived - or finished its last collective - may exit between the last poll of pred() and this check: look again
      //  before calling the collective failed)
      if (r != g->rank && p > 0 && !process_alive(p)) return pred() ? 0 : 1 + r;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return -1;
  }
}

static qd::HostSeg* host_map(int fd, size_t bytes) {
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  return p == MAP_FAILED ? nullptr : static_cast<qd::HostSeg*>(p);
}

// Bootstrap.  Rank 0 removes whatever carries the name, creates the segment exclusively and acknowledges every rank that says hello IN
// THAT SEGMENT; a rank > 0 opens the name, says hello with a token of its own and proceeds only when it reads its token back.  A leftover
// segment of a crashed job (or the one rank 0 is about to replace) never answers: the rank closes it after half a second and opens the
// name again.  No clocks, no modification times.
extern "C" int qd_comm_create_host(const char* name, int rank, int nranks, int device_ordinal, double timeout_s, qd_comm** out) {
  if (!name || !out || nranks < 1 || nranks > qd::kHostMaxRanks || rank < 0 || rank >= nranks)
    return fail(QD_ERR_INVALID, "qd_comm_create_host: bad argument (at most " + std::to_string(qd::kHostMaxRanks) + " ranks)");
  *out = nullptr;
  static_assert(std::atomic<unsigned long long>::is_always_lock_free, "shared-memory flags must be lock-free");
  const std::string shm = host_shm_name(name);
  const size_t bytes = host_seg_bytes(nranks);
  qd::HostSeg* seg = nullptr;
  if (timeout_s <= 0.0) timeout_s = 600.0;
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (rank == 0) {
    (void)shm_unlink(shm.c_str());
    const int fd = shm_open(shm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail(QD_ERR_DEVICE, "qd_comm_create_host: shm_open(" + shm + "): " + strerror(errno));
    if (ftruncate(fd, (off_t)bytes) != 0) {
      close(fd);
      (void)shm_unlink(shm.c_str());
      return fail(QD_ERR_NOMEM, "qd_comm_create_host: cannot size the shared segment");
    }
    seg = host_map(fd, bytes);
    close(fd);
    if (!seg) {
      (void)shm_unlink(shm.c_str());
      return fail(QD_ERR_NOMEM, "qd_comm_create_host: mmap failed");
    }
    seg->token = random_token();
    seg->nranks = nranks;
    seg->magic.store(kHostMagic, std::memory_order_release);
    std::vector<char> acked(nranks, 0);
    acked[0] = 1;
    const bool ok = spin_until([&] {
      bool all = true;
      for (int r = 1; r < nranks; r++) {
        const unsigned long long h = seg->hello[r].v.load(std::memory_order_acquire);
        if (h) {
          seg->ack[r].v.store(h, std::memory_order_release);
          acked[r] = 1;
        }
        all = all && acked[r];
      }
      return all;
    }, timeout_s);
    if (!ok) {
      munmap(seg, bytes);
      (void)shm_unlink(shm.c_str());
      return fail(QD_ERR_STATE, "qd_comm_create_host: timed out waiting for the other ranks");
    }
  } else {
    const unsigned long long mine = random_token();
    for (;;) {
      if (elapsed() > timeout_s) return fail(QD_ERR_STATE, "qd_comm_create_host: timed out waiting for rank 0's segment " + shm);
      const int fd = shm_open(shm.c_str(), O_RDWR, 0600);
      if (fd < 0) {
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        continue;
      }
      struct stat sb;
      if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) {  // not (yet) sized for this job
        close(fd);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        continue;
      }
      seg = host_map(fd, bytes);
      close(fd);
      if (!seg) return fail(QD_ERR_NOMEM, "qd_comm_create_host: mmap failed");
      bool good = spin_until([&] { return seg->magic.load(std::memory_order_acquire) == kHostMagic; }, 0.5) && seg->nranks == nranks;
      if (good) {
        seg->hello[rank].v.store(mine, std::memory_order_release);
        good = spin_until([&] { return seg->ack[rank].v.load(std::memory_order_acquire) == mine; }, 0.5);
      }
      if (good) break;
      munmap(seg, bytes);  // a leftover, or rank 0 has not replaced it yet: look the name up again
      seg = nullptr;
    }
  }
  seg->pid[rank].v.store((unsigned long long)getpid(), std::memory_order_release);
  qd_comm* c = new qd_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = device_ordinal;
  c->backend = 1;
  c->host = new qd::HostRing();
  c->host->seg = seg;
  c->host->bytes = bytes;
  c->host->name = shm;
  c->host->rank = rank;
  c->host->nranks = nranks;
  c->host->timeout_s = timeout_s;
  if (qd::use_device(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    c->stream = nullptr;  // (a host-only caller: the in-stream reductions will fail loudly, the host ones work)
  }
  *out = c;
  return QD_OK;
}

// in-place all-reduce of a host buffer; every rank adds the slots up in rank order, so all ranks hold bit-identical results
static int host_allreduce(qd::HostRing* g, double* buf, size_t n, int op) {
  qd::HostSeg* s = g->seg;
  for (size_t off = 0; off < n; off += qd::kHostSlot) {
    const size_t m = std::min(qd::kHostSlot, n - off);
    const unsigned long long seq = ++g->seq;
    // my slot is free once every rank has finished reading the previous round
    auto why = [&](int w, const char* what) {
      return fail(QD_ERR_STATE, std::string("qd_comm (host): ") + what + (w > 0 ? ": the process of rank " + std::to_string(w - 1) + " is gone" : ": collective time limit (QD_COMM_COLLECTIVE_TIMEOUT_S)"));
    };
    int w = host_wait(g, [&] {
      for (int r = 0; r < g->nranks; r++)
        if (s->done[r].v.load(std::memory_order_acquire) < seq - 1) return false;
      return true;
    });
    if (w) return why(w, "waiting for the previous round to drain");
    std::memcpy(s->slot(g->rank), buf + off, sizeof(double) * m);
    s->arrive[g->rank].v.store(seq, std::memory_order_release);
    w = host_wait(g, [&] {
      for (int r = 0; r < g->nranks; r++)
        if (s->arrive[r].v.load(std::memory_order_acquire) < seq) return false;
      return true;
    });
    if (w) return why(w, "waiting for the other ranks in an all-reduce");
    const double* s0 = s->slot(0);
    if (op == 1) {
      for (size_t i = 0; i < m; i++) {
        double a = s0[i];
        for (int r = 1; r < g->nranks; r++) a = std::max(a, s->slot(r)[i]);
        buf[off + i] = a;
      }
    } else {
      for (size_t i = 0; i < m; i++) {
        double a = s0[i];
        for (int r = 1; r < g->nranks; r++) a += s->slot(r)[i];
        buf[off + i] = a;
      }
    }
    s->done[g->rank].v.store(seq, std::memory_order_release);
  }
  return QD_OK;
}

static void host_destroy(qd_comm* c) {
  if (!c->host) return;
  if (c->host->seg) munmap(c->host->seg, c->host->bytes);
  if (c->rank == 0) (void)shm_unlink(c->host->name.c_str());  // (mappings of the other ranks stay valid until they unmap)
  delete c->host;
  c->host = nullptr;
}

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

// MPI-free bootstrap through a path every rank can see.
//
// Backend: QD_COMM_BACKEND = rccl | host, default: host when there are more ranks than visible GPUs (the ranks then share devices, which
// RCCL refuses; they must be on one node), RCCL otherwise.  Host: the path only names the shared-memory segment (qd_comm_create_host).
//
// RCCL: rank 0 writes the ncclUniqueId to `path` (atomically: temporary name + rename); the file carries {magic, job nonce, token}.  A
// leftover of a crashed run or the file of another job must not be mistaken for this job's id - a rank that calls ncclCommInitRank with
// a dead id blocks forever - so the id is CONFIRMED before anyone uses it: every rank > 0 echoes the token it read into `path.ack<rank>`,
// rank 0 waits for all echoes of ITS token and then publishes `path.go` with the token; a rank proceeds only on a go that carries the token
// it holds and keeps re-reading the id file meanwhile (rank 0 removes leftovers of all three kinds before it publishes, so a rank that
// picked up a stale id sees the new one and echoes again).  No modification times, no clock comparison between hosts.  The job nonce (hash
// of QD_JOB_ID when the launcher sets one) additionally rejects files of other jobs outright.
static unsigned long long job_nonce() {
  const char* j = getenv("QD_JOB_ID");
  if (!j) return 0ull;
  unsigned long long h = 1469598103934665603ull;  // FNV-1a
  for (const char* c = j; *c; c++) h = (h ^ (unsigned char)*c) * 1099511628211ull;
  return h ? h : 1ull;
}

namespace {
struct IdFile {
  char magic[8];
  unsigned long long nonce, token;
  unsigned char id[QD_COMM_ID_BYTES];
};
const char kIdMagic[8] = {'Q', 'D', 'C', 'O', 'M', 'M', '0', '3'};

bool write_atomically(const std::string& path, const void* data, size_t n) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(data, 1, n, f) == n;
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
    (void)remove(tmp.c_str());
    return false;
  }
  return true;
}
bool read_whole(const std::string& path, void* data, size_t n) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const bool ok = fread(data, 1, n, f) == n;
  fclose(f);
  return ok;
}
}  // namespace

extern "C" int qd_comm_create_from_file(const char* path, int rank, int nranks, int device_ordinal, double timeout_s, qd_comm** out) {
  if (!path || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(QD_ERR_INVALID, "qd_comm_create_from_file: bad argument");
  if (timeout_s <= 0.0) timeout_s = 600.0;
  // backend
  bool host = false;
  if (const char* b = getenv("QD_COMM_BACKEND")) {
    if (std::strcmp(b, "host") == 0) host = true;
    else if (std::strcmp(b, "rccl") != 0 && std::strcmp(b, "auto") != 0)
      return fail(QD_ERR_INVALID, std::string("QD_COMM_BACKEND: unknown backend ") + b + " (rccl | host | auto)");
  }
  if (!host && !(getenv("QD_COMM_BACKEND") && std::strcmp(getenv("QD_COMM_BACKEND"), "rccl") == 0)) {
    // automatic: the shared-memory backend only when the ranks OF THIS NODE outnumber its GPUs (they then share devices, which RCCL
    // refuses).  QD_LOCAL_SIZE = ranks on this node (set by the launchers; a launch that spans nodes has local < nranks and keeps RCCL:
    // a POSIX segment does not reach the other nodes); without it the global count stands in.
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
      (void)hipGetLastError();
      ndev = 0;
    }
    // (callers that reach this function through the C or the Python API under a launcher of their own set no QD_LOCAL_SIZE: the usual
    //  launcher variables stand in, as launch_env() of the config-file driver reads them)
    int local = nranks;
    for (const char* key : {"QD_LOCAL_SIZE", "LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE", "PMI_LOCAL_SIZE"}) {
      const char* ls = getenv(key);
      if (ls && atoi(ls) > 0) {
        local = std::min(atoi(ls), nranks);
        break;
      }
    }
    host = local > ndev;
    if (host && local < nranks)
      return fail(QD_ERR_UNSUPPORTED, "qd_comm_create_from_file: more ranks per node than GPUs on a launch that spans nodes (the shared-memory backend is node-local)");
  }
  if (host) {
    // the segment is named after the path (all ranks of a job pass the same one) and the job id
    std::string name = path;
    if (const char* j = getenv("QD_JOB_ID")) name += std::string("_") + j;
    unsigned long long h = 1469598103934665603ull;
    for (char ch : name) h = (h ^ (unsigned char)ch) * 1099511628211ull;
    return qd_comm_create_host(("f" + std::to_string(h)).c_str(), rank, nranks, device_ordinal, timeout_s, out);
  }
  // Handshake: rank 0 publishes {token, id}; rank k echoes {token, n_k} with a nonce n_k of its own; rank 0 answers every echo with a
  // go file FOR THAT RANK carrying {token, n_k}.  A rank only enters ncclCommInitRank on a go file that returns its own nonce, which only
  // a live rank 0 that read this run's echo can have written - the leftovers of a run killed inside ncclCommInitRank (an id file and go
  // files with matching tokens) are never acted upon, whoever starts first.
  const std::string idp = path;
  auto gop = [&](int k) { return idp + ".go" + std::to_string(k); };
  auto ackp = [&](int k) { return idp + ".ack" + std::to_string(k); };
  const unsigned long long nonce = job_nonce();
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
  IdFile f;
  if (rank == 0) {
    // leftovers of an earlier run
    (void)remove(idp.c_str());
    (void)remove((idp + ".go").c_str());  // (the single go file of earlier builds)
    for (int r = 1; r < nranks; r++) {
      (void)remove(gop(r).c_str());
      (void)remove(ackp(r).c_str());
    }
    std::memcpy(f.magic, kIdMagic, sizeof f.magic);
    f.nonce = nonce;
    f.token = random_token();
    int r = qd_comm_unique_id(f.id);
    if (r) return r;
    if (!write_atomically(idp, &f, sizeof f)) return fail(QD_ERR_INVALID, "qd_comm_create_from_file: cannot publish the id file");
    std::vector<unsigned long long> peer(nranks, 0ull);
    for (int k = 1; k < nranks; k++) {
      for (;;) {
        unsigned long long t[2] = {0, 0};
        if (read_whole(ackp(k), t, sizeof t) && t[0] == f.token && t[1]) {
          peer[k] = t[1];
          break;
        }
        if (expired()) {
          (void)remove(idp.c_str());
          return fail(QD_ERR_STATE, "qd_comm_create_from_file: timed out waiting for rank " + std::to_string(k) + " to confirm the id");
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
    }
    for (int k = 1; k < nranks; k++) {
      const unsigned long long g[2] = {f.token, peer[k]};
      if (!write_atomically(gop(k), g, sizeof g)) return fail(QD_ERR_INVALID, "qd_comm_create_from_file: cannot publish a go file");
    }
  } else {
    const unsigned long long mine = random_token();
    unsigned long long echoed = 0;
    for (;;) {
      IdFile g;
      if (read_whole(idp, &g, sizeof g) && std::memcmp(g.magic, kIdMagic, sizeof g.magic) == 0 && (!nonce || g.nonce == nonce)) {
        if (g.token != echoed) {
          f = g;
          const unsigned long long e[2] = {g.token, mine};
          if (write_atomically(ackp(rank), e, sizeof e)) echoed = g.token;
        }
      }
      unsigned long long go[2] = {0, 0};
      if (echoed && read_whole(gop(rank), go, sizeof go) && go[0] == echoed && go[1] == mine) break;
      if (expired())
        return fail(QD_ERR_STATE, "qd_comm_create_from_file: timed out waiting for rank 0 (no id file of this job, or rank 0 never confirmed it)");
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
  }
  int r = qd_comm_create(f.id, rank, nranks, device_ordinal, out);
  // the bootstrap files have served: every rank removes its echo and its go file, rank 0 the id
  if (rank > 0) {
    (void)remove(ackp(rank).c_str());
    (void)remove(gop(rank).c_str());
  } else {
    (void)remove(idp.c_str());
  }
  return r;
}

extern "C" void qd_comm_destroy(qd_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  struct Quiet { ~Quiet() { (void)hipGetLastError(); } } quiet;  // teardown never leaves a sticky error behind
  c->dbuf.release();
  c->hbuf.release();
  host_destroy(c);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm)
    if (Rccl* R = rccl()) R->CommDestroy(c->comm);
  delete c;
}

extern "C" int qd_comm_size(const qd_comm* c) {
  if (!c) return QD_ERR_INVALID;
  if (c->backend == 1) return c->nranks;
  int n = 0;
  Rccl* R = rccl();
  if (!R || R->CommCount(c->comm, &n) != ncclSuccess) return QD_ERR_DEVICE;
  return n;
}
extern "C" int qd_comm_rank(const qd_comm* c) { return c ? c->rank : QD_ERR_INVALID; }
extern "C" int qd_comm_backend(const qd_comm* c) { return c ? c->backend : QD_ERR_INVALID; }

int qd_comm_allreduce_dev(qd_comm* c, double* dbuf, size_t n, int op, hipStream_t st) {
  if (c->backend == 1) {
    // host backend: the buffer makes one round trip through pinned memory; the stream waits for it (what follows on the stream sees the
    // reduced values exactly as after ncclAllReduce)
    int r;
    if (n > c->hbuf.cap) QD_HIP(hipStreamSynchronize(st));  // (an upload from the old staging buffer may still be in flight)
    if ((r = c->hbuf.ensure(n))) return r;
    QD_HIP(hipMemcpyAsync(c->hbuf.p, dbuf, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    QD_HIP(hipStreamSynchronize(st));
    if ((r = host_allreduce(c->host, c->hbuf.p, n, op))) return r;
    QD_HIP(hipMemcpyAsync(dbuf, c->hbuf.p, sizeof(double) * n, hipMemcpyHostToDevice, st));
    return QD_OK;
  }
  QD_RCCL_OR_FAIL(R);
  QD_NCCL(R, R->AllReduce(dbuf, dbuf, n, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, st));
  return QD_OK;
}

// host convenience (timings, test hooks): staged through a device buffer, blocking
extern "C" int qd_comm_allreduce(qd_comm* c, double* buf, int n, int op) {
  if (!c || !buf || n < 0 || (op != 0 && op != 1)) return fail(QD_ERR_INVALID, "qd_comm_allreduce: bad argument");
  if (n == 0) return QD_OK;
  if (c->backend == 1) return host_allreduce(c->host, buf, (size_t)n, op);
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
