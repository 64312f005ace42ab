// api_comm.hip - C ABI: pclmi_comm, the exchange step of the sharded ICP loop below the C ABI.
//
// One process per GPU on one node.  Every rank allocates a mailbox in device memory and maps every peer's mailbox
// through HIP IPC (xGMI peer access); the all-reduce itself is device code (comm_dev.hpp) that runs either as a
// one-workgroup kernel of its own (pclmi_comm_allreduce_f64) or inside the ICP step kernel's last workgroup
// (pclmi_icp_set_comm).  The reference has no counterpart: it is single-process ("Parallelism").
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <signal.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "comm_dev.hpp"
#include "pclmi_internal.h"

using namespace pclmi;

struct pclmi_comm {
  int rank = 0, world = 1, device = 0;
  unsigned long long *box = nullptr;  // local mailbox (+ the sequence counter behind it)
  unsigned int *seq = nullptr;
  void *peer[kCommMaxRanks] = {nullptr};  // IPC mappings (nullptr for the own rank)
  bool connected = false;
  bool dead = false;          // an exchange timed out: the ranks' sequence numbers are out of step for good
  const char *memory = "?";   // flavour of the mailbox allocation (reported with PCLMI_ERR_COMM)
  size_t bulk_bytes = 0;      // bulk window behind the mailbox (the pair mask of the sharded RANSAC step)
  double timeout_s = 10.0;
  static size_t bulk_offset() { return kCommBoxWords * sizeof(unsigned long long) + kCommCtrlBytes; }
  CommView view() const {
    CommView v;
    v.rank = rank;
    v.world = world;
    for (int p = 0; p < kCommMaxRanks; ++p) {
      v.box[p] = p == rank ? box : (unsigned long long *)peer[p];
      v.bulk[p] = v.box[p] ? (unsigned long long *)((char *)v.box[p] + bulk_offset()) : nullptr;
    }
    v.seq = seq;
    v.timeout_ticks = (unsigned long long)(timeout_s * 1e8);  // wall_clock64: 100 MHz
    v.bulk_cap = (bulk_bytes / sizeof(unsigned long long) / (2 * (size_t)world)) & ~(size_t)1;
    return v;
  }
};

namespace pclmi {
CommView comm_view(const pclmi_comm *c) { return c->view(); }
int comm_device(const pclmi_comm *c) { return c->device; }
bool comm_dead(const pclmi_comm *c) { return c->dead; }
void comm_set_dead(pclmi_comm *c) { c->dead = true; }
const char *comm_memory(const pclmi_comm *c) { return c->memory; }
}  // namespace pclmi

namespace {

struct HandleBlob {  // what travels between the ranks: PCLMI_COMM_HANDLE_BYTES
  uint32_t magic;
  int32_t rank, world, device;
  int64_t pid;
  hipIpcMemHandle_t ipc;
};
static_assert(sizeof(HandleBlob) <= PCLMI_COMM_HANDLE_BYTES, "handle blob too large");
constexpr uint32_t kMagic = 0x70636c6du;

__global__ __launch_bounds__(64) void comm_allreduce_kernel(CommView c, double *buf, int count, int *status) {
  const int lane = threadIdx.x;
  const unsigned int seq = *c.seq + 1u;
  double v = lane < count ? buf[lane] : 0.0;
  const bool ok = dev::comm_allreduce_wave(c, v, count, lane, seq);
  if (ok && lane < count) buf[lane] = v;
  if (lane == 0) {
    *c.seq = seq;
    if (!ok) *status = 1;
  }
}

}  // namespace

extern "C" {

int pclmi_comm_begin(int rank, int world, pclmi_comm **out, unsigned char handle[PCLMI_COMM_HANDLE_BYTES]) {
  PCLMI_CHECK(out && handle, PCLMI_ERR_INVALID, "NULL argument");
  *out = nullptr;
  PCLMI_CHECK(world >= 1 && world <= kCommMaxRanks, PCLMI_ERR_UNSUPPORTED, "world size %d outside 1..%d", world,
              kCommMaxRanks);
  PCLMI_CHECK(rank >= 0 && rank < world, PCLMI_ERR_INVALID, "rank %d outside 0..%d", rank, world - 1);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  int dev = 0;
  PCLMI_HIP(hipGetDevice(&dev));
  pclmi_comm *c = new pclmi_comm();
  c->rank = rank;
  c->world = world;
  c->device = dev;
  {
    const char *mb = getenv("PCLMI_COMM_BULK_MB");
    long v = mb ? atol(mb) : 64;
    if (v < 0) v = 0;
    if (v > 4096) v = 4096;
    c->bulk_bytes = world > 1 ? (size_t)v << 20 : 0;
  }
  const size_t bytes = pclmi_comm::bulk_offset() + c->bulk_bytes;
  // uncached device memory: peers write it over xGMI, the owner polls it; nothing of it may linger in an L2.
  // Coarse-grained memory (plain hipMalloc) is NOT a fallback: the owner's polling loads are not guaranteed to see
  // the peers' writes there (intermittent timeouts); PCLMI_COMM_ALLOW_COARSE=1 forces it for experiments.
  hipError_t e = hipExtMallocWithFlags((void **)&c->box, bytes, hipDeviceMallocUncached);
  c->memory = "uncached";
  if (e != hipSuccess) {
    e = hipExtMallocWithFlags((void **)&c->box, bytes, hipDeviceMallocFinegrained);
    c->memory = "fine-grained";
  }
  if (e != hipSuccess && getenv("PCLMI_COMM_ALLOW_COARSE") && atoi(getenv("PCLMI_COMM_ALLOW_COARSE")) != 0) {
    e = hipMalloc((void **)&c->box, bytes);
    c->memory = "coarse-grained (PCLMI_COMM_ALLOW_COARSE)";
  }
  if (e != hipSuccess) {
    delete c;
    (void)hipGetLastError();
    set_error("pclmi_comm: neither uncached nor fine-grained device memory is available for the mailbox (%s)",
              hipGetErrorString(e));
    return PCLMI_ERR_HIP;
  }
  c->seq = (unsigned int *)(c->box + kCommBoxWords);
  if ((e = hipMemset(c->box, 0, bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
    (void)hipFree(c->box);
    delete c;
    return hip_fail(e, "mailbox clear", __FILE__, __LINE__);
  }
  HandleBlob b;
  std::memset(&b, 0, sizeof(b));
  b.magic = kMagic;
  b.rank = rank;
  b.world = world;
  b.device = dev;
  b.pid = (int64_t)getpid();
  if (world > 1 && (e = hipIpcGetMemHandle(&b.ipc, c->box)) != hipSuccess) {
    (void)hipFree(c->box);
    delete c;
    return hip_fail(e, "hipIpcGetMemHandle (mailbox)", __FILE__, __LINE__);
  }
  std::memset(handle, 0, PCLMI_COMM_HANDLE_BYTES);
  std::memcpy(handle, &b, sizeof(b));
  if (world == 1) c->connected = true;
  *out = c;
  return PCLMI_OK;
}

int pclmi_comm_connect(pclmi_comm *c, const unsigned char *all_handles) {
  PCLMI_CHECK(c && all_handles, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->connected || c->world == 1, PCLMI_ERR_INVALID, "communicator is already connected");
  PCLMI_HIP(hipSetDevice(c->device));
  for (int p = 0; p < c->world; ++p) {
    HandleBlob b;
    std::memcpy(&b, all_handles + (size_t)p * PCLMI_COMM_HANDLE_BYTES, sizeof(b));
    PCLMI_CHECK(b.magic == kMagic && b.rank == p && b.world == c->world, PCLMI_ERR_INVALID,
                "handle %d is not rank %d's handle of a %d-rank communicator", p, p, c->world);
    if (p == c->rank) continue;
    PCLMI_CHECK(b.pid != (int64_t)getpid(), PCLMI_ERR_UNSUPPORTED,
                "ranks %d and %d live in one process: pclmi_comm is one process per GPU", c->rank, p);
    hipError_t e = hipIpcOpenMemHandle(&c->peer[p], b.ipc, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail(e, "hipIpcOpenMemHandle (peer mailbox)", __FILE__, __LINE__);
  }
  c->connected = true;
  return PCLMI_OK;
}

// Rendezvous through POSIX shared memory for hosts without another channel (C++ programs started once per GPU):
// /dev/shm/<name> holds one slot per rank; rank 0 creates it, every rank publishes its handle, waits for the others
// and connects; the last rank to finish unlinks the name.
int pclmi_comm_create_shm(int rank, int world, const char *name, pclmi_comm **out) {
  PCLMI_CHECK(out && name && name[0], PCLMI_ERR_INVALID, "NULL argument");
  *out = nullptr;
  unsigned char mine[PCLMI_COMM_HANDLE_BYTES];
  pclmi_comm *c = nullptr;
  RC(pclmi_comm_begin(rank, world, &c, mine));
  if (world == 1) {
    *out = c;
    return PCLMI_OK;
  }
  struct Shm {
    std::atomic<uint32_t> ready[kCommMaxRanks];
    std::atomic<uint32_t> done;
    std::atomic<int64_t> creator_pid;  // rank 0's pid: a segment whose creator is gone is a crashed job's
    unsigned char handles[kCommMaxRanks][PCLMI_COMM_HANDLE_BYTES];
  };
  char path[256];
  snprintf(path, sizeof(path), "/%s%s", name[0] == '/' ? "" : "pclmi_", name[0] == '/' ? name + 1 : name);
  auto fail = [&](const char *what) {
    set_error("pclmi_comm_create_shm(%s): %s", path, what);
    pclmi_comm_destroy(c);
    return (int)PCLMI_ERR_HIP;
  };
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
  Shm *sh = nullptr;
  for (;;) {  // (a non-zero rank starts over when the object it mapped turns out to be a stale one)
    int fd = -1;
    if (rank == 0) {
      shm_unlink(path);  // a stale segment of a crashed job
      fd = shm_open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) return fail("cannot create the segment");
      if (ftruncate(fd, sizeof(Shm)) != 0) {
        close(fd);
        shm_unlink(path);
        return fail("cannot size the segment");
      }
    } else {
      for (;;) {
        fd = shm_open(path, O_RDWR, 0600);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Shm)) break;
        if (fd >= 0) close(fd);
        fd = -1;
        if (std::chrono::steady_clock::now() > deadline) return fail("rank 0 never created the segment");
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    }
    struct stat st0;
    const bool have_ino = fstat(fd, &st0) == 0;
    sh = (Shm *)mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (sh == MAP_FAILED) {
      if (rank == 0) shm_unlink(path);
      return fail("mmap failed");
    }
    if (rank == 0) sh->creator_pid.store((int64_t)getpid(), std::memory_order_release);
    bool stale = false;
    if (rank != 0) {
      // wait for the creator's pid, then make sure that process exists: the segment of a crashed job holds ready
      // flags and IPC handles of dead processes
      int64_t cp = 0;
      while ((cp = sh->creator_pid.load(std::memory_order_acquire)) == 0 && !stale) {
        if (std::chrono::steady_clock::now() > deadline) {
          munmap(sh, sizeof(Shm));
          return fail("rank 0 never published its pid");
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (kill((pid_t)cp, 0) != 0 && errno == ESRCH) stale = true;
    }
    if (!stale) {
      std::memcpy(sh->handles[rank], mine, PCLMI_COMM_HANDLE_BYTES);
      sh->ready[rank].store(1u, std::memory_order_release);
      for (int p = 0; p < world && !stale; ++p)
        while (sh->ready[p].load(std::memory_order_acquire) != 1u) {
          if (std::chrono::steady_clock::now() > deadline) {
            munmap(sh, sizeof(Shm));
            if (rank == 0) shm_unlink(path);
            return fail("a peer never arrived");
          }
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
      if (rank != 0 && have_ino) {
        // the name must still lead to the object that was mapped (rank 0 replaces a stale one when it arrives); once the
        // rendezvous has completed the name may already be gone (the last rank unlinks it): that is fine
        const int fd2 = shm_open(path, O_RDWR, 0600);
        struct stat st1;
        if (fd2 >= 0 && fstat(fd2, &st1) == 0 && st1.st_ino != st0.st_ino) stale = true;
        if (fd2 >= 0) close(fd2);
      }
    }
    if (!stale) break;
    munmap(sh, sizeof(Shm));
    sh = nullptr;
    if (std::chrono::steady_clock::now() > deadline) return fail("only a stale segment of an earlier job was found");
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
  int rc = pclmi_comm_connect(c, &sh->handles[0][0]);
  if (sh->done.fetch_add(1u) + 1u == (uint32_t)world) shm_unlink(path);
  munmap(sh, sizeof(Shm));
  if (rc) {
    pclmi_comm_destroy(c);
    return rc;
  }
  *out = c;
  return PCLMI_OK;
}

int pclmi_comm_rank(const pclmi_comm *c) { return c ? c->rank : -1; }
int pclmi_comm_size(const pclmi_comm *c) { return c ? c->world : 0; }

int pclmi_comm_set_timeout(pclmi_comm *c, double seconds) {
  PCLMI_CHECK(c && seconds > 0.0 && seconds <= 3600.0, PCLMI_ERR_INVALID, "bad timeout");
  c->timeout_s = seconds;
  return PCLMI_OK;
}

int pclmi_comm_allreduce_f64(pclmi_comm *c, double *d_buf, int count, void *stream) {
  PCLMI_CHECK(c && d_buf, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(count >= 1 && count <= kCommMaxCount, PCLMI_ERR_UNSUPPORTED, "count %d outside 1..%d", count,
              kCommMaxCount);
  PCLMI_CHECK(c->connected, PCLMI_ERR_INVALID, "communicator is not connected (pclmi_comm_connect)");
  if (c->world == 1) return PCLMI_OK;
  PCLMI_CHECK(!c->dead, PCLMI_ERR_COMM,
              "an earlier exchange of this communicator timed out: destroy it and create a new one on every rank");
  PCLMI_HIP(hipSetDevice(c->device));
  // the status word lives behind the sequence counter in the (zero-initialised) mailbox block
  int *status = (int *)(c->seq + kCtrlStatus);
  hipLaunchKernelGGL(comm_allreduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c->view(), d_buf, count, status);
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

int pclmi_comm_status(pclmi_comm *c) {
  PCLMI_CHECK(c, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_HIP(hipSetDevice(c->device));
  PCLMI_HIP(hipDeviceSynchronize());
  unsigned int st[kCtrlDead + 1] = {0};
  PCLMI_HIP(hipMemcpy(st, c->seq, sizeof(st), hipMemcpyDeviceToHost));
  if (st[kCtrlStatus] != 0u || st[kCtrlDead] != 0u) c->dead = true;
  PCLMI_CHECK(!c->dead, PCLMI_ERR_COMM,
              "a peer's data did not arrive within %.1f s (rank %d of %d, %s mailbox): the communicator is dead - "
              "destroy it and create a new one on every rank",
              c->timeout_s, c->rank, c->world, c->memory);
  return PCLMI_OK;
}

void pclmi_comm_destroy(pclmi_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (int p = 0; p < kCommMaxRanks; ++p)
    if (c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->box) (void)hipFree(c->box);
  delete c;
}

}  // extern "C"
