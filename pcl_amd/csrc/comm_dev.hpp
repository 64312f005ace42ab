// comm_dev.hpp - device side of pclmi_comm: a one-shot all-reduce of a few fp64 values between the GPUs of one
// node through peer-mapped mailboxes (xGMI), callable from INSIDE a kernel.  gfx950 / wave64 only.
//
// Why not a collective kernel per iteration: the ICP loop exchanges 24 doubles per iteration; as a
// separate launch between the step and the solve that is two more kernel boundaries and a latency-bound collective
// kernel.  Here the workgroup that finishes the step's reduction pushes its sums to every peer, waits for theirs and
// goes straight on to the solve (icp_kernels.hip).
//
// Protocol.  Every rank owns one mailbox (device memory, mapped into every peer through HIP IPC):
//   box[parity][source rank][64] granules of 8 bytes = {tag (high 32 bits) = sequence number, payload (low 32 bits)}.
// A double travels as two granules (low word, high word).  Each granule is written by ONE 8-byte system-scope store
// and read by 8-byte system-scope loads, so a reader sees either the old or the new granule, never a torn one; the tag
// says which exchange the payload belongs to, so no separate flag and no store ordering across the fabric is needed.
// Exchange number s uses parity s & 1: a rank can be at most one exchange ahead of its slowest peer (it needs that
// peer's data of exchange s to finish s), so the granules of exchange s are never overwritten (by s + 2) before every
// peer has read them.  The sums are added in RANK ORDER on every rank: all ranks get the same bits.
// Every wait is bounded (timeout_ticks of the 100 MHz wall clock): a missing peer ends the kernel with an error
// instead of hanging the GPU.
#ifndef PCLMI_COMM_DEV_HPP_
#define PCLMI_COMM_DEV_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pclmi {

constexpr int kCommMaxRanks = 16;
constexpr int kCommGranules = 64;  // per (parity, source): 32 doubles
constexpr int kCommMaxCount = kCommGranules / 2;
constexpr size_t kCommBoxWords = (size_t)2 * kCommMaxRanks * kCommGranules;  // 8-byte words per mailbox

struct CommView {
  int rank = 0, world = 1;
  unsigned long long *box[kCommMaxRanks];  // box[p]: rank p's mailbox (box[rank]: the local one)
  unsigned int *seq = nullptr;             // local: number of exchanges completed
  unsigned long long timeout_ticks = 0;    // wall_clock64() ticks (100 MHz) a wait may last
};

namespace dev {

// explicit global address space: the scoped loads / stores must be global_ instructions, never flat_ ones
typedef __attribute__((address_space(1))) unsigned long long *g_u64p;
typedef __attribute__((address_space(1))) const unsigned long long *g_cu64p;

// All-reduce (sum) of one double per lane over lanes [0, count) of the calling wavefront; count <= kCommMaxCount.
// Every lane of the wave must call.  seq = *c.seq + 1 (the caller stores it back when the exchange has completed).
// Returns false when a peer's data did not arrive in time (v is then undefined).
__device__ __forceinline__ bool comm_allreduce_wave(const CommView &c, double &v, int count, int lane, unsigned int seq) {
  const bool mine = lane < count;
  const int par = (int)(seq & 1u);
  const unsigned long long tag = (unsigned long long)seq << 32;
  const unsigned long long g_lo = tag | (unsigned int)__double2loint(v);
  const unsigned long long g_hi = tag | (unsigned int)__double2hiint(v);
  if (mine) {
    for (int p = 0; p < c.world; ++p) {
      if (p == c.rank) continue;
      unsigned long long *dst = c.box[p] + ((size_t)(par * kCommMaxRanks + c.rank) * kCommGranules + 2 * lane);
      __hip_atomic_store((g_u64p)dst, g_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store((g_u64p)(dst + 1), g_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  const long long t0 = wall_clock64();
  double acc = 0.0;
  bool ok = true;
  for (int p = 0; p < c.world; ++p) {  // rank order: identical sums on every rank
    double x = v;
    if (p != c.rank) {
      const unsigned long long *src =
          c.box[c.rank] + ((size_t)(par * kCommMaxRanks + p) * kCommGranules + 2 * (mine ? lane : 0));
      unsigned long long a = 0, b = 0;
      unsigned int spins = 0;
      for (;;) {
        a = __hip_atomic_load((g_cu64p)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        b = __hip_atomic_load((g_cu64p)(src + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool have = !mine || ((a >> 32) == seq && (b >> 32) == seq);
        if (__all(have)) break;
        if ((++spins & 63u) == 0u &&
            __any((unsigned long long)(wall_clock64() - t0) > c.timeout_ticks)) {  // wave-uniform decision
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (!ok) break;
      x = __hiloint2double((int)(unsigned int)b, (int)(unsigned int)a);
    }
    acc += x;
  }
  v = acc;
  return ok;
}

}  // namespace dev
}  // namespace pclmi
#endif
