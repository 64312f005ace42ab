// comm_dev.hpp - device side of pclmi_comm: one-shot all-reduces of a few 64-bit values between the GPUs of one
// node through peer-mapped mailboxes (xGMI), callable from INSIDE a kernel.  gfx950 / wave64 only.
//
// Why not a collective kernel per iteration: the ICP loop exchanges 24-40 doubles per iteration; as a
// separate launch between the step and the solve that is two more kernel boundaries and a latency-bound collective
// kernel.  Here the workgroup that finishes the step's reduction pushes its sums to every peer, waits for theirs and
// goes straight on to the solve (icp_kernels.hip).
//
// Protocol.  Every rank owns one mailbox (device memory, mapped into every peer through HIP IPC):
//   box[parity][source rank][128] granules of 8 bytes = {tag (high 32 bits) = sequence number, payload (low 32 bits)}.
// A 64-bit value travels as two granules (low word, high word).  Each granule is written by ONE 8-byte system-scope
// store and read by 8-byte system-scope loads, so a reader sees either the old or the new granule, never a torn one;
// the tag says which exchange the payload belongs to, so no separate flag and no store ordering across the fabric is
// needed.  Exchange number s uses parity s & 1: a rank can be at most one exchange ahead of its slowest peer (it needs
// that peer's data of exchange s to finish s), so the granules of exchange s are never overwritten (by s + 2) before
// every peer has read them.  Sums are added in RANK ORDER on every rank: all ranks get the same bits.
// Every wait is bounded (timeout_ticks of the 100 MHz wall clock): a missing peer ends the kernel with an error
// instead of hanging the GPU.  A timed-out exchange leaves the ranks' sequence numbers out of step, so it marks the
// communicator DEAD (a word next to the sequence counter): every later exchange fails at once instead of waiting for
// the full timeout again; the host destroys and re-creates the communicator (pclmi.h).
//
// The bulk window (same allocation, behind the mailbox) carries the same granules for exchanges of many words: the
// bitwise OR of a bit mask over all ranks (the pair mask of the sharded RANSAC step), one thread per word.
#ifndef PCLMI_COMM_DEV_HPP_
#define PCLMI_COMM_DEV_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pclmi {

constexpr int kCommMaxRanks = 16;
constexpr int kCommGranules = 128;  // per (parity, source): 64 values of 64 bits
constexpr int kCommMaxCount = kCommGranules / 2;
constexpr size_t kCommBoxWords = (size_t)2 * kCommMaxRanks * kCommGranules;  // 8-byte words per mailbox
constexpr size_t kCommCtrlBytes = 256;  // behind the mailbox: seq, status, dead, bulk seq, bulk ticket
// words of the control block (unsigned int index from `seq`)
constexpr int kCtrlStatus = 4, kCtrlDead = 8, kCtrlBulkSeq = 12, kCtrlBulkTicket = 16;

struct CommView {
  int rank = 0, world = 1;
  unsigned long long *box[kCommMaxRanks];  // box[p]: rank p's mailbox (box[rank]: the local one)
  unsigned int *seq = nullptr;             // local: number of exchanges completed (+ the control words behind it)
  unsigned long long timeout_ticks = 0;    // wall_clock64() ticks (100 MHz) a wait may last
  // bulk window: bulk[p] = rank p's window, [parity][source rank][bulk_cap] granules
  unsigned long long *bulk[kCommMaxRanks];
  unsigned long long bulk_cap = 0;  // granules per (parity, source)
};

namespace dev {

// explicit global address space: the scoped loads / stores must be global_ instructions, never flat_ ones
typedef __attribute__((address_space(1))) unsigned long long *g_u64p;
typedef __attribute__((address_space(1))) const unsigned long long *g_cu64p;
typedef __attribute__((address_space(1))) unsigned int *g_u32cp;

__device__ __forceinline__ bool comm_is_dead(const CommView &c) {
  return __hip_atomic_load((g_u32cp)(c.seq + kCtrlDead), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ __forceinline__ void comm_mark_dead(const CommView &c) {
  __hip_atomic_store((g_u32cp)(c.seq + kCtrlDead), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

enum CommOp { kCommSumF64 = 0, kCommOrU64 = 1 };

// All-reduce of one 64-bit value per lane over lanes [0, count) of the calling wavefront; count <= kCommMaxCount.
// OP = kCommSumF64: the values are doubles, summed in rank order; kCommOrU64: bitwise OR.
// Every lane of the wave must call.  seq = *c.seq + 1 (the caller stores it back when the exchange has completed).
// Returns false when a peer's data did not arrive in time, or the communicator is dead (bits is then undefined).
template <int OP>
__device__ __forceinline__ bool comm_allreduce_bits(const CommView &c, unsigned long long &bits, int count, int lane,
                                                    unsigned int seq) {
  if (comm_is_dead(c)) return false;  // wave-uniform: one word
  const bool mine = lane < count;
  const int par = (int)(seq & 1u);
  const unsigned long long tag = (unsigned long long)seq << 32;
  const unsigned long long g_lo = tag | (unsigned int)bits;
  const unsigned long long g_hi = tag | (unsigned int)(bits >> 32);
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
  unsigned long long acc_or = 0ull;
  bool ok = true;
  for (int p = 0; p < c.world; ++p) {  // rank order: identical sums on every rank
    unsigned long long x = bits;
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
      x = ((unsigned long long)(unsigned int)b << 32) | (unsigned int)a;
    }
    if (OP == kCommSumF64) acc += __longlong_as_double((long long)x);
    else acc_or |= x;
  }
  if (!ok) {
    if (lane == 0) comm_mark_dead(c);
    return false;
  }
  bits = OP == kCommSumF64 ? (unsigned long long)__double_as_longlong(acc) : acc_or;
  return true;
}

__device__ __forceinline__ bool comm_allreduce_wave(const CommView &c, double &v, int count, int lane, unsigned int seq) {
  unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  const bool ok = comm_allreduce_bits<kCommSumF64>(c, bits, count, lane, seq);
  v = __longlong_as_double((long long)bits);
  return ok;
}

// One word of a bulk OR exchange (bulk sequence number bseq, word index w < bulk_cap / 2): push the local word to
// every peer's window, wait for theirs, return the OR.  Called by one thread per word; ok = false on a timeout.
__device__ __forceinline__ unsigned long long comm_bulk_or_word(const CommView &c, unsigned long long v, size_t w,
                                                               unsigned int bseq, bool &ok) {
  const int par = (int)(bseq & 1u);
  const unsigned long long tag = (unsigned long long)bseq << 32;
  for (int p = 0; p < c.world; ++p) {
    if (p == c.rank) continue;
    unsigned long long *dst = c.bulk[p] + ((size_t)(par * c.world + c.rank) * c.bulk_cap + 2 * w);
    __hip_atomic_store((g_u64p)dst, tag | (unsigned int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((g_u64p)(dst + 1), tag | (unsigned int)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const long long t0 = wall_clock64();
  for (int p = 0; p < c.world && ok; ++p) {
    if (p == c.rank) continue;
    const unsigned long long *src = c.bulk[c.rank] + ((size_t)(par * c.world + p) * c.bulk_cap + 2 * w);
    unsigned int spins = 0;
    for (;;) {
      const unsigned long long a = __hip_atomic_load((g_cu64p)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long b = __hip_atomic_load((g_cu64p)(src + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((a >> 32) == bseq && (b >> 32) == bseq) {
        v |= ((unsigned long long)(unsigned int)b << 32) | (unsigned int)a;
        break;
      }
      if ((++spins & 63u) == 0u && (unsigned long long)(wall_clock64() - t0) > c.timeout_ticks) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  return v;
}

}  // namespace dev
}  // namespace pclmi
#endif
