// icp_kernels.hip - device side of the ICP correspondence loop on gfx950 (MI355X).
//
// One fused kernel per iteration replaces steps A, C (reduction part) and D of
// pcl::IterativeClosestPoint::computeTransformation
// (registration/include/pcl/registration/impl/icp.hpp:84-102,171-174):
//   p <- T_prev * p (the transform estimated by the previous iteration, transforms.hpp:184-191),
//   1-NN of p in the target tree (wave-uniform traversal, traverse.hpp),
//   threshold test d2 < thr2 (strict, icp.hpp:93), d_cur = min(d2, thr2) (icp.hpp:101),
//   fp64 partial sums of n, sum s, sum t, sum s t' (the inputs of
//   TransformationEstimationSVD, impl/transformation_estimation_svd.hpp:114-168) reduced with
//   wavefront shuffles, then an LDS tree per block, then a fixed-order pass over block partials
//   (bit-reproducible; no float atomics).
// Source points live on the device in Morton-cell order, so the 64 queries of a wave are
// neighbours and the loads/stores of the point array are fully coalesced.
#include <hipcub/hipcub.hpp>

#include "comm_dev.hpp"
#include "pclmi_internal.h"
#include "rigid_solve.hpp"
#include "traverse.hpp"

namespace pclmi {
using namespace dev;

namespace {

__device__ __forceinline__ float4 apply_xf(const Xf &T, float4 p) {
  // rot * p + trans, one rounding per operation, Eigen's column-sum order
  // ((c0*x + c1*y) + c2*z) + t  (common/include/pcl/common/impl/transforms.hpp:184-191)
  float4 o;
  o.x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.m[0], p.x), __fmul_rn(T.m[1], p.y)), __fmul_rn(T.m[2], p.z)), T.m[3]);
  o.y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.m[4], p.x), __fmul_rn(T.m[5], p.y)), __fmul_rn(T.m[6], p.z)), T.m[7]);
  o.z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.m[8], p.x), __fmul_rn(T.m[9], p.y)), __fmul_rn(T.m[10], p.z)), T.m[11]);
  o.w = p.w;
  return o;
}

__device__ __forceinline__ bool finite3(float4 p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

__global__ void transform_kernel(float4 *__restrict__ pts, int n, Xf T) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  if (finite3(p)) pts[i] = apply_xf(T, p);  // the !is_dense branch skips non-finite points (:196-204)
}

// strided host-layout points -> float4 (x, y, z, 1), optionally through an index list
__global__ void pack_kernel(const float *__restrict__ in, int stride_f, const int *__restrict__ rows, int n,
                            float4 *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *p = in + (size_t)(rows ? rows[i] : i) * stride_f;
  out[i] = make_float4(p[0], p[1], p[2], 1.f);
}

__global__ void permute_kernel(const float4 *__restrict__ in, const int *__restrict__ order, int n,
                               float4 *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[order[i]];
}

__global__ void unpermute_kernel(const float4 *__restrict__ in, const int *__restrict__ order, int n,
                                 float4 *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[order[i]] = in[i];
}

template <typename T>
__global__ void unpermute_scalar_kernel(const T *__restrict__ in, const int *__restrict__ order, int n,
                                        T *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[order[i]] = in[i];
}

// wave shuffle reduction of NV doubles, then one LDS row per wave, then thread j sums row entries j
template <int NV, int WPB>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NV], double (*lds)[NV], int lane, int wib,
                                                   double *__restrict__ partial_row) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double v = acc[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    acc[j] = v;
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[wib][j] = acc[j];
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < NV) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < WPB; ++w) s += lds[w][tid];
    partial_row[tid] = s;
  }
}

// wave sum of 8 doubles (lanes with use == false contribute zero) -> row[0..8).  Reduce-scatter butterfly: at
// the exchanges over 32, 16 and 8 lanes every lane hands HALF of its values to its partner and keeps the other
// half (8 -> 4 -> 2 -> 1 values per lane), then the one value left goes through the last three exchanges: 10
// 64-bit shuffles per lane instead of 48 for eight independent butterflies.  The order of the additions is fixed
// by the lane numbers, so the sums are reproducible bit for bit.
__device__ __forceinline__ void wave_reduce_to_row(double (&v)[8], bool use, double *row, int lane) {
  double x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = use ? v[j] : 0.0;
  const bool h32 = (lane & 32) != 0, h16 = (lane & 16) != 0, h8 = (lane & 8) != 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // lanes 0-31 go on with values 0-3, lanes 32-63 with values 4-7
    const double send = h32 ? x[j] : x[j + 4], keep = h32 ? x[j + 4] : x[j];
    x[j] = keep + __shfl_xor(send, 32);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const double send = h16 ? x[j] : x[j + 2], keep = h16 ? x[j + 2] : x[j];
    x[j] = keep + __shfl_xor(send, 16);
  }
  {
    const double send = h8 ? x[0] : x[1], keep = h8 ? x[1] : x[0];
    x[0] = keep + __shfl_xor(send, 8);
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) x[0] += __shfl_xor(x[0], o);
  // the lane whose low three bits are zero holds the total of value 4 * bit5 + 2 * bit4 + bit3
  if ((lane & 7) == 0) row[(lane >> 3) & 7] = x[0];
}

constexpr int NS = PCLMI_ICP_NSUMS;  // 24
constexpr int NSP = PCLMI_ICP_NSUMS_PLANE;  // 32: the point-to-plane loop (layout: pclmi.h)
constexpr int ICP_WPB = 4;
// EST: 0 = TransformationEstimationSVD, 1 = TransformationEstimationPointToPlaneLLS
// RS: the step of the RANSAC loop - one more group of eight sums: the moments of the PAIRED source points about the
// origin, from which sample_dist_thresh_ of the iteration's SampleConsensusModelRegistration is computed
// (sac_model_registration.h:257-279) without another pass over the points
template <int EST, bool RS = false>
struct SumsOf {
  static constexpr int n = (EST ? NSP : NS) + (RS ? 8 : 0);
  static constexpr int pairs = EST ? 27 : 0;     // pairs under the distance threshold
  static constexpr int dsum = EST ? 28 : 16;     // sum min(d2, thr2)
  static constexpr int fit_num = EST ? 29 : 17;  // sum (d_now - d_prev)
  static constexpr int points = EST ? 30 : 18;   // points searched
  // RS only: sum s (3 values; EST = 0 has them anyway) and sum s s' (xx, xy, xz, yy, yz, zz) over the pairs
  static constexpr int mom1 = EST ? 31 : 1;
  static constexpr int mom2 = EST ? 34 : 24;
  // the sums a refit launch of the RANSAC loop takes over from the step (IcpTail::carry): [carry0, carry1)
  static constexpr int carry0 = EST ? 28 : 16, carry1 = EST ? 32 : 24;
};

// One lane turns the (all-reduced) sums into the next transform and evaluates the convergence criteria
// (icp.hpp:103-108,171-196): the device-side twin of the loop body of pclmi_icp_align.
// What the solve reads of the device state, fetched by 40 lanes at once (in flight during the reduction / exchange)
// instead of by the solving lane one dependent global load after the other - the solve sits on the critical path of
// every iteration.  It is read only after the rotation has been computed, so none of it is live (and spilled, under
// the step kernel's 64-VGPR budget) across that computation.
// State words that a LATER ITERATION OF THE SAME LAUNCH reads back (the resident loop): agent-scope accesses, which
// go past the vector L1 and the scalar cache (both may hold the word's value from before the store); AG = false: plain
template <bool AG, typename V>
__device__ __forceinline__ void st_state(V *p, V v) {
  typedef __attribute__((address_space(1))) V *gp;
  if (AG) __hip_atomic_store((gp)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool AG, typename V>
__device__ __forceinline__ V ld_state(const V *p) {
  typedef __attribute__((address_space(1))) const V *gp;
  if (AG) return __hip_atomic_load((gp)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

struct SolveIn {
  float T[16], F[16];
  int min_corr, iterations, max_iter, pad;
  double teps, feps;
};
template <bool AG = false>
__device__ __forceinline__ void fetch_solve_in(const IcpDevState *__restrict__ st, SolveIn *in, int lane) {
  if (lane < 16) in->T[lane] = ld_state<AG>(&st->T[lane]);
  else if (lane < 32) in->F[lane - 16] = ld_state<AG>(&st->final_T[lane - 16]);
  else if (lane == 32) in->min_corr = st->min_correspondences;
  else if (lane == 33) in->iterations = ld_state<AG>(&st->iterations);
  else if (lane == 34) in->max_iter = st->max_iterations;
  else if (lane == 35) in->teps = st->transformation_epsilon;
  else if (lane == 36) in->feps = st->euclidean_fitness_epsilon;
}

// One lane turns the (all-reduced) sums into the next transform and evaluates the convergence criteria
// (icp.hpp:103-108,171-196): the device-side twin of the loop body of pclmi_icp_align.
template <int EST>
__device__ __forceinline__ void icp_solve_body(const double *__restrict__ sums, IcpDevState *__restrict__ st,
                                               double *work /* 42 doubles of LDS, rigid_solve.hpp svd3 */,
                                               const SolveIn *in) {
  const double *h = sums;
  const long long cnt = llround(h[SumsOf<EST>::pairs]);
  st->last_pairs = (int)cnt;
  st->last_inliers = (int)cnt;
  if (cnt < (long long)in->min_corr) {  // icp.hpp:103-108
    st->status = PCLMI_ERR_NOT_ENOUGH;
    st->done = 1;
    return;
  }
  float Tprev[16], T[16], F[16];
#ifdef PCLMI_SOLVE_TICKS
  st->walk[0] = (unsigned long long)wall_clock64();
#endif
  if (!(EST ? rs::solve_point_to_plane_sums(h, T, work) : rs::solve_from_sums(h, T, work))) {
    st->status = PCLMI_ERR_NOT_ENOUGH;
    st->done = 1;
    return;
  }
#ifdef PCLMI_SOLVE_TICKS
  st->walk[1] = (unsigned long long)wall_clock64();
#endif
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    Tprev[i] = in->T[i];
    F[i] = in->F[i];
  }
  rs::mat4_mul(T, F, F);  // icp.hpp:177
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    st->T[i] = T[i];
    st->final_T[i] = F[i];
  }
  st->apply = 1;  // icp.hpp:174, carried out by the next step (or at the end)
  const int it = in->iterations + 1;
  st->iterations = it;
  const double tdelta = (double)rs::transform_delta(T, Tprev);
  const double n_points = h[SumsOf<EST>::points] > 0 ? h[SumsOf<EST>::points] : 1.0;
  const double fit = h[SumsOf<EST>::fit_num] / n_points;  // registration.hpp:60-68
  st->tdelta = tdelta;
  st->fit = fit;
  if (it >= in->max_iter || tdelta < in->teps || fabs(fit) <= in->feps) {  // icp.hpp:193-196
    st->converged = 1;
    st->done = 1;
  }
}


// agent-scope (sc1) accesses in the global address space: how workgroups of one launch hand data to each other
// without cache maintenance (MI355X_MICROARCH.md, "Valid forms": every store and every load of the handed-off bytes
// sc1, the storing waves drain their stores, one lane then signals with an agent-scope atomic add)
typedef __attribute__((address_space(1))) double *g_f64p;
typedef __attribute__((address_space(1))) const double *g_cf64p;
typedef __attribute__((address_space(1))) unsigned int *g_u32p;
__device__ __forceinline__ void st_agent(double *p, double v) {
  __hip_atomic_store((g_f64p)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
  return __hip_atomic_load((g_cf64p)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// RandomSampleConsensus::computeModel's bookkeeping (ransac.hpp:71-121) over one batch's trials, in order: ready(b)
// says whether trial b has a model (a degenerate sample is skipped), count(b) its inliers.  Carries the rule's state in
// rl across the batches of an iteration; returns whether the rule has settled (otherwise st->hold asks the host for
// another batch).  One lane.
template <typename ReadyF, typename CountF>
__device__ __forceinline__ bool ransac_consume(RansacLoopState *rl, IcpDevState *st, const RansacTrial *trials, int batch,
                                               int max_trials, ReadyF ready, CountF count) {
  int iterations = rl->iterations, best = rl->best, have = rl->have;
  unsigned int skipped = rl->skipped;
  double k = rl->k;
  const unsigned int max_skip = (unsigned int)min((long long)max_trials * 10ll, 0xffffffffll);
  bool stop = false;
  int b = 0;
  for (; b < batch && iterations < k && skipped < max_skip; ++b) {
    if (!ready(b)) {  // degenerate sample
      ++skipped;
      continue;
    }
    const int inl = count(b);
    if (inl > best) {
      best = inl;
      have = 1;
      rl->best_trial = b;
      for (int i = 0; i < 16; ++i) rl->Tbest[i] = trials[b].T[i];
      const double w = (double)best / (double)rl->cnt;
      double p_no = 1.0 - pow(w, 3.0);
      p_no = fmax(2.220446049250313e-16, p_no);
      p_no = fmin(1.0 - 2.220446049250313e-16, p_no);
      k = log(1.0 - 0.99) / log(p_no);  // ransac.hpp:102-107
    }
    ++iterations;
    if (iterations > max_trials) {
      stop = true;
      break;
    }
  }
  // every model of the batch consumed and the rule still asks for more: no further sample exists once 1000 draws in
  // a row were bad (computeModel breaks out, ransac.hpp:72-76)
  if (!stop && b >= batch && rl->dead) stop = true;
  const bool settled = stop || !(iterations < k && skipped < max_skip);
  rl->iterations = iterations;
  rl->best = best;
  rl->have = have;
  rl->skipped = skipped;
  rl->k = k;
  rl->settled = settled ? 1 : 0;
  if (settled) rl->trials_last = iterations;
  else st->hold = 1;
  return settled;
}

// What the workgroup that finishes a step's reduction goes on to do (IcpTail, see icp_step_tail).
constexpr int ICP_GROUP = 64;  // workgroups per reduction group

// Sum of column `col` over `cnt` entries spaced 1 apart, by the 8 lanes of one aligned lane group (r = lane & 7):
// entries r, r + 8, ... in ascending order, then a fixed butterfly.  The order depends on cnt only.
__device__ __forceinline__ double tail_column_sum(const double *col_base, int cnt, int r, bool active,
                                                  size_t stride = 1) {
  double s = 0.0;
  if (active)
    for (int i = r; i < cnt; i += 8) s += ld_agent(col_base + (size_t)i * stride);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 1);
  return s;
}

// Tail of the step kernel: the block partials become the NS sums of the launch WITHOUT another kernel.
// Two levels of arrival tickets: the workgroup that arrives last in its group of ICP_GROUP adds up the group's
// partials; the group that finishes last adds up the group sums (fixed order: the result does not depend on which
// workgroup happens to be last), exchanges them with the other ranks when a communicator is attached (comm_dev.hpp),
// writes sums_out and, when asked, runs the solve.  Nobody waits for anybody inside the GPU: a workgroup that is not
// last simply ends.  Returns to the caller only in the last workgroup's first wave... never: everything happens here.
// DECIDE: the first batch of the RANSAC loop (ransac_first_kernel): columns cnt0.. of the sums are the inlier counts
// of the batch's models; the stopping rule runs here and the solve only when it settles on the refit sums' model
template <int EST, bool RS = false, bool DECIDE = false>
__device__ __forceinline__ void icp_step_tail(const IcpTail &tail, double *__restrict__ partials,
                                              double (*lds)[SumsOf<EST, RS>::n]) {
  constexpr int NS = SumsOf<EST, RS>::n;  // shadows the SVD loop's count
  __shared__ int s_last;
  __shared__ double fin[NS];
  __shared__ double svd_work[42];
  __shared__ SolveIn sin;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nblk = gridDim.x;
  if (tid < NS) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < ICP_WPB; ++w) s += lds[w][tid];
    // row-major (sum j of block b at [b * NS + j]): a workgroup's sums are one contiguous 192..320-byte store that no
    // other workgroup's share a cache line with (column-major, the neighbours of a line came from eight XCDs at once)
    st_agent(&partials[(size_t)blockIdx.x * NS + tid], s);
  }
  drain_stores();
  __syncthreads();
  const int g = blockIdx.x / ICP_GROUP;
  const int gsize = min(ICP_GROUP, nblk - g * ICP_GROUP);
  if (tid == 0) {
    const unsigned int t = __hip_atomic_fetch_add((g_u32p)&tail.gticket[(size_t)g * kTicketStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned int)(gsize - 1)) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  const int r = tid & 7;
#pragma unroll
  for (int c0 = 0; c0 < NS; c0 += 32) {  // 32 columns per round (8 lanes each): one round up to 32 sums
    const int col = c0 + (tid >> 3);
    const bool colth = col < NS;
    const double s = tail_column_sum(partials + (size_t)g * ICP_GROUP * NS + (colth ? col : 0), gsize, r, colth, NS);
    if (colth && r == 0) st_agent(&tail.gpart[(size_t)col * tail.ngroups + g], s);
  }
  drain_stores();
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store((g_u32p)&tail.gticket[(size_t)g * kTicketStride], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    const unsigned int t = __hip_atomic_fetch_add((g_u32p)tail.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned int)(tail.ngroups - 1)) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  const unsigned long long tk1 = (unsigned long long)wall_clock64();
#pragma unroll
  for (int c0 = 0; c0 < NS; c0 += 32) {
    const int col = c0 + (tid >> 3);
    const bool colth = col < NS;
    const double s = tail_column_sum(tail.gpart + (size_t)(colth ? col : 0) * tail.ngroups, tail.ngroups, r, colth);
    if (colth && r == 0) fin[col] = s;
  }
  __syncthreads();
  if (tid >= 64) return;  // the first wave goes on alone
  if (tail.st && tail.solve) fetch_solve_in(tail.st, &sin, lane);  // in flight during the exchange
  double v = lane < NS ? fin[lane] : 0.0;
  bool ok = true;
  if (tail.comm.world > 1) {
    const unsigned int seq = *tail.comm.seq + 1u;
    ok = comm_allreduce_wave(tail.comm, v, NS, lane, seq);
    if (lane == 0) *tail.comm.seq = seq;
  }
  // refit launch of the RANSAC loop: what is summed over ALL points (and was exchanged by the step) is the step's
  long long step_pairs = -1;
  if (tail.carry) {
    if (lane >= SumsOf<EST>::carry0 && lane < SumsOf<EST>::carry1) v = tail.carry[lane];
    step_pairs = llround(tail.carry[SumsOf<EST>::pairs]);
  }
  if (lane < NS) {
    tail.sums_out[lane] = v;  // read by later launches / the host only: a plain store
    fin[lane] = v;
  }
  if (lane == 0) {
    __hip_atomic_store((g_u32p)tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    IcpDevState *st = tail.st;
    if (st) {
      st->ticks[1] = tk1;
      st->ticks[2] = (unsigned long long)wall_clock64();
      st->apply = 0;  // this launch has consumed the pending transform
      if (!ok) {
        st->status = PCLMI_ERR_COMM;
        st->done = 1;
      } else if (tail.solve) {
        if constexpr (DECIDE) {
          RansacLoopState *rl = tail.rl;
          bool solve_now = true;  // fewer than 3 pairs: no rejection step, the sums are over every pair
          if (rl->active) {
            constexpr int cnt0 = EST ? 32 : 24;
            const int batch = rl->ntrials;
            int b0 = 0;  // the model the refit sums belong to: the first trial that has one
            while (b0 < batch && tail.trials[b0].status != 0) ++b0;
            const bool settled = ransac_consume(rl, st, tail.trials, batch, tail.max_trials,
                                                [&](int b) { return tail.trials[b].status == 0; },
                                                [&](int b) { return (int)llround(fin[cnt0 + b]); });
            solve_now = settled && rl->have && rl->best_trial == b0;
            if (settled && !solve_now) {  // settled on another model (or on none): the host enqueues the refit
              rl->need_refit = 1;
              st->hold = 1;
            }
          }
          if (!solve_now) {
            st->ticks[3] = (unsigned long long)wall_clock64();
            return;
          }
        }
        icp_solve_body<EST>(fin, st, svd_work, &sin);
        st->last_inliers = st->last_pairs;
        if (step_pairs >= 0) {
          st->last_pairs = (int)step_pairs;
          st->hold = 0;  // the iteration the host was asked to finish is complete
        }
      }
      st->ticks[3] = (unsigned long long)wall_clock64();
    }
  }
}

// ---- the RESIDENT loop's hand-offs (icp_step_kernel<.., LOOP>) --------------------------------------------------
// Every workgroup of the launch is on the chip at once, so a workgroup may WAIT for another one.  That replaces the
// arrival tickets (store, drain, atomic, read: four memory round trips per level) by data that announces itself, the
// scheme of comm_dev.hpp: a double travels as two 8-byte granules {tag = iteration number since the context began,
// 32 payload bits}, each written by one agent-scope store and polled with agent-scope loads; a granule is either the
// old one or the new one, the tag says which.  Workgroup b leaves its NS block sums in row b; the first workgroup of
// every group of ICP_GROUP polls its group's rows and leaves the group sums in the group's row; workgroup 0 polls
// those, exchanges with the other ranks, solves and PUBLISHES the next transform (12 floats + the stop flag) the same
// way; every wavefront of the launch polls the publication and goes on with the next iteration.  The additions are
// the ticket path's, in the same order (tail_column_sum): the sums are the per-iteration launches' bit for bit.
// Every wait is bounded; a wait that runs out ends the launch with an error in the state.
constexpr int kLoopRow = 64;  // granules per row: 2 * (sums of the point-to-plane loop)
typedef __attribute__((address_space(1))) unsigned long long *gl_u64p;
typedef __attribute__((address_space(1))) const unsigned long long *gl_cu64p;
__device__ __forceinline__ void put_granules(unsigned long long *dst, double v, unsigned long long tg) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store((gl_u64p)dst, tg | (unsigned int)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((gl_u64p)(dst + 1), tg | (unsigned int)(bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long *p) {
  return __hip_atomic_load((gl_cu64p)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tail_column_sum over rows that announce themselves: entry i of the column is the granule pair at base + i * kLoopRow.
// All of a lane's entries (<= 8: cnt <= 64) are fetched in one batch and the batch is repeated until every tag of the
// wavefront matches.  ok = false when the data did not arrive within `limit` ticks.
__device__ __forceinline__ double poll_column_sum(const unsigned long long *base, int cnt, int r, bool active,
                                                  unsigned int tag, unsigned long long limit, bool &ok) {
  unsigned long long a[8], b[8];
  const long long t0 = wall_clock64();
  unsigned int spins = 0;
  for (;;) {
    bool all = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = r + 8 * j;
      const bool mine = active && i < cnt;
      const unsigned long long *src = base + (size_t)(mine ? i : 0) * kLoopRow;
      a[j] = mine ? ld_granule(src) : ((unsigned long long)tag << 32);
      b[j] = mine ? ld_granule(src + 1) : ((unsigned long long)tag << 32);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) all = all && (unsigned int)(a[j] >> 32) == tag && (unsigned int)(b[j] >> 32) == tag;
    if (__all(all)) break;
    if ((++spins & 63u) == 0u && __any((unsigned long long)(wall_clock64() - t0) > limit)) {
      ok = false;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = r + 8 * j;
    if (active && i < cnt)
      s += __longlong_as_double((long long)(((unsigned long long)(unsigned int)b[j] << 32) | (unsigned int)a[j]));
  }
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 1);
  return s;
}

// Cheap first wait of a summing workgroup: thread i < cnt watches ONE granule of row i (the last one written) - a
// sweep is cnt loads, not cnt * 2 NS - and the full batches of poll_column_sum only start when every row has shown up
// (their own tag checks still decide: a row's stores may land in any order).
__device__ __forceinline__ void wait_rows(const unsigned long long *base, int cnt, int last_granule, unsigned int tag,
                                          unsigned long long limit) {
  if ((int)threadIdx.x < cnt) {
    const unsigned long long *src = base + (size_t)threadIdx.x * kLoopRow + last_granule;
    const long long t0 = wall_clock64();
    unsigned int spins = 0;
    while ((unsigned int)(ld_granule(src) >> 32) != tag) {
      if ((++spins & 63u) == 0u && (unsigned long long)(wall_clock64() - t0) > limit) break;  // poll_column_sum reports
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

// The solve as a function of its own: inlined into the loop kernel its matrices share the 64 registers with whatever
// the register allocator keeps alive across the iterations, and the polar iteration ran out of scratch memory
template <int EST>
__device__ __attribute__((noinline)) void loop_solve(const double *fin, IcpDevState *st, double *work, const SolveIn *in) {
  icp_solve_body<EST>(fin, st, work, in);
}

// Returns true when the launch goes on with another iteration (T = the transform to apply in it).
template <int EST>
__device__ __forceinline__ bool icp_loop_tail(const IcpTail &tail, double (*lds)[SumsOf<EST>::n], unsigned int tag,
                                              Xf &T) {
  constexpr int NS = SumsOf<EST>::n;
  static_assert(2 * NS <= kLoopRow, "row too short");
  __shared__ double fin[NS];
  __shared__ double svd_work[42];
  __shared__ SolveIn sin;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nblk = gridDim.x;
  unsigned long long *rows = tail.loopbox;                                 // [nblk][kLoopRow]
  unsigned long long *grows = rows + (size_t)nblk * kLoopRow;              // [ngroups][kLoopRow]
  unsigned long long *pub = grows + (size_t)tail.ngroups * kLoopRow;       // [kLoopRow]
  const unsigned long long tg = (unsigned long long)tag << 32;
  const unsigned long long limit =
      tail.comm.timeout_ticks > 250000000ull ? 2ull * tail.comm.timeout_ticks : 500000000ull;  // >= 5 s
  if (tid < NS) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < ICP_WPB; ++w) s += lds[w][tid];
    put_granules(rows + (size_t)blockIdx.x * kLoopRow + 2 * tid, s, tg);
  }
  const int g = blockIdx.x / ICP_GROUP;
  const int r = tid & 7;
  bool ok = true;
  if (blockIdx.x % ICP_GROUP == 0) {  // the group's first workgroup adds up the group
    const int gsize = min(ICP_GROUP, nblk - g * ICP_GROUP);
    wait_rows(rows + (size_t)g * ICP_GROUP * kLoopRow, gsize, 2 * NS - 1, tag, limit);
#pragma unroll
    for (int c0 = 0; c0 < NS; c0 += 32) {
      const int col = c0 + (tid >> 3);
      const bool colth = col < NS;
      const double s = poll_column_sum(rows + (size_t)g * ICP_GROUP * kLoopRow + 2 * (colth ? col : 0), gsize, r, colth,
                                       tag, limit, ok);
      if (colth && r == 0) {
        // one group: its sum goes through the second level's additions (s + 0 + 0 + 0) without the trip through memory
        if (tail.ngroups == 1) fin[col] = ((s + 0.0) + 0.0) + 0.0;
        else put_granules(grows + (size_t)g * kLoopRow + 2 * col, s, tg);
      }
    }
  }
  if (blockIdx.x == 0) {
    if (tail.ngroups > 1) {
      wait_rows(grows, tail.ngroups, 2 * NS - 1, tag, limit);
#pragma unroll
      for (int c0 = 0; c0 < NS; c0 += 32) {
        const int col = c0 + (tid >> 3);
        const bool colth = col < NS;
        const double s = poll_column_sum(grows + 2 * (colth ? col : 0), tail.ngroups, r, colth, tag, limit, ok);
        if (colth && r == 0) fin[col] = s;
      }
    }
    const bool failed = __syncthreads_or(ok ? 0 : 1) != 0;
    if (tid < 64) {
      IcpDevState *st = tail.st;
      const unsigned long long tk1 = (unsigned long long)wall_clock64();
      fetch_solve_in<true>(st, &sin, lane);  // workgroup 0 wrote these itself in the last iteration
      double v = lane < NS ? fin[lane] : 0.0;
      bool good = !failed;
      int status = failed ? PCLMI_ERR_HIP : PCLMI_OK;
      if (good && tail.comm.world > 1) {
        const unsigned int seq = ld_state<true>(tail.comm.seq) + 1u;
        good = comm_allreduce_wave(tail.comm, v, NS, lane, seq);
        if (lane == 0) st_state<true>(tail.comm.seq, seq);
        if (!good) status = PCLMI_ERR_COMM;
      }
      if (lane < NS) {
        tail.sums_out[lane] = v;
        fin[lane] = v;
      }
      if (lane == 0) {
        st->ticks[1] = tk1;
        st->ticks[2] = (unsigned long long)wall_clock64();
        st->apply = 0;  // this iteration has consumed the pending transform
        if (!good) {
          st->status = status;
          st->done = 1;
        } else {
          loop_solve<EST>(fin, st, svd_work, &sin);
          st->last_inliers = st->last_pairs;
        }
        st->ticket = tag;
        st->ticks[3] = (unsigned long long)wall_clock64();
      }
      // publish: lanes 0-11 the transform's three rows, lane 12 the stop flag (lane 0's stores above are this
      // wavefront's own: ordered before these loads)
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      unsigned int word = 0;
      if (lane < 12) word = __float_as_uint(ld_state<true>(&st->T[lane]));
      else if (lane == 12) word = (unsigned int)ld_state<true>(&st->done);
      if (lane < 13) __hip_atomic_store((gl_u64p)(pub + lane), tg | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // wait for the publication: the workgroup's LAST wavefront polls (the first one of workgroup 0 is the publisher)
  // and hands the words to the others through LDS - a quarter of the polling traffic of every wavefront for itself
  __shared__ unsigned int s_pub[16];
  if (tid >= 64 * (ICP_WPB - 1)) {
    unsigned long long w = tg;
    const long long t0 = wall_clock64();
    unsigned int spins = 0;
    bool seen = true;
    for (;;) {
      if (lane < 13) w = ld_granule(pub + lane);
      if (__all((unsigned int)(w >> 32) == tag)) break;
      if ((++spins & 63u) == 0u && __any((unsigned long long)(wall_clock64() - t0) > 2ull * limit)) {
        seen = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane < 13) s_pub[lane] = (unsigned int)w;
    if (lane == 13) s_pub[13] = seen ? 1u : 0u;
  }
  __syncthreads();
  if (!s_pub[13]) {  // the publication never came (workgroup 0 not on the chip?): say so, the host must not go on
    if (tid == 0) {
      st_state<true>(&tail.st->status, (int)PCLMI_ERR_HIP);
      st_state<true>(&tail.st->done, 1);
    }
    return false;
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) T.m[j] = __uint_as_float(s_pub[j]);
  const bool go_on = s_pub[12] == 0u;
  __syncthreads();  // s_pub is read before the next iteration's polling wave overwrites it
  return go_on;
}

// One ICP iteration over the (Morton-ordered) source points.  See file header.
// LANE: per-lane traversal (traverse.hpp lane_nearest) with `levels` rows of LDS per wave, started from the
// distance to the previous step's neighbour (nn_slot, when have_prev) below the wave's common subtree;
// !LANE: wave-uniform traversal (development switch PCLMI_NN1=union).
// EST = 1 (TransformationEstimationPointToPlaneLLS, impl/transformation_estimation_point_to_plane_lls.hpp:141-245):
// instead of the moments of the SVD estimator every pair adds its row of the linearised point-to-plane system to the
// 21 + 6 sums of the 6x6 normal equations; tn_slot holds the target normals in bucket-slot order (next to the point a
// lane has just found), src_skip (nullable) flags source points whose own normal is not finite (:158-174 skips them).
// RS: the step of the RANSAC loop (pclmi_icp_align with ransac_iterations > 0): also leaves keep_slot[slot] = "this
// point has a pair" for the passes of the rejection step and adds the paired source points' second moments.
// LOOP: the RESIDENT loop - up to loop_iters iterations in ONE launch (see icp_loop_tail): a shard of <= ~0.25M points
// pays no launch boundary (8-15 us) per iteration.  The host checks the grid against the occupancy (icp_loop_fits);
// with every workgroup resident the occupancy that hides the walk's latency elsewhere is not needed: 128 registers, no
// spills on the walk and in the solve.
template <bool LANE, int EST, bool RS, bool SC, bool LOOP>
__global__ __launch_bounds__(64 * ICP_WPB) __attribute__((amdgpu_waves_per_eu(LOOP ? 4 : 8, LOOP ? 4 : 8))) void icp_step_kernel(TreeView t, float4 *__restrict__ pts, int n,
                                                                 int apply, Xf T, float thr2f, double thr2,
                                                                 int *__restrict__ nn_idx, int *__restrict__ nn_slot,
                                                                 float *__restrict__ dist, double ox, double oy,
                                                                 double oz, double *__restrict__ partials,
                                                                 const IcpDevState *__restrict__ st, int levels,
                                                                 int have_prev, IcpTail tail,
                                                                 const float4 *__restrict__ tn_slot,
                                                                 const unsigned char *__restrict__ src_skip,
                                                                 unsigned char *__restrict__ keep_slot,
                                                                 float4 *__restrict__ tq,
                                                                 const float4 *__restrict__ raw_tgt, float ax, float ay,
                                                                 float az, int loop_iters) {
  static_assert(LANE || !EST, "the point-to-plane loop reads the target normals by bucket slot");
  static_assert(!SC || !EST, "the rescaled search is the SVD estimator's");
  static_assert(!LOOP || (LANE && !RS && !SC), "the resident loop: per-lane walk, no RANSAC step, no rescaled search");
  extern __shared__ float far_lds[];
  __shared__ double lds[ICP_WPB][SumsOf<EST, RS>::n];
  if (st) {  // device-resident loop: the pending transform and the stop flag live in device memory
    if (st->done || (RS && st->hold)) return;
    apply = st->apply;
#pragma unroll
    for (int j = 0; j < 12; ++j) T.m[j] = st->T[j];
  }
  // LOOP: iterations completed when this launch began (written by an earlier launch: every workgroup reads the same)
  const unsigned int epoch0 = LOOP ? st->ticket : 0u;
  for (int it = 0; it < (LOOP ? loop_iters : 1); ++it) {
  if (LOOP && it > 0) {
    apply = 1;
    have_prev = 1;
  }
  if (tail.st && blockIdx.x == 0 && threadIdx.x == 0) {
    tail.st->ticks[0] = (unsigned long long)wall_clock64();
#ifdef PCLMI_LOOP_TICKS
    tail.st->walk[it & 7] = tail.st->ticks[0];
#endif
  }
#ifdef PCLMI_ICP_TICKS
  const bool stamp = tail.st && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
#define PCLMI_WALK_STAMP(i)                                                   \
  do {                                                                        \
    __builtin_amdgcn_s_waitcnt(0);                                            \
    if (stamp) tail.st->walk[i] = (unsigned long long)wall_clock64();         \
  } while (0)
#else
#define PCLMI_WALK_STAMP(i) \
  do {                      \
  } while (0)
#endif
  PCLMI_WALK_STAMP(0);
#ifdef PCLMI_ICP_TICKS
  const unsigned long long wave_t0 = (unsigned long long)wall_clock64();
#endif
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int per = (gridDim.x + 7) >> 3;
  const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);  // XCD-contiguous chunks (speed only)
  const int slot = (blk * ICP_WPB + wib) * 64 + lane;
  const bool wave_live = __builtin_amdgcn_readfirstlane(slot) < n;
  const float4 *__restrict__ rows = reinterpret_cast<const float4 *>(t.slot);  // bucket order (x, y, z, idx bits)
  bool paired = false, counted = false;
  float4 p = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), 1.f);
  float4 q = p;
  float4 m = p;  // EST: the target normal of the pair
  bool contrib = false;
  float dc = 0.f, prev = 0.f;
  if (wave_live) {
    const bool alive = slot < n;
    bool moved = false;
    int j_prev = -1;      // the three per-point loads go out together: one round trip instead of three in a row
    float d_prev = 0.f;
    if (alive) {
      if (have_prev) j_prev = nn_slot[slot];
      d_prev = dist[slot];
      p = pts[slot];
      if (apply && finite3(p)) {
        p = apply_xf(T, p);
        moved = true;  // stored with the results below: a store up here sits in the in-order memory counter in front
      }                //  of every load of the search, whose first wait would then also wait for the write's round trip
    }
    const bool searchable = alive && finite3(p);
    if (!searchable) p.x = p.y = p.z = __builtin_nanf("");
    // the search runs in the tree's point representation (a per-dimension rescale of x, y, z: Registration::
    // setPointRepresentation, registration.h:362-370 -> kdtree.h:204-218 vectorises the query); the estimator below
    // reads the points themselves.  !SC: the representation is the identity (its own instantiation: the plain step
    // keeps its registers).
    float vx = p.x, vy = p.y, vz = p.z;
    if (SC) {
      vx = __fmul_rn(p.x, ax);
      vy = __fmul_rn(p.y, ay);
      vz = __fmul_rn(p.z, az);
    }
    PCLMI_WALK_STAMP(1);
    unsigned long long best = make_key(3.402823466e+38f, 0x7fffffff);
    int bslot = 0;
    float tau0 = __builtin_inff();
    int start_leaf = 0;  // heap id of the previous neighbour's bucket: where the per-lane walk starts
    // certified: the point is still closer to its previous neighbour t_j than 0.49 x the distance from t_j to ITS
    // nearest other point - then t_j is the nearest neighbour, strictly - or than 0.49 x the distance to its SECOND
    // nearest other point - then it is t_j or that nearest other point (TreeView::cert): nothing to walk for.  A
    // converged iteration on a cloud whose noise is small against its spacing certifies nearly every point.
    bool certified = false;
    int prev_j = -1;
    if (have_prev && searchable) {
      const int j = j_prev;
      if (j >= 0) {  // the previous neighbour is an indexed point: its distance bounds the new nearest one
        const float4 o = rows[j];
        const float4 c = t.cert ? reinterpret_cast<const float4 *>(t.cert)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        tau0 = sq3(vx - o.x, vy - o.y, vz - o.z);
        start_leaf = t.M + j / L;
        if (tau0 < (t.cert_levels >= 2 ? c.y : c.x)) {
          certified = true;
          prev_j = j;
          best = make_key(tau0, __float_as_int(o.w));  // the key the walk would return: same fp32 expression
          bslot = j;
          if (!(tau0 < c.x)) {  // the nearest other point may have come closer
            const int k = __float_as_int(c.z);
            const float4 o2 = rows[k];
            const unsigned long long key2 = make_key(sq3(vx - o2.x, vy - o2.y, vz - o2.z), __float_as_int(o2.w));
            if (key2 < best) {
              best = key2;
              bslot = k;
            }
          }
        }
      }
    }
    PCLMI_WALK_STAMP(2);
    const bool walks = searchable && !certified;
    if (__any(walks)) {
      const int root = have_prev ? wave_common_root(t, vx, vy, vz, walks, tau0) : 1;
      PCLMI_WALK_STAMP(3);
      unsigned long long wbest;
      int wslot = 0;
      if (LANE) {
        wbest = lane_nearest(t, vx, vy, vz, walks, tau0,
                             reinterpret_cast<float(*)[64]>(far_lds + (size_t)wib * levels * 64), lane, root, wslot,
                             start_leaf);
      } else {
        // wave-uniform walk below the common root; lanes without a previous neighbour (first step, or the point was
        // not searchable then) have an infinite bound, which only keeps the root at 1
        wbest = wave_nearest(t, vx, vy, vz, walks, lane, have_prev != 0, tau0, root, &wslot);
      }
      if (!certified) {
        best = wbest;
        bslot = wslot;
      }
    }
    PCLMI_WALK_STAMP(4);
    if (alive) {
      if (moved) pts[slot] = p;
      const int idx = searchable ? (int)(unsigned int)(best & 0xffffffffull) : -1;
      const float d = searchable ? __uint_as_float((unsigned int)(best >> 32)) : __builtin_nanf("");
      if (!(certified && bslot == prev_j)) {  // (a certified point that kept its neighbour has both in place)
        nn_idx[slot] = idx;
        nn_slot[slot] = searchable ? bslot : -1;
      }
      if (RS) keep_slot[slot] = (searchable && (double)d < thr2) ? 1 : 0;
      if (searchable) {
        dc = fminf(d, thr2f);  // icp.hpp:101
        prev = d_prev;
        dist[slot] = dc;
        counted = true;
        if ((double)d < thr2) {  // strict, icp.hpp:93
          q = SC ? raw_tgt[idx] : rows[bslot];
          paired = true;
          if (RS) tq[slot] = q;  // the rejection step's passes read the matched point in storage order
          if (EST) {
            m = tn_slot[bslot];
            contrib = finite3(m) && !(src_skip && src_skip[slot]);
          }
        }
      }
    }
  }
  PCLMI_WALK_STAMP(5);
#ifdef PCLMI_ICP_TICKS
  if (tail.st && (threadIdx.x & 63) == 0) {  // latest wave start and longest wave, over the launch ([6], [7]: the
    __builtin_amdgcn_s_waitcnt(0);           //  host clears them between launches)
    const unsigned long long now = (unsigned long long)wall_clock64();
    atomicMax(&tail.st->walk[6], wave_t0);
    atomicMax(&tail.st->walk[7], now - wave_t0);
  }
#endif
  if constexpr (EST != 0) {
    // fp32 expressions widened on assignment, as the reference writes them (:186-188, :219)
    const double a = (double)__fsub_rn(__fmul_rn(m.z, p.y), __fmul_rn(m.y, p.z));
    const double b = (double)__fsub_rn(__fmul_rn(m.x, p.z), __fmul_rn(m.z, p.x));
    const double c = (double)__fsub_rn(__fmul_rn(m.y, p.x), __fmul_rn(m.x, p.y));
    const double nx = m.x, ny = m.y, nz = m.z;
    float df = __fadd_rn(__fadd_rn(__fmul_rn(m.x, q.x), __fmul_rn(m.y, q.y)), __fmul_rn(m.z, q.z));
    df = __fsub_rn(__fsub_rn(__fsub_rn(df, __fmul_rn(m.x, p.x)), __fmul_rn(m.y, p.y)), __fmul_rn(m.z, p.z));
    const double dd = df;
    {
      double v[8] = {a * a, a * b, a * c, a * nx, a * ny, a * nz, b * b, b * c};
      wave_reduce_to_row(v, contrib, &lds[wib][0], lane);
    }
    {
      double v[8] = {b * nx, b * ny, b * nz, c * c, c * nx, c * ny, c * nz, nx * nx};
      wave_reduce_to_row(v, contrib, &lds[wib][8], lane);
    }
    {
      double v[8] = {nx * ny, nx * nz, ny * ny, ny * nz, nz * nz, a * dd, b * dd, c * dd};
      wave_reduce_to_row(v, contrib, &lds[wib][16], lane);
    }
    const double sx = (double)p.x - ox, sy = (double)p.y - oy, sz = (double)p.z - oz;  // RS only
    {
      double v[8] = {contrib ? nx * dd : 0.0, contrib ? ny * dd : 0.0, contrib ? nz * dd : 0.0, paired ? 1.0 : 0.0,
                     (double)dc,              (double)dc - (double)prev,  1.0,                      (RS && paired) ? sx : 0.0};
      wave_reduce_to_row(v, counted, &lds[wib][24], lane);
    }
    if constexpr (RS) {
      double v[8] = {sy, sz, sx * sx, sx * sy, sx * sz, sy * sy, sy * sz, sz * sz};
      wave_reduce_to_row(v, paired, &lds[wib][32], lane);
    }
    __syncthreads();
    if constexpr (LOOP) {
      if (!icp_loop_tail<1>(tail, lds, epoch0 + (unsigned int)it + 1u, T)) return;
    } else {
      icp_step_tail<1, RS>(tail, partials, lds);
    }
  } else {
  // fp64 moments, reduced in three groups of eight so that at most 16 accumulator registers are live at a
  // time (the register budget is 64 VGPRs = 8 waves per SIMD: the search is latency-bound)
  const double sx = (double)p.x - ox, sy = (double)p.y - oy, sz = (double)p.z - oz;
  const double tx = (double)q.x - ox, ty = (double)q.y - oy, tz = (double)q.z - oz;
  const double one_p = paired ? 1.0 : 0.0, one_c = counted ? 1.0 : 0.0;
  {
    double v[8] = {one_p, sx, sy, sz, tx, ty, tz, sx * tx};
    wave_reduce_to_row(v, paired, &lds[wib][0], lane);
  }
  {
    double v[8] = {sx * ty, sx * tz, sy * tx, sy * ty, sy * tz, sz * tx, sz * ty, sz * tz};
    wave_reduce_to_row(v, paired, &lds[wib][8], lane);
  }
  {
    double v[8] = {(double)dc, (double)dc - (double)prev, one_c, ox, oy, oz, 0.0, 0.0};
    wave_reduce_to_row(v, counted, &lds[wib][16], lane);
  }
  if constexpr (RS) {
    double v[8] = {sx * sx, sx * sy, sx * sz, sy * sy, sy * sz, sz * sz, 0.0, 0.0};
    wave_reduce_to_row(v, paired, &lds[wib][24], lane);
  }
  __syncthreads();
  if constexpr (LOOP) {
    if (!icp_loop_tail<0>(tail, lds, epoch0 + (unsigned int)it + 1u, T)) return;
  } else {
    icp_step_tail<0, RS>(tail, partials, lds);
  }
  }
  }  // iterations of the launch
}

// fixed-order sum of one column of block partials by one workgroup of 256: element r of column j sits at
// partials[r * row_stride + j * col_stride]
__device__ __forceinline__ double column_sum(const double *__restrict__ partials, int nrows, size_t row_stride,
                                             size_t col_stride, int j, double *sh) {
  double s = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 256) s += partials[(size_t)r * row_stride + (size_t)j * col_stride];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  return sh[0];
}

// block j of the grid produces out[j]
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ partials, int nrows,
                                                              size_t row_stride, size_t col_stride,
                                                              double *__restrict__ out) {
  __shared__ double sh[256];
  const int j = blockIdx.x;
  const double s = column_sum(partials, nrows, row_stride, col_stride, j, sh);
  if (threadIdx.x == 0) out[j] = s;
}

template <int EST>
__global__ __launch_bounds__(64) void icp_solve_kernel(const double *__restrict__ sums, IcpDevState *__restrict__ st) {
  __shared__ double work[42];
  __shared__ SolveIn in;
  if (st->done) return;  // uniform
  fetch_solve_in(st, &in, threadIdx.x);
  __syncthreads();
  if (threadIdx.x != 0) return;
  icp_solve_body<EST>(sums, st, work, &in);
}

__global__ void transform_pending_kernel(float4 *__restrict__ pts, int n, const IcpDevState *__restrict__ st) {
  if (!st->apply) return;
  Xf T;
#pragma unroll
  for (int j = 0; j < 12; ++j) T.m[j] = st->T[j];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  if (finite3(p)) pts[i] = apply_xf(T, p);
}

__global__ void clear_apply_kernel(IcpDevState *st) { st->apply = 0; }

// sums over explicitly paired points (TransformationEstimationSVD on a correspondence list,
// impl/transformation_estimation_svd.hpp:101-111; also the RANSAC inlier refit)
constexpr int PR_WPB = 4;
__global__ __launch_bounds__(64 * PR_WPB) void pair_reduce_kernel(const float4 *__restrict__ src,
                                                                   const int *__restrict__ sidx,
                                                                   const float4 *__restrict__ tgt,
                                                                   const int *__restrict__ tidx,
                                                                   const unsigned char *__restrict__ flags, int n,
                                                                   double ox, double oy, double oz,
                                                                   double *__restrict__ partials) {
  __shared__ double lds[PR_WPB][NS];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  double acc[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) acc[j] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (flags && !flags[i]) continue;
    const float4 p = src[sidx ? sidx[i] : i];
    const float4 q = tgt[tidx ? tidx[i] : i];
    const double sx = (double)p.x - ox, sy = (double)p.y - oy, sz = (double)p.z - oz;
    const double tx = (double)q.x - ox, ty = (double)q.y - oy, tz = (double)q.z - oz;
    acc[0] += 1.0;
    acc[1] += sx; acc[2] += sy; acc[3] += sz;
    acc[4] += tx; acc[5] += ty; acc[6] += tz;
    acc[7] += sx * tx; acc[8] += sx * ty; acc[9] += sx * tz;
    acc[10] += sy * tx; acc[11] += sy * ty; acc[12] += sy * tz;
    acc[13] += sz * tx; acc[14] += sz * ty; acc[15] += sz * tz;
    acc[18] += 1.0;
    acc[19] += ox; acc[20] += oy; acc[21] += oz;
  }
  block_reduce_store<NS, PR_WPB>(acc, lds, lane, wib, partials + (size_t)blockIdx.x * NS);
}

// the 21 + 6 sums of the point-to-plane normal equations over explicitly paired points (the RANSAC inlier refit of the
// point-to-plane loop): src by slot, tgt and its normals by cloud row; layout as the step kernel's (pclmi.h)
__global__ __launch_bounds__(64 * PR_WPB) void pair_reduce_plane_kernel(const float4 *__restrict__ src,
                                                                         const int *__restrict__ sidx,
                                                                         const float4 *__restrict__ tgt,
                                                                         const int *__restrict__ tidx,
                                                                         const float4 *__restrict__ tn,
                                                                         const unsigned char *__restrict__ src_skip,
                                                                         const unsigned char *__restrict__ flags, int n,
                                                                         double *__restrict__ partials) {
  __shared__ double lds[PR_WPB][NSP];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  double acc[NSP];
#pragma unroll
  for (int j = 0; j < NSP; ++j) acc[j] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (flags && !flags[i]) continue;
    const int si = sidx ? sidx[i] : i, ti = tidx ? tidx[i] : i;
    acc[27] += 1.0;
    const float4 p = src[si], q = tgt[ti], m = tn[ti];
    if (!finite3(p) || !finite3(q) || !finite3(m) || (src_skip && src_skip[si])) continue;
    const double a = (double)__fsub_rn(__fmul_rn(m.z, p.y), __fmul_rn(m.y, p.z));
    const double b = (double)__fsub_rn(__fmul_rn(m.x, p.z), __fmul_rn(m.z, p.x));
    const double c = (double)__fsub_rn(__fmul_rn(m.y, p.x), __fmul_rn(m.x, p.y));
    float df = __fadd_rn(__fadd_rn(__fmul_rn(m.x, q.x), __fmul_rn(m.y, q.y)), __fmul_rn(m.z, q.z));
    df = __fsub_rn(__fsub_rn(__fsub_rn(df, __fmul_rn(m.x, p.x)), __fmul_rn(m.y, p.y)), __fmul_rn(m.z, p.z));
    const double v[6] = {a, b, c, (double)m.x, (double)m.y, (double)m.z};
    const double dd = df;
    int o = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int k = r; k < 6; ++k) acc[o++] += v[r] * v[k];
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[21 + r] += v[r] * dd;
  }
  block_reduce_store<NSP, PR_WPB>(acc, lds, lane, wib, partials + (size_t)blockIdx.x * NSP);
}

// target normals (cloud rows) -> bucket-slot order: the row a lane's search ends at sits next to its normal
__global__ void normals_to_slots_kernel(const float4 *__restrict__ slot_rows, int nslots,
                                        const float4 *__restrict__ tn_rows, float4 *__restrict__ tn_slot) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nslots) return;
  const int idx = __float_as_int(slot_rows[j].w);
  const float nanv = __builtin_nanf("");
  tn_slot[j] = idx >= 0 ? tn_rows[idx] : make_float4(nanv, nanv, nanv, 0.f);
}

// the same sums over a correspondence list with the clouds in their caller layout (strided rows):
// pairs (src[index_query], tgt[index_match]); moments about the first pair's target point
__global__ __launch_bounds__(64 * PR_WPB) void corr_reduce_kernel(const float *__restrict__ src, int src_stride_f,
                                                                   const float *__restrict__ tgt, int tgt_stride_f,
                                                                   const pclmi_correspondence *__restrict__ corr, int n,
                                                                   double *__restrict__ partials) {
  __shared__ double lds[PR_WPB][NS];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const float *t0 = tgt + (size_t)corr[0].index_match * tgt_stride_f;
  const double ox = t0[0], oy = t0[1], oz = t0[2];
  double acc[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) acc[j] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const pclmi_correspondence c = corr[i];
    const float *p = src + (size_t)c.index_query * src_stride_f;
    const float *q = tgt + (size_t)c.index_match * tgt_stride_f;
    const double sx = (double)p[0] - ox, sy = (double)p[1] - oy, sz = (double)p[2] - oz;
    const double tx = (double)q[0] - ox, ty = (double)q[1] - oy, tz = (double)q[2] - oz;
    acc[0] += 1.0;
    acc[1] += sx; acc[2] += sy; acc[3] += sz;
    acc[4] += tx; acc[5] += ty; acc[6] += tz;
    acc[7] += sx * tx; acc[8] += sx * ty; acc[9] += sx * tz;
    acc[10] += sy * tx; acc[11] += sy * ty; acc[12] += sy * tz;
    acc[13] += sz * tx; acc[14] += sz * ty; acc[15] += sz * tz;
    acc[18] += 1.0;
    acc[19] += ox; acc[20] += oy; acc[21] += oz;
  }
  block_reduce_store<NS, PR_WPB>(acc, lds, lane, wib, partials + (size_t)blockIdx.x * NS);
}

// RANSAC support -------------------------------------------------------------------------------
// countWithinDistance / selectWithinDistance (sample_consensus/.../impl/sac_model_registration.hpp
// :131-224): pairs with |T s - t|^2 < thr2 in fp32.  flags (nullable) receives 0/1 per pair.
__global__ __launch_bounds__(256) void count_within_kernel(const float4 *__restrict__ src,
                                                           const int *__restrict__ sidx,
                                                           const float4 *__restrict__ tgt,
                                                           const int *__restrict__ tidx, int n, Xf T, double thr2,
                                                           unsigned int *__restrict__ count,
                                                           unsigned char *__restrict__ flags,
                                                           const unsigned char *__restrict__ keep) {
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool in = false;
    if (!keep || keep[i]) {  // sidx == nullptr: pair i is (src[i], tgt[tidx[i]]); keep: pairs that exist
      const float4 p = apply_xf(T, src[sidx ? sidx[i] : i]);
      const float4 q = tgt[tidx[i]];
      const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      in = (double)d < thr2;
    }
    c += in ? 1 : 0;
    if (flags) flags[i] = in ? 1 : 0;
  }
  // one atomic per workgroup: with every pair an inlier (a converged ICP iteration) an atomic per wavefront meant
  // 16 000 adds to one address, 150 of the kernel's 187 us
  __shared__ int part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = part[0] + part[1] + part[2] + part[3];
    if (tot) atomicAdd(count, (unsigned int)tot);
  }
}

// ---- RANSAC trials in batches ------------------------------------------------------------------------------------
// The reference draws a sample, fits it, counts its inliers and only then draws the next (ransac.hpp:71-121) - but the
// draws never depend on the counts (only WHEN to stop does).  So the sampling loop runs here, on the device, where the
// points are: lane 0 replays the generator and the index shuffle (drawIndexSample, sac_model.h:401-412) and checks
// isSampleGood (sac_model_registration.hpp:47-60) for up to kRansacBatch trials, the lanes then fit the three-point
// models in parallel (the same rigid_solve.hpp code as the host's, same bits), and ONE pass over the pairs counts the
// inliers of every model of the batch.  The host consumes the trials in order and applies the reference's stopping
// rule; what it discards beyond the stop was drawn from a generator that is re-seeded on every call anyway.
// sample_dist_thresh_ = (mean of the square roots of the eigenvalues of the source points' covariance)^2
// (computeSampleDistanceThreshold, sac_model_registration.h:257-279; cyclic Jacobi stands in for pcl::eigen33).
// m: the sums of moment_reduce_kernel (second moments 0..5, first moments 6..8, count 9).
// one Jacobi rotation of the symmetric 3x3 (c00 c01 c02; . c11 c12; . . c22) that zeroes element (P, Q); R is the
// third index.  Compile-time indices: everything stays in registers.
template <int P, int Q>
__device__ __forceinline__ void sym3_rotate(double (&d)[3], double (&o)[3]) {
  // o[0] = c01, o[1] = c02, o[2] = c12; element (p, q) lives at o[p + q - 1]
  constexpr int R = 3 - P - Q;
  constexpr int ipq = P + Q - 1, ipr = P + R - 1, iqr = Q + R - 1;
  const double apq = o[ipq];
  if (fabs(apq) < 1e-300) return;
  const double theta = (d[Q] - d[P]) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
  d[P] -= t * apq;
  d[Q] += t * apq;
  o[ipq] = 0.0;
  const double arp = o[ipr], arq = o[iqr];
  o[ipr] = cs * arp - sn * arq;
  o[iqr] = sn * arp + cs * arq;
}

__device__ __forceinline__ double sample_threshold_from_moments(const double *m) {
  const double inv_n = 1.0 / m[9];
  double a[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = m[i] * inv_n;
  double d[3] = {a[0] - a[6] * a[6], a[3] - a[7] * a[7], a[5] - a[8] * a[8]};
  double o[3] = {a[1] - a[6] * a[7], a[2] - a[6] * a[8], a[4] - a[7] * a[8]};
  for (int sweep = 0; sweep < 60; ++sweep) {
    // converged: the off-diagonal part is below the last bit of the diagonal, further rotations change nothing
    const double off = fabs(o[0]) + fabs(o[1]) + fabs(o[2]);
    if (off <= 1e-22 * (fabs(d[0]) + fabs(d[1]) + fabs(d[2])) || off < 1e-300) break;
    sym3_rotate<0, 1>(d, o);
    sym3_rotate<0, 2>(d, o);
    sym3_rotate<1, 2>(d, o);
  }
  double sthr = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) sthr += sqrt(d[i] > 0 ? d[i] : 0.0);
  sthr /= 3.0;
  return sthr * sthr;
}

__global__ void ransac_sample_threshold_kernel(const double *__restrict__ m, double *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  out[0] = sample_threshold_from_moments(m);
}

__device__ __forceinline__ unsigned int mt_next(unsigned int *mt, int &idx) {  // one lane; mt in LDS
  if (idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const unsigned int y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
      mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    idx = 0;
  }
  unsigned int y = mt[idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// loads / stores of the shuffle array that never see a stale line of the vector L1 (a later draw reads what an
// earlier one stored)
__device__ __forceinline__ int shuf_ld(const int *p) {
  return __hip_atomic_load((__attribute__((address_space(1))) const int *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void shuf_st(int *p, int v) {
  __hip_atomic_store((__attribute__((address_space(1))) int *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rng_in: the generator state to continue from - the freshly seeded one (kept per RansacDev: the seed never changes) for
// the first batch of a call, rng itself afterwards
__global__ __launch_bounds__(64) void ransac_draw_kernel(const RansacRng *__restrict__ rng_in, RansacRng *__restrict__ rng,
                                                         int *__restrict__ shuf, int cnt,
                                                         const float4 *__restrict__ src, const int *__restrict__ sidx,
                                                         const float4 *__restrict__ tgt, const int *__restrict__ tidx,
                                                         const double *__restrict__ sthr_p, int batch,
                                                         RansacTrial *__restrict__ trials) {
  __shared__ unsigned int mt[624];
  __shared__ float4 six[kRansacBatch][6];
  __shared__ int status[kRansacBatch];
  __shared__ double work[kRansacBatch][42];
  const int lane = threadIdx.x;
  for (int i = lane; i < 624; i += 64) mt[i] = rng_in->mt[i];
  if (lane < kRansacBatch) status[lane] = 2;
  __syncthreads();
  if (lane == 0) {
    const double sthr = sthr_p[0];
    int idx = rng_in->idx;
    int head[3] = {shuf_ld(shuf + 0), shuf_ld(shuf + 1), shuf_ld(shuf + 2)};  // positions 0..2 live in registers
    bool dead = false;
    for (int b = 0; b < batch && !dead; ++b) {
      bool good = false;
      for (unsigned int iter = 0; iter < 1000u && !good; ++iter) {  // max_sample_checks_ (sac_model.h:470)
        // the three swaps of drawIndexSample: shuf[i] <-> shuf[i + rnd() % (cnt - i)], i = 0, 1, 2; the far ends are
        // loaded together and the swaps replayed on the six entries (0, 1, 2, j0, j1, j2; equal positions share one)
        int pos[6] = {0, 1, 2, 0, 0, 0}, val[6], canon[6] = {0, 1, 2, 3, 4, 5};
        for (int i = 0; i < 3; ++i) {
          const unsigned int r = mt_next(mt, idx) >> 1;
          pos[3 + i] = i + (int)(r % (unsigned int)(cnt - i));
        }
        for (int i = 0; i < 3; ++i) val[i] = head[i];
        for (int i = 3; i < 6; ++i) val[i] = pos[i] >= 3 ? shuf_ld(shuf + pos[i]) : 0;
        for (int i = 3; i < 6; ++i)
          for (int e = 0; e < i; ++e)
            if (pos[e] == pos[i]) {
              canon[i] = canon[e];
              break;
            }
        for (int i = 0; i < 3; ++i) {
          const int a = i, c = canon[3 + i];
          const int t = val[a];
          val[a] = val[c];
          val[c] = t;
        }
        for (int i = 3; i < 6; ++i)
          if (canon[i] == i && pos[i] >= 3) shuf_st(shuf + pos[i], val[i]);
        for (int i = 0; i < 3; ++i) head[i] = val[i];
        float4 sp[3], tp[3];
        for (int i = 0; i < 3; ++i) {
          sp[i] = src[sidx[head[i]]];
          tp[i] = tgt[tidx[head[i]]];
        }
        auto sq = [](const float4 &u, const float4 &v) {
          const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z;
          return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        };
        good = (double)sq(sp[1], sp[0]) > sthr && (double)sq(sp[2], sp[0]) > sthr && (double)sq(sp[2], sp[1]) > sthr;
        if (good) {
          for (int i = 0; i < 3; ++i) {
            six[b][2 * i] = sp[i];
            six[b][2 * i + 1] = tp[i];
          }
          status[b] = 0;
        }
      }
      dead = !good;  // ransac.hpp:72-76: computeModel gives up (this and every later trial keep status 2)
    }
    for (int i = 0; i < 3; ++i) shuf_st(shuf + i, head[i]);
    rng->idx = idx;
  }
  __syncthreads();
  for (int i = lane; i < 624; i += 64) rng->mt[i] = mt[i];
  if (lane < batch) {
    RansacTrial tr;
    for (int i = 0; i < 16; ++i) tr.T[i] = (i % 5 == 0) ? 1.f : 0.f;
    tr.status = status[lane];
    tr.count = 0u;
    tr.pad[0] = tr.pad[1] = 0;
    if (tr.status == 0) {
      // three-point rigid fit (estimateRigidTransformationSVD, sac_model_registration.hpp:257-302): the sums the
      // host used to build (api_icp.hip ransac_core), in the same order
      const float4 *h6 = six[lane];
      double sm[PCLMI_ICP_NSUMS];
      for (int i = 0; i < PCLMI_ICP_NSUMS; ++i) sm[i] = 0.0;
      const double o[3] = {h6[1].x, h6[1].y, h6[1].z};
      for (int i = 0; i < 3; ++i) {
        const double sx = h6[2 * i].x - o[0], sy = h6[2 * i].y - o[1], sz = h6[2 * i].z - o[2];
        const double tx = h6[2 * i + 1].x - o[0], ty = h6[2 * i + 1].y - o[1], tz = h6[2 * i + 1].z - o[2];
        sm[0] += 1;
        sm[1] += sx; sm[2] += sy; sm[3] += sz;
        sm[4] += tx; sm[5] += ty; sm[6] += tz;
        sm[7] += sx * tx; sm[8] += sx * ty; sm[9] += sx * tz;
        sm[10] += sy * tx; sm[11] += sy * ty; sm[12] += sy * tz;
        sm[13] += sz * tx; sm[14] += sz * ty; sm[15] += sz * tz;
        sm[18] += 1;
        sm[19] += o[0]; sm[20] += o[1]; sm[21] += o[2];
      }
      if (!rs::solve_from_sums(sm, tr.T, work[lane])) tr.status = 1;
    }
    trials[lane] = tr;
  }
}

__global__ __launch_bounds__(256) void count_within_batch_kernel(const float4 *__restrict__ src,
                                                                 const int *__restrict__ sidx,
                                                                 const float4 *__restrict__ tgt,
                                                                 const int *__restrict__ tidx, int n, double thr2,
                                                                 int batch, RansacTrial *__restrict__ trials,
                                                                 const unsigned char *__restrict__ keep) {
  __shared__ Xf xf[kRansacBatch];
  __shared__ int ready[kRansacBatch];
  __shared__ unsigned int cnt[kRansacBatch];
  if ((int)threadIdx.x < batch) {
    const RansacTrial &tr = trials[threadIdx.x];
    for (int i = 0; i < 12; ++i) xf[threadIdx.x].m[i] = tr.T[i];
    ready[threadIdx.x] = tr.status == 0;
    cnt[threadIdx.x] = 0u;
  }
  __syncthreads();
  unsigned int mine = 0u;  // lane b of each wave keeps the wave's running count for model b (and b + 64 ... none: batch <= 64)
  const int lane = threadIdx.x & 63;
  for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + (int)threadIdx.x;
    const bool live = i < n && (!keep || keep[i]);
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
    if (live) {
      p = src[sidx ? sidx[i] : i];
      q = tgt[tidx[i]];
    }
    for (int b = 0; b < batch; ++b) {
      if (!ready[b]) continue;  // block-uniform
      const float4 m = apply_xf(xf[b], p);
      const float dx = m.x - q.x, dy = m.y - q.y, dz = m.z - q.z;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      const unsigned int c = (unsigned int)__popcll(__ballot(live && (double)d < thr2));
      if (lane == b) mine += c;
    }
  }
  if (lane < batch && mine) atomicAdd(&cnt[lane], mine);
  __syncthreads();
  if ((int)threadIdx.x < batch && cnt[threadIdx.x]) atomicAdd(&trials[threadIdx.x].count, cnt[threadIdx.x]);
}

// fp32 second moments of the source points of the pair list (computeMeanAndCovarianceMatrix,
// common/include/pcl/common/impl/centroid.hpp:553-617), accumulated in fp64
// sidx == nullptr: entry i is src[i]; keep (nullable): entries whose byte is 0 do not exist
__global__ __launch_bounds__(64 * PR_WPB) void moment_reduce_kernel(const float4 *__restrict__ src,
                                                                     const int *__restrict__ sidx, int n,
                                                                     const unsigned char *__restrict__ keep,
                                                                     double *__restrict__ partials) {
  __shared__ double lds[PR_WPB][NS];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  double acc[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) acc[j] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (keep && !keep[i]) continue;
    const float4 p = src[sidx ? sidx[i] : i];
    const double x = p.x, y = p.y, z = p.z;
    acc[0] += x * x; acc[1] += x * y; acc[2] += x * z;
    acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
    acc[6] += x; acc[7] += y; acc[8] += z;
    acc[9] += 1.0;
  }
  block_reduce_store<NS, PR_WPB>(acc, lds, lane, wib, partials + (size_t)blockIdx.x * NS);
}

// correspondences of the last step in ORIGINAL source order: keep[orig] = d2 < thr2
// rows / row_of (nullable): the matched point's coordinates come from rows[row_of[i]] instead of cloud[nn_idx[i]] -
// the bucket-slot copy, where the neighbours of neighbouring source points sit next to each other
__global__ void corr_flags_kernel(const int *__restrict__ nn_idx, const float4 *__restrict__ pts,
                                  const float4 *__restrict__ cloud, const float4 *__restrict__ rows,
                                  const int *__restrict__ row_of, const int *__restrict__ order, int n,
                                  double thr2, int *__restrict__ tgt_of_orig, unsigned char *__restrict__ keep,
                                  unsigned char *__restrict__ keep_slot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = order[i];
  const int j = nn_idx[i];
  bool k = false;
  if (j >= 0) {
    const float4 p = pts[i], q = rows ? rows[row_of[i]] : cloud[j];
    const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    k = (double)d < thr2;
  }
  tgt_of_orig[o] = j;
  keep[o] = k ? 1 : 0;
  if (keep_slot) keep_slot[i] = k ? 1 : 0;  // the same mask in the order the points are stored in
}

__global__ void gather_u8_kernel(const unsigned char *__restrict__ in, const int *__restrict__ idx, int n,
                                 unsigned char *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}

__global__ void iota_kernel(int *__restrict__ v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

__global__ void gather_int_kernel(const int *__restrict__ in, const int *__restrict__ idx, int n,
                                  int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}

struct Pos3 {
  int p[3];
};
// the three sampled pairs of a RANSAC trial: out[2i] = source point, out[2i+1] = its target point
__global__ void gather_points_kernel(const float4 *__restrict__ a, const int *__restrict__ ia,
                                     const float4 *__restrict__ b, const int *__restrict__ ib, Pos3 pos,
                                     float4 *__restrict__ out) {
  int i = threadIdx.x;
  if (i < 3) {
    out[2 * i] = a[ia[pos.p[i]]];
    out[2 * i + 1] = b[ib[pos.p[i]]];
  }
}

// Registration::getFitnessScore (registration.hpp:83-103): sum / count of d2 <= max_range
__global__ __launch_bounds__(256) void fitness_kernel(const float *__restrict__ d2, int n, double max_range,
                                                      double *__restrict__ out /* [2] */) {
  double s = 0.0, c = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float d = d2[i];
    if (d == d && !((double)d > max_range)) {  // registration.hpp:94 skips d > max_range
      s += fabs((double)d);
      c += 1.0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    c += __shfl_xor(c, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], s);
    atomicAdd(&out[1], c);
  }
}

// ---- RANSAC loop: the rejection step of icp.hpp:113-150 without the host -----------------------------------------
// One ICP iteration with the reference's default ransac_iterations_ = 1000 is four launches, none of which the host
// waits for: the step (icp_step_kernel<.., RS = true>: pairs, keep mask, the step's sums incl. the source moments),
// ransac_draw_loop_kernel (the sampling stream + the three-point models of one batch), ransac_count_loop_kernel (the
// inliers of every model of the batch in one pass; its last workgroup applies the reference's stopping rule,
// ransac.hpp:71-121) and ransac_refit_kernel (selectWithinDistance with the best model fused with the refit sums of
// icp.hpp:171, then the step's tail: reduction, exchange, solve, convergence criteria).  When the first batch does
// not satisfy the stopping rule (few inliers: the first iterations of an align), the count kernel sets st->hold, every
// later launch of the speculatively enqueued chunk is a no-op, and the host enqueues further batches (first = 0).
// Positions of the pair list: the reference's list is in ORIGINAL source order; position -> row is the identity when
// every row has a pair, otherwise a rank / select over the pair bit mask by row (ransac_keep_words_kernel: one word
// per 64 rows, one count per 1024) - no compaction, no gathers over the whole list.
// shuffled_indices_ (sac_model.h) is reset for every iteration's model: entries carry the iteration's tag, an entry
// with another tag reads as its own position; what a batch changes lives in registers until its end.
constexpr int kSbRows = 1024;  // rows per count of the select structure

// the comm failed under a kernel of the RANSAC loop: the align ends with PCLMI_ERR_COMM
__device__ __forceinline__ void ransac_comm_failed(const RansacLoopArgs &a) {
  a.st->status = PCLMI_ERR_COMM;
  a.st->done = 1;
}

// local pair mask by row (rows of other ranks: 0), pushed to every peer's bulk window when sharded; nobody waits here
__global__ __launch_bounds__(kSbRows) void ransac_keep_words_kernel(RansacLoopArgs a) {
  if (a.st->done || a.st->hold) return;
  if (llround(a.sums[a.est ? 27 : 0]) == (long long)a.n_rows) return;  // every row has a pair
  const int row = blockIdx.x * kSbRows + threadIdx.x;
  bool k = false;
  if (row < a.n_rows) {
    const int sl = a.inv[row];
    k = sl >= 0 && a.keep_slot[sl] != 0;
  }
  const unsigned long long w = __ballot(k);
  __shared__ int c16[kSbRows / 64];
  if (a.comm.world == 1) {  // one rank: the block's pair count is final (sharded: ransac_keep_count_kernel)
    if ((threadIdx.x & 63) == 0) c16[threadIdx.x >> 6] = __popcll(w);
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int i = 0; i < kSbRows / 64; ++i) t += c16[i];
      a.sb_cnt[blockIdx.x] = (unsigned int)t;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    a.keep_bits[row >> 6] = w;  // nsb * 16 words are allocated
    if (a.comm.world > 1) {
      const CommView &c = a.comm;
      const unsigned int bseq = c.seq[kCtrlBulkSeq] + 1u;
      const unsigned long long tag = (unsigned long long)bseq << 32;
      const size_t wi = (size_t)(row >> 6);
      for (int p = 0; p < c.world; ++p) {
        if (p == c.rank) continue;
        unsigned long long *dst = c.bulk[p] + ((size_t)((bseq & 1u) * c.world + c.rank) * c.bulk_cap + 2 * wi);
        __hip_atomic_store((g_u64p)dst, tag | (unsigned int)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((g_u64p)(dst + 1), tag | (unsigned int)(w >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// the peers' words OR-ed in (sharded), then the pair count of every 1024 rows.  A SMALL grid on purpose: its waves
// wait for the peers' pushes, and a grid that fills the GPU would starve a peer that shares it (tests: two ranks on
// one device); 16 lanes per block of 1024 rows.
__global__ __launch_bounds__(64) void ransac_keep_count_kernel(RansacLoopArgs a) {
  __shared__ int s_last;
  if (a.st->done || a.st->hold) return;
  if (llround(a.sums[a.est ? 27 : 0]) == (long long)a.n_rows) return;
  const CommView &c = a.comm;
  const int lane = threadIdx.x;
  const unsigned int bseq = c.world > 1 ? c.seq[kCtrlBulkSeq] + 1u : 0u;
  bool ok = c.world > 1 ? !comm_is_dead(c) : true;
  for (int sb = blockIdx.x * 4 + (lane >> 4); sb - (lane >> 4) < a.nsb; sb += gridDim.x * 4) {
    int cnt = 0;
    if (sb < a.nsb) {
      const size_t wi = (size_t)sb * (kSbRows / 64) + (lane & 15);
      unsigned long long w = a.keep_bits[wi];
      if (c.world > 1 && ok) {  // the local word is pushed already (ransac_keep_words_kernel): only the wait
        const long long t0 = wall_clock64();
        for (int p = 0; p < c.world && ok; ++p) {
          if (p == c.rank) continue;
          const unsigned long long *src = c.bulk[c.rank] + ((size_t)((bseq & 1u) * c.world + p) * c.bulk_cap + 2 * wi);
          unsigned int spins = 0;
          for (;;) {
            const unsigned long long x = __hip_atomic_load((g_cu64p)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long y = __hip_atomic_load((g_cu64p)(src + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((x >> 32) == bseq && (y >> 32) == bseq) {
              w |= ((unsigned long long)(unsigned int)y << 32) | (unsigned int)x;
              break;
            }
            if ((++spins & 63u) == 0u && (unsigned long long)(wall_clock64() - t0) > c.timeout_ticks) {
              ok = false;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        a.keep_bits[wi] = w;
      }
      cnt = __popcll(w);
    }
    cnt += __shfl_xor(cnt, 8);
    cnt += __shfl_xor(cnt, 4);
    cnt += __shfl_xor(cnt, 2);
    cnt += __shfl_xor(cnt, 1);
    if (sb < a.nsb && (lane & 15) == 0) a.sb_cnt[sb] = (unsigned int)cnt;
  }
  if (c.world == 1) return;
  if (__any(!ok)) {
    if (lane == 0) {
      comm_mark_dead(c);
      ransac_comm_failed(a);
    }
  }
  __threadfence();
  if (lane == 0) {
    const unsigned int t = __hip_atomic_fetch_add((g_u32p)(c.seq + kCtrlBulkTicket), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (s_last && lane == 0) {
    __hip_atomic_store((g_u32p)(c.seq + kCtrlBulkTicket), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.seq[kCtrlBulkSeq] = bseq;
  }
}

// what one batch has changed of the shuffle array: up to 256 (position, value) entries spread over the lanes'
// registers; every operation is wave-uniform (all lanes call with the same arguments)
struct ShufMap {
  int pos[4], val[4];
  int count;
};
__device__ __forceinline__ bool map_get(const ShufMap &m, int j, int &v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned long long hit = __ballot(m.pos[e] == j);
    if (hit) {
      v = __shfl(m.val[e], __ffsll((long long)hit) - 1);
      return true;
    }
  }
  return false;
}
__device__ __forceinline__ void map_put(ShufMap &m, int j, int v, int lane) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned long long hit = __ballot(m.pos[e] == j);
    if (hit) {
      if (m.pos[e] == j) m.val[e] = v;
      return;
    }
  }
  const int e = m.count >> 6, l = m.count & 63;
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2)
    if (e2 == e && lane == l) {
      m.pos[e2] = j;
      m.val[e2] = v;
    }
  ++m.count;
}

// position in the pair list -> source row: the pos-th set bit of the pair mask
// lpre: the exclusive prefix in LDS (up to kSbLds blocks of 1024 rows), else nullptr: read from a.sb_prefix
constexpr int kSbLds = 4096;
// AG: the mask words were written by other workgroups of THIS launch (ransac_words_draw_kernel): agent-scope loads
template <bool AG = false>
__device__ __forceinline__ int select_row(const RansacLoopArgs &a, const unsigned int *lpre, int pos) {
  int lo = 0, hi = a.nsb - 1;  // last block whose prefix is <= pos
  auto pre = [&](int i) {
    return lpre ? (int)lpre[i]
                : (int)__hip_atomic_load((g_u32p)&a.sb_prefix[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pre(mid) <= pos) lo = mid;
    else hi = mid - 1;
  }
  int rem = pos - pre(lo);
  unsigned long long wv[kSbRows / 64];
  if (AG) {
#pragma unroll
    for (int i = 0; i < kSbRows / 64; ++i) wv[i] = ld_state<true>(a.keep_bits + (size_t)lo * (kSbRows / 64) + i);
  } else {
    const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(a.keep_bits + (size_t)lo * (kSbRows / 64));
#pragma unroll
    for (int i = 0; i < kSbRows / 128; ++i) {  // the block's 16 words, all loads in flight
      const ulonglong2 t = w2[i];
      wv[2 * i] = t.x;
      wv[2 * i + 1] = t.y;
    }
  }
  int wi = kSbRows / 64;
  unsigned long long bits = 0;
  bool found = false;
#pragma unroll
  for (int i = 0; i < kSbRows / 64; ++i) {
    const int c = __popcll(wv[i]);
    if (!found && rem < c) {
      found = true;
      wi = i;
      bits = wv[i];
    }
    if (!found) rem -= c;
  }
  if (!found) return -1;  // cannot happen for pos < the pair count
  for (; rem > 0; --rem) bits &= bits - 1;
  return lo * kSbRows + wi * 64 + (__ffsll((long long)bits) - 1);
}

template <bool AG>
__device__ __forceinline__ void ransac_draw_body(const RansacLoopArgs &a, int first, int draws) {
  __shared__ unsigned int mt[624];
  __shared__ int jpos[kRansacBatch][3], gval[kRansacBatch][3], dpos[kRansacBatch][3];
  __shared__ float4 six[kRansacBatch][6];
  __shared__ int trial_of[kRansacBatch];
  __shared__ double work[kRansacBatch][42];
  __shared__ double s_sthr;
  __shared__ unsigned int lpre[kSbLds];
  RansacLoopState *rl = a.rl;
  IcpDevState *st = a.st;
  const int lane = threadIdx.x;
  const int i_pairs = a.est ? 27 : 0;
  // everything this wave needs from memory before it can start: ONE round trip (the loads below do not depend on
  // each other; the early returns come after them)
  const RansacRng *rin = first ? a.rng0 : a.rng;
  unsigned int mtv[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) mtv[i] = lane + 64 * i < 624 ? rin->mt[lane + 64 * i] : 0u;
  int idx = rin->idx;
  const int st_done = st->done, st_hold = st->hold, rl_settled = rl->settled;
  const unsigned int tag_in = rl->tag;
  const int run_in = rl->bad_run, dead_in = rl->dead, cnt_in = rl->cnt;
  const double sthr_in = rl->sthr, pairs_in = a.sums[i_pairs];
  double mom_in = 0.0;  // lane i < 9: moment i of the paired source points
  {
    const int mom1 = a.est ? 31 : 1, mom2 = a.est ? 34 : 24;
    if (lane < 9) mom_in = a.sums[lane < 6 ? mom2 + lane : mom1 + (lane - 6)];
  }
  if (st_done) return;
  if (first ? st_hold != 0 : (st_hold == 0 || rl_settled != 0)) return;
  int cnt;
  double sthr;
  if (lane == 0) rl->dbg[0] = (unsigned long long)wall_clock64();
  const unsigned int tag = first ? tag_in + 1u : tag_in;
  const int run0 = first ? 0 : run_in, dead0 = first ? 0 : dead_in;
  if (first) {
    const long long c64 = llround(pairs_in);
    cnt = (int)c64;
    if (c64 < (long long)a.min_corr) {  // icp.hpp:103-108, before the rejection step
      if (lane == 0) {
        st->last_pairs = cnt;
        st->last_inliers = cnt;
        st->status = PCLMI_ERR_NOT_ENOUGH;
        st->done = 1;
      }
      return;
    }
    double m[10];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = __shfl(mom_in, i);
    m[9] = pairs_in;
    if (lane == 0) {
      s_sthr = cnt >= 3 ? sample_threshold_from_moments(m) : 0.0;
      rl->tag = tag;
      rl->cnt = cnt;
      rl->active = cnt >= 3 ? 1 : 0;  // fewer: getSamples clears the selection, computeModel fails (ransac.hpp:72-76)
      rl->all_kept = cnt == a.n_rows ? 1 : 0;
      rl->ntrials = 0;
      rl->bad_run = 0;
      rl->dead = 0;
      rl->iterations = 0;
      rl->best = -INT32_MAX;
      rl->have = 0;
      rl->best_trial = -1;
      rl->need_refit = 0;
      rl->settled = cnt >= 3 ? 0 : 1;
      rl->batches = 0;
      rl->skipped = 0u;
      rl->k = 1.0;
      rl->sthr = s_sthr;
    }
    if (cnt < 3) return;
    __syncthreads();
    sthr = s_sthr;
  } else {
    cnt = cnt_in;
    sthr = sthr_in;
  }
  const bool all_kept = cnt == a.n_rows;
  const bool fresh = first != 0;
  const bool pre_lds = a.nsb <= kSbLds;
  if (!all_kept && (first || pre_lds)) {  // exclusive prefix of the per-1024-row pair counts
    // first batch: from the counts; a further batch with the prefix in LDS: the same again (cheap)
    const int chunk = (a.nsb + 63) / 64;
    const int b0 = lane * chunk, b1 = min(a.nsb, b0 + chunk);
    if (pre_lds) {
      for (int i = lane; i < a.nsb; i += 64) lpre[i] = ld_state<AG>(&a.sb_cnt[i]);  // independent loads, all in flight
      __syncthreads();
    }
    unsigned int sum = 0;
    for (int i = b0; i < b1; ++i) sum += pre_lds ? lpre[i] : ld_state<AG>(&a.sb_cnt[i]);
    unsigned int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    unsigned int run = incl - sum;
    for (int i = b0; i < b1; ++i) {
      const unsigned int c = pre_lds ? lpre[i] : ld_state<AG>(&a.sb_cnt[i]);
      if (pre_lds) lpre[i] = run;
      else a.sb_prefix[i] = run;
      run += c;
    }
    if (!pre_lds) __threadfence();  // select_row (other lanes of this wave) reads the prefix with agent-scope loads
    __syncthreads();
  }
#define PCLMI_DSTAMP(i)                                                    \
  do {                                                                    \
    if (lane == 0) rl->dbg[i] = (unsigned long long)wall_clock64();       \
  } while (0)
  PCLMI_DSTAMP(1);
#pragma unroll
  for (int i = 0; i < 10; ++i)
    if (lane + 64 * i < 624) mt[lane + 64 * i] = mtv[i];
  __syncthreads();
  // (1) the generator's numbers of the whole batch: they depend on nothing but the pair count
  for (int d = 0; d < draws; ++d)
    for (int i = 0; i < 3; ++i) {
      const unsigned int r = mt_next(mt, idx) >> 1;  // every lane computes the same (LDS state, same-value stores)
      if (lane == 0) jpos[d][i] = i + (int)(r % (unsigned int)(cnt - i));
    }
  __syncthreads();
  PCLMI_DSTAMP(2);
  // (2) the far ends' entries as earlier batches of this iteration left them, all loads in flight together
  for (int t = lane; t < 3 * draws; t += 64) {
    const int j = jpos[t / 3][t % 3];
    int v = j;
    if (!fresh) {
      const unsigned long long g = a.shuf[j];
      if ((unsigned int)(g >> 32) == tag) v = (int)(unsigned int)g;
    }
    gval[t / 3][t % 3] = v;
  }
  int head[3] = {0, 1, 2};
  if (!fresh)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned long long g = a.shuf[i];
      if ((unsigned int)(g >> 32) == tag) head[i] = (int)(unsigned int)g;
    }
  __syncthreads();
  // (3) drawIndexSample (sac_model.h:401-412) replayed: shuf[i] <-> shuf[i + rnd() % (cnt - i)], i = 0, 1, 2.
  // The swaps of a batch only depend on each other where two of them touch one position.  With far ends that are
  // all different and all beyond the three heads (the rule on a large cloud: 192 positions out of a million) every
  // sample is simply what its far ends held, and each far end receives the head of the draw before - no replay.
  __shared__ int htab[512];
  for (int i = lane; i < 512; i += 64) htab[i] = -1;
  __syncthreads();
  bool conflict = false;
  for (int t = lane; t < 3 * draws; t += 64) {
    const int j = jpos[t / 3][t % 3];
    if (j < 3) {
      conflict = true;
    } else {
      unsigned int h = ((unsigned int)j * 2654435761u) >> 23;
      for (int probe = 0; probe < 512; ++probe) {
        const int old = atomicCAS(&htab[h], -1, j);
        if (old == -1) break;
        if (old == j) {
          conflict = true;
          break;
        }
        h = (h + 1u) & 511u;
      }
    }
  }
  const bool replay = __any(conflict) != 0;
  const int head0[3] = {head[0], head[1], head[2]};
  ShufMap map;
#pragma unroll
  for (int e = 0; e < 4; ++e) map.pos[e] = -1, map.val[e] = 0;
  map.count = 0;
  if (!replay) {
    for (int t = lane; t < 3 * draws; t += 64) dpos[t / 3][t % 3] = gval[t / 3][t % 3];
    if (draws > 0) {
      head[0] = gval[draws - 1][0];
      head[1] = gval[draws - 1][1];
      head[2] = gval[draws - 1][2];
    }
  }
  for (int d = 0; replay && d < draws; ++d) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // (unrolled: head[] stays in registers)
      const int j = jpos[d][i];
      if (j < 3) {
        const int t = head[i];
        // j >= i; the compiler keeps head[] in registers with these selects
        const int hj = j == 0 ? head[0] : (j == 1 ? head[1] : head[2]);
        head[i] = hj;
        if (j == 0) head[0] = t;
        else if (j == 1) head[1] = t;
        else head[2] = t;
      } else {
        int v;
        if (!map_get(map, j, v)) v = gval[d][i];
        map_put(map, j, head[i], lane);
        head[i] = v;
      }
    }
    if (lane == 0) {
      dpos[d][0] = head[0];
      dpos[d][1] = head[1];
      dpos[d][2] = head[2];
    }
  }
  __syncthreads();
  PCLMI_DSTAMP(3);
  // (4) the sampled pairs' coordinates, one lookup chain per lane
  for (int t = lane; t < 3 * draws; t += 64) {
    const int pos = dpos[t / 3][t % 3];
    const int row = all_kept ? pos : select_row<AG>(a, pre_lds ? lpre : nullptr, pos);
    const int sl = row >= 0 ? a.inv[row] : -1;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
    if (sl >= 0) {
      p = a.pts[sl];
      q = a.tq[sl];
    }
    six[t / 3][2 * (t % 3)] = p;
    six[t / 3][2 * (t % 3) + 1] = q;
  }
  __syncthreads();
  if (a.comm.world > 1) {
    // sharded: a sampled row belongs to ONE rank, which holds its point as it stands now and its match; the others
    // contributed zeros above.  Bitwise OR over the ranks, three 64-bit words per (draw, point): exact for any bits.
    const int nwords = 9 * draws;
    unsigned int seq = *a.comm.seq;
    bool ok = true;
    for (int w0 = 0; w0 < nwords && ok; w0 += kCommMaxCount) {
      const int w = w0 + lane, cntw = min(kCommMaxCount, nwords - w0);
      unsigned long long bits = 0ull;
      const float *f = reinterpret_cast<const float *>(&six[0][0]);  // item t = w / 3: floats 8 t + {0,1,2} and 8 t + {4,5,6}
      const int t = w / 3, part = w % 3;
      if (lane < cntw) {
        const int i0 = part == 0 ? 0 : (part == 1 ? 2 : 5), i1 = part == 0 ? 1 : (part == 1 ? 4 : 6);
        bits = ((unsigned long long)__float_as_uint(f[8 * t + i1]) << 32) | __float_as_uint(f[8 * t + i0]);
      }
      ++seq;
      ok = comm_allreduce_bits<kCommOrU64>(a.comm, bits, cntw, lane, seq);
      if (ok && lane < cntw) {
        const int i0 = part == 0 ? 0 : (part == 1 ? 2 : 5), i1 = part == 0 ? 1 : (part == 1 ? 4 : 6);
        float *fw = reinterpret_cast<float *>(&six[0][0]);
        fw[8 * t + i0] = __uint_as_float((unsigned int)bits);
        fw[8 * t + i1] = __uint_as_float((unsigned int)(bits >> 32));
      }
    }
    if (lane == 0) *a.comm.seq = seq;
    if (!ok) {
      if (lane == 0) ransac_comm_failed(a);
      return;
    }
    __syncthreads();
  }
  PCLMI_DSTAMP(4);
  // (5) isSampleGood (sac_model_registration.hpp:47-60); a trial is a good sample, at most 1000 draws apart
  bool good = false;
  if (lane < draws) {
    auto sq = [](const float4 &u, const float4 &v) {
      const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z;
      return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    };
    const float4 s0 = six[lane][0], s1 = six[lane][2], s2 = six[lane][4];
    good = (double)sq(s1, s0) > sthr && (double)sq(s2, s0) > sthr && (double)sq(s2, s1) > sthr;
  }
  const unsigned long long gm = __ballot(good);
  int nt = 0, run = run0, dead = dead0;
  for (int d = 0; d < draws && !dead; ++d) {
    if ((gm >> d) & 1ull) {
      if (lane == 0) trial_of[nt] = d;
      ++nt;
      run = 0;
    } else if (++run >= 1000) {  // max_sample_checks_ (sac_model.h:470)
      dead = 1;
    }
  }
  __syncthreads();
  // (6) three-point models (estimateRigidTransformationSVD, sac_model_registration.hpp:257-302), one per lane
  if (lane < nt) {
    RansacTrial tr;
    for (int i = 0; i < 16; ++i) tr.T[i] = (i % 5 == 0) ? 1.f : 0.f;
    tr.status = 0;
    tr.count = 0u;
    tr.pad[0] = tr.pad[1] = 0;
    const float4 *h6 = six[trial_of[lane]];
    double sm[PCLMI_ICP_NSUMS];
    for (int i = 0; i < PCLMI_ICP_NSUMS; ++i) sm[i] = 0.0;
    const double o[3] = {h6[1].x, h6[1].y, h6[1].z};
    for (int i = 0; i < 3; ++i) {
      const double sx = h6[2 * i].x - o[0], sy = h6[2 * i].y - o[1], sz = h6[2 * i].z - o[2];
      const double tx = h6[2 * i + 1].x - o[0], ty = h6[2 * i + 1].y - o[1], tz = h6[2 * i + 1].z - o[2];
      sm[0] += 1;
      sm[1] += sx; sm[2] += sy; sm[3] += sz;
      sm[4] += tx; sm[5] += ty; sm[6] += tz;
      sm[7] += sx * tx; sm[8] += sx * ty; sm[9] += sx * tz;
      sm[10] += sy * tx; sm[11] += sy * ty; sm[12] += sy * tz;
      sm[13] += sz * tx; sm[14] += sz * ty; sm[15] += sz * tz;
      sm[18] += 1;
      sm[19] += o[0]; sm[20] += o[1]; sm[21] += o[2];
    }
    if (!rs::solve_from_sums(sm, tr.T, work[lane])) tr.status = 1;
    a.trials[lane] = tr;
  }
  __builtin_amdgcn_s_waitcnt(0);
  PCLMI_DSTAMP(5);
  // (7) what the batch leaves behind: generator, shuffle entries, the run of bad samples
  for (int i = lane; i < 624; i += 64) a.rng->mt[i] = mt[i];
  const unsigned long long tg = (unsigned long long)tag << 32;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (map.pos[e] >= 0) a.shuf[map.pos[e]] = tg | (unsigned int)map.val[e];
  if (!replay)  // far end (d, i) received the head of the draw before
    for (int t = lane; t < 3 * draws; t += 64) {
      const int d = t / 3, i = t % 3;
      const int v = d == 0 ? (i == 0 ? head0[0] : (i == 1 ? head0[1] : head0[2])) : gval[d - 1][i];
      a.shuf[jpos[d][i]] = tg | (unsigned int)v;
    }
  if (lane < 3) a.shuf[lane] = tg | (unsigned int)(lane == 0 ? head[0] : (lane == 1 ? head[1] : head[2]));
  if (lane == 0) {
    a.rng->idx = idx;
    rl->ntrials = nt;
    rl->bad_run = run;
    rl->dead = dead;
    rl->batches += 1;
  }
  __builtin_amdgcn_s_waitcnt(0);
  PCLMI_DSTAMP(6);
#undef PCLMI_DSTAMP
}

__global__ __launch_bounds__(64) void ransac_draw_loop_kernel(RansacLoopArgs a, int first, int draws) {
  ransac_draw_body<false>(a, first, draws);
}

// One rank, first batch of an iteration: the mask words AND the draws in ONE launch (one launch boundary less per
// iteration).  Every workgroup builds the words of its 1024 rows - nothing to do when every row has a pair, the state
// of a converging align - and workgroup 0's first wavefront then draws; it waits for the other workgroups' words only
// when there are any (they depend on nothing, so they always arrive; the wait is bounded all the same).  The words
// and counts are handed over with agent-scope stores, drained, and an arrival count.
__global__ __launch_bounds__(kSbRows) void ransac_words_draw_kernel(RansacLoopArgs a, int draws) {
  if (a.st->done || a.st->hold) return;
  const bool all_kept = llround(a.sums[a.est ? 27 : 0]) == (long long)a.n_rows;
  if (!all_kept) {
    const int row = blockIdx.x * kSbRows + threadIdx.x;
    bool k = false;
    if (row < a.n_rows) {
      const int sl = a.inv[row];
      k = sl >= 0 && a.keep_slot[sl] != 0;
    }
    const unsigned long long w = __ballot(k);
    __shared__ int c16[kSbRows / 64];
    if ((threadIdx.x & 63) == 0) {
      c16[threadIdx.x >> 6] = __popcll(w);
      st_state<true>(&a.keep_bits[row >> 6], w);  // nsb * 16 words are allocated
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int i = 0; i < kSbRows / 64; ++i) t += c16[i];
      st_state<true>(&a.sb_cnt[blockIdx.x], (unsigned int)t);
    }
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add((g_u32p)&a.rl->wticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  if (!all_kept) {
    const long long t0 = wall_clock64();
    unsigned int spins = 0;
    bool ok = true;
    while (ld_state<true>(&a.rl->wticket) != gridDim.x) {
      if ((++spins & 63u) == 0u && (unsigned long long)(wall_clock64() - t0) > 500000000ull) {  // 5 s
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0) st_state<true>(&a.rl->wticket, 0u);
    if (!ok) {
      if (threadIdx.x == 0) {
        a.st->status = PCLMI_ERR_HIP;
        a.st->done = 1;
      }
      return;
    }
  }
  ransac_draw_body<true>(a, 1, draws);
}

// countWithinDistance of every model of the batch in one pass over the stored source points (masked by keep_slot),
// then - in the workgroup that finishes last - the trials are consumed in order under the reference's stopping rule
__global__ __launch_bounds__(256) void ransac_count_loop_kernel(RansacLoopArgs a, int first) {
  // the models in PAIRS: coefficient i of models 2 k and 2 k + 1 side by side, for the packed fp32 instructions below
  __shared__ f2v xf2[kRansacBatch / 2][12];
  __shared__ int ready[kRansacBatch];
  __shared__ unsigned int cnt[kRansacBatch];
  __shared__ int s_last;
  RansacLoopState *rl = a.rl;
  IcpDevState *st = a.st;
  if (st->done || !rl->active || rl->settled) return;
  if (first ? st->hold != 0 : st->hold == 0) return;
  const int batch = rl->ntrials;
  if ((int)threadIdx.x < kRansacBatch) {
    const bool in = (int)threadIdx.x < batch;
    const RansacTrial &tr = a.trials[in ? threadIdx.x : 0];
    float *half = reinterpret_cast<float *>(&xf2[threadIdx.x >> 1][0]) + (threadIdx.x & 1);
    for (int i = 0; i < 12; ++i) half[2 * i] = in ? tr.T[i] : 0.f;
    ready[threadIdx.x] = in && tr.status == 0;
    cnt[threadIdx.x] = 0u;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  if (batch > 0) {
    unsigned int mine = 0u;  // lane b of each wave keeps the wave's running count for model b
    const int n = a.n;
    for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
      const int i = i0 + (int)threadIdx.x;
      const bool live = i < n && a.keep_slot[i] != 0;
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
      if (live) {
        p = a.pts[i];
        q = a.tq[i];
      }
      // two models per instruction (v_pk_mul_f32 / v_pk_add_f32: each half an ordinary IEEE fp32 operation, so the
      // distances are apply_xf's and the scalar expression's to the bit): the kernel is bound by its VALU work
      const f2v px = {p.x, p.x}, py = {p.y, p.y}, pz = {p.z, p.z};
      const f2v qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
      for (int b = 0; b < batch; b += 2) {
        if (!ready[b] && !ready[b + 1]) continue;  // block-uniform
        const f2v *T = xf2[b >> 1];
        f2v d;
        {
#pragma clang fp contract(off)
          const f2v mx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[3];
          const f2v my = ((T[4] * px + T[5] * py) + T[6] * pz) + T[7];
          const f2v mz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11];
          const f2v dx = mx - qx, dy = my - qy, dz = mz - qz;
          d = (dx * dx + dy * dy) + dz * dz;
        }
        if (ready[b]) {
          const unsigned int c = (unsigned int)__popcll(__ballot(live && (double)d.x < a.inlier_thr2));
          if (lane == b) mine += c;
        }
        if (ready[b + 1]) {
          const unsigned int c = (unsigned int)__popcll(__ballot(live && (double)d.y < a.inlier_thr2));
          if (lane == b + 1) mine += c;
        }
      }
    }
    if (lane < batch && mine) atomicAdd(&cnt[lane], mine);
    __syncthreads();
    if ((int)threadIdx.x < batch && cnt[threadIdx.x])
      __hip_atomic_fetch_add((g_u32p)&a.trials[threadIdx.x].count, cnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the hand-off of icp_step_tail: agent-scope atomics (performed at the device's coherence point), the issuing wave
  // (wave 0: batch <= 64) drains them, then ITS lane 0 takes the ticket - no cache write-back / invalidate, which an
  // acquire-release ticket would cost every one of the 1024 workgroups
  drain_stores();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add((g_u32p)&rl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last || threadIdx.x >= 64) return;
  {
    // the batch's counts: this rank's, or summed over the ranks (exact: integers below 2^53 as doubles)
    unsigned int mine_total = 0u;
    if (lane < batch) mine_total = __hip_atomic_load((g_u32p)&a.trials[lane].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.comm.world > 1) {
      double v = (double)mine_total;
      const unsigned int seq = *a.comm.seq + 1u;
      const bool ok = comm_allreduce_wave(a.comm, v, kRansacBatch, lane, seq);  // the same count on every rank
      if (lane == 0) *a.comm.seq = seq;
      if (!ok) {
        if (lane == 0) {
          __hip_atomic_store((g_u32p)&rl->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ransac_comm_failed(a);
        }
        return;
      }
      mine_total = (unsigned int)llround(v);
    }
    if (lane < kRansacBatch) cnt[lane] = mine_total;
  }
  if (threadIdx.x != 0) return;
  __hip_atomic_store((g_u32p)&rl->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ransac_consume(rl, st, a.trials, batch, a.max_trials, [&](int b) { return ready[b] != 0; },
                     [&](int b) { return (int)cnt[b]; }))
    rl->need_refit = 1;
}

// selectWithinDistance with the best model (ransac.hpp:124) fused with the sums the estimator needs of the inliers
// (icp.hpp:130-171); no model (computeModel failed) keeps every pair (icp.hpp:128-132).  Then the step's tail.
template <int EST>
__global__ __launch_bounds__(64 * ICP_WPB) void ransac_refit_kernel(RansacLoopArgs a, int first, double ox, double oy,
                                                                     double oz, double *__restrict__ partials,
                                                                     IcpTail tail, const float4 *__restrict__ tn_slot,
                                                                     const unsigned char *__restrict__ src_skip,
                                                                     unsigned char *__restrict__ flags) {
  __shared__ double lds[ICP_WPB][SumsOf<EST>::n];
  const RansacLoopState *rl = a.rl;
  if (a.st->done || !rl->settled || !rl->need_refit) return;  // (the first batch's fused sums usually did it)
  if (first ? a.st->hold != 0 : a.st->hold == 0) return;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const bool have = rl->have != 0;
  Xf T;
#pragma unroll
  for (int j = 0; j < 12; ++j) T.m[j] = rl->Tbest[j];
  if (threadIdx.x < SumsOf<EST>::n - SumsOf<EST>::carry0) {  // the carried columns: zeros from this launch
#pragma unroll
    for (int w = 0; w < ICP_WPB; ++w) lds[w][SumsOf<EST>::carry0 + threadIdx.x] = 0.0;
  }
  // every thread adds up its slots (a few: the grid is a fraction of the points, so that the reduction, the hand-off
  // of the partial sums and the arrival ticket are paid once per workgroup and not once per 256 points)
  constexpr int NA = EST ? 28 : 16;
  double acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = 0.0;
  for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < a.n; slot += gridDim.x * blockDim.x) {
    bool paired = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f), q = p;
    int js = 0;
    if (a.keep_slot[slot] != 0) {
      p = a.pts[slot];
      if (EST) js = a.nn_slot[slot];
      q = a.tq[slot];
      paired = true;
      if (have) {
        const float4 mp = apply_xf(T, p);
        const float dx = mp.x - q.x, dy = mp.y - q.y, dz = mp.z - q.z;
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        paired = (double)d < a.inlier_thr2;
      }
    }
    if (flags) flags[slot] = paired ? 1 : 0;
    if (!paired) continue;
    if constexpr (EST != 0) {
      acc[27] += 1.0;
      const float4 m = tn_slot[js];
      if (!finite3(m) || (src_skip && src_skip[slot])) continue;
      const double v[6] = {(double)__fsub_rn(__fmul_rn(m.z, p.y), __fmul_rn(m.y, p.z)),
                           (double)__fsub_rn(__fmul_rn(m.x, p.z), __fmul_rn(m.z, p.x)),
                           (double)__fsub_rn(__fmul_rn(m.y, p.x), __fmul_rn(m.x, p.y)),
                           (double)m.x, (double)m.y, (double)m.z};
      float df = __fadd_rn(__fadd_rn(__fmul_rn(m.x, q.x), __fmul_rn(m.y, q.y)), __fmul_rn(m.z, q.z));
      df = __fsub_rn(__fsub_rn(__fsub_rn(df, __fmul_rn(m.x, p.x)), __fmul_rn(m.y, p.y)), __fmul_rn(m.z, p.z));
      const double dd = df;
      int o = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int k = r; k < 6; ++k) acc[o++] += v[r] * v[k];
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[21 + r] += v[r] * dd;
    } else {
      const double sx = (double)p.x - ox, sy = (double)p.y - oy, sz = (double)p.z - oz;
      const double tx = (double)q.x - ox, ty = (double)q.y - oy, tz = (double)q.z - oz;
      acc[0] += 1.0;
      acc[1] += sx; acc[2] += sy; acc[3] += sz;
      acc[4] += tx; acc[5] += ty; acc[6] += tz;
      acc[7] += sx * tx; acc[8] += sx * ty; acc[9] += sx * tz;
      acc[10] += sy * tx; acc[11] += sy * ty; acc[12] += sy * tz;
      acc[13] += sz * tx; acc[14] += sz * ty; acc[15] += sz * tz;
    }
  }
  __syncthreads();  // the zeros of the carried columns are in place
#pragma unroll
  for (int g0 = 0; g0 < NA; g0 += 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g0 + j < NA ? acc[g0 + j] : 0.0;
    wave_reduce_to_row(v, true, &lds[wib][g0], lane);
  }
  __syncthreads();
  icp_step_tail<EST>(tail, partials, lds);
}

// The first batch of an iteration in ONE pass (the usual case: a converging align's first trial settles the stopping
// rule): the inlier counts of the batch's (at most kRansacFirst) models AND the refit sums over the inliers of the
// first of them go through the step's tail together - reduction, ONE exchange when sharded, then the stopping rule and,
// when it settles on that first model, the solve.  Otherwise the host-fed batches / ransac_refit_kernel take over.
constexpr int kRansacFirst = 4;
template <int EST>
__global__ __launch_bounds__(64 * ICP_WPB) __attribute__((amdgpu_waves_per_eu(EST ? 2 : 4, EST ? 3 : 4))) void ransac_first_kernel(RansacLoopArgs a, double ox, double oy, double oz,
                                                                     double *__restrict__ partials, IcpTail tail,
                                                                     const float4 *__restrict__ tn_slot,
                                                                     const unsigned char *__restrict__ src_skip) {
  __shared__ double lds[ICP_WPB][SumsOf<EST, true>::n];
  __shared__ Xf xf[kRansacFirst];
  __shared__ int ready[kRansacFirst];
  const RansacLoopState *rl = a.rl;
  if (a.st->done || a.st->hold) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.st->ticks[0] = (unsigned long long)wall_clock64();
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const bool active = rl->active != 0;
  const int batch = active ? min(rl->ntrials, kRansacFirst) : 0;
  if ((int)threadIdx.x < kRansacFirst) {
    const bool r = (int)threadIdx.x < batch && a.trials[threadIdx.x].status == 0;
    ready[threadIdx.x] = r ? 1 : 0;
    for (int i = 0; i < 12; ++i) xf[threadIdx.x].m[i] = r ? a.trials[threadIdx.x].T[i] : 0.f;
  }
  constexpr int NS_ = SumsOf<EST, true>::n;
  for (int j = threadIdx.x; j < ICP_WPB * NS_; j += blockDim.x) lds[j / NS_][j % NS_] = 0.0;
  __syncthreads();
  int b0 = 0;  // the model the refit sums are taken over: the first trial that has one (none: every pair, icp.hpp:128-132)
  while (b0 < batch && !ready[b0]) ++b0;
  const bool have0 = b0 < batch;
  constexpr int NA = EST ? 28 : 16;
  double acc[NA];
  double cntv[kRansacFirst];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = 0.0;
#pragma unroll
  for (int b = 0; b < kRansacFirst; ++b) cntv[b] = 0.0;
  constexpr int IT = EST ? 2 : 4;  // slots per thread and round, their loads in flight together (128 VGPRs: 4 workgroups per CU)
  for (int base = blockIdx.x * (blockDim.x * IT) + threadIdx.x; base - (int)threadIdx.x < a.n;
       base += gridDim.x * blockDim.x * IT) {
    bool kp[IT];
    float4 p[IT], q[IT];
    int js[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int slot = base + j * (int)blockDim.x;
      kp[j] = slot < a.n && a.keep_slot[slot] != 0;
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {  // (not under kp: these loads go out together with the mask's)
      const int slot = min(base + j * (int)blockDim.x, a.n - 1);
      p[j] = a.pts[slot];
      q[j] = a.tq[slot];
      js[j] = EST ? a.nn_slot[slot] : 0;
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      if (!kp[j]) continue;
      bool in0 = !have0;
#pragma unroll
      for (int b = 0; b < kRansacFirst; ++b) {
        if (!ready[b]) continue;  // block-uniform
        const float4 mp = apply_xf(xf[b], p[j]);
        const float dx = mp.x - q[j].x, dy = mp.y - q[j].y, dz = mp.z - q[j].z;
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const bool in = (double)d < a.inlier_thr2;
        cntv[b] += in ? 1.0 : 0.0;
        if (b == b0) in0 = in;
      }
      if (!in0) continue;
      if constexpr (EST != 0) {
        acc[27] += 1.0;
        const int slot = base + j * (int)blockDim.x;
        const float4 m = tn_slot[js[j]];
        if (!finite3(m) || (src_skip && src_skip[slot])) continue;
        const float4 pp = p[j], qq = q[j];
        const double v[6] = {(double)__fsub_rn(__fmul_rn(m.z, pp.y), __fmul_rn(m.y, pp.z)),
                             (double)__fsub_rn(__fmul_rn(m.x, pp.z), __fmul_rn(m.z, pp.x)),
                             (double)__fsub_rn(__fmul_rn(m.y, pp.x), __fmul_rn(m.x, pp.y)),
                             (double)m.x, (double)m.y, (double)m.z};
        float df = __fadd_rn(__fadd_rn(__fmul_rn(m.x, qq.x), __fmul_rn(m.y, qq.y)), __fmul_rn(m.z, qq.z));
        df = __fsub_rn(__fsub_rn(__fsub_rn(df, __fmul_rn(m.x, pp.x)), __fmul_rn(m.y, pp.y)), __fmul_rn(m.z, pp.z));
        const double dd = df;
        int o = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int k = r; k < 6; ++k) acc[o++] += v[r] * v[k];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[21 + r] += v[r] * dd;
      } else {
        const double sx = (double)p[j].x - ox, sy = (double)p[j].y - oy, sz = (double)p[j].z - oz;
        const double tx = (double)q[j].x - ox, ty = (double)q[j].y - oy, tz = (double)q[j].z - oz;
        acc[0] += 1.0;
        acc[1] += sx; acc[2] += sy; acc[3] += sz;
        acc[4] += tx; acc[5] += ty; acc[6] += tz;
        acc[7] += sx * tx; acc[8] += sx * ty; acc[9] += sx * tz;
        acc[10] += sy * tx; acc[11] += sy * ty; acc[12] += sy * tz;
        acc[13] += sz * tx; acc[14] += sz * ty; acc[15] += sz * tz;
      }
    }
  }
#pragma unroll
  for (int g0 = 0; g0 < NA; g0 += 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g0 + j < NA ? acc[g0 + j] : 0.0;
    wave_reduce_to_row(v, true, &lds[wib][g0], lane);
  }
  {
    double v[8] = {cntv[0], cntv[1], cntv[2], cntv[3], 0.0, 0.0, 0.0, 0.0};
    wave_reduce_to_row(v, true, &lds[wib][EST ? 32 : 24], lane);
  }
  __syncthreads();
  icp_step_tail<EST, true, true>(tail, partials, lds);
}

inline unsigned blocks_for(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

// ---- launchers ---------------------------------------------------------------------------------
void launch_transform(float4 *d_pts, size_t n, const Xf &T, hipStream_t s) {
  if (n) hipLaunchKernelGGL(transform_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_pts, (int)n, T);
}

void launch_transform_pending(float4 *d_pts, size_t n, IcpDevState *st, hipStream_t s) {
  if (n) hipLaunchKernelGGL(transform_pending_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_pts, (int)n, st);
  hipLaunchKernelGGL(clear_apply_kernel, dim3(1), dim3(1), 0, s, st);
}

void launch_icp_solve(const double *d_sums, IcpDevState *st, hipStream_t s, bool plane) {
  if (plane) hipLaunchKernelGGL(icp_solve_kernel<1>, dim3(1), dim3(64), 0, s, d_sums, st);
  else hipLaunchKernelGGL(icp_solve_kernel<0>, dim3(1), dim3(64), 0, s, d_sums, st);
}

void launch_pack(const float *d_in, size_t stride_f, const int *d_rows, size_t n, float4 *d_out, hipStream_t s) {
  if (n)
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_in, (int)stride_f, d_rows, (int)n,
                       d_out);
}

void launch_permute(const float4 *in, const int *order, size_t n, float4 *out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(permute_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, in, order, (int)n, out);
}

void launch_unpermute(const float4 *in, const int *order, size_t n, float4 *out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(unpermute_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, in, order, (int)n, out);
}

unsigned icp_step_blocks(size_t n) {
  unsigned waves = (unsigned)((n + 63) / 64);
  unsigned blocks = (waves + ICP_WPB - 1) / ICP_WPB;
  blocks = (blocks + 7) & ~7u;
  return blocks ? blocks : 8u;  // an empty shard still launches: its sums are zeros, its tail runs the exchange
}

// How many workgroups of the resident loop's step kernel fit on the chip at once (0: the loop is not available).
// A launch of the loop needs ALL its workgroups resident: they wait for each other.
size_t icp_loop_capacity(const pclmi_tree *t, bool plane) {
  if (!(plane || nn1_per_lane())) return 0;
  static const bool off = [] { const char *e = getenv("PCLMI_ICP_LOOP"); return e && e[0] == '0'; }();
  if (off) return 0;
  const int levels = lane_nearest_levels(t->M);
  const size_t lds_bytes = (size_t)ICP_WPB * levels * 64 * sizeof(float);
  int per_cu = 0, dev = 0;
  hipError_t e = plane ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, icp_step_kernel<true, 1, false, false, true>,
                                                                      64 * ICP_WPB, lds_bytes)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, icp_step_kernel<true, 0, false, false, true>,
                                                                      64 * ICP_WPB, lds_bytes);
  hipDeviceProp_t prop;
  if (e != hipSuccess || hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  // PCLMI_DEVICE_SHARE = n: n ranks of one job run on this GPU (rehearsals of the sharded loop on one device): their
  // resident loops wait for each other's sums, so all of them must fit on the chip together
  static const int share = [] { const char *e = getenv("PCLMI_DEVICE_SHARE"); const int v = e ? atoi(e) : 1; return v > 1 ? v : 1; }();
  // (workgroup 0 adds up the group sums in one batch of <= 8 entries per lane: at most 64 groups of ICP_GROUP)
  return std::min((size_t)per_cu * (size_t)prop.multiProcessorCount / (size_t)share, (size_t)64 * ICP_GROUP);
}
bool icp_loop_fits(const pclmi_tree *t, size_t n, bool plane) { return icp_step_blocks(n) <= icp_loop_capacity(t, plane); }
size_t icp_loop_box_words(size_t n) { return ((size_t)icp_step_blocks(n) + icp_step_groups(n) + 1) * kLoopRow; }

unsigned icp_step_groups(size_t n) { return (icp_step_blocks(n) + ICP_GROUP - 1) / ICP_GROUP; }

void launch_icp_step(const pclmi_tree *t, float4 *d_pts, size_t n, bool apply, const Xf &T, double thr2,
                     int *d_nn, int *d_nn_slot, float *d_dist, const double origin[3], double *d_partials,
                     hipStream_t s, IcpDevState *st, bool have_prev, const IcpTail &tail, const float4 *d_tn_slot,
                     const unsigned char *d_src_skip, unsigned char *d_keep_slot, float4 *d_tq, const float4 *d_raw_tgt,
                     const float *scale, int loop_iters) {
  const float sc[3] = {scale ? scale[0] : 1.f, scale ? scale[1] : 1.f, scale ? scale[2] : 1.f};
  const unsigned blocks = icp_step_blocks(n);
  const bool plane = d_tn_slot != nullptr;  // the point-to-plane loop always walks per lane
  const bool rsv = d_keep_slot != nullptr;  // the RANSAC loop's step
  const int levels = (plane || nn1_per_lane()) ? lane_nearest_levels(t->M) : 0;
  TimingScope ts(PCLMI_TIMER_ICP_STEP, s);
  const size_t lds_bytes = (size_t)ICP_WPB * levels * 64 * sizeof(float);
  const bool scl = d_raw_tgt != nullptr;    // searches in a rescaled space (api_icp.hip refuses it with the plane loop)
  const bool loop = loop_iters > 1;  // the caller has checked icp_loop_fits() and set tail.loopbox
  // (PCLMI_ICP_CERT_LEVELS=1: the first certificate only, for A/B runs)
  static const int force_levels = [] { const char *e = getenv("PCLMI_ICP_CERT_LEVELS"); return e ? atoi(e) : 0; }();
  TreeView view = t->view();
  view.cert_levels = force_levels ? force_levels : 2;
#define PCLMI_STEP_LAUNCH(LANE_, EST_, RS_, SC_, LDS_) PCLMI_STEP_LAUNCH5(LANE_, EST_, RS_, SC_, false, LDS_)
#define PCLMI_STEP_LAUNCH5(LANE_, EST_, RS_, SC_, LOOP_, LDS_)                                                              \
  hipLaunchKernelGGL((icp_step_kernel<LANE_, EST_, RS_, SC_, LOOP_>), dim3(blocks), dim3(64 * ICP_WPB), LDS_, s, view, d_pts, \
                     (int)n, apply ? 1 : 0, T, (float)thr2, thr2, d_nn, d_nn_slot, d_dist, origin[0], origin[1],        \
                     origin[2], d_partials, (const IcpDevState *)st, levels, have_prev ? 1 : 0, tail, d_tn_slot,        \
                     d_src_skip, d_keep_slot, d_tq, d_raw_tgt, sc[0], sc[1], sc[2], loop_iters)
  if (plane) {
    if (rsv) PCLMI_STEP_LAUNCH(true, 1, true, false, lds_bytes);
    else if (loop) PCLMI_STEP_LAUNCH5(true, 1, false, false, true, lds_bytes);
    else PCLMI_STEP_LAUNCH(true, 1, false, false, lds_bytes);
  } else if (levels > 0) {
    if (scl) {
      if (rsv) PCLMI_STEP_LAUNCH(true, 0, true, true, lds_bytes);
      else PCLMI_STEP_LAUNCH(true, 0, false, true, lds_bytes);
    } else {
      if (rsv) PCLMI_STEP_LAUNCH(true, 0, true, false, lds_bytes);
      else if (loop) PCLMI_STEP_LAUNCH5(true, 0, false, false, true, lds_bytes);
      else PCLMI_STEP_LAUNCH(true, 0, false, false, lds_bytes);
    }
  } else {
    if (scl) {
      if (rsv) PCLMI_STEP_LAUNCH(false, 0, true, true, 0);
      else PCLMI_STEP_LAUNCH(false, 0, false, true, 0);
    } else {
      if (rsv) PCLMI_STEP_LAUNCH(false, 0, true, false, 0);
      else PCLMI_STEP_LAUNCH(false, 0, false, false, 0);
    }
  }
#undef PCLMI_STEP_LAUNCH
#undef PCLMI_STEP_LAUNCH5
}

unsigned pair_reduce_blocks(size_t n) {
  unsigned b = blocks_for(n, 64 * PR_WPB);
  return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}

void launch_pair_reduce(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx,
                        const unsigned char *flags, size_t n, const double origin[3], double *d_partials,
                        double *d_sums, hipStream_t s) {
  const unsigned blocks = pair_reduce_blocks(n);
  hipLaunchKernelGGL(pair_reduce_kernel, dim3(blocks), dim3(64 * PR_WPB), 0, s, src, sidx, tgt, tidx, flags, (int)n,
                     origin[0], origin[1], origin[2], d_partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(NS), dim3(256), 0, s, d_partials, (int)blocks, (size_t)NS, (size_t)1,
                     d_sums);
}

void launch_pair_reduce_plane(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx, const float4 *tn,
                              const unsigned char *src_skip, const unsigned char *flags, size_t n, double *d_partials,
                              double *d_sums, hipStream_t s) {
  const unsigned blocks = pair_reduce_blocks(n);
  hipLaunchKernelGGL(pair_reduce_plane_kernel, dim3(blocks), dim3(64 * PR_WPB), 0, s, src, sidx, tgt, tidx, tn, src_skip,
                     flags, (int)n, d_partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(NSP), dim3(256), 0, s, d_partials, (int)blocks, (size_t)NSP, (size_t)1,
                     d_sums);
}

void launch_normals_to_slots(const pclmi_tree *t, const float4 *tn_rows, float4 *tn_slot, hipStream_t s) {
  const size_t nslots = (size_t)t->M * kLeafSize;
  if (nslots)
    hipLaunchKernelGGL(normals_to_slots_kernel, dim3(blocks_for(nslots, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4 *>(t->d_slot), (int)nslots, tn_rows, tn_slot);
}

void launch_corr_reduce(const float *src, size_t src_stride_f, const float *tgt, size_t tgt_stride_f,
                        const pclmi_correspondence *corr, size_t n, double *d_partials, double *d_sums, hipStream_t s) {
  const unsigned blocks = pair_reduce_blocks(n);
  hipLaunchKernelGGL(corr_reduce_kernel, dim3(blocks), dim3(64 * PR_WPB), 0, s, src, (int)src_stride_f, tgt,
                     (int)tgt_stride_f, corr, (int)n, d_partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(NS), dim3(256), 0, s, d_partials, (int)blocks, (size_t)NS, (size_t)1,
                     d_sums);
}

void launch_moment_reduce(const float4 *src, const int *sidx, size_t n, double *d_partials, double *d_sums,
                          hipStream_t s, const unsigned char *keep) {
  const unsigned blocks = pair_reduce_blocks(n);
  hipLaunchKernelGGL(moment_reduce_kernel, dim3(blocks), dim3(64 * PR_WPB), 0, s, src, sidx, (int)n, keep, d_partials);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(NS), dim3(256), 0, s, d_partials, (int)blocks, (size_t)NS, (size_t)1,
                     d_sums);
}

void launch_count_within(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx, size_t n,
                         const Xf &T, double thr2, unsigned int *d_count, unsigned char *d_flags, hipStream_t s,
                         const unsigned char *keep) {
  (void)hipMemsetAsync(d_count, 0, sizeof(unsigned int), s);
  unsigned b = blocks_for(n, 256);
  b = b < 1 ? 1 : (b > 1024 ? 1024 : b);
  hipLaunchKernelGGL(count_within_kernel, dim3(b), dim3(256), 0, s, src, sidx, tgt, tidx, (int)n, T, thr2, d_count,
                     d_flags, keep);
}

// The freshly seeded generator, first twist done (it happens once, here, instead of at the head of every call's first
// draw): computed on the host - 1248 dependent steps took one GPU lane 84 us - and copied.
static RansacRng seeded_generator(unsigned int seed) {
  RansacRng h;
  unsigned int x = seed;
  h.mt[0] = x;
  for (int i = 1; i < 624; ++i) {
    x = 1812433253u * (x ^ (x >> 30)) + (unsigned int)i;
    h.mt[i] = x;
  }
  for (int i = 0; i < 624; ++i) {
    const unsigned int y = (h.mt[i] & 0x80000000u) | (h.mt[(i + 1) % 624] & 0x7fffffffu);
    h.mt[i] = h.mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  h.idx = 0;
  return h;
}
void launch_ransac_seed(RansacRng *d_rng, unsigned int seed, hipStream_t s) {
  static const RansacRng ref = seeded_generator(12345u);  // the reference's seed (sac_model.h:95): lives as long as the process
  if (seed == 12345u) {
    (void)hipMemcpyAsync(d_rng, &ref, sizeof(ref), hipMemcpyHostToDevice, s);
    return;
  }
  const RansacRng h = seeded_generator(seed);
  (void)hipMemcpyAsync(d_rng, &h, sizeof(h), hipMemcpyHostToDevice, s);
  (void)hipStreamSynchronize(s);  // the source is a local
}

void launch_ransac_draw(const RansacRng *d_rng_in, RansacRng *d_rng, int *d_shuf, int cnt, const float4 *src,
                        const int *sidx, const float4 *tgt, const int *tidx, const double *d_sample_dist_thresh, int batch,
                        RansacTrial *d_trials, hipStream_t s) {
  hipLaunchKernelGGL(ransac_draw_kernel, dim3(1), dim3(64), 0, s, d_rng_in, d_rng, d_shuf, cnt, src, sidx, tgt, tidx,
                     d_sample_dist_thresh, batch, d_trials);
}

void launch_ransac_sample_threshold(const double *d_moment_sums, double *d_out, hipStream_t s) {
  hipLaunchKernelGGL(ransac_sample_threshold_kernel, dim3(1), dim3(64), 0, s, d_moment_sums, d_out);
}

void launch_count_within_batch(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx, size_t n,
                               double thr2, int batch, RansacTrial *d_trials, hipStream_t s, const unsigned char *keep) {
  unsigned b = blocks_for(n, 256);
  b = b < 1 ? 1 : (b > 1024 ? 1024 : b);
  hipLaunchKernelGGL(count_within_batch_kernel, dim3(b), dim3(256), 0, s, src, sidx, tgt, tidx, (int)n, thr2, batch,
                     d_trials, keep);
}

void launch_corr_flags(const int *d_nn, const float4 *d_pts, const float4 *d_cloud, const int *d_order, size_t n,
                       double thr2, int *d_tgt_of_orig, unsigned char *d_keep, hipStream_t s,
                       unsigned char *d_keep_slot, const float4 *d_rows, const int *d_row_of) {
  if (n)
    hipLaunchKernelGGL(corr_flags_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_nn, d_pts, d_cloud, d_rows,
                       d_row_of, d_order, (int)n, thr2, d_tgt_of_orig, d_keep, d_keep_slot);
}

void launch_gather_u8(const unsigned char *in, const int *idx, size_t n, unsigned char *out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(gather_u8_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, in, idx, (int)n, out);
}

__global__ void inverse_permutation_kernel(const int *__restrict__ order, int n, int *__restrict__ inv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[order[i]] = i;
}
void launch_inverse_permutation(const int *order, size_t n, int *inv, hipStream_t s) {
  if (n) hipLaunchKernelGGL(inverse_permutation_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, order, (int)n, inv);
}

void launch_iota(int *v, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, (int)n);
}

void launch_gather_int(const int *in, const int *idx, size_t n, int *out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(gather_int_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, in, idx, (int)n, out);
}

void launch_gather_points(const float4 *a, const int *ia, const float4 *b, const int *ib, const int pos[3],
                          float4 *out, hipStream_t s) {
  Pos3 p3{{pos[0], pos[1], pos[2]}};
  hipLaunchKernelGGL(gather_points_kernel, dim3(1), dim3(64), 0, s, a, ia, b, ib, p3, out);
}

int select_flagged(const int *d_in, const unsigned char *d_flags, size_t n, int *d_out, int *d_num, void *d_tmp,
                   size_t tmp_bytes, size_t *need, hipStream_t s) {
  size_t bytes = tmp_bytes;
  hipError_t e = hipcub::DeviceSelect::Flagged(d_tmp, bytes, d_in, d_flags, d_out, d_num, (int)n, s);
  if (need) *need = bytes;
  if (e != hipSuccess) return hip_fail(e, "DeviceSelect::Flagged", __FILE__, __LINE__);
  return PCLMI_OK;
}

void launch_fitness(const float *d_d2, size_t n, double max_range, double *d_out2, hipStream_t s) {
  (void)hipMemsetAsync(d_out2, 0, 2 * sizeof(double), s);
  unsigned b = blocks_for(n, 256);
  b = b < 1 ? 1 : (b > 2048 ? 2048 : b);
  hipLaunchKernelGGL(fitness_kernel, dim3(b), dim3(256), 0, s, d_d2, (int)n, max_range, d_out2);
}

void launch_ransac_loop_batch(const RansacLoopArgs &a, int first, int draws, hipStream_t s) {
  if (draws > kRansacBatch) draws = kRansacBatch;
  static const bool fuse = [] { const char *e = getenv("PCLMI_RANSAC_FUSE_WORDS"); return !(e && e[0] == '0'); }();
  if (first && a.nsb > 0 && a.comm.world == 1 && fuse) {
    hipLaunchKernelGGL(ransac_words_draw_kernel, dim3((unsigned)a.nsb), dim3(kSbRows), 0, s, a, draws);
    return;  // the first batch's models are counted by launch_ransac_loop_first
  }
  if (first && a.nsb > 0) {
    hipLaunchKernelGGL(ransac_keep_words_kernel, dim3((unsigned)a.nsb), dim3(kSbRows), 0, s, a);
    if (a.comm.world > 1)
      hipLaunchKernelGGL(ransac_keep_count_kernel, dim3((unsigned)std::min(64, (a.nsb + 3) / 4)), dim3(64), 0, s, a);
  }
  hipLaunchKernelGGL(ransac_draw_loop_kernel, dim3(1), dim3(64), 0, s, a, first, draws);
  if (first) return;  // the first batch's models are counted by launch_ransac_loop_first
  unsigned b = blocks_for((size_t)a.n, 256);
  b = b < 1 ? 1 : (b > 1024 ? 1024 : b);
  hipLaunchKernelGGL(ransac_count_loop_kernel, dim3(b), dim3(256), 0, s, a, first);
}

void launch_ransac_loop_first(const RansacLoopArgs &a, const double origin[3], double *d_partials, const IcpTail &tail,
                              const float4 *d_tn_slot, const unsigned char *d_src_skip, hipStream_t s) {
  // one round of workgroups: 4 per CU x 256 CUs are resident at 128 VGPRs
  unsigned blocks = blocks_for((size_t)a.n, 64 * ICP_WPB * (d_tn_slot ? 2 : 4));
  const unsigned cap = d_tn_slot ? 512u : 1024u;  // (the point-to-plane instance: 2 workgroups per CU)
  blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
  IcpTail tl = tail;
  tl.ngroups = (int)((blocks + ICP_GROUP - 1) / ICP_GROUP);
  tl.rl = a.rl;
  tl.trials = a.trials;
  tl.max_trials = a.max_trials;
  if (d_tn_slot)
    hipLaunchKernelGGL(ransac_first_kernel<1>, dim3(blocks), dim3(64 * ICP_WPB), 0, s, a, origin[0], origin[1], origin[2],
                       d_partials, tl, d_tn_slot, d_src_skip);
  else
    hipLaunchKernelGGL(ransac_first_kernel<0>, dim3(blocks), dim3(64 * ICP_WPB), 0, s, a, origin[0], origin[1], origin[2],
                       d_partials, tl, d_tn_slot, d_src_skip);
}

void launch_ransac_loop_refit(const RansacLoopArgs &a, const pclmi_tree *t, int first, const double origin[3],
                              double *d_partials, const IcpTail &tail, const float4 *d_tn_slot,
                              const unsigned char *d_src_skip, unsigned char *d_flags, hipStream_t s) {
  (void)t;
  // a fraction of the step's grid (every thread adds up a few slots); the tail's groups follow the grid
  const unsigned blocks = std::min(icp_step_blocks((size_t)a.n), 1024u);
  IcpTail tl = tail;
  tl.ngroups = (int)((blocks + ICP_GROUP - 1) / ICP_GROUP);
  if (d_tn_slot)
    hipLaunchKernelGGL(ransac_refit_kernel<1>, dim3(blocks), dim3(64 * ICP_WPB), 0, s, a, first, origin[0], origin[1],
                       origin[2], d_partials, tl, d_tn_slot, d_src_skip, d_flags);
  else
    hipLaunchKernelGGL(ransac_refit_kernel<0>, dim3(blocks), dim3(64 * ICP_WPB), 0, s, a, first, origin[0], origin[1],
                       origin[2], d_partials, tl, d_tn_slot, d_src_skip, d_flags);
}

}  // namespace pclmi
