// api_icp.hip - C ABI: correspondences, rigid-transform estimation, transform, ICP.
#include <memory>
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <random>
#include <unordered_map>
#include <vector>

#include "pclmi_internal.h"

using namespace pclmi;

namespace {

bool is_identity(const float *T) {
  for (int i = 0; i < 16; ++i)
    if (T[i] != ((i % 5 == 0) ? 1.f : 0.f)) return false;
  return true;
}

void set_identity(float *T) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
}

// host points (strided) -> device float4 (x, y, z, 1), optionally through an index list
// lo / hi (nullable, with guess nullable): the bounding box of the finite rows after `guess`, taken in the same pass
int upload_packed(const float *h, size_t n_rows, size_t stride_bytes, const int32_t *rows, size_t n_sel,
                  DevBuf<float4> &out, hipStream_t s, float *lo = nullptr, float *hi = nullptr,
                  const float *guess = nullptr) {
  const size_t n = rows ? n_sel : n_rows;
  // (not a std::vector: its value-initialisation would write the 16 bytes per point once more)
  std::unique_ptr<float4[]> tmp(new (std::nothrow) float4[n ? n : 1]);
  PCLMI_CHECK(tmp != nullptr, PCLMI_ERR_OOM, "out of host memory");
  const size_t sf = stride_bytes / 4;
  for (size_t i = 0; i < n; ++i) {
    size_t r = rows ? (size_t)rows[i] : i;
    if (r >= n_rows) {
      set_error("index %zu out of range (%zu rows)", r, n_rows);
      return PCLMI_ERR_INVALID;
    }
    const float *p = h + r * sf;
    tmp[i] = make_float4(p[0], p[1], p[2], 1.f);
    if (lo) {
      if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
      float q[3] = {p[0], p[1], p[2]};
      if (guess)
        for (int d = 0; d < 3; ++d)
          q[d] = guess[d * 4 + 0] * p[0] + guess[d * 4 + 1] * p[1] + guess[d * 4 + 2] * p[2] + guess[d * 4 + 3];
      for (int d = 0; d < 3; ++d) {
        lo[d] = std::min(lo[d], q[d]);
        hi[d] = std::max(hi[d], q[d]);
      }
    }
  }
  RC(out.alloc(n));
  if (n) PCLMI_HIP(hipMemcpyAsync(out.p, tmp.get(), n * sizeof(float4), hipMemcpyHostToDevice, s));
  PCLMI_HIP(hipStreamSynchronize(s));  // tmp goes out of scope
  return PCLMI_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
struct pclmi_icp_ctx {
  const pclmi_tree *tree = nullptr;
  size_t n = 0;
  DevBuf<float4> pts;     // source points, Morton-cell order, transformed in place every step
  DevBuf<int> order;      // slot -> original row
  DevBuf<int> inv;        // original row -> slot
  DevBuf<int> nn;         // per slot: index of the nearest target row (-1: not searchable)
  DevBuf<int> nn_slot;    // per slot: bucket slot of that row (start bound of the next step's search)
  DevBuf<float> dist;     // per slot: min(d2, thr2) of the previous step (correspondence_distances_)
  DevBuf<double> partials;
  DevBuf<double> sums;
  DevBuf<double> gpart;          // group sums of the step kernel's tail
  DevBuf<unsigned int> tickets;  // [groups] group tickets, then the launch ticket
  DevBuf<unsigned long long> loopbox;  // the resident loop's rows of tagged granules (allocated at its first launch)
  size_t loop_hold = 0;  // workgroups this context has reserved on its device for resident launches (loop_reserve)
  unsigned groups = 1;
  pclmi_comm *comm = nullptr;    // exchange partner of pclmi_icp_run_dev (pclmi_icp_set_comm)
  double origin[3] = {0, 0, 0};
  bool pending = false;   // a transform waits to be applied by the next step / by end()
  float T[16];
  hipStream_t stream = nullptr;
  hipStream_t user_stream = nullptr;  // stream of the last pclmi_icp_step_dev, when not ours
  double last_thr2 = 0.0;
  bool stepped = false;       // nn holds the neighbours of a previous step
  DevBuf<IcpDevState> state;  // device-resident loop state (pclmi_icp_solve_dev)
  bool dev_mode = false;      // the transform chain lives in `state`, not in T / pending
  // sharded contexts (pclmi_icp_begin_sharded): this rank's contiguous run of the Hilbert-ordered source
  bool sharded = false;
  size_t n_full = 0;                // rows of the whole source
  std::vector<int> shard_rows;      // slot -> row of the whole source
  // point-to-plane loop (pclmi_icp_set_point_to_plane)
  bool plane = false;
  DevBuf<float4> tn_rows;           // target normals by cloud row (RANSAC refit)
  DevBuf<float4> tn_slot;           // target normals by bucket slot (step kernel)
  DevBuf<unsigned char> src_skip;   // per slot: the source normal is not finite (allocated only when one is)
  bool has_skip = false;
  // rescaling point representation (pclmi_icp_set_search_scale): the tree holds scale * target point
  bool scaled = false;
  float scale[3] = {1.f, 1.f, 1.f};
  DevBuf<float4> raw_tgt;  // the target's own rows, for the estimator
  int nsums() const { return plane ? PCLMI_ICP_NSUMS_PLANE : PCLMI_ICP_NSUMS; }
  int i_pairs() const { return plane ? 27 : 0; }
  int i_fit() const { return plane ? 29 : 17; }
  int i_points() const { return plane ? 30 : 18; }
};

extern "C" {

int pclmi_transform(float *xyz, size_t n, size_t stride_bytes, const float T[16]) {
  PCLMI_CHECK(T != nullptr && (n == 0 || xyz != nullptr), PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad stride %zu", stride_bytes);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  if (n == 0) return PCLMI_OK;
  DevBuf<float4> d;
  RC(upload_packed(xyz, n, stride_bytes, nullptr, 0, d, nullptr));
  launch_transform(d.p, n, make_xf(T), nullptr);
  std::vector<float4> h(n);
  PCLMI_HIP(hipMemcpy(h.data(), d.p, n * sizeof(float4), hipMemcpyDeviceToHost));
  const size_t sf = stride_bytes / 4;
  for (size_t i = 0; i < n; ++i) {
    xyz[i * sf + 0] = h[i].x;
    xyz[i * sf + 1] = h[i].y;
    xyz[i * sf + 2] = h[i].z;
  }
  return PCLMI_OK;
}

int pclmi_estimate_rigid_svd(const float *src, size_t n_src, size_t src_stride_bytes, const int32_t *src_idx,
                             const float *tgt, size_t n_tgt, size_t tgt_stride_bytes, const int32_t *tgt_idx,
                             size_t n, float T[16]) {
  PCLMI_CHECK(src && tgt && T, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(src_stride_bytes >= 12 && src_stride_bytes % 4 == 0 && tgt_stride_bytes >= 12 &&
                  tgt_stride_bytes % 4 == 0,
              PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  set_identity(T);
  PCLMI_CHECK(n > 0, PCLMI_ERR_NOT_ENOUGH, "no point pairs");
  PCLMI_CHECK(src_idx || n <= n_src, PCLMI_ERR_INVALID, "n exceeds the source cloud");
  PCLMI_CHECK(tgt_idx || n <= n_tgt, PCLMI_ERR_INVALID, "n exceeds the target cloud");
  DevBuf<float4> ds, dt;
  RC(upload_packed(src, n_src, src_stride_bytes, src_idx, n, ds, nullptr));
  RC(upload_packed(tgt, n_tgt, tgt_stride_bytes, tgt_idx, n, dt, nullptr));
  // moments are taken about the first target point (keeps the one-pass sums well conditioned)
  const float *t0 = tgt + (size_t)(tgt_idx ? tgt_idx[0] : 0) * (tgt_stride_bytes / 4);
  const double origin[3] = {t0[0], t0[1], t0[2]};
  DevBuf<double> partials, sums;
  RC(partials.alloc((size_t)pair_reduce_blocks(n) * PCLMI_ICP_NSUMS));
  RC(sums.alloc(PCLMI_ICP_NSUMS));
  launch_pair_reduce(ds.p, nullptr, dt.p, nullptr, nullptr, n, origin, partials.p, sums.p, nullptr);
  double h[PCLMI_ICP_NSUMS];
  PCLMI_HIP(hipMemcpy(h, sums.p, sizeof(h), hipMemcpyDeviceToHost));
  return solve_rigid_from_sums(h, T);
}

int pclmi_estimate_rigid_svd_dev(const float *d_src, size_t n_src, size_t src_stride_bytes, const float *d_tgt,
                                 size_t n_tgt, size_t tgt_stride_bytes, const pclmi_correspondence *d_corr, size_t n,
                                 float T[16], void *stream) {
  PCLMI_CHECK(d_src && d_tgt && T, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(src_stride_bytes >= 12 && src_stride_bytes % 4 == 0 && tgt_stride_bytes >= 12 &&
                  tgt_stride_bytes % 4 == 0,
              PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  set_identity(T);
  PCLMI_CHECK(n > 0 && d_corr, PCLMI_ERR_NOT_ENOUGH, "no point pairs");
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 correspondences");
  (void)n_src;
  (void)n_tgt;
  hipStream_t s = (hipStream_t)stream;
  // partial sums + the 24 results: a small pooled block (the result-buffer cache recycles it from call to call)
  const size_t bytes = ((size_t)pair_reduce_blocks(n) + 1) * PCLMI_ICP_NSUMS * sizeof(double);
  double *buf = (double *)dev_cache_get(bytes);
  if (!buf) return PCLMI_ERR_OOM;
  double *sums = buf + (size_t)pair_reduce_blocks(n) * PCLMI_ICP_NSUMS;
  launch_corr_reduce(d_src, src_stride_bytes / 4, d_tgt, tgt_stride_bytes / 4, d_corr, n, buf, sums, s);
  double h[PCLMI_ICP_NSUMS];
  hipError_t e = hipMemcpyAsync(h, sums, sizeof(h), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  dev_cache_put(buf);
  if (e != hipSuccess) return hip_fail(e, "rigid estimate", __FILE__, __LINE__);
  return solve_rigid_from_sums(h, T);
}

// ---- correspondence estimation -------------------------------------------------------------
static int nn1_of(const pclmi_tree *tree, const float *pts, size_t n_rows, size_t stride_bytes,
                  const int32_t *rows, size_t n_sel, std::vector<int32_t> &idx, std::vector<float> &d2) {
  const size_t n = rows ? n_sel : n_rows;
  idx.assign(n, -1);
  d2.assign(n, 0.f);
  if (n == 0 || tree->n == 0) return PCLMI_OK;
  if (!rows) return pclmi_knn(tree, pts, n, stride_bytes, 1, idx.data(), d2.data(), nullptr);  // pooled staging
  // selected rows: gather on the host, then the same path
  std::vector<float> sel(n * 3);
  const size_t sf = stride_bytes / 4;
  for (size_t i = 0; i < n; ++i) {
    const size_t r = (size_t)rows[i];
    PCLMI_CHECK(r < n_rows, PCLMI_ERR_INVALID, "index %zu out of range (%zu rows)", r, n_rows);
    const float *p = pts + r * sf;
    sel[i * 3 + 0] = p[0];
    sel[i * 3 + 1] = p[1];
    sel[i * 3 + 2] = p[2];
  }
  return pclmi_knn(tree, sel.data(), n, 12, 1, idx.data(), d2.data(), nullptr);
}

int pclmi_correspondences(const pclmi_tree *target, const float *src, size_t n_src, size_t stride_bytes,
                          const int32_t *src_indices, size_t n_indices, double max_distance,
                          pclmi_correspondence *out, size_t *n_out) {
  PCLMI_CHECK(target && n_out && (out || n_src == 0), PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(n_src == 0 || src, PCLMI_ERR_INVALID, "src is NULL");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad stride %zu", stride_bytes);
  *n_out = 0;
  std::vector<int32_t> idx;
  std::vector<float> d2;
  RC(nn1_of(target, src, n_src, stride_bytes, src_indices, n_indices, idx, d2));
  const double max2 = max_distance * max_distance;
  size_t c = 0;
  for (size_t i = 0; i < idx.size(); ++i) {
    if (idx[i] < 0) continue;               // empty tree / non-finite query
    if ((double)d2[i] > max2) continue;     // correspondence_estimation.hpp:108 (non-strict keep)
    out[c].index_query = src_indices ? src_indices[i] : (int32_t)i;
    out[c].index_match = idx[i];
    out[c].distance = d2[i];
    ++c;
  }
  *n_out = c;
  return PCLMI_OK;
}

int pclmi_correspondences_reciprocal(const pclmi_tree *target, const pclmi_tree *source_tree, const float *src,
                                     size_t n_src, size_t src_stride_bytes, const int32_t *src_indices,
                                     size_t n_indices, const float *tgt, size_t n_tgt, size_t tgt_stride_bytes,
                                     double max_distance, pclmi_correspondence *out, size_t *n_out) {
  PCLMI_CHECK(target && source_tree && n_out && tgt, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(n_src == 0 || (src && out), PCLMI_ERR_INVALID, "src/out is NULL");
  PCLMI_CHECK(src_stride_bytes >= 12 && src_stride_bytes % 4 == 0 && tgt_stride_bytes >= 12 &&
                  tgt_stride_bytes % 4 == 0,
              PCLMI_ERR_INVALID, "bad stride");
  *n_out = 0;
  std::vector<int32_t> idx, ridx;
  std::vector<float> d2, rd2;
  RC(nn1_of(target, src, n_src, src_stride_bytes, src_indices, n_indices, idx, d2));
  const double max2 = max_distance * max_distance;
  // second search: the matched target points back into the source tree (:187)
  std::vector<int32_t> rows;
  std::vector<size_t> who;
  for (size_t i = 0; i < idx.size(); ++i) {
    if (idx[i] < 0 || (double)d2[i] > max2) continue;
    PCLMI_CHECK((size_t)idx[i] < n_tgt, PCLMI_ERR_INVALID, "tgt does not match the target tree");
    rows.push_back(idx[i]);
    who.push_back(i);
  }
  RC(nn1_of(source_tree, tgt, n_tgt, tgt_stride_bytes, rows.data(), rows.size(), ridx, rd2));
  size_t c = 0;
  for (size_t a = 0; a < who.size(); ++a) {
    const size_t i = who[a];
    const int32_t q = src_indices ? src_indices[i] : (int32_t)i;
    if (ridx[a] < 0 || (double)rd2[a] > max2 || ridx[a] != q) continue;  // :188
    out[c].index_query = q;
    out[c].index_match = idx[i];
    out[c].distance = d2[i];
    ++c;
  }
  *n_out = c;
  return PCLMI_OK;
}

// ---- sharded ICP steps ---------------------------------------------------------------------
int pclmi_icp_begin(const pclmi_tree *target, const float *src, size_t n_src, size_t stride_bytes,
                    const float *guess, pclmi_icp_ctx **out) {
  return pclmi_icp_begin_sharded(target, src, n_src, stride_bytes, guess, 0, 1, out);
}

// The whole source is ordered along the Hilbert curve once (every rank computes the same order), then cut into
// `world` contiguous runs: a rank's shard is spatially compact, so its searches touch one part of the target tree
// (SURVEY 8e).  world == 1: the plain context.
int pclmi_icp_begin_sharded(const pclmi_tree *target, const float *src, size_t n_src, size_t stride_bytes,
                            const float *guess, int rank, int world, pclmi_icp_ctx **out) {
  PCLMI_CHECK(target && out, PCLMI_ERR_INVALID, "NULL argument");
  *out = nullptr;
  PCLMI_CHECK(world >= 1 && rank >= 0 && rank < world, PCLMI_ERR_INVALID, "rank %d of %d", rank, world);
  PCLMI_CHECK(n_src == 0 || src, PCLMI_ERR_INVALID, "src is NULL");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad stride %zu", stride_bytes);
  PCLMI_CHECK(n_src < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 source points");
  PCLMI_CHECK(target->n > 0, PCLMI_ERR_INVALID, "the target tree is empty");
  PCLMI_HIP(hipSetDevice(target->device));
  (void)tree_ensure_cert(target);  // once per tree; without the certificates the step walks for every point (same results)
  pclmi_icp_ctx *c = new pclmi_icp_ctx();
  c->tree = target;
  c->n = n_src;
  c->stream = target->stream;
  set_identity(c->T);
  auto fail = [&](int code) {
    delete c;
    return code;
  };
  int rc;
  // output <- input with w = 1 (registration.hpp:149-163), then the guess (icp.hpp:60-66)
  DevBuf<float4> raw;
  // Morton-cell order over the source's own bounding box (taken in the upload's pass over the points)
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if ((rc = upload_packed(src, n_src, stride_bytes, nullptr, 0, raw, c->stream, lo, hi,
                          (guess && !is_identity(guess)) ? guess : nullptr)))
    return fail(rc);
  if (guess && !is_identity(guess)) launch_transform(raw.p, n_src, make_xf(guess), c->stream);
  if (!(lo[0] <= hi[0])) lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.f;
  if ((rc = c->pts.alloc(n_src)) || (rc = c->order.alloc(n_src)) || (rc = c->inv.alloc(n_src)) ||
      (rc = c->nn.alloc(n_src)) || (rc = c->nn_slot.alloc(n_src)) || (rc = c->dist.alloc(n_src)) ||
      (rc = c->partials.alloc((size_t)std::max(icp_step_blocks(n_src), pair_reduce_blocks(n_src)) * kIcpMaxSums)) ||
      (rc = c->sums.alloc(kIcpMaxSums)))
    return fail(rc);
  c->groups = icp_step_groups(n_src);
  if ((rc = c->gpart.alloc((size_t)c->groups * kIcpMaxSums)) || (rc = c->tickets.alloc((size_t)c->groups * kTicketStride + 1))) return fail(rc);
  if (hipMemsetAsync(c->tickets.p, 0, ((size_t)c->groups * kTicketStride + 1) * sizeof(unsigned int), c->stream) != hipSuccess)
    return fail(hip_fail(hipGetLastError(), "memset", __FILE__, __LINE__));
  {
    WsGuard g(target, c->stream);
    const int32_t *d_order = nullptr;
    // stable: the slot order fixes the summation order of the fp64 moments, so it must not vary between runs
    if ((rc = launch_morton_order(target, g.w, (const float *)raw.p, n_src, 4, lo, hi, c->stream, &d_order, true)))
      return fail(rc);
    if (d_order) {
      if (hipMemcpyAsync(c->order.p, d_order, n_src * 4, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
        return fail(hip_fail(hipGetLastError(), "order copy", __FILE__, __LINE__));
    } else {
      launch_iota(c->order.p, n_src, c->stream);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess)
      return fail(hip_fail(hipGetLastError(), "sync", __FILE__, __LINE__));
  }
  c->n_full = n_src;
  if (world > 1) {
    // this rank's contiguous run of the ordered source
    const size_t per = (n_src + (size_t)world - 1) / (size_t)world;
    const size_t lo = std::min(n_src, per * (size_t)rank), hi = std::min(n_src, lo + per);
    c->sharded = true;
    c->n = hi - lo;
    c->shard_rows.resize(c->n);
    if (c->n) {
      // on the context's stream (a non-blocking stream is not ordered against the null stream's copies)
      if (hipMemcpyAsync(c->shard_rows.data(), c->order.p + lo, c->n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
          hipStreamSynchronize(c->stream) != hipSuccess ||
          (lo && hipMemcpyAsync(c->order.p, c->order.p + lo, c->n * 4, hipMemcpyDeviceToDevice, c->stream) != hipSuccess))
        return fail(hip_fail(hipGetLastError(), "shard cut", __FILE__, __LINE__));
    }
    c->groups = icp_step_groups(c->n);  // buffers were sized for the whole source: large enough
  }
  launch_permute(raw.p, c->order.p, c->n, c->pts.p, c->stream);
  if (c->sharded) {
    // inv[row of the whole source] = this rank's slot, -1 for the other ranks' rows (the sharded RANSAC step: the
    // owner of a sampled row contributes its coordinates)
    std::vector<int> h_inv(n_src ? n_src : 1, -1);
    for (size_t s = 0; s < c->n; ++s) h_inv[c->shard_rows[s]] = (int)s;
    if (n_src && (hipMemcpyAsync(c->inv.p, h_inv.data(), n_src * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                  hipStreamSynchronize(c->stream) != hipSuccess))
      return fail(hip_fail(hipGetLastError(), "inv upload", __FILE__, __LINE__));
  }
  // inv[order[slot]] = slot (RANSAC pair lists)
  if (!c->sharded) launch_inverse_permutation(c->order.p, n_src, c->inv.p, c->stream);
  if (c->n && hipMemsetAsync(c->dist.p, 0, c->n * 4, c->stream) != hipSuccess)  // previous distances start at 0
    return fail(hip_fail(hipGetLastError(), "memset", __FILE__, __LINE__));
  // moments are taken about the centre of the target's bounding box: identical on every rank
  for (int d = 0; d < 3; ++d) c->origin[d] = 0.5 * ((double)target->bb_lo[d] + (double)target->bb_hi[d]);
  {
    IcpDevState h;
    std::memset(&h, 0, sizeof(h));
    set_identity(h.T);
    set_identity(h.final_T);
    if (guess && !is_identity(guess)) std::memcpy(h.final_T, guess, sizeof(h.final_T));  // icp.hpp:63
    h.max_iterations = INT32_MAX;  // raw stepping: no criterion fires until pclmi_icp_set_criteria
    h.min_correspondences = 1;
    h.transformation_epsilon = 0.0;
    h.euclidean_fitness_epsilon = -std::numeric_limits<double>::max();
    if ((rc = c->state.alloc(1))) return fail(rc);
    if (hipMemcpyAsync(c->state.p, &h, sizeof(h), hipMemcpyHostToDevice, c->stream) != hipSuccess)
      return fail(hip_fail(hipGetLastError(), "state upload", __FILE__, __LINE__));
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess)
    return fail(hip_fail(hipGetLastError(), "sync", __FILE__, __LINE__));
  *out = c;
  return PCLMI_OK;
}

// Resident launches of ONE process must fit on the chip TOGETHER: two of them that each hold part of the wavefront
// slots and wait for workgroups that cannot start would stall each other until their bounded waits run out.  A
// context reserves its workgroups before its first resident launch and gives them back when the host has waited for
// its stream (pclmi_icp_state / pclmi_icp_end); a context that does not get them runs one launch per iteration.
static std::atomic<long long> g_loop_blocks[64];
// The resident loop is USED up to this many workgroups (16k points).  It was built when a launch per iteration cost a
// 125k-point shard 8 us more than the kernel ran; since the step's arrival tickets sit in address ranges of their own
// (kTicketStride) and most points skip the walk, one launch per iteration is as fast or faster from ~30k points up
// (250k: 22.1 us against 26.0 resident, 125k 18.7 / 19.0, 31k 16.8 / 17.5) and only the smallest shards still gain
// (15.6k: 15.3 / 13.9, 7.8k 14.0 / 13.1, 977 points 12.3 / 12.0).  PCLMI_ICP_LOOP_MAX=<workgroups> moves the limit
// (the tests run with 4096: every size the loop fits).  Read on every call: a test may change it.
static size_t loop_prefer_blocks() {
  const char *e = getenv("PCLMI_ICP_LOOP_MAX");
  return e ? (size_t)atoll(e) : (size_t)64;
}
static bool loop_reserve(pclmi_icp_ctx *c) {
  if (c->loop_hold) return true;
  const size_t need = icp_step_blocks(c->n);
  if (need > loop_prefer_blocks()) return false;
  const size_t cap = icp_loop_capacity(c->tree, c->plane);
  const int dev = c->tree->device & 63;
  if (need > cap) return false;
  const long long before = g_loop_blocks[dev].fetch_add((long long)need);
  if ((size_t)before + need > cap) {
    g_loop_blocks[dev].fetch_sub((long long)need);
    return false;
  }
  c->loop_hold = need;
  return true;
}
static void loop_release(pclmi_icp_ctx *c) {
  if (!c->loop_hold) return;
  g_loop_blocks[c->tree->device & 63].fetch_sub((long long)c->loop_hold);
  c->loop_hold = 0;
}

// fused: the launch's last workgroup also exchanges the sums with the other ranks (when a communicator is attached)
// and runs the solve; otherwise it only leaves the rank's sums in d_sums
static int icp_step_enqueue(pclmi_icp_ctx *c, double max_correspondence_distance, double *d_sums, void *stream,
                            bool fused, unsigned char *d_keep_slot = nullptr, float4 *d_tq = nullptr,
                            int loop_iters = 1) {
  PCLMI_CHECK(c && d_sums, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!(fused && c->comm && comm_dead(c->comm)), PCLMI_ERR_COMM,
              "an earlier exchange of this communicator timed out: destroy it and create a new one on every rank");
  PCLMI_HIP(hipSetDevice(c->tree->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  c->user_stream = stream ? s : nullptr;
  const double thr2 = max_correspondence_distance * max_correspondence_distance;  // icp.hpp:57
  c->last_thr2 = thr2;
  IcpTail tail;
  tail.gpart = c->gpart.p;
  tail.gticket = c->tickets.p;
  tail.ticket = c->tickets.p + (size_t)c->groups * kTicketStride;
  tail.sums_out = d_sums;
  tail.st = c->dev_mode ? c->state.p : nullptr;
  tail.solve = (fused && !d_keep_slot) ? 1 : 0;  // the RANSAC loop's step: the refit launch solves
  tail.ngroups = (int)c->groups;
  if (fused && c->comm) tail.comm = comm_view(c->comm);
  if (loop_iters > 1) {
    if (!c->loopbox.p) {  // zeros: no tag of a launch (iteration numbers from 1) matches
      const size_t words = icp_loop_box_words(c->n);
      RC(c->loopbox.alloc(words));
      PCLMI_HIP(hipMemsetAsync(c->loopbox.p, 0, words * sizeof(unsigned long long), s));
    }
    tail.loopbox = c->loopbox.p;
  }
  launch_icp_step(c->tree, c->pts.p, c->n, c->pending, make_xf(c->T), thr2, c->nn.p, c->nn_slot.p, c->dist.p, c->origin,
                  c->partials.p, s, c->dev_mode ? c->state.p : nullptr, c->stepped, tail,
                  c->plane ? c->tn_slot.p : nullptr, c->has_skip ? c->src_skip.p : nullptr, d_keep_slot, d_tq,
                  c->scaled ? c->raw_tgt.p : nullptr, c->scaled ? c->scale : nullptr, loop_iters);
  c->pending = false;
  c->stepped = true;
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

int pclmi_icp_step_dev(pclmi_icp_ctx *c, double max_correspondence_distance, double *d_sums, void *stream) {
  return icp_step_enqueue(c, max_correspondence_distance, d_sums, stream, false);
}

// h_sums: c->nsums() doubles
static int icp_step_host(pclmi_icp_ctx *c, double max_correspondence_distance, double *h_sums) {
  RC(pclmi_icp_step_dev(c, max_correspondence_distance, c->sums.p, nullptr));
  PCLMI_HIP(hipMemcpyAsync(h_sums, c->sums.p, (size_t)c->nsums() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PCLMI_HIP(hipStreamSynchronize(c->stream));
  return PCLMI_OK;
}

int pclmi_icp_step(pclmi_icp_ctx *c, double max_correspondence_distance, double h_sums[PCLMI_ICP_NSUMS]) {
  PCLMI_CHECK(c && h_sums, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->plane, PCLMI_ERR_UNSUPPORTED,
              "a point-to-plane context produces PCLMI_ICP_NSUMS_PLANE sums: use pclmi_icp_step_dev / pclmi_icp_run_dev");
  return icp_step_host(c, max_correspondence_distance, h_sums);
}

// TransformationEstimationPointToPlaneLLS inside the loop (pclmi.h "Point-to-plane ICP")
int pclmi_icp_set_point_to_plane(pclmi_icp_ctx *c, const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes,
                                 const float *src_normals, size_t n_src, size_t sn_stride_bytes) {
  PCLMI_CHECK(c && tgt_normals, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->stepped && !c->dev_mode, PCLMI_ERR_INVALID, "the estimator must be chosen before the first step");
  PCLMI_CHECK(tn_stride_bytes >= 12 && tn_stride_bytes % 4 == 0 && (!src_normals || (sn_stride_bytes >= 12 && sn_stride_bytes % 4 == 0)),
              PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(n_tgt >= c->tree->rows, PCLMI_ERR_INVALID, "%zu target normals for a target cloud of %zu rows", n_tgt,
              (size_t)c->tree->rows);
  PCLMI_CHECK(!src_normals || n_src == c->n_full, PCLMI_ERR_INVALID, "%zu source normals for a source of %zu rows", n_src,
              c->n_full);
  PCLMI_HIP(hipSetDevice(c->tree->device));
  RC(upload_packed(tgt_normals, c->tree->rows, tn_stride_bytes, nullptr, 0, c->tn_rows, c->stream));
  RC(c->tn_slot.alloc((size_t)c->tree->M * kLeafSize));
  launch_normals_to_slots(c->tree, c->tn_rows.p, c->tn_slot.p, c->stream);
  PCLMI_HIP(hipGetLastError());
  c->has_skip = false;
  if (src_normals && c->n) {
    // per slot: is the source normal of that row finite?  (the normals never change: transformPointCloud moves x, y, z)
    std::vector<int> rows(c->n);
    if (c->sharded) rows = c->shard_rows;
    else PCLMI_HIP(hipMemcpyAsync(rows.data(), c->order.p, c->n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PCLMI_HIP(hipStreamSynchronize(c->stream));
    std::vector<unsigned char> skip(c->n, 0);
    const size_t sf = sn_stride_bytes / 4;
    bool any = false;
    for (size_t i = 0; i < c->n; ++i) {
      const float *m = src_normals + (size_t)rows[i] * sf;
      if (!std::isfinite(m[0]) || !std::isfinite(m[1]) || !std::isfinite(m[2])) skip[i] = 1, any = true;
    }
    if (any) {
      RC(c->src_skip.alloc(c->n));
      PCLMI_HIP(hipMemcpyAsync(c->src_skip.p, skip.data(), c->n, hipMemcpyHostToDevice, c->stream));
      c->has_skip = true;
    }
  }
  PCLMI_HIP(hipStreamSynchronize(c->stream));
  c->plane = true;
  return PCLMI_OK;
}

// Registration::setPointRepresentation with a per-dimension rescale (pclmi.h "Point representations in ICP")
int pclmi_icp_set_search_scale(pclmi_icp_ctx *c, const float scale[3], const float *tgt, size_t n_tgt,
                               size_t tgt_stride_bytes) {
  PCLMI_CHECK(c && scale && tgt, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->stepped && !c->dev_mode, PCLMI_ERR_INVALID, "the search scale must be set before the first step");
  PCLMI_CHECK(!c->plane, PCLMI_ERR_UNSUPPORTED, "the point-to-plane loop has no rescaled search");
  PCLMI_CHECK(tgt_stride_bytes >= 12 && tgt_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(n_tgt >= c->tree->rows, PCLMI_ERR_INVALID, "%zu target rows for a tree built from %zu", n_tgt,
              (size_t)c->tree->rows);
  PCLMI_HIP(hipSetDevice(c->tree->device));
  RC(upload_packed(tgt, c->tree->rows, tgt_stride_bytes, nullptr, 0, c->raw_tgt, c->stream));
  for (int d = 0; d < 3; ++d) c->scale[d] = scale[d];
  c->scaled = true;
  return PCLMI_OK;
}

// ---- device-resident iteration -----------------------------------------------------------------
int pclmi_icp_set_criteria(pclmi_icp_ctx *c, const pclmi_icp_params *p) {
  PCLMI_CHECK(c && p, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->dev_mode, PCLMI_ERR_INVALID, "criteria must be set before the first pclmi_icp_solve_dev");
  PCLMI_HIP(hipSetDevice(c->tree->device));
  IcpDevState h;
  PCLMI_HIP(hipMemcpy(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost));
  h.max_iterations = p->max_iterations;
  h.min_correspondences = p->min_correspondences;
  h.transformation_epsilon = p->transformation_epsilon;
  h.euclidean_fitness_epsilon = p->euclidean_fitness_epsilon;
  PCLMI_HIP(hipMemcpy(c->state.p, &h, sizeof(h), hipMemcpyHostToDevice));
  return PCLMI_OK;
}

int pclmi_icp_solve_dev(pclmi_icp_ctx *c, const double *d_sums, void *stream) {
  PCLMI_CHECK(c && d_sums, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->pending, PCLMI_ERR_INVALID,
              "pclmi_icp_solve_dev after pclmi_icp_set_transform: one context is driven either from the host or "
              "from the device");
  PCLMI_HIP(hipSetDevice(c->tree->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  c->user_stream = stream ? s : nullptr;
  c->dev_mode = true;
  launch_icp_solve(d_sums, c->state.p, s, c->plane);
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

int pclmi_icp_run_dev(pclmi_icp_ctx *c, double max_correspondence_distance, int iterations, void *stream) {
  PCLMI_CHECK(c, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->pending, PCLMI_ERR_INVALID,
              "pclmi_icp_run_dev after pclmi_icp_set_transform: one context is driven either from the host or "
              "from the device");
  // ONE launch per iteration: the step kernel's last workgroup sums the block partials, exchanges the sums with the
  // other ranks when a communicator is attached, and runs the solve (icp_kernels.hip icp_step_tail).  A shard whose
  // workgroups are all resident at once runs the iterations inside ONE launch (the resident loop): no launch boundary
  // between iterations.  Same arithmetic in the same order: the results are the per-iteration launches' bit for bit.
  c->dev_mode = true;
  if (iterations > 1 && !c->scaled && loop_reserve(c))
    return icp_step_enqueue(c, max_correspondence_distance, c->sums.p, stream, true, nullptr, nullptr, iterations);
  for (int i = 0; i < iterations; ++i) RC(icp_step_enqueue(c, max_correspondence_distance, c->sums.p, stream, true));
  return PCLMI_OK;
}

// development aid (not in pclmi.h): device wall-clock ticks (100 MHz) of the context's last step launch
int pclmi_debug_icp_ticks(pclmi_icp_ctx *c, unsigned long long out[4]) {
  PCLMI_CHECK(c && out, PCLMI_ERR_INVALID, "NULL argument");
  hipStream_t s = c->user_stream ? c->user_stream : c->stream;
  IcpDevState h;
  PCLMI_HIP(hipMemcpyAsync(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < 4; ++i) out[i] = h.ticks[i];
  return PCLMI_OK;
}

// the same for the walk stamps of a -DPCLMI_ICP_TICKS build (zeros otherwise)
int pclmi_debug_icp_walk(pclmi_icp_ctx *c, unsigned long long out[8]) {
  PCLMI_CHECK(c && out, PCLMI_ERR_INVALID, "NULL argument");
  hipStream_t s = c->user_stream ? c->user_stream : c->stream;
  IcpDevState h;
  PCLMI_HIP(hipMemcpyAsync(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < 8; ++i) out[i] = h.walk[i];
  h.walk[6] = h.walk[7] = 0;  // the launch-wide maxima start afresh
  PCLMI_HIP(hipMemcpyAsync((char *)c->state.p + offsetof(IcpDevState, walk) + 6 * sizeof(unsigned long long), &h.walk[6],
                           2 * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  return PCLMI_OK;
}

int pclmi_icp_set_comm(pclmi_icp_ctx *c, pclmi_comm *comm) {
  PCLMI_CHECK(c, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!comm || comm_device(comm) == c->tree->device, PCLMI_ERR_INVALID,
              "the communicator lives on device %d, the context on device %d", comm ? comm_device(comm) : -1,
              c->tree->device);
  c->comm = (comm && pclmi_comm_size(comm) > 1) ? comm : nullptr;
  return PCLMI_OK;
}

int pclmi_icp_state(pclmi_icp_ctx *c, float final_T[16], pclmi_icp_stats *stats) {
  PCLMI_CHECK(c, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_HIP(hipSetDevice(c->tree->device));
  hipStream_t s = c->user_stream ? c->user_stream : c->stream;
  IcpDevState h;
  PCLMI_HIP(hipMemcpyAsync(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  loop_release(c);  // every resident launch of this context has ended
  if (final_T) std::memcpy(final_T, h.final_T, sizeof(h.final_T));
  if (stats) {
    stats->iterations = h.iterations;
    stats->converged = h.converged;
    stats->last_correspondences = h.last_pairs;
    stats->last_inliers = h.last_inliers;
    stats->fitness_delta = h.fit;
    stats->transform_delta = h.tdelta;
    std::memcpy(stats->last_transform, h.T, sizeof(h.T));
  }
  if (h.status == PCLMI_ERR_COMM) {
    if (c->comm) comm_set_dead(c->comm);
    set_error("the exchange with the other ranks timed out in iteration %d (%s mailbox): the communicator is dead - "
              "destroy it and create a new one on every rank",
              h.iterations + 1, c->comm ? comm_memory(c->comm) : "no");
    return PCLMI_ERR_COMM;
  }
  if (h.status == PCLMI_ERR_NOT_ENOUGH) {
    set_error("not enough correspondences (%d < the configured minimum): relax the threshold parameters",
              h.last_pairs);
    return PCLMI_ERR_NOT_ENOUGH;
  }
  if (h.status == PCLMI_ERR_HIP) {
    set_error("a resident ICP launch gave up waiting for its own workgroups in iteration %d (other work held the GPU "
              "for more than 5 s?): PCLMI_ICP_LOOP=0 runs one launch per iteration",
              h.iterations + 1);
    return PCLMI_ERR_HIP;
  }
  return PCLMI_OK;
}

int pclmi_icp_set_transform(pclmi_icp_ctx *c, const float T[16]) {
  PCLMI_CHECK(c && T, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(!c->dev_mode, PCLMI_ERR_INVALID,
              "pclmi_icp_set_transform after pclmi_icp_solve_dev: one context is driven either from the host or "
              "from the device");
  if (c->pending) {  // two transforms without a step in between: apply the first one now
    PCLMI_HIP(hipSetDevice(c->tree->device));
    if (c->user_stream) PCLMI_HIP(hipStreamSynchronize(c->user_stream));
    launch_transform(c->pts.p, c->n, make_xf(c->T), c->stream);
    PCLMI_HIP(hipStreamSynchronize(c->stream));
  }
  std::memcpy(c->T, T, 16 * sizeof(float));
  c->pending = true;
  return PCLMI_OK;
}

int pclmi_icp_end(pclmi_icp_ctx *c, float *aligned) {
  if (!c) return PCLMI_OK;
  int rc = PCLMI_OK;
  (void)hipSetDevice(c->tree->device);
  if (c->user_stream) (void)hipStreamSynchronize(c->user_stream);  // steps enqueued on the caller's stream
  if (aligned && c->n && c->sharded) {
    // own rows only, at their positions in the whole source (the other rows belong to other ranks)
    if (c->pending) launch_transform(c->pts.p, c->n, make_xf(c->T), c->stream);
    if (c->dev_mode) launch_transform_pending(c->pts.p, c->n, c->state.p, c->stream);
    std::vector<float4> h(c->n);
    hipError_t e = hipMemcpyAsync(h.data(), c->pts.p, c->n * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = hip_fail(e, "aligned download", __FILE__, __LINE__);
    else
      for (size_t i = 0; i < c->n; ++i) std::memcpy(aligned + (size_t)c->shard_rows[i] * 4, &h[i], sizeof(float4));
  } else if (aligned && c->n) {
    if (c->pending) launch_transform(c->pts.p, c->n, make_xf(c->T), c->stream);
    if (c->dev_mode) launch_transform_pending(c->pts.p, c->n, c->state.p, c->stream);
    DevBuf<float4> outb;
    rc = outb.alloc(c->n);
    if (!rc) {
      launch_unpermute(c->pts.p, c->order.p, c->n, outb.p, c->stream);
      hipError_t e = hipMemcpyAsync(aligned, outb.p, c->n * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) rc = hip_fail(e, "aligned download", __FILE__, __LINE__);
    }
  } else {
    (void)hipStreamSynchronize(c->stream);
  }
  if (c->loop_hold) {
    (void)hipStreamSynchronize(c->stream);
    loop_release(c);
  }
  delete c;
  return rc;
}

size_t pclmi_icp_shard_size(const pclmi_icp_ctx *c) { return c ? c->n : 0; }

// ---- RANSAC rejection as invoked from ICP (icp.hpp:113-150) ------------------------------------
// Host control flow restates RandomSampleConsensus::computeModel
// (sample_consensus/include/pcl/sample_consensus/impl/ransac.hpp:44-133) over
// SampleConsensusModelRegistration; every O(N) pass (moments, countWithinDistance,
// selectWithinDistance, the refit sums) runs on the GPU.  Returns the number of inliers, or -1 when
// no model was found (the caller then keeps all pairs, icp.hpp:128-132).
namespace {

// Generic core: pairs (src[sidx[i]], tgt[tidx[i]]), i < cnt.  On success *have = true, Tbest holds the best
// three-point model, d_flags the inlier mask of selectWithinDistance and *n_inliers its population.
struct RansacDev {
  DevBuf<unsigned int> count;
  DevBuf<float4> six;
  DevBuf<double> partials, sums;
  DevBuf<RansacRng> rng, rng0;  // rng0: the freshly seeded state (the seed never changes), set up once
  DevBuf<RansacTrial> trials;
  DevBuf<double> sthr;
  DevBuf<int> shuf;  // shuffled_indices_ (sac_model.h)
  size_t cap_shuf = 0;
  int init(size_t n) {
    if (n > cap_shuf) {
      RC(shuf.alloc(n));
      cap_shuf = n;
    }
    if (count.p) return PCLMI_OK;
    RC(rng.alloc(1));
    RC(rng0.alloc(1));
    launch_ransac_seed(rng0.p, 12345u, nullptr);  // boost::mt19937 seed 12345 (sac_model.h:95)
    PCLMI_HIP(hipStreamSynchronize(nullptr));
    RC(trials.alloc(kRansacBatch));
    RC(sthr.alloc(1));
    RC(count.alloc(1));
    RC(six.alloc(6));
    RC(partials.alloc((size_t)pair_reduce_blocks(n) * PCLMI_ICP_NSUMS));
    RC(sums.alloc(PCLMI_ICP_NSUMS));
    return PCLMI_OK;
  }
};

// The passes over ALL pairs (moments, inlier counts, the inlier mask) may run over a superset of the pair list in
// storage order instead: entry i is (src[i], tgt[tidx[i]]) and exists iff keep[i] (ICP: the source points in the order
// they are stored in, coalesced, where the pair list in the reference's original-source order is a random gather).
// d_flags is then indexed like keep.  The samples are always drawn from the pair list (sidx, tidx, cnt).
struct RansacBulk {
  const unsigned char *keep;
  const float4 *tgt;  // rows the bulk passes read the matched points from (may be another copy than the samples')
  const int *tidx;
  size_t n;
};

int ransac_core(RansacDev &rd, const float4 *src, const int *sidx, const float4 *tgt, const int *tidx, int cnt,
                double inlier_threshold, int max_iterations, unsigned char *d_flags, hipStream_t s, bool *have_out,
                float Tbest[16], unsigned int *n_inliers, int *trials, const RansacBulk *bulk = nullptr) {
  *have_out = false;
  *n_inliers = 0;
  if (trials) *trials = 0;
  set_identity(Tbest);
  if (cnt < 3) return PCLMI_OK;  // getSamples clears the selection, computeModel fails (ransac.hpp:72-76)
  RC(rd.init(bulk ? std::max(bulk->n, (size_t)cnt) : (size_t)cnt));
  const int *b_sidx = bulk ? nullptr : sidx, *b_tidx = bulk ? bulk->tidx : tidx;
  const float4 *b_tgt = bulk ? bulk->tgt : tgt;
  const unsigned char *b_keep = bulk ? bulk->keep : nullptr;
  const size_t b_n = bulk ? bulk->n : (size_t)cnt;
  // sample_dist_thresh_ (sac_model_registration.h:257-279): stays on the device, where the draws use it
  launch_moment_reduce(src, b_sidx, b_n, rd.partials.p, rd.sums.p, s, b_keep);
  launch_ransac_sample_threshold(rd.sums.p, rd.sthr.p, s);

  // The trials, in batches (icp_kernels.hip "RANSAC trials in batches"): the generator (boost::mt19937, seed 12345,
  // sac_model.h:95; rnd() == gen() >> 1), the index shuffle and isSampleGood run on the device, the three-point models
  // of a batch are fitted in parallel and ONE pass over the pairs counts all their inliers; the host consumes the
  // trials in order under the reference's stopping rule (ransac.hpp:71-121).  One round trip per batch instead of two
  // per trial - with few inliers (a first ICP iteration) the reference's rule asks for hundreds of trials.
  launch_iota(rd.shuf.p, (size_t)cnt, s);
  const double thresh = inlier_threshold * inlier_threshold;
  int iterations = 0, best = -INT32_MAX;
  double k = 1.0;
  unsigned skipped = 0;
  const unsigned max_skip = (unsigned)std::min<long long>((long long)max_iterations * 10, 0xffffffffll);
  bool have = false, stop = false;
  RansacTrial ht[kRansacBatch];
  int batch = 2;  // a converged ICP iteration needs ONE trial (99.9 % inliers: k < 1 after it)
  bool first = true;
  while (!stop && iterations < k && skipped < max_skip) {
    launch_ransac_draw(first ? rd.rng0.p : rd.rng.p, rd.rng.p, rd.shuf.p, cnt, src, sidx, tgt, tidx, rd.sthr.p, batch,
                       rd.trials.p, s);
    first = false;
    launch_count_within_batch(src, b_sidx, b_tgt, b_tidx, b_n, thresh, batch, rd.trials.p, s, b_keep);
    PCLMI_HIP(hipMemcpyAsync(ht, rd.trials.p, (size_t)batch * sizeof(RansacTrial), hipMemcpyDeviceToHost, s));
    PCLMI_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < batch && iterations < k && skipped < max_skip; ++b) {
      if (ht[b].status == 2) {  // no good sample in 1000 draws: computeModel gives up (ransac.hpp:72-76)
        stop = true;
        break;
      }
      if (ht[b].status == 1) {  // degenerate sample
        ++skipped;
        continue;
      }
      const unsigned int inl = ht[b].count;
      if ((int)inl > best) {
        best = (int)inl;
        have = true;
        std::memcpy(Tbest, ht[b].T, sizeof(ht[b].T));
        const double w = (double)best / (double)cnt;
        double p_no = 1.0 - std::pow(w, 3.0);
        p_no = std::max(std::numeric_limits<double>::epsilon(), p_no);
        p_no = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no);
        k = std::log(1.0 - 0.99) / std::log(p_no);  // ransac.hpp:102-107
      }
      ++iterations;
      if (iterations > max_iterations) {
        stop = true;
        break;
      }
    }
    const double left = k - (double)iterations;
    batch = left > (double)kRansacBatch - 2.0 ? kRansacBatch : std::max(2, (int)std::ceil(left) + 2);
  }
  if (trials) *trials = iterations;
  if (!have) return PCLMI_OK;
  // selectWithinDistance with the best model (ransac.hpp:124): its population is the count the trial returned (the
  // same predicate over the same pairs), so the mask is enqueued and nobody waits for it here
  launch_count_within(src, b_sidx, b_tgt, b_tidx, b_n, make_xf(Tbest), thresh, rd.count.p, d_flags, s, b_keep);
  *n_inliers = (unsigned int)best;
  *have_out = true;
  return PCLMI_OK;
}

struct RansacScratch {
  DevBuf<int> ids, pair_src, pair_slot, pair_tgt, num;
  DevBuf<unsigned char> keep, flags;
  DevBuf<unsigned char> keep_slot;  // keep in the order the source points are stored in
  DevBuf<unsigned char> flags_pos;  // the inlier mask by position in the pair list (for the observer)
  DevBuf<int> tgt_of_orig;
  DevBuf<unsigned char> tmp;
  size_t tmp_bytes = 0;
  RansacDev rd;
  bool all_kept = false;  // the pair list of the current iteration is the identity (every source point has a pair)
  const int *src_of_pair() const { return all_kept ? ids.p : pair_src.p; }
  const int *tgt_of_pair() const { return all_kept ? tgt_of_orig.p : pair_tgt.p; }
  const int *slot_of_pair(const pclmi_icp_ctx *c) const { return all_kept ? c->inv.p : pair_slot.p; }
};

// step_pairs: the step's pair count (its sums) - the same predicate over the same pairs as the flags below, so the
// length of the pair list is known without waiting for the compaction
int ransac_icp(pclmi_icp_ctx *c, RansacScratch &rs, const pclmi_icp_params *p, int64_t step_pairs, int64_t *n_inliers,
               double *sums_out /* c->nsums() */) {
  *n_inliers = -1;
  const size_t n = c->n;
  hipStream_t s = c->stream;
  const float4 *cloud = (const float4 *)c->tree->d_cloud;
  if (!rs.ids.p) {
    RC(rs.ids.alloc(n));
    RC(rs.pair_src.alloc(n));
    RC(rs.pair_slot.alloc(n));
    RC(rs.pair_tgt.alloc(n));
    RC(rs.num.alloc(1));
    RC(rs.keep.alloc(n));
    RC(rs.keep_slot.alloc(n));
    RC(rs.flags_pos.alloc(n));
    RC(rs.flags.alloc(n));
    RC(rs.tgt_of_orig.alloc(n));
    launch_iota(rs.ids.p, n, s);
    size_t need = 0;
    RC(select_flagged(rs.ids.p, rs.keep.p, n, rs.pair_src.p, rs.num.p, nullptr, 0, &need, s));
    rs.tmp_bytes = need + 256;
    RC(rs.tmp.alloc(rs.tmp_bytes));
  }
  // kept pairs in ORIGINAL source order (the order the reference's index lists have)
  const float4 *rows = (const float4 *)c->tree->d_slot;  // bucket-slot copy of the target (same coordinates)
  launch_corr_flags(c->nn.p, c->pts.p, cloud, c->order.p, n, c->last_thr2, rs.tgt_of_orig.p, rs.keep.p, s,
                    rs.keep_slot.p, rows, c->nn_slot.p);
  const int cnt = (int)step_pairs;
  int cnt_check = -1;  // arrives with ransac_core's first round trip
  // the pair list the samples are drawn from: (source row, its slot, target row) of the kept pairs in original-source
  // order.  When EVERY source point has a pair (a generous distance threshold: the usual state of a converging align)
  // the list is the identity - no compaction, no gathers
  rs.all_kept = (size_t)cnt == n;
  if (rs.all_kept) {
    cnt_check = cnt;
  } else {
    RC(select_flagged(rs.ids.p, rs.keep.p, n, rs.pair_src.p, rs.num.p, rs.tmp.p, rs.tmp_bytes, nullptr, s));
    PCLMI_HIP(hipMemcpyAsync(&cnt_check, rs.num.p, 4, hipMemcpyDeviceToHost, s));
  }
  if (cnt < 3) return PCLMI_OK;
  if (!rs.all_kept) {
    launch_gather_int(c->inv.p, rs.pair_src.p, cnt, rs.pair_slot.p, s);
    launch_gather_int(rs.tgt_of_orig.p, rs.pair_src.p, cnt, rs.pair_tgt.p, s);
  }
  bool have = false;
  float Tbest[16];
  unsigned int inl = 0;
  // the passes over all pairs walk the source points as they are stored (c->pts with their neighbours c->nn and the
  // mask keep_slot): coalesced, where the pair list in original-source order is a random gather; rs.flags comes back
  // indexed the same way
  const RansacBulk bulk{rs.keep_slot.p, rows, c->nn_slot.p, n};
  RC(ransac_core(rs.rd, c->pts.p, rs.slot_of_pair(c), cloud, rs.tgt_of_pair(), cnt, p->ransac_outlier_threshold,
                 p->ransac_iterations, rs.flags.p, s, &have, Tbest, &inl, nullptr, &bulk));
  PCLMI_CHECK(cnt_check == cnt, PCLMI_ERR_HIP, "internal: the step counted %d pairs, the pair list holds %d", cnt,
              cnt_check);
  if (!have) return PCLMI_OK;
  // the sums of the inlier pairs, for the refit (icp.hpp:171 on the *_good index lists)
  if (c->plane)
    launch_pair_reduce_plane(c->pts.p, nullptr, rows, c->nn_slot.p, c->tn_slot.p, c->has_skip ? c->src_skip.p : nullptr,
                             rs.flags.p, n, c->partials.p, c->sums.p, s);
  else
    launch_pair_reduce(c->pts.p, nullptr, rows, c->nn_slot.p, rs.flags.p, n, c->origin, c->partials.p, c->sums.p, s);
  PCLMI_HIP(hipMemcpyAsync(sums_out, c->sums.p, (size_t)c->nsums() * sizeof(double), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  *n_inliers = inl;
  return PCLMI_OK;
}

}  // namespace

namespace {

// ---- RANSAC rejection inside the device-resident loop (icp_kernels.hip "RANSAC loop") ---------------------------
struct RansacLoopBufs {
  DevBuf<RansacLoopState> rl;
  DevBuf<RansacRng> rng0, rng;
  DevBuf<unsigned long long> shuf, keep_bits;
  DevBuf<unsigned int> sb_cnt, sb_prefix;
  DevBuf<RansacTrial> trials;
  DevBuf<unsigned char> keep_slot;
  DevBuf<float4> tq;     // matched target point per slot
  DevBuf<double> sums2;  // the refit launch's sums
  RansacLoopArgs a;
  int init(pclmi_icp_ctx *c, const pclmi_icp_params *p) {
    hipStream_t s = c->stream;
    const size_t n_rows = c->n_full;
    const int nsb = (int)((n_rows + 1023) / 1024);
    RC(rl.alloc(1));
    RC(rng0.alloc(1));
    RC(rng.alloc(1));
    RC(shuf.alloc(n_rows));
    RC(keep_bits.alloc((size_t)nsb * 16));
    RC(sb_cnt.alloc(nsb));
    RC(sb_prefix.alloc(nsb));
    RC(trials.alloc(kRansacBatch));
    RC(keep_slot.alloc(c->n));
    RC(tq.alloc(c->n));
    RC(sums2.alloc(kIcpMaxSums));
    PCLMI_HIP(hipMemsetAsync(rl.p, 0, sizeof(RansacLoopState), s));
    PCLMI_HIP(hipMemsetAsync(shuf.p, 0, n_rows * sizeof(unsigned long long), s));  // tag 0: no entry is current
    launch_ransac_seed(rng0.p, 12345u, s);  // boost::mt19937 seed 12345 (sac_model.h:95)
    a.rl = rl.p;
    a.st = c->state.p;
    a.sums = c->sums.p;
    a.est = c->plane ? 1 : 0;
    a.n = (int)c->n;
    a.n_rows = (int)n_rows;
    a.min_corr = p->min_correspondences;
    a.max_trials = p->ransac_iterations;
    a.inlier_thr2 = p->ransac_outlier_threshold * p->ransac_outlier_threshold;
    a.rng0 = rng0.p;
    a.rng = rng.p;
    a.shuf = shuf.p;
    a.inv = c->inv.p;
    a.pts = c->pts.p;
    a.nn_slot = c->nn_slot.p;
    a.rows = (const float4 *)c->tree->d_slot;
    a.keep_slot = keep_slot.p;
    a.tq = tq.p;
    a.keep_bits = keep_bits.p;
    a.sb_cnt = sb_cnt.p;
    a.sb_prefix = sb_prefix.p;
    a.nsb = nsb;
    a.trials = trials.p;
    a.comm = CommView();
    if (c->comm) {
      a.comm = comm_view(c->comm);
      // the pair mask travels through the communicator's bulk window: two granules per word, per (parity, source rank)
      PCLMI_CHECK((size_t)nsb * 16 * 2 <= a.comm.bulk_cap, PCLMI_ERR_UNSUPPORTED,
                  "a source of %zu rows needs a bulk window of %zu MB for the sharded RANSAC step: set "
                  "PCLMI_COMM_BULK_MB before the communicator is created (or ransac_iterations = 0)",
                  n_rows, ((size_t)nsb * 16 * 2 * 8 * 2 * (size_t)a.comm.world >> 20) + 1);
    }
    return PCLMI_OK;
  }
};

// one iteration of the RANSAC loop, enqueued without waiting: step, first batch of trials, refit + solve
int ransac_loop_enqueue(pclmi_icp_ctx *c, RansacLoopBufs &rb, const pclmi_icp_params *p, int first) {
  hipStream_t s = c->stream;
  if (first) {
    RC(icp_step_enqueue(c, p->max_correspondence_distance, c->sums.p, nullptr, true, rb.keep_slot.p, rb.tq.p));
    launch_ransac_loop_batch(rb.a, 1, 4, s);  // a converged iteration needs ONE trial (99.9 % inliers: k < 1)
  }
  IcpTail tail;
  tail.gpart = c->gpart.p;
  tail.gticket = c->tickets.p;
  tail.ticket = c->tickets.p + (size_t)c->groups * kTicketStride;
  tail.sums_out = rb.sums2.p;
  tail.st = c->state.p;
  tail.solve = 1;
  tail.ngroups = (int)c->groups;
  tail.carry = c->sums.p;
  if (c->comm) tail.comm = comm_view(c->comm);
  if (first)  // counts + refit sums of the first model + stopping rule + (usually) the solve
    launch_ransac_loop_first(rb.a, c->origin, c->partials.p, tail, c->plane ? c->tn_slot.p : nullptr,
                             c->has_skip ? c->src_skip.p : nullptr, s);
  else  // the rule settled in a host-fed batch, or on another model than the first: refit on that model's inliers
    launch_ransac_loop_refit(rb.a, c->tree, 0, c->origin, c->partials.p, tail, c->plane ? c->tn_slot.p : nullptr,
                             c->has_skip ? c->src_skip.p : nullptr, nullptr, s);
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

// the whole align with the rejection step on; the host only reads the state between chunks of iterations and feeds
// further batches of trials to an iteration whose first batch did not satisfy the stopping rule
int ransac_loop_align(pclmi_icp_ctx *c, const pclmi_icp_params *p, float final_T[16], pclmi_icp_stats *st) {
  RansacLoopBufs rb;
  RC(rb.init(c, p));
  RC(pclmi_icp_set_criteria(c, p));
  c->dev_mode = true;
  hipStream_t s = c->stream;
  int chunk = 1;
  for (;;) {
    const int left = std::max(1, p->max_iterations - st->iterations);
    const int now = std::min(chunk, left);
    for (int i = 0; i < now; ++i) RC(ransac_loop_enqueue(c, rb, p, 1));
    IcpDevState h;
    RansacLoopState hr;
    PCLMI_HIP(hipMemcpyAsync(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost, s));
    PCLMI_HIP(hipMemcpyAsync(&hr, rb.rl.p, sizeof(hr), hipMemcpyDeviceToHost, s));
    PCLMI_HIP(hipStreamSynchronize(s));
    const bool stalled = h.hold != 0;
    while (h.hold && !h.done) {  // the iteration that waits: further batches (ransac.hpp:71-121), then the refit
      if (!hr.settled) {
        // as many batches as the rule asks for right now (k only falls as better models turn up: batches past the
        // point where it settles are no-ops), and the refit behind them - one round trip for the whole iteration
        // when the estimate holds, instead of one per batch of 64 trials
        const double want = std::min(hr.k, (double)p->ransac_iterations) - (double)hr.iterations;
        const int batches = (int)std::max(1.0, std::min(16.0, std::ceil(want / (double)kRansacBatch)));
        for (int b = 0; b < batches; ++b) launch_ransac_loop_batch(rb.a, 0, kRansacBatch, s);
        RC(ransac_loop_enqueue(c, rb, p, 0));  // runs only once the rule has settled
        PCLMI_HIP(hipMemcpyAsync(&hr, rb.rl.p, sizeof(hr), hipMemcpyDeviceToHost, s));
        PCLMI_HIP(hipMemcpyAsync(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost, s));
        PCLMI_HIP(hipStreamSynchronize(s));
        continue;
      }
      RC(ransac_loop_enqueue(c, rb, p, 0));
      PCLMI_HIP(hipMemcpyAsync(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost, s));
      PCLMI_HIP(hipStreamSynchronize(s));
      PCLMI_CHECK(h.hold == 0 || h.done, PCLMI_ERR_HIP, "internal: the RANSAC loop's refit did not run");
    }
    const int rc = pclmi_icp_state(c, final_T, st);
    if (rc) return rc;
    if (h.done) break;
    chunk = stalled ? 1 : std::min(8, chunk * 2);
  }
  if (getenv("PCLMI_RANSAC_TICKS")) {  // development aid: where the last draw kernel spent its time (10 ns ticks)
    RansacLoopState hr;
    PCLMI_HIP(hipMemcpy(&hr, rb.rl.p, sizeof(hr), hipMemcpyDeviceToHost));
    fprintf(stderr, "[pclmi] draw kernel ticks: sthr+reset %llu, prefix %llu, rng %llu, shuffle %llu, lookups %llu, "
                    "good+fit %llu, write-back %llu\n",
            hr.dbg[1] - hr.dbg[0], 0ull, hr.dbg[2] - hr.dbg[1], hr.dbg[3] - hr.dbg[2], hr.dbg[4] - hr.dbg[3],
            hr.dbg[5] - hr.dbg[4], hr.dbg[6] - hr.dbg[5]);
    IcpDevState h;
    PCLMI_HIP(hipMemcpy(&h, c->state.p, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[pclmi] last count+refit launch: last workgroup arrived %llu, sums ready +%llu, solved +%llu (ticks "
                    "after its first workgroup started)\n",
            h.ticks[1] - h.ticks[0], h.ticks[2] - h.ticks[1], h.ticks[3] - h.ticks[2]);
  }
  return PCLMI_OK;
}

struct PlaneArgs {
  const float *src_normals;
  size_t sn_stride;
  const float *tgt_normals;
  size_t n_tgt, tn_stride;
};
struct ScaleArgs {  // a rescaling point representation: the tree indexes scale * target point
  const float *scale;
  const float *tgt;
  size_t n_tgt, tgt_stride;
};

// Per-iteration observer (Registration::registerVisualizationCallback, registration.h update_visualizer_, called at
// icp.hpp:182-183): after every iteration the moved source and the pairs the estimator was given go to the host.
struct Observer {
  pclmi_icp_iteration_fn fn = nullptr;
  void *user = nullptr;
  DevBuf<float4> outb;
  DevBuf<int> tgt_of_orig;
  DevBuf<unsigned char> keep;
  std::vector<float> cloud;
  std::vector<int32_t> si, ti;

  // moves the source by the transform that is still pending (so that the next step does not apply it again),
  // downloads it in the caller's row order and calls the observer
  int emit(pclmi_icp_ctx *c, int iteration) {
    hipStream_t s = c->stream;
    if (c->pending) {
      launch_transform(c->pts.p, c->n, make_xf(c->T), s);
      c->pending = false;
    }
    if (c->dev_mode) launch_transform_pending(c->pts.p, c->n, c->state.p, s);
    if (!outb.p) RC(outb.alloc(c->n));
    cloud.resize(c->n * 4);
    launch_unpermute(c->pts.p, c->order.p, c->n, outb.p, s);
    PCLMI_HIP(hipMemcpyAsync(cloud.data(), outb.p, c->n * sizeof(float4), hipMemcpyDeviceToHost, s));
    PCLMI_HIP(hipStreamSynchronize(s));
    fn(user, iteration, cloud.data(), si.data(), ti.data(), si.size());
    return PCLMI_OK;
  }
  // pairs under the distance threshold of the last step, in source order (call BEFORE the new transform is applied)
  int pairs_from_step(pclmi_icp_ctx *c) {
    hipStream_t s = c->stream;
    if (!keep.p) {
      RC(keep.alloc(c->n));
      RC(tgt_of_orig.alloc(c->n));
    }
    launch_corr_flags(c->nn.p, c->pts.p, (const float4 *)c->tree->d_cloud, c->order.p, c->n, c->last_thr2, tgt_of_orig.p,
                      keep.p, s);
    return pairs_from_flags(c, keep.p, tgt_of_orig.p);
  }
  int pairs_from_flags(pclmi_icp_ctx *c, const unsigned char *d_keep, const int *d_tgt) {
    std::vector<unsigned char> hk(c->n);
    std::vector<int32_t> ht(c->n);
    PCLMI_HIP(hipMemcpyAsync(hk.data(), d_keep, c->n, hipMemcpyDeviceToHost, c->stream));
    PCLMI_HIP(hipMemcpyAsync(ht.data(), d_tgt, c->n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PCLMI_HIP(hipStreamSynchronize(c->stream));
    si.clear();
    ti.clear();
    for (size_t i = 0; i < c->n; ++i)
      if (hk[i]) {
        si.push_back((int32_t)i);
        ti.push_back(ht[i]);
      }
    return PCLMI_OK;
  }
  // the inliers of the RANSAC step: entries of the pair list whose flag is set
  int pairs_from_inliers(pclmi_icp_ctx *c, const int *d_pair_src, const int *d_pair_tgt, const unsigned char *d_flags,
                         size_t cnt) {
    std::vector<unsigned char> hf(cnt);
    std::vector<int32_t> hs(cnt), ht(cnt);
    PCLMI_HIP(hipMemcpyAsync(hf.data(), d_flags, cnt, hipMemcpyDeviceToHost, c->stream));
    PCLMI_HIP(hipMemcpyAsync(hs.data(), d_pair_src, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PCLMI_HIP(hipMemcpyAsync(ht.data(), d_pair_tgt, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PCLMI_HIP(hipStreamSynchronize(c->stream));
    si.clear();
    ti.clear();
    for (size_t i = 0; i < cnt; ++i)
      if (hf[i]) {
        si.push_back(hs[i]);
        ti.push_back(ht[i]);
      }
    return PCLMI_OK;
  }
};

// Registration::align for both estimators, on one GPU or over the ranks of `comm`
int align_core(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src, size_t stride_bytes,
               const float *guess, pclmi_comm *comm, const PlaneArgs *plane, float final_T[16], float *aligned,
               pclmi_icp_stats *stats, pclmi_icp_iteration_fn on_iteration = nullptr, void *user = nullptr,
               const ScaleArgs *scale = nullptr) {
  PCLMI_CHECK(p && target && final_T, PCLMI_ERR_INVALID, "NULL argument");
  Observer obs;
  obs.fn = on_iteration;
  obs.user = user;
  set_identity(final_T);
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    set_identity(stats->last_transform);
  }
  PCLMI_CHECK(n_src > 0 && src, PCLMI_ERR_INVALID, "empty source cloud");
  const int world = comm ? pclmi_comm_size(comm) : 1;
  PCLMI_CHECK(world == 1 || !comm_dead(comm), PCLMI_ERR_COMM,
              "an earlier exchange of this communicator timed out: destroy it and create a new one on every rank");
  PCLMI_CHECK(!(scale && on_iteration), PCLMI_ERR_UNSUPPORTED,
              "the per-iteration observer is not available with a rescaled search");
  PCLMI_CHECK(!on_iteration || world == 1, PCLMI_ERR_UNSUPPORTED,
              "the per-iteration observer sees the whole source: it is not available for a sharded align");
  pclmi_icp_ctx *c = nullptr;
  RC(pclmi_icp_begin_sharded(target, src, n_src, stride_bytes, guess, world > 1 ? pclmi_comm_rank(comm) : 0, world, &c));
  if (guess && !is_identity(guess)) std::memcpy(final_T, guess, 16 * sizeof(float));  // icp.hpp:63
  int rc = PCLMI_OK;
  if (plane)
    rc = pclmi_icp_set_point_to_plane(c, plane->tgt_normals, plane->n_tgt, plane->tn_stride, plane->src_normals, n_src,
                                      plane->sn_stride);
  if (!rc && scale) rc = pclmi_icp_set_search_scale(c, scale->scale, scale->tgt, scale->n_tgt, scale->tgt_stride);
  if (!rc && p->ransac_iterations <= 0) {
    // no RANSAC step: the whole loop runs on the device, ONE kernel per iteration (step, reduction, exchange with the
    // other ranks, solve); the host only looks at the state every few iterations to stop enqueueing once a criterion
    // has fired (iterations enqueued past that point are no-ops on the device).  Every rank enqueues the same chunks:
    // the criteria see identical sums, so all ranks stop in the same iteration.
    rc = pclmi_icp_set_criteria(c, p);
    if (!rc && world > 1) rc = pclmi_icp_set_comm(c, comm);
    pclmi_icp_stats st;
    std::memset(&st, 0, sizeof(st));
    set_identity(st.last_transform);
    int enqueued = 0;
    // an observer is called after every single iteration; the resident loop (pclmi_icp_run_dev) stops by itself
    // (the reservation made here is the one pclmi_icp_run_dev finds; pclmi_icp_state gives it back)
    while (!rc) {
      const bool resident = !obs.fn && !c->scaled && loop_reserve(c);
      const int chunk_max = obs.fn ? 1 : resident ? (1 << 30) : world > 1 ? 16 : 8;
      const int chunk = std::max(1, std::min(chunk_max, p->max_iterations - enqueued));
      const int before = st.iterations;
      if ((rc = pclmi_icp_run_dev(c, p->max_correspondence_distance, chunk, nullptr))) break;
      enqueued += chunk;
      rc = pclmi_icp_state(c, final_T, &st);
      if (obs.fn && !rc && st.iterations > before) {
        if ((rc = obs.pairs_from_step(c))) break;
        if ((rc = obs.emit(c, st.iterations))) break;
      }
      if (st.converged) break;
    }
    if (stats) *stats = st;
    int rc2 = pclmi_icp_end(c, aligned);
    return rc ? rc : rc2;
  }
  if (!rc && !obs.fn) {
    // the reference's default: RANSAC rejection on.  Device-resident as well (icp_kernels.hip "RANSAC loop"), on one
    // GPU or over the ranks of `comm`: every rank draws the same samples (same generator, same pair count), the owner
    // of a sampled row contributes its coordinates, inlier counts and refit sums are exchanged inside the kernels
    if (world > 1) rc = pclmi_icp_set_comm(c, comm);
    pclmi_icp_stats st;
    std::memset(&st, 0, sizeof(st));
    set_identity(st.last_transform);
    if (!rc) rc = ransac_loop_align(c, p, final_T, &st);
    if (stats) *stats = st;
    int rc2 = pclmi_icp_end(c, aligned);
    return rc ? rc : rc2;
  }
  // with an observer the pairs of every iteration go to the host anyway: the host-driven loop
  float T[16], Tprev[16];
  set_identity(T);
  RansacScratch rs;
  int iters = 0, converged = 0;
  double sums[PCLMI_ICP_NSUMS_PLANE];
  double fit = 0.0, tdelta = 0.0;
  int64_t cnt = 0, used = 0;
  while (!rc && !converged) {  // icp.hpp:72
    std::memcpy(Tprev, T, sizeof(T));
    if ((rc = icp_step_host(c, p->max_correspondence_distance, sums))) break;
    cnt = (int64_t)std::llround(sums[c->i_pairs()]);
    if (cnt < p->min_correspondences) {  // icp.hpp:103-108
      set_error("not enough correspondences (%lld < %d): relax the threshold parameters", (long long)cnt,
                p->min_correspondences);
      rc = PCLMI_ERR_NOT_ENOUGH;
      break;
    }
    used = cnt;
    const double *solve_sums = sums;
    double rsums[PCLMI_ICP_NSUMS_PLANE];
    {  // icp.hpp:113-150
      int64_t ninl = -1;
      if ((rc = ransac_icp(c, rs, p, cnt, &ninl, rsums))) break;
      if (ninl >= 0) {
        used = ninl;
        solve_sums = rsums;  // pair sums of the inliers only
      }
      if (obs.fn) {  // the pairs the estimator is about to get (icp.hpp:130-149)
        if (ninl >= 0) launch_gather_u8(rs.flags.p, rs.slot_of_pair(c), (size_t)cnt, rs.flags_pos.p, c->stream);
        rc = ninl >= 0 ? obs.pairs_from_inliers(c, rs.src_of_pair(), rs.tgt_of_pair(), rs.flags_pos.p, (size_t)cnt)
                       : obs.pairs_from_flags(c, rs.keep.p, rs.tgt_of_orig.p);
        if (rc) break;
      }
      if (used < p->min_correspondences) {  // icp.hpp:153-159
        set_error("not enough correspondences after RANSAC (%lld < %d)", (long long)used, p->min_correspondences);
        rc = PCLMI_ERR_NOT_ENOUGH;
        break;
      }
    }
    const double n_points = sums[c->i_points()] > 0 ? sums[c->i_points()] : (double)c->n;
    if ((rc = c->plane ? solve_point_to_plane(solve_sums, T) : solve_rigid_from_sums(solve_sums, T))) break;  // icp.hpp:171
    if ((rc = pclmi_icp_set_transform(c, T))) break;    // icp.hpp:174 (applied by the next step / end)
    mat4_mul_f32(T, final_T, final_T);                  // icp.hpp:177
    ++iters;
    if (obs.fn && (rc = obs.emit(c, iters))) break;     // icp.hpp:182-183
    tdelta = 0.0;
    {
      float fs = 0.f;
      for (int i = 0; i < 16; ++i) fs += std::fabs(T[i] - Tprev[i]);
      tdelta = fs;
    }
    fit = sums[c->i_fit()] / n_points;  // registration.hpp:60-68
    if (iters >= p->max_iterations || tdelta < p->transformation_epsilon ||
        std::fabs(fit) <= p->euclidean_fitness_epsilon)  // icp.hpp:193-196
      converged = 1;
  }
  if (stats) {
    stats->iterations = iters;
    stats->converged = converged;
    stats->last_correspondences = (int)cnt;
    stats->last_inliers = (int)used;
    stats->fitness_delta = fit;
    stats->transform_delta = tdelta;
    std::memcpy(stats->last_transform, T, sizeof(T));
  }
  int rc2 = pclmi_icp_end(c, aligned);
  return rc ? rc : rc2;
}

}  // namespace

int pclmi_icp_align(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src,
                    size_t stride_bytes, const float *guess, float final_T[16], float *aligned,
                    pclmi_icp_stats *stats) {
  return align_core(p, target, src, n_src, stride_bytes, guess, nullptr, nullptr, final_T, aligned, stats);
}

int pclmi_icp_align_scaled(const pclmi_icp_params *p, const pclmi_tree *scaled_target, const float scale[3],
                           const float *tgt, size_t n_tgt, size_t tgt_stride_bytes, const float *src, size_t n_src,
                           size_t stride_bytes, const float *guess, pclmi_comm *comm, float final_T[16], float *aligned,
                           pclmi_icp_stats *stats) {
  PCLMI_CHECK(scale && tgt, PCLMI_ERR_INVALID, "NULL argument");
  const ScaleArgs sa{scale, tgt, n_tgt, tgt_stride_bytes};
  return align_core(p, scaled_target, src, n_src, stride_bytes, guess, comm, nullptr, final_T, aligned, stats, nullptr,
                    nullptr, &sa);
}

int pclmi_icp_align_sharded(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src,
                            size_t stride_bytes, const float *guess, pclmi_comm *comm, float final_T[16],
                            float *aligned, pclmi_icp_stats *stats) {
  PCLMI_CHECK(comm, PCLMI_ERR_INVALID, "NULL argument");
  return align_core(p, target, src, n_src, stride_bytes, guess, comm, nullptr, final_T, aligned, stats);
}

int pclmi_icp_align_point_to_plane(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src,
                                   size_t stride_bytes, const float *src_normals, size_t sn_stride_bytes,
                                   const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes, const float *guess,
                                   pclmi_comm *comm, float final_T[16], float *aligned, pclmi_icp_stats *stats) {
  PCLMI_CHECK(tgt_normals, PCLMI_ERR_INVALID, "tgt_normals is NULL");
  const PlaneArgs pa{src_normals, sn_stride_bytes, tgt_normals, n_tgt, tn_stride_bytes};
  return align_core(p, target, src, n_src, stride_bytes, guess, comm, &pa, final_T, aligned, stats);
}

int pclmi_icp_align_observed(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src,
                             size_t stride_bytes, const float *src_normals, size_t sn_stride_bytes,
                             const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes, const float *guess,
                             pclmi_icp_iteration_fn on_iteration, void *user, float final_T[16], float *aligned,
                             pclmi_icp_stats *stats) {
  PCLMI_CHECK(on_iteration, PCLMI_ERR_INVALID, "on_iteration is NULL");
  const PlaneArgs pa{src_normals, sn_stride_bytes, tgt_normals, n_tgt, tn_stride_bytes};
  return align_core(p, target, src, n_src, stride_bytes, guess, nullptr, tgt_normals ? &pa : nullptr, final_T, aligned,
                    stats, on_iteration, user);
}

// CorrespondenceRejectorSampleConsensus::getRemainingCorrespondences
// (registration/include/pcl/registration/impl/correspondence_rejection_sample_consensus.hpp:44-110)
int pclmi_reject_sample_consensus(const float *src, size_t n_src, size_t src_stride_bytes, const float *tgt,
                                  size_t n_tgt, size_t tgt_stride_bytes, const pclmi_correspondence *in, size_t n,
                                  double inlier_threshold, int max_iterations, pclmi_correspondence *out,
                                  size_t *n_out, float best_T[16]) {
  PCLMI_CHECK(n_out && best_T, PCLMI_ERR_INVALID, "NULL argument");
  *n_out = 0;
  set_identity(best_T);
  PCLMI_CHECK(n == 0 || (in && out && src && tgt), PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(src_stride_bytes >= 12 && src_stride_bytes % 4 == 0 && tgt_stride_bytes >= 12 &&
                  tgt_stride_bytes % 4 == 0,
              PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 correspondences");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  auto keep_all = [&]() {  // computeModel failed or < 3 inliers (:78-95)
    std::memcpy(out, in, n * sizeof(pclmi_correspondence));
    *n_out = n;
    return PCLMI_OK;
  };
  if (n < 3) return n ? keep_all() : PCLMI_OK;
  std::vector<int32_t> si(n), ti(n);
  for (size_t i = 0; i < n; ++i) {
    si[i] = in[i].index_query;
    ti[i] = in[i].index_match;
    PCLMI_CHECK(si[i] >= 0 && (size_t)si[i] < n_src && ti[i] >= 0 && (size_t)ti[i] < n_tgt, PCLMI_ERR_INVALID,
                "correspondence %zu (%d -> %d) is outside the clouds", i, si[i], ti[i]);
  }
  DevBuf<float4> ds, dt;
  RC(upload_packed(src, n_src, src_stride_bytes, nullptr, 0, ds, nullptr));
  RC(upload_packed(tgt, n_tgt, tgt_stride_bytes, nullptr, 0, dt, nullptr));
  DevBuf<int> dsi, dti;
  DevBuf<unsigned char> flags;
  RC(dsi.alloc(n));
  RC(dti.alloc(n));
  RC(flags.alloc(n));
  PCLMI_HIP(hipMemcpy(dsi.p, si.data(), n * 4, hipMemcpyHostToDevice));
  PCLMI_HIP(hipMemcpy(dti.p, ti.data(), n * 4, hipMemcpyHostToDevice));
  RansacDev rd;
  bool have = false;
  unsigned int inl = 0;
  float T[16];
  RC(ransac_core(rd, ds.p, dsi.p, dt.p, dti.p, (int)n, inlier_threshold, max_iterations, flags.p, nullptr, &have, T,
                 &inl, nullptr));
  if (!have || inl < 3) return keep_all();
  std::vector<unsigned char> hf(n);
  PCLMI_HIP(hipMemcpy(hf.data(), flags.p, n, hipMemcpyDeviceToHost));
  size_t c = 0;
  for (size_t i = 0; i < n; ++i)
    if (hf[i]) out[c++] = in[i];
  *n_out = c;
  std::memcpy(best_T, T, sizeof(T));  // the best sample's model coefficients (:103-108)
  return PCLMI_OK;
}

int pclmi_fitness_score(const pclmi_tree *target, const float *src, size_t n_src, size_t stride_bytes,
                        const float T[16], double max_range, double *score) {
  PCLMI_CHECK(target && src && T && score, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad stride %zu", stride_bytes);
  *score = std::numeric_limits<double>::max();
  if (n_src == 0 || target->n == 0) return PCLMI_OK;
  PCLMI_HIP(hipSetDevice(target->device));
  DevBuf<float4> d;
  RC(upload_packed(src, n_src, stride_bytes, nullptr, 0, d, nullptr));
  launch_transform(d.p, n_src, make_xf(T), nullptr);  // registration.hpp:78
  PCLMI_HIP(hipDeviceSynchronize());
  DevBuf<int> di;
  DevBuf<float> dd;
  DevBuf<double> out2;
  RC(di.alloc(n_src));
  RC(dd.alloc(n_src));
  RC(out2.alloc(2));
  RC(pclmi_knn_dev(target, (const float *)d.p, n_src, 16, 1, di.p, dd.p, nullptr));
  launch_fitness(dd.p, n_src, max_range, out2.p, nullptr);
  double h[2];
  PCLMI_HIP(hipMemcpy(h, out2.p, sizeof(h), hipMemcpyDeviceToHost));
  if (h[1] > 0) *score = h[0] / h[1];
  return PCLMI_OK;
}

}  // extern "C"
