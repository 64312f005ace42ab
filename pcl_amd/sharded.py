"""Sharded ICP driver: one process per GPU, source points sharded, target tree replicated.

Mirrors the loop of pcl::IterativeClosestPoint::computeTransformation
(registration/include/pcl/registration/impl/icp.hpp:72-209) with RANSAC off: each rank runs the
fused GPU step on its shard (pclmi_icp_step_dev), the 24 fp64 partial sums are combined with ONE
all-reduce per iteration (RCCL when the process group is "nccl"; gloo on CPU in the tests), and every
rank runs the identical host solve (pclmi_icp_solve), so all ranks hold bit-identical transforms with
no broadcast.  torch is plumbing here: the sums buffer and the collective.

icp_align_gpu_sharded_dev is the same loop with NO host round trip per iteration: the solve runs as a
kernel on the stream (pclmi_icp_solve_dev - the same code compiled for the GPU, same bits), the
all-reduce is enqueued between step and solve, and the host looks at the device-side state only every
`check_every` iterations (iterations enqueued after a criterion fired are no-ops on the device).
"""
import numpy as np

from . import capi


def allreduce_sums(local_sums, group=None):
    """Sum of the per-rank fp64 moment vectors over the process group (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_sums
    if isinstance(local_sums, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(local_sums, np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.numpy()
    dist.all_reduce(local_sums, op=dist.ReduceOp.SUM, group=group)
    return local_sums


def mat4_mul_f32(A, B):
    """A * B in fp32 with the library's summation order (icp.hpp:177 final = T * final)."""
    A = np.asarray(A, np.float32)
    B = np.asarray(B, np.float32)
    out = np.zeros((4, 4), np.float32)
    for k in range(4):
        out = (out + A[:, k:k + 1] * B[k:k + 1, :]).astype(np.float32)
    return out


def converged(iteration, T, T_prev, fitness_delta, params):
    """icp.hpp:193-196."""
    tdelta = float(np.abs(T - T_prev).astype(np.float32).sum(dtype=np.float32))
    return (iteration >= params.max_iterations or tdelta < params.transformation_epsilon
            or abs(fitness_delta) <= params.euclidean_fitness_epsilon)


def icp_align_sharded(step_fn, set_transform_fn, params, guess=None, group=None):
    """Runs the ICP loop over shards.

    step_fn() -> this rank's PCLMI_ICP_NSUMS fp64 sums of one iteration (numpy array or torch tensor);
    set_transform_fn(T) hands the solved transform back to the shard.  Returns (status, final_T,
    iterations, is_converged, last_pairs): status 0, or 6 (PCLMI_ERR_NOT_ENOUGH) as icp.hpp:103-108.
    """
    final_T = np.eye(4, dtype=np.float32) if guess is None else np.array(guess, np.float32).reshape(4, 4)
    T = np.eye(4, dtype=np.float32)
    it, done, pairs = 0, False, 0
    while not done:
        T_prev = T
        sums = allreduce_sums(step_fn(), group)
        h = sums.cpu().numpy() if hasattr(sums, "cpu") else np.asarray(sums)
        pairs = int(round(h[0]))
        if pairs < params.min_correspondences:
            return 6, final_T, it, False, pairs
        T = capi.icp_solve(h)
        set_transform_fn(T)
        final_T = mat4_mul_f32(T, final_T)  # icp.hpp:177
        it += 1
        n_points = h[18] if h[18] > 0 else 1.0
        done = converged(it, T, T_prev, h[17] / n_points, params)
    return 0, final_T, it, True, pairs


def icp_align_gpu_sharded(target_tree, src_shard, params, guess=None, group=None, stream=None):
    """The GPU instantiation: `src_shard` is this rank's part of the source cloud."""
    import torch
    ctx = capi.IcpShard(target_tree, src_shard, guess)
    dev = torch.device("cuda", torch.cuda.current_device())
    sums = torch.zeros(capi.ICP_NSUMS, dtype=torch.float64, device=dev)
    # the step, the collective and the download of the sums must be ordered on ONE stream: a caller's raw stream
    # handle becomes torch's current stream for the duration of the loop
    ext = torch.cuda.ExternalStream(stream, device=dev) if stream else None
    with torch.cuda.stream(ext if ext is not None else torch.cuda.current_stream()):
        s = torch.cuda.current_stream().cuda_stream

        def step():
            ctx.step_dev(params.max_correspondence_distance, sums.data_ptr(), s or None)
            if not s:
                torch.cuda.synchronize()  # the library's own stream was used: order it before the collective
            return sums

        rc, T, it, ok, pairs = icp_align_sharded(step, ctx.set_transform, params, guess, group)
        aligned = ctx.end()
    return rc, T, aligned, it, ok, pairs


def icp_align_gpu_sharded_dev(target_tree, src_shard, params, guess=None, group=None, check_every=8,
                              force_collective=False):
    """Device-resident variant of icp_align_gpu_sharded: same results, one host sync per `check_every`
    iterations.  Runs on torch's current stream (the collective is ordered on it too).
    force_collective: issue the all-reduce even in a group of one rank (tests: RCCL between the library's kernels)."""
    import torch
    import torch.distributed as dist
    ctx = capi.IcpShard(target_tree, src_shard, guess)
    ctx.set_criteria(params)
    dev = torch.device("cuda", torch.cuda.current_device())
    sums = torch.zeros(capi.ICP_NSUMS, dtype=torch.float64, device=dev)
    multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
    cur = torch.cuda.current_stream()
    own = None
    if not cur.cuda_stream:
        # the legacy null stream: the library would fall back to its own stream; use an explicit one instead
        own = torch.cuda.Stream(device=dev)
        own.wait_stream(cur)
    with torch.cuda.stream(own if own is not None else cur):
        s = torch.cuda.current_stream().cuda_stream
        enqueued = 0
        while True:
            chunk = max(1, min(check_every, params.max_iterations - enqueued))
            for _ in range(chunk):
                ctx.step_dev(params.max_correspondence_distance, sums.data_ptr(), s)
                if multi:
                    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
                ctx.solve_dev(sums.data_ptr(), s)
            enqueued += chunk
            rc, T, st = ctx.state()
            if rc or st.converged:
                break
        aligned = ctx.end()
    if own is not None:
        cur.wait_stream(own)
    return rc, T, aligned, st.iterations, bool(st.converged), st.last_correspondences
