#!/bin/bash
# 2-rank rehearsal of bench.py's N > 1 path on ONE GPU (both ranks on device 0; torch's barrier over gloo, the ICP
# exchange through pclmi_comm / HIP IPC).  Run ONCE per change; output under gpurun_out/.
mkdir -p gpurun_out
PCLMI_BENCH_REHEARSAL=1 PCLMI_BENCH_WATCHDOG=120 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu "$@" \
  > gpurun_out/reh_r2.json 2> gpurun_out/reh_r2.err
rc=$?
grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/reh_r2.err | tail -30
cat gpurun_out/reh_r2.json
exit $rc
