#!/bin/bash
# GPU parity tests, one process, per-test timeout; output under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread "$@" > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/gpu_tests.log
exit $rc
