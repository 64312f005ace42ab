#!/bin/bash
# development: A/B of library variants on the ICP numbers (1M bench line + shard times), one GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  lib=pcl_amd/lib/libpclmi_$v.so; [ "$v" = default ] && lib=pcl_amd/lib/libpclmi.so
  echo "== $v"
  PCLMI_LIB=$lib timeout -k 10 200 python3 bench.py --no-extras --no-cpu 2>/dev/null | grep -o "ms_per_iteration\": [0-9.]*\|step_kernel_ms_steady\": [0-9.]*\|step_kernel_ms_first_iteration\": [0-9.]*"
  PCLMI_LIB=$lib WORLDS=4,8,64 timeout -k 10 200 python3 tools/icp_shard_times.py 2>/dev/null | grep world
done
