#!/bin/bash
# rocprofv3 PMC passes for the k=16 traversal kernel, one counter group per run (the TCC block
# holds FETCH_SIZE or WRITE_SIZE, not both).  Output: gpurun_out/pmc_<name>/ + pmc_summary.json
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
K=${1:-16}
run() {
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python3 tools/prof_knn.py 1e6 1e6 $K 3 > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; exit 1; }
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD && \
python3 tools/pmc_summary.py gpurun_out knn_union gpurun_out/pmc_summary_k$K.json
