#!/bin/bash
# rocprofv3 SQ counter pass over icp_step_kernel (tools/prof_icp.py); output gpurun_out/pmc_icp_*/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  name=$1; shift
  rm -rf gpurun_out/pmc_icp_$name
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_icp_$name -- python3 tools/prof_icp.py 10 > gpurun_out/pmc_icp_$name.log 2>&1 || { tail -5 gpurun_out/pmc_icp_$name.log; exit 1; }
}
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY && \
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVES && \
run fetch FETCH_SIZE && run write WRITE_SIZE && run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
python3 - <<'PY'
import csv, glob, json
acc = {}
for f in glob.glob("gpurun_out/pmc_icp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "icp_step_kernel" not in r["Kernel_Name"]:
            continue
        a = acc.setdefault(r["Counter_Name"], {})
        key = (f, r["Dispatch_Id"])
        a[key] = a.get(key, 0.0) + float(r["Counter_Value"])
res = {k: sum(v.values()) / len(v) for k, v in acc.items()}
json.dump(res, open("gpurun_out/pmc_icp_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
