#!/bin/bash
# rocprofv3 PMC passes over any helper program, one counter group per run (the TCC block holds FETCH_SIZE or
# WRITE_SIZE, not both), reduced per kernel to per-launch averages.
# usage: tools/pmc_kernel.sh NAME "kernel_substring[,kernel_substring...]" python3 tools/<helper>.py args...
# output: gpurun_out/pmc_NAME_<pass>/ and gpurun_out/pmc_NAME.json ({kernel_substring: {counter: mean per launch}})
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
NAME=$1; KERNELS=$2; shift 2
run() {
  pass=$1; shift
  rm -rf gpurun_out/pmc_${NAME}_$pass
  timeout -s KILL 170 rocprofv3 --pmc $COUNTERS --output-format csv -d gpurun_out/pmc_${NAME}_$pass -- "$@" > gpurun_out/pmc_${NAME}_$pass.log 2>&1 || { tail -5 gpurun_out/pmc_${NAME}_$pass.log; exit 1; }
  echo "pass $pass done"
}
COUNTERS="FETCH_SIZE" run fetch "$@" && COUNTERS="WRITE_SIZE" run write "$@" && \
COUNTERS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" run tcc "$@" && \
COUNTERS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" run sq "$@" && \
COUNTERS="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" run sq2 "$@" && \
python3 - "$NAME" "$KERNELS" <<'PY'
import csv, glob, json, sys
name, kernels = sys.argv[1], sys.argv[2].split(",")
out = {}
for kern in kernels:
    acc = {}
    for f in glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]:
                continue
            a = acc.setdefault(r["Counter_Name"], {})
            key = (f, r["Dispatch_Id"])
            a[key] = a.get(key, 0.0) + float(r["Counter_Value"])
    res = {c: sum(v.values()) / len(v) for c, v in acc.items()}
    res["launches_averaged"] = max([len(v) for v in acc.values()] or [0])
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:  # KiB; FETCH doubled per the guide's gfx950 note
        res["hbm_bytes_per_launch_raw"] = (res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
        res["hbm_bytes_per_launch"] = (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res:
        res["l2_hit_rate"] = res["TCC_HIT_sum"] / max(res["TCC_HIT_sum"] + res["TCC_MISS_sum"], 1)
    if res.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in res:
        res["wait_any_over_wave_cycles"] = res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"]
    out[kern] = res
json.dump(out, open("gpurun_out/pmc_%s.json" % name, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
