# development: A/B of library variants on the GPU box (timing + one SQ counter pass each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
mkdir -p gpurun_out
L=pcl_amd/lib
SQ="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"
: > gpurun_out/ab.log
for v in "$@"; do
  lib=$L/libpclmi_$v.so; [ "$v" = default ] && lib=$L/libpclmi.so
  PCLMI_LIB=$lib timeout -k 10 200 python3 tools/ab_knn.py 1e6 1e6 ${KS:-8,16,32} >> gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  if [ -n "$PMC" ]; then
    export PCLMI_LIB=$lib
    pass=0
    IFS=';' read -ra SETS <<< "${PMCSETS:-$SQ}"
    for set in "${SETS[@]}"; do
      pass=$((pass + 1))
      rm -rf gpurun_out/pmc_ab_${v}_$pass
      timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_ab_${v}_$pass -- python3 tools/prof_knn.py 1e6 1e6 ${PMCK:-16} 3 > gpurun_out/pmc_ab_${v}_$pass.log 2>&1 || { tail -5 gpurun_out/pmc_ab_${v}_$pass.log; exit 1; }
    done
    unset PCLMI_LIB
    python3 - $v <<'PY' >> gpurun_out/ab.log
import csv, glob, sys, collections
v = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_ab_%s_*/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "knn_union" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(v, {k: round(sum(x) / len(x)) for k, x in sorted(acc.items())})
PY
  fi
done
grep -v -E "amdgpu.ids|Exception ignored|Traceback|pcl_oracle.py|TypeError" gpurun_out/ab.log
