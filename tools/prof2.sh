cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nq in 1e6 1.1e6 4e6; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nq$nq -- python3 tools_prof_knn.py 1e6 $nq 16 5 > gpurun_out/prof_nq.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_nq$nq/*/*kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name']; short = 'knn' if 'knn_union' in n else ('morton' if 'morton' in n else n[60:150])
    print('$nq', short[:80], r['Calls'], r['AverageNs']); tot+=float(r['TotalDurationNs'])
print('$nq total per call us', tot/5/1e3)
PY
done
