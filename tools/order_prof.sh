# development: rocprofv3 kernel stats of the ordering kernels and the k=16 kernel for library variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default bt512 bt1024; do
  lib=pcl_amd/lib/libpclmi_$v.so; [ "$v" = default ] && lib=pcl_amd/lib/libpclmi.so
  export PCLMI_LIB=$lib
  rm -rf gpurun_out/prof_o_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_o_$v -- python3 tools/prof_knn.py 1e6 1e6 16 5 > /dev/null 2>&1
  echo $v; python3 tools/stats_summary.py gpurun_out/prof_o_$v | grep -E "knn_union|tile_|bin_sort"
done
