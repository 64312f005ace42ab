"""Copies the summaries of the last GPU profiling run (gpurun_out/: tools/gpu_bench_prof.sh, tools/pmc_collect.sh 16,
tools/pmc_icp.sh) into profiles/ under this round's names, each stamped with the commit the tree was at.
usage: save_profiles.py r02 [commit]"""
import glob
import json
import os
import shutil
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(
    ["git", "log", "-1", "--format=%h %cI"], cwd=root, text=True).strip()
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def stamp(src, dst, extra=None):
    d = json.load(open(src))
    d["source_commit"] = commit
    if extra:
        d.update(extra)
    json.dump(d, open(dst, "w"), indent=1)
    print("wrote", os.path.relpath(dst, root))


stamp(os.path.join(go, "bench.json"), os.path.join(pr, tag + "_bench_latest.json"))
stamp(os.path.join(go, "pmc_summary_k16.json"), os.path.join(pr, tag + "_knn16_pmc.json"),
      {"collected_by": "tools/pmc_collect.sh 16 (rocprofv3 --pmc, one counter group per run)"})
stamp(os.path.join(go, "pmc_icp_summary.json"), os.path.join(pr, tag + "_icp_step_pmc.json"),
      {"collected_by": "tools/pmc_icp.sh (rocprofv3 --pmc, one counter group per run, 10 iterations averaged)"})
stats = glob.glob(os.path.join(go, "prof_bench", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(pr, tag + "_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(go, "prof_bench", "summary.txt"), os.path.join(pr, tag + "_bench_kernel_stats_summary.txt"))
    print("wrote kernel stats")
