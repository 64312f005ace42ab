"""Copies the summaries of the last GPU profiling run (gpurun_out/: tools/gpu_bench_prof.sh, tools/pmc_all.sh) into
profiles/ under this round's names, each stamped with the commit the tree was at.
usage: save_profiles.py r03 [commit]"""
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


def stamp(d, dst, extra=None):
    d = dict(d)
    d["source_commit"] = commit
    if extra:
        d.update(extra)
    json.dump(d, open(dst, "w"), indent=1)
    print("wrote", os.path.relpath(dst, root))


# (tools/pmc_all.sh's run name, kernel substring) -> file name under profiles/, helper command
PMC = {
    ("knn16", "knn_union_ix_kernel<16"): ("knn16", "prof_knn.py 1e6 1e6 16 3"),
    ("knn32", "knn_union_ix_kernel<32"): ("knn32", "prof_knn.py 1e7 1.25e7 32 2 (10M-point target, one 12.5M-query batch)"),
    ("radius", "radius_union_kernel"): ("radius_union", "prof_radius.py 1e6 1e6 3"),
    ("radius", "prov_sort_kernel"): ("prov_sort", "prof_radius.py 1e6 1e6 3"),
    ("icp", "icp_step_kernel"): ("icp_step", "prof_icp.py 10 (RANSAC off)"),
    ("ransac", "icp_step_kernel"): ("ransac_rs_step", "prof_icp_ransac.py (two default 30-iteration aligns)"),
    ("ransac", "ransac_words_draw_kernel"): ("ransac_words_draw", "prof_icp_ransac.py (pair-mask words + the first batch's draws)"),
    ("ransac", "ransac_draw_loop_kernel"): ("ransac_draw_loop", "prof_icp_ransac.py (the draws of further batches)"),
    ("ransac", "ransac_first_kernel"): ("ransac_first", "prof_icp_ransac.py"),
}
for (run, kern), (name, cmd) in PMC.items():
    src = os.path.join(go, "pmc_%s.json" % run)
    if not os.path.exists(src):
        print("missing", src)
        continue
    d = json.load(open(src)).get(kern)
    if d and d.get("launches_averaged"):
        stamp(d, os.path.join(pr, "%s_%s_pmc.json" % (tag, name)),
              {"kernel": kern, "collected_by": "tools/pmc_kernel.sh (rocprofv3 --pmc, one counter group per run) over tools/" + cmd})
if os.path.exists(os.path.join(go, "bench.json")):
    stamp(json.load(open(os.path.join(go, "bench.json"))), os.path.join(pr, tag + "_bench_latest.json"))
for sub, suffix in (("prof_bench", "_bench_kernel_stats"), ("prof_bench_headline", "_bench_headline_kernel_stats")):
    stats = glob.glob(os.path.join(go, sub, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(pr, tag + suffix + ".csv"))
        txt = open(os.path.join(go, sub, "summary.txt")).read()
        open(os.path.join(pr, tag + suffix + "_summary.txt"), "w").write("# source_commit: %s\n%s" % (commit, txt))
        print("wrote", tag + suffix)
