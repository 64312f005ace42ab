"""Condenses a rocprofv3 --kernel-trace --stats CSV directory into a short table (stdout + <dir>/summary.txt)."""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        short = name.replace("void ", "").replace("pclmi::(anonymous namespace)::", "").split("(")[0]
        if len(short) > 90:
            short = short[:40] + "..." + short[-45:]
        rows.append((float(r["TotalDurationNs"]), short, int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"])))
rows.sort(reverse=True)
lines = ["%-92s %8s %14s %8s" % ("kernel", "calls", "avg_us", "pct")]
for tot, short, calls, avg, pct in rows[:25]:
    lines.append("%-92s %8d %14.2f %8.2f" % (short, calls, avg / 1e3, pct))
txt = "\n".join(lines)
print(txt)
open(d + "/summary.txt", "w").write(txt + "\n")
