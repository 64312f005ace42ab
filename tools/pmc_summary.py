"""Reduces the rocprofv3 --pmc CSVs of tools/pmc_collect.sh to per-launch averages of one kernel.
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so the read side is
doubled (an upper bound for this kernel, whose reads are 32/64-byte scalar loads)."""
import csv
import glob
import json
import sys

root, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
acc = {}
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]:
            continue
        a = acc.setdefault(r["Counter_Name"], {})
        key = (f, r["Dispatch_Id"])
        a[key] = a.get(key, 0.0) + float(r["Counter_Value"])
res = {name: sum(v.values()) / len(v) for name, v in acc.items()}
res["launches_averaged"] = {name: len(v) for name, v in acc.items()}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["hbm_bytes_per_launch_raw"] = (res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
    res["hbm_bytes_per_launch"] = (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res:
    res["l2_hit_rate"] = res["TCC_HIT_sum"] / max(res["TCC_HIT_sum"] + res["TCC_MISS_sum"], 1)
res["kernel"] = kernel
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
