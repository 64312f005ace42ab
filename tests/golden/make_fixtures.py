#!/usr/bin/env python3
"""Regenerate tests/golden/* from the reference tree (run in the build container only).

The fixtures are DATA the reference's own tests hold for the kNN/ICP path:
  * test/bun0.pcd, test/bun4.pcd                       (ASCII PCD clouds, 397 / 361 points)
  * test/registration/test_registration_api_data.h     (index-pair tables + 4x4 goldens)
  * test/registration/test_registration.cpp:169-187    (full-ICP golden 4x4 + tolerances)
  * test/kdtree/test_kdtree.cpp:339-396                (10-point known-answer kNN, 3 variants)
Only numbers are extracted; no reference source text is kept.  /root/reference does not
exist on the GPU box, so nothing at test time reads it - tests read the JSON/PCD written here.
"""
import json, re, shutil, sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent


def int_table(text, name):
    m = re.search(r"const\s+int\s+%s\s*\[\d+\]\s*\[2\]\s*=\s*\{(.*?)\};" % name, text, re.S)
    pairs = re.findall(r"\{\s*(-?\d+)\s*,\s*(-?\d+)\s*\}", m.group(1))
    return [[int(a), int(b)] for a, b in pairs]


def mat_table(text, name):
    m = re.search(r"const\s+float\s+%s\s*\[4\]\s*\[4\]\s*=\s*\{(.*?)\};" % name, text, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    return [[float(v.strip().rstrip("f")) for v in r.split(",") if v.strip()] for r in rows]


def scalar(text, name):
    m = re.search(r"const\s+\w+\s+%s\s*=\s*([-0-9.eE]+)f?\s*;" % name, text)
    return float(m.group(1))


def main():
    for f in ("bun0.pcd", "bun4.pcd"):
        shutil.copyfile(REF / "test" / f, OUT / f)

    h = (REF / "test/registration/test_registration_api_data.h").read_text()
    api = {
        "source": "test/registration/test_registration_api_data.h",
        "correspondences_original": int_table(h, "correspondences_original"),
        "correspondences_reciprocal": int_table(h, "correspondences_reciprocal"),
        "rej_dist_max_dist": scalar(h, "rej_dist_max_dist"),
        "correspondences_dist": int_table(h, "correspondences_dist"),
        "correspondences_one_to_one": int_table(h, "correspondences_one_to_one"),
        "rej_sac_max_dist": scalar(h, "rej_sac_max_dist"),
        "rej_sac_max_iter": int(scalar(h, "rej_sac_max_iter")),
        "correspondences_sac": int_table(h, "correspondences_sac"),
        "rej_trimmed_overlap": scalar(h, "rej_trimmed_overlap"),
        "correspondences_trimmed": int_table(h, "correspondences_trimmed"),
        "transform_from_SAC": mat_table(h, "transform_from_SAC"),
        "transform_from_SVD": mat_table(h, "transform_from_SVD"),
    }
    assert len(api["correspondences_original"]) == 397
    assert len(api["correspondences_reciprocal"]) == 53
    assert len(api["correspondences_dist"]) == 97
    assert len(api["correspondences_one_to_one"]) == 103
    assert len(api["correspondences_sac"]) == 97
    assert len(api["correspondences_trimmed"]) == 198
    (OUT / "registration_api.json").write_text(json.dumps(api, indent=0))

    # full ICP golden: EXPECT_NEAR (transformation (r, c), value, tol) lines 169-182
    t = (REF / "test/registration/test_registration.cpp").read_text().splitlines()[153:188]
    vals = {}
    for line in t:
        m = re.search(r"EXPECT_NEAR \(transformation \((\d), (\d)\),\s*([-0-9.eE]+),\s*([-0-9.eE]+)\)", line)
        if m:
            vals[(int(m.group(1)), int(m.group(2)))] = (float(m.group(3)), float(m.group(4)))
    assert len(vals) == 12
    icp = {
        "source": "test/registration/test_registration.cpp:154-188",
        "max_iterations": 50, "transformation_epsilon": 1e-8, "max_correspondence_distance": 0.05,
        "matrix": [[vals[(r, c)][0] for c in range(4)] for r in range(3)] + [[0.0, 0.0, 0.0, 1.0]],
        "tol": [[vals[(r, c)][1] for c in range(4)] for r in range(3)] + [[0.0, 0.0, 0.0, 0.0]],
    }
    (OUT / "icp_bun0_bun4.json").write_text(json.dumps(icp, indent=1))

    # 10-point known-answer kNN (test/kdtree/test_kdtree.cpp:339-396)
    k = (REF / "test/kdtree/test_kdtree.cpp").read_text().splitlines()[335:398]
    pts = []
    gts_i, gts_d = [], []
    for line in k:
        m = re.search(r"push_back \(MyPoint \(\s*([-0-9.]+)f,\s*([-0-9.]+)f,\s*([-0-9.]+)f\)\)", line)
        if m:
            pts.append([float(m.group(i)) for i in (1, 2, 3)])
        m = re.search(r"gt_indices\[10\]\s*=\s*\{([^}]*)\}", line)
        if m:
            gts_i.append([int(v) for v in m.group(1).split(",")])
    joined = " ".join(k)
    for m in re.finditer(r"gt_distances\[10\]\s*=\s*\{([^}]*)\}", joined):
        gts_d.append([float(v.strip().rstrip("f")) for v in m.group(1).split(",")])
    assert len(pts) == 10 and len(gts_i) == 3 and len(gts_d) == 3
    kat = {
        "source": "test/kdtree/test_kdtree.cpp:336-398",
        "points": pts, "query": [50.0, 50.0, 50.0], "k": 10, "distance_tol": 0.1,
        "cases": [
            {"name": "xyz", "dims": 3, "alpha": [1.0, 1.0, 1.0], "indices": gts_i[0], "distances": gts_d[0]},
            {"name": "xy_only", "dims": 2, "alpha": [1.0, 1.0, 1.0], "indices": gts_i[1], "distances": gts_d[1]},
            {"name": "rescaled_1_2_3", "dims": 3, "alpha": [1.0, 2.0, 3.0], "indices": gts_i[2], "distances": gts_d[2]},
        ],
    }
    (OUT / "kdtree_kat.json").write_text(json.dumps(kat, indent=1))
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
