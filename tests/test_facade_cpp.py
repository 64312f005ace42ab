"""Runs the C++ facade test binary (tests/cpp/test_facade.cpp: the reference's own kdtree / search /
registration tests re-expressed against include/pcl/**).  The binary is built by
__graft_entry__.build() (make -C tests/cpp) and travels to the GPU box with the tree."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "tests" / "cpp" / "test_facade"


def _run():
    assert EXE.exists(), "tests/cpp/test_facade is not built: run __graft_entry__.build()"
    return subprocess.run([str(EXE), str(ROOT / "tests" / "golden")], capture_output=True, text=True, timeout=300)


@pytest.mark.gpu
def test_reference_tests_through_the_cpp_facade():
    r = _run()
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 failed" in r.stdout and r.stdout.count("[  OK  ]") == 15


def test_facade_binary_links_and_refuses_to_run_without_a_gpu():
    from pcl_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    r = _run()
    assert r.returncode == 2 and "no CPU path" in r.stdout


def test_pcd_io_round_trips(tmp_path):
    """include/pcl/io/pcd_io.h (host only): fixture loads, ASCII / binary files read back bit-identically."""
    exe = ROOT / "tests" / "cpp" / "test_pcd_io"
    assert exe.exists(), "tests/cpp/test_pcd_io is not built: run __graft_entry__.build()"
    r = subprocess.run([str(exe), str(ROOT / "tests" / "golden"), str(tmp_path)], capture_output=True, text=True,
                       timeout=60)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    # and the files the facade wrote are readable by the tests' independent reader
    import numpy as np
    from tests import util
    raw = (tmp_path / "b.pcd").read_bytes()
    body = raw[raw.index(b"DATA binary\n") + len(b"DATA binary\n"):]
    assert np.array_equal(np.frombuffer(body, np.float32).reshape(-1, 3), util.bun0())
    assert np.array_equal(util.read_pcd_ascii(tmp_path / "a.pcd"), util.bun0())


@pytest.mark.gpu
def test_pcl_icp_command_line_tool(tmp_path):
    """tools/pcl_icp.cpp = the reference's tools/icp.cpp (BASELINE configs[0]: bun0.pcd -> bun4.pcd, -d 0.05
    -r 0.05 -i 50) through the facade: prints the golden transform of test_registration.cpp:169-187 and writes
    the moved cloud as a binary PCD."""
    import numpy as np
    from pcl_amd import capi
    from tests import util
    exe = ROOT / "tests" / "cpp" / "pcl_icp"
    assert exe.exists(), "tests/cpp/pcl_icp is not built: run __graft_entry__.build()"
    g = ROOT / "tests" / "golden"
    # the reference tool registers the SECOND file to the first: model = bun4 (target), data = bun0 (source)
    r = subprocess.run([str(exe), "-d", "0.05", "-r", "0.05", "-i", "50", "--out-dir", str(tmp_path),
                        str(g / "bun4.pcd"), str(g / "bun0.pcd")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    at = lines.index("Using IterativeClosestPoint")
    T = np.array([[float(v) for v in ln.split()] for ln in lines[at + 1:at + 5]], np.float32)
    gold = util.golden("icp_bun0_bun4.json")
    assert np.all(np.abs(T[:3] - np.array(gold["matrix"])[:3]) <= np.array(gold["tol"])[:3]), T
    assert T[3].tolist() == [0, 0, 0, 1]
    # same numbers as the C ABI called directly with the tool's parameters (RANSAC on, 1000 iterations)
    p = capi.icp_default_params()
    p.max_iterations, p.max_correspondence_distance, p.ransac_outlier_threshold = 50, 0.05, 0.05
    rc, T2, out, st = capi.icp_align(capi.Tree(util.bun4()), util.bun0(), p)
    assert rc == 0 and np.allclose(T, T2, atol=2e-6)  # printed with 6 significant digits
    raw = (tmp_path / "bun0.pcd").read_bytes()
    body = raw[raw.index(b"DATA binary\n") + len(b"DATA binary\n"):]
    moved = np.frombuffer(body, np.float32).reshape(-1, 3)
    assert moved.shape == (397, 3) and np.allclose(moved, out[:, :3], atol=1e-6)
    assert np.array_equal(util.read_pcd_ascii(tmp_path / "bun4.pcd"), util.bun4())
