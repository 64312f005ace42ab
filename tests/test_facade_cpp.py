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
    assert "0 failed" in r.stdout and r.stdout.count("[  OK  ]") == 6


def test_facade_binary_links_and_refuses_to_run_without_a_gpu():
    from pcl_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    r = _run()
    assert r.returncode == 2 and "no CPU path" in r.stdout
