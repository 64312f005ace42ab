import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The library uses its resident ICP loop up to 64 workgroups (where it still gains); the tests exercise it at every
# size it fits (multi-group hand-offs, the reservation between contexts, two ranks on one GPU) - worker processes
# inherit this
os.environ.setdefault("PCLMI_ICP_LOOP_MAX", "4096")
