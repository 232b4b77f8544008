import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """Recorded reference traces (quality_*.npz holds element records, not a trace)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR)
                  if f.endswith(".npz") and not f.startswith(("quality_", "move_", "smooth", "samples_", "front_", "plot_")))


def move_golden_names():
    """Recorded traces of the reference's move() API (oracle/gen_golden.py --move-only)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("move_"))


def smooth_golden_names():
    """Recorded step() traces with smooth_pave(interior=True) calls in between (oracle/gen_golden.py --smooth-only)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("smooth_"))


def front_smooth_golden_names():
    """Recorded smooth_pave(interior=False) + find_next_state calls of the reference (oracle/gen_golden.py --smooth-only)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("smoothfront_"))


def final_smooth_golden_names():
    """Recorded smooth() calls of the reference on finished meshes (oracle/gen_golden.py --smooth-only)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("smoothfinal_"))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ref_lib
    return ref_lib.lib()
