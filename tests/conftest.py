import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle runs on the host cores.  A container that sees 256 cores under a 16-CPU quota gets torch's default pool of one
    # thread per visible core throttled by the kernel for most of every 100 ms period (utils.fit_cpu_threads): size it to the quota.
    from tacotron2_subword_amd.utils import fit_cpu_threads
    fit_cpu_threads()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
