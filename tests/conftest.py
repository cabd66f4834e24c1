import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: (re)build it if sources are newer / missing
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
    if not os.path.exists(os.path.join(ROOT, "orb_line_slam_amd", "csrc", "libolf_synth.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "orb_line_slam_amd", "csrc"), "libolf_synth.so"], check=True)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib
