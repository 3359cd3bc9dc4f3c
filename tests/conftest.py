import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# every Aligner of the test suite also formats its SAM text on the device (fq_emit.h) and api.Aligner.sam_text() holds it to the host formatter's bytes
os.environ.setdefault("FASTQUICK_API_EMIT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "refbuild: needs oracle/_ref (the reference compiled in the build container)")


@pytest.fixture(scope="session")
def golden_cases(tmp_path_factory):
    """Materialise every tests/golden/<case> (index files + sparse bitmap list + FASTQ) once per session."""
    import golden_util
    base = tmp_path_factory.mktemp("golden")
    return {tag: golden_util.materialise(tag, str(base / tag)) for tag in golden_util.case_tags()}


@pytest.fixture(scope="session")
def emu_cli():
    """the command line over the host-loop library (CPU tier; test infrastructure only)"""
    import subprocess
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "libfq_emu.so", "FASTQuick_emu"])
    return os.path.join(emu, "FASTQuick_emu")
