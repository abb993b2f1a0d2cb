import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _legs_lines():
    from tests import extlibs
    from tests.test_h5file import h5_used
    used = dict(extlibs.used, **h5_used)
    return [f"independent decoder {name}: {'present' if ok else 'ABSENT'} ({what}); calls in this session: {used.get(name, 0)}"
            for name, (ok, what) in extlibs.legs().items()]


def pytest_report_header(config):
    """which independent decoders (base-image libraries, not reference code) the parity tests can lean on here"""
    return [ln.split(";")[0] for ln in _legs_lines()]


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """... and how often each was really called: the oracle's own decoder always runs, these legs only where loadable"""
    terminalreporter.section("independent decoders", sep="-")
    for ln in _legs_lines():
        terminalreporter.write_line(ln)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def fixture_text():
    import gzip
    return gzip.open(os.path.join(GOLDEN, "chr22.filtered.vcf.gz"), "rb").read()


@pytest.fixture(scope="session")
def fixture_golden():
    import json
    return json.load(open(os.path.join(GOLDEN, "fixture_golden.json")))


@pytest.fixture(scope="session")
def ctx():
    """hhgt device context (GPU tests only)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from haplohyped_varawareml_amd.device import Context
    c = Context(0)
    yield c
    c.close()
