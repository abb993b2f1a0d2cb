import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
