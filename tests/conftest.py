import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def known_answers():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_1kg():
    """(input VCF bytes, expected sorted body lines, expected header line) of the reference's own
    regression pair (previous_out_check/, see tests/golden/README.md)."""
    import gzip

    g = os.path.join(ROOT, "tests", "golden")
    with gzip.open(os.path.join(g, "1kg_chr1_20klines.vcf.gz"), "rb") as f:
        vcf = f.read()
    with gzip.open(os.path.join(g, "1kg_chr1_20klines.expected.tsv.gz"), "rb") as f:
        exp = f.read()
    lines = exp.split(b"\n")
    assert lines[-1] == b""
    return vcf, sorted(lines[1:-1]), lines[0]
