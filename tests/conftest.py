"""pytest config: registers the ``gpu`` marker and puts the product package dir on sys.path.

The product package directory is named after the repo
(``multimodal-framework-for-speaker-emotion-recognition_amd``), which is not a Python identifier; it is a
*path entry* whose children mirror the reference's import names (``models.lsthm_sps``, ``models.encoder``,
``attention.SelfAttention``, ``model_trainer``, ``loss``) plus ``mser`` (the ctypes binding to the C-ABI).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
