import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

# PyTorch ships its own copy of the HIP runtime; libhabdec_amd.so links the system one.  Whichever is loaded first serves the whole process, and
# torch finds no GPU when the system copy came first -- so where both are used (the GPU tests stage inputs with torch), torch is imported first.
try:
    import torch  # noqa: F401
except Exception:      # the CPU-only suite does not need it
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_ref: needs oracle/_ref built from /root/reference (skipped elsewhere)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    """The oracle is test infrastructure: (re)build it on demand; the reference harness only where the
    reference checkout exists (it never travels to the GPU box as source)."""
    from oracle import pyoracle
    if not (ROOT / "oracle" / "liboracle.so").exists():
        pyoracle.build()
    if Path("/root/reference/code/Decoder").exists() and not pyoracle.have_ref():
        pyoracle.build(ref=True)
    yield


def pytest_collection_modifyitems(config, items):
    from oracle import pyoracle
    have = pyoracle.have_ref() or Path("/root/reference/code/Decoder").exists()
    skip = pytest.mark.skip(reason="oracle/_ref not built (no /root/reference here)")
    for it in items:
        if "needs_ref" in it.keywords and not have:
            it.add_marker(skip)
