"""The C-ABI library loads and exports every symbol include/*.h declares (no compute call, no GPU needed)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared(header: Path):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(hd_[a-z0-9_]+)\s*\(", text)) - {"hd_sentence_cb", "hd_chars_cb"})


@pytest.fixture(scope="module")
def lib():
    from habdec_amd.build import build
    build()
    import habdec_amd
    return habdec_amd.lib()


@pytest.mark.parametrize("header", ["habdec_amd.h", "habdec_amd_host.h"])
def test_every_declared_symbol_is_exported(lib, header):
    names = declared(ROOT / "include" / header)
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_tables_cover_the_headers():
    from habdec_amd import capi
    assert set(declared(ROOT / "include" / "habdec_amd.h")) == set(capi.ENGINE_API)
    assert set(declared(ROOT / "include" / "habdec_amd_host.h")) == set(capi.HOST_API)


def test_engine_refuses_to_run_without_a_gpu_and_never_falls_back(lib):
    """On a box without an MI355X creating an engine must fail loudly (there is no CPU path in the product)."""
    import ctypes
    import habdec_amd
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(habdec_amd.HabdecError, match="no HIP device"):
        habdec_amd.Engine(n_streams=1)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/."""
    for path in (ROOT / "habdec_amd").rglob("*"):
        if path.suffix in {".py", ".cpp", ".hip", ".h", ".hpp", ".inc"} and path.name != "build.py":
            text = path.read_text(errors="ignore")
            assert "oracle" not in text.replace("the oracle's std::regex", "").replace("the CPU oracle", "").lower() or path.name in {"sentence.hpp", "__init__.py"}, path
    text = (ROOT / "habdec_amd" / "capi.py").read_text() + (ROOT / "habdec_amd" / "engine.py").read_text()
    assert "pyoracle" not in text and "liboracle" not in text


def test_fault_injection_library_loads_and_binds():
    """tests/test_gpu_fault.py's library (the product sources with -DHD_RING_FAULT, built by __graft_entry__.build()) must load here and export the
    whole C ABI plus its arming hook -- so that the GPU test cannot be skipped or fail for a link error."""
    import ctypes
    from habdec_amd import capi
    path = capi.LIB_PATH.with_name("libhabdec_amd_fault.so")
    from habdec_amd.build import build
    build(variant="fault")                                   # (returns at once unless the library is missing or older than its sources)
    L = ctypes.CDLL(str(path))
    for table in (capi.ENGINE_API, capi.HOST_API):
        for name in table:
            getattr(L, name)
    getattr(L, "hd_debug_ring_fault_arm")
    getattr(L, "hd_debug_ring_fault_arm_starve")
    getattr(L, "hd_debug_tail_fault_arm_no_tag")
    prod = ctypes.CDLL(str(capi.LIB_PATH))
    assert not hasattr(prod, "hd_debug_ring_fault_arm")      # the hook is compiled out of the product
