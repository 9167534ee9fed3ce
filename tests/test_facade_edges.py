"""habdec::Decoder<float> facade, stage-plan edges (reference code/Decoder/Decoder.h:268-332 and :336-412), on the CPU.

The reference clears its stage list BEFORE it looks the factor up (Decoder.h:281-284), so a factor without a table (1, or anything that is not
a power of two) leaves "no stages, factor 1" behind and returns 0 (:317-319); factors outside [1, 256] leave the plan alone and return the
current factor (:272-276); a supported factor prints its stages as "/32/2" (:324-329).  setupDecimationStagesBW with an input rate already
at or below the bound runs its loop zero times: no stages, factor 1, returns 1 (:351,:406-411).  Decoder.h itself cannot be compiled here
(fftw3.h, ssdv.h absent), so the transcript below is pinned by reading those lines, not by running them."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

EXPECTED = """\
Decoder::setupDecimationStagesFactor Decimation Stages: /32/2
Decoder::setupDecimationStagesFactor Post Decimation Sampling Rate = 32000, decimation factor = 64
R 64 64
Unsupported decimation factor: 3
R 0 1
Decoder::setupDecimationStagesFactor Decimation Stages: /8/2
Decoder::setupDecimationStagesFactor Post Decimation Sampling Rate = 128000, decimation factor = 16
R 16 16
Unsupported decimation factor: 1
R 0 1
Decoder::setupDecimationStagesFactor Decimation Stages: /8
Decoder::setupDecimationStagesFactor Post Decimation Sampling Rate = 256000, decimation factor = 8
R 8 8
Unsupported decimation factor: 0
R 8 8
Unsupported decimation factor: 512
R 8 8
Decoder::setupDecimationStagesBW Decimation Stages: /32/2
Decoder::setupDecimationStagesBW Post Decimation Sampling Rate = 32000, decimation factor = 64
R 64 64
Decoder::setupDecimationStagesBW Decimation Stages: 
Decoder::setupDecimationStagesBW Post Decimation Sampling Rate = 2.048e+06, decimation factor = 1
R 1 1
Decoder::setupDecimationStagesBW more than /256 needed for 100 Hz: unsupported, keeping /1
R 0 1
"""


def test_stage_plan_edges_through_the_facade(tmp_path):
    lib = ROOT / "habdec_amd" / "libhabdec_amd.so"
    if not lib.exists() or not shutil.which("g++"):
        pytest.skip("library or g++ missing")
    exe = tmp_path / "facade_edges"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", str(ROOT / "habdec_amd" / "include"), "-I", str(ROOT / "include"),
                    str(ROOT / "tests" / "cpp" / "facade_edges.cpp"), "-o", str(exe), "-L", str(lib.parent), "-lhabdec_amd",
                    f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out == EXPECTED
