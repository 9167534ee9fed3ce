"""HD_LIB_VARIANT=<label>: make the diagnostic scripts of this directory load gpurun_in/variants/libhd_<label>.so (an experiment build made with
HD_BUILD_VARIANT=<label> HD_EXTRA_FLAGS=... python3 -m habdec_amd.build) instead of the product library.  Import before habdec_amd.lib() is used."""
import os
from pathlib import Path
from habdec_amd import capi

v = os.environ.get("HD_LIB_VARIANT")
if v:
    capi.LIB_PATH = Path(__file__).resolve().parents[2] / "gpurun_in" / "variants" / f"libhd_{v}.so"
