"""Build libhabdec_amd.so (HIP kernels + C-ABI engine) for gfx950, in-tree.

    python -m habdec_amd.build          # or: from habdec_amd.build import build; build()

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the FIR/decimator sums must be separate
IEEE multiply and add (no FMA) to be bit-identical to the reference's CPU arithmetic.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libhabdec_amd.so"
FAULT_OUT = HERE / "libhabdec_amd_fault.so"
SOURCES = ["engine.cpp", "host_api.cpp", "kernels/decimate.hip", "kernels/fir_demod.hip", "kernels/backend.hip", "kernels/spectrum.hip", "kernels/spectrum_wave.hip", "kernels/symbols.hip", "kernels/tail.hip"]
# the translation units that carry a FIR of the chain are compiled once per arithmetic mode (kernels/arith.h): as they are -> namespace hd::exact
# (separately rounded multiply and add), with -DHD_FAST_ARITH -> namespace hd::fast (fused multiply-add); hd_engine_config.arith picks at run time
MODE_SOURCES = ["kernels/decimate.hip", "kernels/fir_demod.hip", "kernels/backend.hip", "kernels/tail.hip", "kernels/symbols.hip"]
ARCH = "gfx950"


def source_id() -> str:
    """Identity of the product library's SOURCES: sha256 over every file under csrc/ and include/habdec_amd.h (relative path + contents), first 16 hex digits.
    tools/collect_profiles.py stores it beside the PMC traffic it records; bench.py quotes that traffic only while the sources are still the ones profiled."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(p for p in CSRC.rglob("*") if p.is_file() and p.suffix in {".hip", ".cpp", ".h", ".hpp", ".inc"}) + [HERE.parent / "include" / "habdec_amd.h"]
    for f in files:
        h.update(str(f.relative_to(HERE.parent)).encode()); h.update(b"\0"); h.update(f.read_bytes()); h.update(b"\0")
    return h.hexdigest()[:16]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found")


def _stale(out: Path = OUT) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    deps = list(CSRC.rglob("*.hip")) + list(CSRC.rglob("*.cpp")) + list(CSRC.rglob("*.h")) + list(CSRC.rglob("*.hpp")) + \
        list(CSRC.rglob("*.inc")) + [HERE.parent / "include" / "habdec_amd.h", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False, variant: str | None = None) -> Path:
    """`variant` (or HD_BUILD_VARIANT): an experiment build beside the product library -- objects in build/<variant>/, the library in
    gpurun_in/variants/libhd_<variant>.so (tools/micro/ab_step.py loads several of them into one process); flags from HD_EXTRA_FLAGS."""
    variant = variant or os.environ.get("HD_BUILD_VARIANT") or None
    target = OUT
    if variant == "fault":
        # the fault-injection library of tests/test_gpu_fault.py: one LDS-DMA loader of the process drops a tile's publish (-DHD_RING_FAULT), so
        # that the bounded waits, the give-up word and the engine's failed state are exercised end to end.  In-tree beside the product library.
        target = FAULT_OUT
        if not force and target.exists() and not _stale(target):
            return target
    elif variant:
        target = HERE.parent / "gpurun_in" / "variants" / f"libhd_{variant}.so"
        target.parent.mkdir(parents=True, exist_ok=True)
    elif not force and not _stale():
        return OUT
    objs = []
    build_dir = HERE / "build" / variant if variant else HERE / "build"
    build_dir.mkdir(parents=True, exist_ok=True)
    # (variant "timingexp": the timing-experiment build of tools/micro/joules.py -- HD_CU_EXP=1/2 runs step launches without their tails / without stage 1,
    # results wrong; that switch does not exist in the product library)
    extra = os.environ.get("HD_EXTRA_FLAGS", "").split() + (["-DHD_RING_FAULT"] if variant == "fault" else []) + (["-DHD_TIMING_EXPERIMENT"] if variant == "timingexp" else [])
    common = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
              # (the atomic optimiser rewrites lane 0's ticket draw in the step kernel into a form that needs the old value at once: the wave
              # would wait for the round trip it issues a tile early precisely not to wait for)
              "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", *extra,
              "-Wall", "-Wno-unused-function", "-I", str(HERE.parent / "include"), "-I", str(CSRC)]
    procs = []
    units = [(src, "", []) for src in SOURCES] + [(src, ".fast", ["-DHD_FAST_ARITH"]) for src in MODE_SOURCES]
    units.sort(key=lambda u: 0 if "decimate" in u[0] else 1)      # the long compiles first
    for src, tag, defs in units:
        obj = build_dir / (src.replace("/", "_") + tag + ".o")
        cmd = common + defs + ["-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src + tag, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(str(obj))
    bad = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            bad = True
            sys.stderr.write(f"--- {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if bad:
        raise RuntimeError("hipcc failed")
    link = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(target)] + objs + \
        ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(link, check=True)
    return target


def build_facade_demo() -> Path:
    """tests/cpp/decoder_thread_demo: the reference's DECODER_THREAD loop on top of habdec::Decoder<float> (the facade)."""
    build()
    src = HERE.parent / "tests" / "cpp" / "decoder_thread_demo.cpp"
    out = HERE.parent / "tests" / "cpp" / "decoder_thread_demo"
    if out.exists() and out.stat().st_mtime > max(src.stat().st_mtime, OUT.stat().st_mtime, (HERE / "include" / "habdec" / "Decoder.h").stat().st_mtime):
        return out
    cmd = ["g++", "-std=c++17", "-O2", "-I", str(HERE / "include"), "-I", str(HERE.parent / "include"), str(src), "-o", str(out),
           "-L", str(HERE), "-lhabdec_amd", f"-Wl,-rpath,{HERE}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"]
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
