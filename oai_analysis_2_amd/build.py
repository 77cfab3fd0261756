"""Build liboai_hip.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc.

No cmake/ninja, no torch extension machinery: the library has no torch dependency.  Objects go to
``oai_analysis_2_amd/csrc/_obj`` and the shared object next to this file, so it travels to the GPU
box with the repo snapshot (``*.so`` is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "liboai_hip.so")
ARCH = "gfx950"
# -packed-fp32-ops (device code only; the host pass prints "not a recognized feature", filtered below): no v_pk_fma_f32 /
# v_pk_mul_f32 / v_pk_add_f32.  Measured on MI355X (profiles/r02_packed_fp32_hazard.md): a gather kernel whose fp32 arithmetic
# hipcc had SLP-packed into v_pk_* produced wrong values in 16-lane groups whenever it ran on a side stream beside the MFMA
# convolution kernels (5 of 12 volumes), and never with scalar fp32 VALU ops -- same source, same launches.  Packed fp32 is also
# slower beside MFMAs (MI355X_MICROARCH.md, "price of one filler").  Results are unchanged: a packed op is two IEEE ops.
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build liboai_hip.so")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True, extra_flags=(), lib_path: str = None, obj_dir: str = None) -> str:
    """Production build by default.  ``build_diag()`` passes -DOAI_DIAG and other output paths: the diagnostic library (timing
    ablations with wrong results, variant selection by environment) is a separate file that only scripts/ load via OAI_LIB_PATH."""
    OBJ, LIB = obj_dir or globals()["OBJ"], lib_path or globals()["LIB"]
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "oai_hip.h"))
    hipcc = _hipcc()
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    # diagnostic builds only: OAI_PACKED_TU=warp.hip compiles the named sources WITH packed fp32 ops (hipcc's default), to re-measure
    # the hazard of profiles/r02_packed_fp32_hazard.md against the current kernels (scripts/stress_overlap.py)
    packed_tu = set(filter(None, os.environ.get("OAI_PACKED_TU", "").split(","))) if "-DOAI_DIAG" in extra_flags else set()

    def compile_one(job):
        src, obj = job
        flags = [f for f in FLAGS if f not in ("-Xclang", "-target-feature", "-packed-fp32-ops")] if os.path.basename(src) in packed_tu else FLAGS
        cmd = [hipcc, *flags, *extra_flags, "-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        err = "\n".join(ln for ln in r.stderr.splitlines() if "not a recognized feature for this target" not in ln)
        if verbose and err.strip():
            print(err, file=sys.stderr)
        return obj

    if jobs:
        if verbose:
            print(f"[oai build] compiling {len(jobs)} source(s) for {ARCH}", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s) + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[oai build] linked {LIB}", file=sys.stderr)
        if not packed_tu:
            check_no_packed_fp32(LIB)
        check_no_sgpr_hazard(LIB)
    return LIB


def check_no_sgpr_hazard(lib: str) -> None:
    """The build FAILS if a VMEM instruction reads an SGPR that a VALU instruction (v_readfirstlane / v_readlane / v_cmp) wrote fewer than
    five wait states earlier.  hipcc pads its own code; it does not look inside inline assembly (gload16_asm: `global_load_dwordx4 v, v,
    s[b:b+1]`), where the stale SGPR is a wild address (csrc/unet_sres2.h: sgpr_settle)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sgpr_hazard_scan", os.path.join(os.path.dirname(HERE), "scripts", "sgpr_hazard_scan.py"))
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"      # (located like check_no_packed_fp32 does)
    if not os.path.exists(spec.origin) or not os.path.exists(objdump):
        os.remove(lib)
        raise RuntimeError("the SGPR-hazard guard cannot run (scripts/sgpr_hazard_scan.py or llvm-objdump missing): refusing to ship an unchecked library")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hits = mod.scan_library(lib)
    if hits:
        os.remove(lib)
        raise RuntimeError("VALU-written SGPR read by a VMEM instruction too early (inline assembly?):\n" + "\n".join(hits))


def check_no_packed_fp32(lib: str) -> None:
    """The build FAILS if any gfx950 code object of the library contains a packed fp32 VALU instruction: a future float2
    expression, an inline-asm v_pk op or a toolchain change would otherwise bring the hazard of profiles/r03_packed_fp32_hazard.md back
    silently (a packed op that reads src1 across halves returns a zero product in lanes 48..63 beside 16-bit MFMAs)."""
    import glob
    import re
    import tempfile
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        os.remove(lib)
        raise RuntimeError("no llvm-objdump: the packed-fp32 guard cannot run, refusing to ship an unchecked library")
    with tempfile.TemporaryDirectory() as work:
        shutil.copy(lib, os.path.join(work, "lib.so"))
        subprocess.run([objdump, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
        n = 0
        for o in glob.glob(os.path.join(work, "lib.so.*gfx950*")):
            text = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout
            n += len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", text))
    if n:
        os.remove(lib)
        raise RuntimeError(f"{n} packed fp32 VALU instruction(s) in {lib}: refusing to ship it (see profiles/r03_packed_fp32_hazard.md)")


DIAG_LIB = os.path.join(os.path.dirname(HERE), "build", "diag", "liboai_hip_diag.so")


def build_diag(force: bool = False, extra_flags=(), tag: str = "") -> str:
    lib = DIAG_LIB.replace(".so", f"{tag}.so")
    return build_library(force, True, ["-DOAI_DIAG", *extra_flags], lib, os.path.join(os.path.dirname(DIAG_LIB), "_obj" + tag))


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag(force="--force" in sys.argv))
    else:
        build_library(force="--force" in sys.argv)
