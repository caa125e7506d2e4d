"""Build the C-ABI shared library (include/klab_mm.h) for gfx950 with hipcc, in-tree.

`python -m klab_multimodalmodel_amd.build` or `build_library()`; hipcc cross-compiles without a GPU.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libklab_mm.so")
OBJDIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "klab_mm.h")]
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        checked = os.path.basename(src) == "gemm.hip"  # hand-scheduled inline-asm loops: the ISA itself is verified (below)
        cmd = [HIPCC] + FLAGS + (["-save-temps=obj"] if checked else []) + ["-x", "hip", "-c", src, "-o", obj]
        stem = os.path.splitext(os.path.basename(src))[0]
        if checked:  # stale temporaries of an earlier compile must not be what gets checked
            for f in os.listdir(OBJDIR):
                if f.startswith(stem) and f != os.path.basename(obj):
                    os.remove(os.path.join(OBJDIR, f))
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJDIR)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if checked:
            # tools/check_gemm_asm.py on the device assembly of THIS compile: no read of an asm-loaded LDS fragment before its wait,
            # no read of an inline-asm MFMA's result before the pipe has drained (hipcc may place copies at control-flow joins)
            # (fails CLOSED: this check is the only guard against an inline-asm MFMA result being read early)
            asm = [os.path.join(OBJDIR, f) for f in os.listdir(OBJDIR) if f.startswith(stem) and "gfx950" in f and f.endswith(".s")]
            tool = os.path.join(ROOT, "tools", "check_gemm_asm.py")
            if len(asm) != 1 or not os.path.exists(tool):
                os.remove(obj)
                raise RuntimeError(f"ISA check of {os.path.basename(src)} could not run: device assembly {asm or 'not found'} "
                                   f"(hipcc -save-temps naming changed?), checker {'present' if os.path.exists(tool) else 'MISSING'}")
            c = subprocess.run([sys.executable, tool, asm[0]], capture_output=True, text=True)
            for f in os.listdir(OBJDIR):  # the temporaries are large (tens of MB): keep only objects
                if f.startswith(stem) and f != os.path.basename(obj):
                    os.remove(os.path.join(OBJDIR, f))
            import re
            nk = re.search(r"checked (\d+) kernels", c.stdout)
            if c.returncode != 0 or not nk or int(nk.group(1)) < 1:
                os.remove(obj)
                raise RuntimeError("tools/check_gemm_asm.py rejected the GEMM kernels' assembly (or checked nothing):\n" + c.stdout[-3000:] + c.stderr[-1000:])
            if verbose:
                print("ISA check:", "; ".join(c.stdout.strip().splitlines()[-2:]), file=sys.stderr)
        if verbose:
            print("compiled", os.path.basename(src), file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("linked", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
