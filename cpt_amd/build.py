"""Build libcpt_hip.so (gfx950 only) in-tree with hipcc.  No JIT cache: the .so sits next to the
package so it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcpt_hip.so")
LIB_ABL = os.path.join(HERE, "libcpt_hip_abl.so")      # development build (-DCPT_ABLATION): kernel-variant switches live (include/cpt_hip_debug.h)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOSTCXX = os.environ.get("CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-Wno-unused-result", "-I" + os.path.join(os.path.dirname(HERE), "include")]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, ablation=False):
    """Product library (default) or, with ablation=True, the development library libcpt_hip_abl.so (-DCPT_ABLATION; objects under csrc/abl/)."""
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "cpt_hip.h"))
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "cpt_io.h"))
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "cpt_hip_debug.h"))
    odir = os.path.join(CSRC, "abl") if ablation else CSRC
    os.makedirs(odir, exist_ok=True)
    lib = LIB_ABL if ablation else LIB
    flags = FLAGS + (["-DCPT_ABLATION"] if ablation else [])
    objs, jobs = [], []
    for f in _sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(odir, f[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[cpt_amd.build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
        return r.stdout

    # host-only C++ translation units with their own ISA flags (b64_avx2.cpp: the AVX2 base64 loop, called behind a run-time CPU check)
    for f, extra in (("b64_avx2.cpp", ["-mavx2"]),):
        src = os.path.join(CSRC, f)
        obj = os.path.join(odir, f[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src]):
            jobs.append([HOSTCXX, "-O3", "-std=c++17", "-fPIC"] + extra + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", lib] + objs + ["-ldl"])
    return lib


def build_tools(verbose=True):
    """tools/yardstick.bin: hipBLASLt's best plain bf16 GEMM of the four encoder projections + HBM streaming figures -- the ceiling probe bench.py runs
    beside its own kernels ("re-measure, do not trust").  A measurement tool: it links hipBLASLt, the product library does not."""
    root = os.path.dirname(HERE)
    src, out = os.path.join(root, "tools", "yardstick.hip"), os.path.join(root, "tools", "yardstick.bin")
    if not os.path.exists(src):
        return None
    if _stale(out, [src]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-lhipblaslt", "-ldl", "-o", out]
        if verbose:
            print("[cpt_amd.build]", " ".join(cmd[-6:]), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed (tools/yardstick.hip):\n" + r.stdout)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, ablation="--ablation" in sys.argv))
