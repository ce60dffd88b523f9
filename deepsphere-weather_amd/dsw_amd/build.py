"""Build ``libdsw_hip.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
SOURCES = ["dsw_api.hip", "dsw_spmm.hip", "dsw_gemm.hip", "dsw_gemm_x3.hip", "dsw_spmm2.hip", "dsw_wgrad_x3.hip",
           "dsw_gemm_x3s.hip", "dsw_narrow.hip", "dsw_elementwise.hip", "dsw_pool.hip", "dsw_fwd3.hip", "dsw_spmm1s.hip", "dsw_bwd3d.hip"]
OUT = os.path.join(HERE, "libdsw_hip.so")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "dsw_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    """Compile every .hip translation unit for gfx950 (in parallel) and link libdsw_hip.so."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(CSRC, "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    if os.environ.get("DSW_BUILD_DIAG") == "1":      # diagnostics build: enables the DSW_* run-time switches of csrc/
        flags.append("-DDSW_DIAG")

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc_path(), *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[dsw build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
    if verbose:
        print("[dsw build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
