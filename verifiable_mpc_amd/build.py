"""Build libvmpc_hip.so (gfx950) in-tree: python -m verifiable_mpc_amd.build

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so that it
travels with the repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvmpc_hip.so")
UNITS = ["api", "msm", "msm_sort", "msm_reduce_tree", "msm_short", "exact", "frvec", "format", "sha256", "bn256", "bn256_g1", "bn256_g2_bucket", "bn256_g2_reduce", "bn256_g2_final", "bn256_g2_table", "fold_jump", "prover", "comm", "probe", "bn256_probe"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("VMPC_EXTRA_FLAGS", "").split()


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "vmpc.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_unit(u):
        src = os.path.join(CSRC, u + ".hip")
        obj = os.path.join(objdir, u + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_unit, UNITS))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
