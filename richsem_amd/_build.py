"""Build librichsem_msda.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the built
.so travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librichsem_msda.so")
SOURCES = ["msda_api.hip", "ffn_mfma.hip", "rows_api.hip", "cls_mfma.hip", "conv_mfma.hip", "conv_wgrad.hip", "lin256_mfma.hip", "attn_mfma.hip"]
def _headers():
    """every header the library is built from: csrc/*.h and include/*.h (globbed, so a new kernel header can never be
    forgotten by the staleness check)"""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(_PKG, "..", "include", "*.h")))
HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("RICHSEM_HIPCC_EXTRA", "").split()      # (diagnostic builds: -DRPS_ROUTE_ABLATION, -DCONV_RING_ABLATE=...)
OBJ_DIR = os.path.join(LIB_DIR, "obj")
STAMP = os.path.join(LIB_DIR, "build_flags.txt")      # the hipcc flags the library on disk was built with (part of the staleness key)


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    try:
        if open(STAMP).read() != " ".join(HIPCC_FLAGS):      # (a diagnostic build left behind, or RICHSEM_HIPCC_EXTRA changed)
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + _headers() + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources.  Returns its path.  Every translation unit is compiled
    to an object of its own (in parallel; only the stale ones unless `force`), then linked."""
    if not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    try:
        flags_changed = open(STAMP).read() != " ".join(HIPCC_FLAGS)
    except OSError:
        flags_changed = True
    force = force or flags_changed      # (objects built with other flags are not reused)
    hipcc = find_hipcc()
    newest_header = max(os.path.getmtime(h) for h in _headers() + [os.path.abspath(__file__)])

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(OBJ_DIR, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", "-o", obj, path]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(STAMP, "w") as f:
        f.write(" ".join(HIPCC_FLAGS))
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
