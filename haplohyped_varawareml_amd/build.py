"""Builds libhhgt.so (HIP, gfx950 only) in-tree with hipcc.  No CMake, no JIT cache: the .so sits
next to this file so it travels with the repository snapshot to the GPU box.  Every source is compiled
to its own object under build/obj (in parallel, only when it or a header changed) and linked once."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhhgt.so")
OBJ = os.path.join(ROOT, "build", "obj")
SOURCES = ["scan.hip", "index.hip", "encode.hip", "lz4.hip", "lz4bits.hip", "frame.hip", "decode.hip", "inflate.hip", "synth.hip",
           "reader.hip", "ingest.hip", "onehot.hip", "api.hip"]
# -mssse3: host side only (pshufb in csrc/fast_inflate.h; every x86-64 host of an MI355X has it)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-mssse3"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libhhgt.so cannot be built (there is no CPU fallback)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def headers():
    inc = os.path.join(ROOT, "include")
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return hs


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src) + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def needs_build():
    return _stale(LIB, sources() + headers())


def build(force=False, verbose=False, extra_flags=(), lib=LIB):
    """extra_flags / lib: development builds (-DHHGT_LZ4_STATS ...) go to another .so and bypass the object cache"""
    if not force and not extra_flags and not needs_build():
        return lib
    cc = _hipcc()
    hs = headers()
    objdir = OBJ if not extra_flags else os.path.join(ROOT, "build", "obj_" + str(abs(hash(tuple(extra_flags))) % 10**8))
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in sources():
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        if force or _stale(o, [s] + hs):
            jobs.append([cc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in sources()]
    run([cc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs + ["-lz", "-lpthread"])
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
