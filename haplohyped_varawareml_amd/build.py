"""Builds libhhgt.so (HIP, gfx950 only) in-tree with hipcc.  No CMake, no JIT cache: the .so sits
next to this file so it travels with the repository snapshot to the GPU box."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhhgt.so")
SOURCES = ["scan.hip", "index.hip", "encode.hip", "lz4.hip", "frame.hip", "decode.hip", "inflate.hip", "synth.hip",
           "reader.hip", "onehot.hip", "api.hip"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libhhgt.so cannot be built (there is no CPU fallback)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "hhgt.h"),
                        os.path.join(HERE, "..", "include", "hhgt_synth.h"),
                        os.path.join(HERE, "..", "include", "hhgt_reader.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + sources() + ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
