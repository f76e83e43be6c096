"""Build the native library IN-TREE (debigulator_amd/lib/libdebigulator_hip.so).

    python -m debigulator_amd.build        # or  __graft_entry__.build()

hipcc cross-compiles gfx950 code objects without a GPU.  The host-side C sources
(the drop-in inflate/decode_png/decode_gz layer) are compiled as C by the same
driver and linked into the one shared library.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdebigulator_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_defs=(), out=None):
    """out: alternative library name (diagnostic builds, e.g. -DDEBIG_PROFILE)"""
    os.makedirs(LIBDIR, exist_ok=True)
    target = LIB if out is None else os.path.join(LIBDIR, out)
    root = os.path.dirname(HERE)
    deps = (glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(CSRC, "host", "*")) +
            glob.glob(os.path.join(CSRC, "compat", "*")) +
            glob.glob(os.path.join(root, "include", "*.h")))
    if not force and not _newer(target, deps):
        return target
    objs = []
    odir = os.path.join(LIBDIR, "obj")
    os.makedirs(odir, exist_ok=True)
    inc = ["-I" + os.path.join(root, "include")]
    defs = ["-D" + d for d in extra_defs]
    for c in sorted(glob.glob(os.path.join(CSRC, "host", "*.c"))):
        o = os.path.join(odir, os.path.basename(c) + ("" if out is None else "_" + out) + ".o")  # (a variant build has its own objects)
        cmd = ["gcc", "-O2", "-fPIC", "-std=c11", "-D_GNU_SOURCE", "-pthread", "-Wall", "-Wextra", "-fvisibility=hidden"] + inc + defs + ["-c", c, "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    o = os.path.join(odir, "debig_hip" + ("" if out is None else "_" + out) + ".o")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + inc + defs + [
        "-c", os.path.join(CSRC, "debig_hip.hip"), "-o", o]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    objs.append(o)
    # -Bsymbolic: our internal calls must never bind to zlib's `inflate`
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", target] + objs + ["-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if out is None:
        # optional static archive with the literal `inflate` symbol (csrc/compat/debig_compat.c)
        co = os.path.join(odir, "debig_compat.o")
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-Wall", "-Wextra", "-c",
                               os.path.join(CSRC, "compat", "debig_compat.c"), "-o", co])
        ar = os.path.join(LIBDIR, "libdebig_compat.a")
        if os.path.exists(ar):
            os.remove(ar)
        subprocess.check_call(["ar", "rcs", ar, co])
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
