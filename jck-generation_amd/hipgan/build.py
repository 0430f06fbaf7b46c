"""Builds libjckgan_hip.so (gfx950) in-tree with hipcc.  `python -m hipgan.build` or via __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libjckgan_hip.so")
SOURCES = ["ops.hip", "engine.hip", "infer.hip", "comm.hip"]


def source_id():
    """What jck_version() of a library built from the present sources returns: a hash of every file under csrc/ and of
    include/jckgan.h (>= 100).  The loader compares it with the binary's answer: a stale .so is refused, not used."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))
    files.append(os.path.join(os.path.dirname(PKG), "include", "jckgan.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return 100 + int.from_bytes(h.digest()[:4], "little") % (2**31 - 101)


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(PKG), "include", "jckgan.h")]
    bid = source_id()
    idfile = os.path.join(LIBDIR, "BUILD_ID")
    built = open(idfile).read().strip() if os.path.exists(idfile) else ""
    if not force and os.path.exists(LIB) and built == str(bid) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DJCK_BUILD_ID={bid}"] + \
              (["-DJCK_DIAG"] if os.environ.get("JCK_DIAG") == "1" else []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(idfile, "w") as f:
        f.write(str(bid))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
