"""Builds libltrx.so (the C-ABI library of HIP kernels) for gfx950, in-tree.

    python -m allrank_amd.build            # or: __graft_entry__.build()

hipcc cross-compiles without a GPU.  The .so lands next to this file (git-ignored, but it travels to the GPU
box with the gpurun snapshot).  Sources: allrank_amd/csrc/*.hip, public header: include/ltrx.h.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libltrx.so")
STAMP = os.path.join(HERE, ".libltrx.stamp")
ARCH = "gfx950"


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    files = _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files.append(os.path.join(os.path.dirname(HERE), "include", "ltrx.h"))
    for f in files:
        h.update(os.path.basename(f).encode())       # content-addressed: the same tree at another path is up to date
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        if verbose:
            print("[allrank_amd.build] libltrx.so up to date")
        return LIB
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("[allrank_amd.build] FAILED %s\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out.strip():
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    if verbose:
        print("[allrank_amd.build] built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
