"""Builds ``lib/libvdetr_hip.so`` (gfx950 code objects + the C-ABI of include/vdetr_hip.h) with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting shared
object travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvdetr_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", f"--offload-arch={ARCH}"]
FLAGS += os.environ.get("VDETR_EXTRA_HIPCC_FLAGS", "").split()  # probe builds (-DVDETR_SP_PROBE=1 ...); part of the fingerprint


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _fingerprint():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    root = os.path.dirname(HERE)
    for f in sorted(os.listdir(CSRC)) + [os.path.join(root, "include", "vdetr_hip.h")]:
        p = f if os.path.isabs(f) else os.path.join(CSRC, f)
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile if the sources changed since the last build.  Returns the path of the shared object."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, ".fingerprint")
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == fp:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvdetr_hip.so")
    objs = []
    procs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(fp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
