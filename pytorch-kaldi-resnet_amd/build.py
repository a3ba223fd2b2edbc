"""Build libspkhip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspkhip.so")
SOURCES = ["err.cpp", "conv_mfma.hip", "conv_wgrad.hip", "stem.hip", "bn.hip", "pool.hip", "gemm.hip", "head.hip",
           "sgd.hip", "score.hip", "conv_split.hip", "conv_wgrad_split.hip", "pack.hip", "conv_ws.hip", "conv_pipe.hip", "conv_wgrad_pipe.hip", "conv_wgrad_1x1.hip", "conv_wgrad_wm.hip"]


def csrc_fingerprint():
    """sha256 over the device sources: ties a committed profile (profiles/pmc_traffic.json) to the kernels it measured."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if not os.path.isfile(os.path.join(CSRC, f)):
            continue
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in os.listdir(CSRC):
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    io_lib = os.path.join(HERE, "libspkio.so")
    if not os.path.exists(io_lib):
        return True
    for f in os.listdir(os.path.join(HERE, "csrc_io")):
        if os.path.getmtime(os.path.join(HERE, "csrc_io", f)) > os.path.getmtime(io_lib):
            return True
    return False


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", os.path.join(CSRC, s), "-o", o] + os.environ.get("SPK_CXXFLAGS", "").split()
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % s)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    # host-only ingest library (no device code)
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O3", "-fPIC", "-shared", "-std=c++17", "-pthread",
                           os.path.join(HERE, "csrc_io", "ark_reader.cpp"), os.path.join(HERE, "csrc_io", "vec_writer.cpp"),
                           "-o", os.path.join(HERE, "libspkio.so")])
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
