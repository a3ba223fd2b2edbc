"""Build libspkhip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspkhip.so")
SOURCES = ["err.cpp", "conv_mfma.hip", "conv_wgrad.hip", "stem.hip", "bn.hip", "pool.hip", "gemm.hip", "head.hip",
           "sgd.hip", "score.hip", "conv_split.hip", "conv_wgrad_split.hip", "pack.hip", "conv_pipe.hip", "conv_wgrad_1x1.hip",
           "conv_wgrad_wm.hip", "conv_wgrad_wm16.hip", "conv1x1_stream.hip", "conv3x3_c32_stream.hip"]
# kernel forms that were measured and did not pay: producer / consumer convolution and weight gradient,
# in-wave pipelined weight gradient, in-wave pipelined fused-BatchNorm-backward data gradient.  Kept as source for reference and
# for the bit-identity tests; compiled only with SPK_EXPERIMENTAL=1 (-DSPK_EXPERIMENTAL; spk_build_flags() & 1)
SOURCES_EXPERIMENTAL = ["conv_ws.hip", "conv_wgrad_pipe.hip"]


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


def build(force=False, verbose=True, experimental=None, lib=None):
    """experimental (default: env SPK_EXPERIMENTAL == "1"): also compile the measured, not-faster kernel forms; lib: output path
    (default: the in-tree libspkhip.so)"""
    if experimental is None:
        experimental = os.environ.get("SPK_EXPERIMENTAL", "0") == "1"
    lib = lib or LIB
    if not force and lib == LIB and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build_exp" if experimental else "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    extra = ["-DSPK_EXPERIMENTAL"] if experimental else []
    for s in SOURCES + (SOURCES_EXPERIMENTAL if experimental else []):
        o = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", os.path.join(CSRC, s), "-o", o] + extra + os.environ.get("SPK_CXXFLAGS", "").split()
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % s)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    subprocess.check_call(cmd)
    # host-only ingest library (no device code)
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O3", "-fPIC", "-shared", "-std=c++17", "-pthread",
                           os.path.join(HERE, "csrc_io", "ark_reader.cpp"), os.path.join(HERE, "csrc_io", "vec_writer.cpp"),
                           "-o", os.path.join(HERE, "libspkio.so")])
    if verbose:
        print("built", lib)
    return lib


if __name__ == "__main__":
    if "--experimental" in sys.argv:       # the variant library with every kernel form: variants/libspkhip_exp.so (SPK_LIB=...)
        os.makedirs(os.path.join(HERE, "variants"), exist_ok=True)
        build(force=True, experimental=True, lib=os.path.join(HERE, "variants", "libspkhip_exp.so"))
    else:
        build(force="--force" in sys.argv)
