"""hipcc driver: compiles csrc/ into the in-tree libngp_hip.so (gfx950 only; cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libngp_hip.so")
SOURCES = ["nerf_kernels.hip", "wide_kernels.hip", "mesh_kernels.hip", "train_kernels.hip", "ngp_api.cpp", "ngp_mesh.cpp", "ngp_train.cpp", "ngp_multi.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


LIB_LEGACY = os.path.join(HERE, "libngp_hip_legacy.so")  # -DNGP_TCNN_LEGACY_ENCODE: the grid encode's corner sum as tcnn had it before its tvec refactor


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    for f in os.listdir(CSRC):
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    inc = os.path.join(HERE, "..", "include", "ngp_hip.h")
    return os.path.getmtime(inc) > t


def build(force=False, verbose=False, legacy=False):
    """The shipped library sums the grid encoding's corners as tvec-era tiny-cuda-nn does (`fma((T)weight, val, result)`),
    bit for bit the oracle's "fma" mode. legacy=True builds the variant with the older published sequence
    (`result[f] += (T)(weight * val[f])`, oracle mode "legacy"): tiny-cuda-nn is un-pinned in the reference, so both are
    kept testable."""
    lib = LIB_LEGACY if legacy else LIB
    if not force and not needs_build(lib):
        return lib
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # link to a private name and rename: another process (a rank of the same job, a test worker) may be dlopen-ing the
    # library at this moment and must see either the old file or the complete new one
    tmp = "%s.%d.tmp" % (lib, os.getpid())
    cmd = [hipcc()] + FLAGS + (["-DNGP_TCNN_LEGACY_ENCODE"] if legacy else []) + os.environ.get("NGP_BUILD_DEFINES", "").split() + ["-o", tmp] + srcs + ["-lz"]  # NGP_BUILD_DEFINES: experiments only
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed")
    os.replace(tmp, lib)
    return lib


def pyngp_path():
    import sysconfig

    return os.path.join(HERE, "pyngp" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pyngp(force=False, verbose=False):
    """pybind11 module `pyngp` (csrc/pyngp.cpp over the C++ Testbed shim), linked against libngp_hip.so."""
    import pybind11
    import sysconfig

    out = pyngp_path()
    srcs = [os.path.join(CSRC, "pyngp.cpp"), os.path.join(CSRC, "testbed_shim.h"), os.path.join(HERE, "..", "include", "ngp_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    build()
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-I", sysconfig.get_paths()["include"], "-I", pybind11.get_include(),
           srcs[0], "-o", out + ".%d.tmp" % os.getpid(), "-L", HERE, "-lngp_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building pyngp failed")
    os.replace(out + ".%d.tmp" % os.getpid(), out)
    return out


def main_path():
    return os.path.join(HERE, "ngp_hip_main")


def build_main(force=False, verbose=False):
    """Headless command line (csrc/ngp_main.cpp: the flags of the reference's src/main.cu), linked against libngp_hip.so."""
    out = main_path()
    srcs = [os.path.join(CSRC, "ngp_main.cpp"), os.path.join(CSRC, "testbed_shim.h"), os.path.join(CSRC, "minijson.h"), os.path.join(HERE, "..", "include", "ngp_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    build()
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", srcs[0], "-o", out + ".%d.tmp" % os.getpid(), "-L", HERE, "-lngp_hip", "-lz", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building ngp_hip_main failed")
    os.replace(out + ".%d.tmp" % os.getpid(), out)
    return out


def import_pyngp():
    """Import the in-tree pyngp extension (build it first if needed)."""
    import importlib.util

    path = build_pyngp()
    spec = importlib.util.spec_from_file_location("pyngp", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_pyngp(force="--force" in sys.argv, verbose=True))
