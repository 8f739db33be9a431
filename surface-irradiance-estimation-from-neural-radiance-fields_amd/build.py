"""hipcc driver: compiles csrc/ into the in-tree libngp_hip.so (gfx950 only; cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libngp_hip.so")
SOURCES = ["nerf_kernels.hip", "mesh_kernels.hip", "train_kernels.hip", "ngp_api.cpp", "ngp_mesh.cpp", "ngp_train.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


LIB_EXACT = os.path.join(HERE, "libngp_hip_exact.so")  # -DNGP_EXACT_TCNN_ENCODE: tcnn's fp16 rounding sequence in the grid encode


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    for f in os.listdir(CSRC):
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    inc = os.path.join(HERE, "..", "include", "ngp_hip.h")
    return os.path.getmtime(inc) > t


def build(force=False, verbose=False, exact=False):
    """exact=True builds the variant whose hash-grid encode reproduces tcnn's rounding sequence bit for bit (the parity
    tests' reference point for gathers and layout); the default build accumulates with packed fmas, 5 % faster."""
    lib = LIB_EXACT if exact else LIB
    if not force and not needs_build(lib):
        return lib
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc()] + FLAGS + (["-DNGP_EXACT_TCNN_ENCODE"] if exact else []) + ["-o", lib] + srcs + ["-lz"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed")
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
           srcs[0], "-o", out, "-L", HERE, "-lngp_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building pyngp failed")
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
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", srcs[0], "-o", out, "-L", HERE, "-lngp_hip", "-lz", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building ngp_hip_main failed")
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
