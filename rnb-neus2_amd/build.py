"""Compiles csrc/ into librnb_neus2_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo)."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "rnb_neus2_hip.hip")
OUT = os.path.join(PKG_DIR, "librnb_neus2_hip.so")
DEPS = [os.path.join(PKG_DIR, "csrc", f) for f in ("rnb_neus2_hip.hip", "common.cuh", "mlp.cuh", "chain.cuh", "kernels_net.cuh", "kernels_ray.cuh", "kernels_mesh.cuh")] + [
    os.path.join(PKG_DIR, "host", f) for f in ("mesh.hpp", "mc_table.hpp")] + [
    os.path.join(os.path.dirname(PKG_DIR), "include", "rnb_neus2.h")]

# -ffp-contract=off: the index/ray arithmetic must match the CPU checker bit for bit (no FMA contraction).
# -packed-fp32-ops (device target feature): no v_pk_{mul,add,fma}_f32. Measured on MI355X / ROCm 7.2 with tools/march_determinism.py: the
# march of the NEXT step runs on a side stream beside this step's backward pass, and with packed fp32 instructions in it a few
# rays of wavefront lanes 48-63 came out with a wrong direction (one component of R * d_cam, which the compiler had put on the
# packed pipe) in 0.8-3 % of the launches -- never on an idle GPU, and never (0 of 2000 launches) once, in addition, the march
# kernel stopped keeping ballot masks in SGPRs that spilled through VGPR lanes (rnb_neus2_hip.hip, launch_premarch). The x86 half of the
# compilation does not know the feature and says so on stderr; build() drops those lines.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    noise = "is not a recognized feature for this target"
    rest = "\n".join(line for line in res.stderr.splitlines() if noise not in line)
    if rest:
        print(rest)
    if res.returncode != 0:
        raise subprocess.CalledProcessError(res.returncode, cmd)
    return OUT


ROOT = os.path.dirname(PKG_DIR)
TESTBED_SRC = os.path.join(PKG_DIR, "host", "testbed_main.cpp")
TESTBED_OUT = os.path.join(ROOT, "build", "testbed")
TESTBED_DEPS = [os.path.join(PKG_DIR, "host", f) for f in ("testbed_main.cpp", "dataset.hpp", "json_min.hpp", "png16.hpp", "msgpack_min.hpp", "mesh.hpp", "mc_table.hpp", "dist_transport.hpp")] + [
    os.path.join(ROOT, "include", "rnb_neus2.h")]


def build_testbed(force=False, verbose=False):
    """`build/testbed`: the reference's command line (src/main.cu) over the C-ABI; plain g++, links the HIP library + zlib."""
    if not force and os.path.exists(TESTBED_OUT):
        t = os.path.getmtime(TESTBED_OUT)
        if not any(os.path.getmtime(d) > t for d in TESTBED_DEPS):
            return TESTBED_OUT
    os.makedirs(os.path.dirname(TESTBED_OUT), exist_ok=True)
    # RNB_WITH_RCCL: one process per GPU over RCCL (tools/launch_testbed.sh); the CPU-checker build of the same file (tests/) leaves it out,
    # and so does a ROCm installation without the RCCL development files (RNB_NO_RCCL=1 forces that): the single-GPU command line needs none of it
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    rccl_libdir = next((os.path.join(rocm, d) for d in ("lib", "lib64") if os.path.exists(os.path.join(rocm, d, "librccl.so"))), None)
    with_rccl = not os.environ.get("RNB_NO_RCCL") and rccl_libdir is not None and os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h"))
    hip_libdir = next((os.path.join(rocm, d) for d in ("lib", "lib64") if os.path.exists(os.path.join(rocm, d, "libamdhip64.so"))), os.path.join(rocm, "lib"))
    # the HIP headers / runtime are linked whether or not RCCL is there (the file may call HIP outside its RNB_WITH_RCCL blocks)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-D__HIP_PLATFORM_AMD__", "-DRNB_WITH_HIP", "-I" + os.path.join(rocm, "include"), TESTBED_SRC, "-o", TESTBED_OUT,
           "-L" + PKG_DIR, "-lrnb_neus2_hip", "-lz", "-L" + hip_libdir, "-lamdhip64", "-Wl,-rpath,$ORIGIN/../rnb-neus2_amd", "-Wl,-rpath," + hip_libdir]
    if with_rccl:
        cmd += ["-DRNB_WITH_RCCL", "-L" + rccl_libdir, "-lrccl", "-Wl,-rpath," + rccl_libdir]
    elif verbose:
        print("build_testbed: no RCCL development files under %s -- single-GPU command line only" % rocm)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return TESTBED_OUT


HOSTLIB_SRC = os.path.join(PKG_DIR, "host", "hostlib.cpp")
HOSTLIB_OUT = os.path.join(PKG_DIR, "librnb_host.so")
HOSTLIB_DEPS = [HOSTLIB_SRC, os.path.join(PKG_DIR, "host", "png16.hpp"), os.path.join(ROOT, "include", "rnb_host.h")]


def build_hostlib(force=False, verbose=False):
    """librnb_host.so: CPU helpers of the Python preparation stages (PNG codec, mesh ray casting); g++ + zlib + OpenMP."""
    if not force and os.path.exists(HOSTLIB_OUT):
        t = os.path.getmtime(HOSTLIB_OUT)
        if not any(os.path.getmtime(d) > t for d in HOSTLIB_DEPS):
            return HOSTLIB_OUT
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wall", HOSTLIB_SRC, "-o", HOSTLIB_OUT, "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOSTLIB_OUT


if __name__ == "__main__":
    print(build_hostlib(force=True, verbose=True))
    print(build(force=True, verbose=True))
    print(build_testbed(force=True, verbose=True))
