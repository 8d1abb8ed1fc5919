"""Builds the native pieces of the package in-tree (the .so files travel to the GPU box with the repo).

  libapriltag_amd.so    HIP kernels + C ABI (hipcc, gfx950 only)         <- csrc/detector.hip
  libapriltag_synth.so  deterministic frame renderer (host C)            <- csrc/synth_render.c
  libapriltag_node.so   ROS-free C++ mirror of the reference node shell  <- csrc/node_shell.cpp (if present)
  examples/multi_stream_host   C++ multi-GPU front end (RCCL broadcast)  <- examples/multi_stream_host.cpp
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_AMD = os.path.join(_HERE, "libapriltag_amd.so")
LIB_SYNTH = os.path.join(_HERE, "libapriltag_synth.so")
LIB_NODE = os.path.join(_HERE, "libapriltag_node.so")


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(*exts):
    inc = os.path.join(os.path.dirname(_HERE), "include")
    out = []
    for d in (_CSRC, inc):
        for f in sorted(os.listdir(d)):
            if f.endswith(exts):
                out.append(os.path.join(d, f))
    return out


def build_amd(force=False):
    srcs = _sources(".hip", ".h")
    if force or _newer(LIB_AMD, srcs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-value", "-Wno-unused-function", "-Wno-pass-failed",
               os.path.join(_CSRC, "detector.hip"), "-o", LIB_AMD]
        subprocess.check_call(cmd)
    return LIB_AMD


def build_amd_variant(tag, defines):
    """Measurement builds for tools/ (e.g. -DAMDAT_FQ_PROFILE: per-phase cycle counters in the quad-fit kernel).
    The product library carries none of this; the variant is written next to it as libapriltag_amd_<tag>.so.
    An entry of `defines` that starts with "-" is passed to the compiler as it stands (e.g. "-mllvm", "-amdgpu-sched-strategy=max-ilp")."""
    out = os.path.join(_HERE, "libapriltag_amd_%s.so" % tag)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-fast-math", "-Wno-unused-value", "-Wno-unused-function", "-Wno-pass-failed"] + \
          [d if d.startswith("-") else "-D" + d for d in defines] + \
          [os.path.join(_CSRC, "detector.hip"), "-o", out]
    subprocess.check_call(cmd)
    return out


def build_synth(force=False):
    src = os.path.join(_CSRC, "synth_render.c")
    if force or _newer(LIB_SYNTH, [src] + _sources(".h")):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=gnu99", "-ffp-contract=off", "-shared", "-o", LIB_SYNTH,
                               src, "-lm"])
    return LIB_SYNTH


def build_node(force=False):
    src = os.path.join(_CSRC, "node_shell.cpp")
    if not os.path.exists(src):
        return None
    if force or _newer(LIB_NODE, [src] + _sources(".h", ".hpp")):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", LIB_NODE, src, "-ldl"])
    return LIB_NODE


HOST_BIN = os.path.join(os.path.dirname(_HERE), "examples", "multi_stream_host")


def build_host(force=False):
    """C++ multi-GPU front end (one handle per device, RCCL broadcast of the parameter block)."""
    src = os.path.join(os.path.dirname(_HERE), "examples", "multi_stream_host.cpp")
    if not os.path.exists(src):
        return None
    if force or _newer(HOST_BIN, [src, LIB_AMD] + _sources(".h")):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(os.path.dirname(_HERE), "include"),
                               src, "-L" + _HERE, "-lapriltag_amd", "-lrccl", "-Wl,-rpath," + _HERE, "-o", HOST_BIN])
    return HOST_BIN


LIB_STRESS = os.path.join(_HERE, "libapriltag_amd_stress.so")


def build_stress(force=False):
    """Stress build for the GPU suite (tests/test_gpu_parity.py::test_long_staging_records...): a 64 x 16 tile's emission list
    is cut to 768 entries, so that on ordinary frames a good share of the boundary points takes the long-record path that the
    product build only enters above two emissions per pixel of a tile; and the cluster list starts at 1024 entries instead of
    65 536, so that the noisy 1080p frames (4 000 clusters) make it grow twice.  Never loaded by the product path."""
    if force or _newer(LIB_STRESS, [os.path.join(_CSRC, "detector.hip")] + _sources(".h")):
        build_amd_variant("stress", ["PT_ELIST=768", "AMDAT_LCAP_DIV=1", "AMDAT_CCAP0=1024u"])
    return LIB_STRESS


MUTANTS = (1, 2, 3)   # csrc/tools_hooks.h, AMDAT_MUTATE


def lib_mutant(n):
    return os.path.join(_HERE, "libapriltag_amd_mut%d.so" % n)


def build_mutants(force=False):
    """Deliberately WRONG builds of the same sources (csrc/tools_hooks.h: -DAMDAT_MUTATE=1 the launch sequence without k_fit_small,
    2 one row constant off in k_cc_local<4>, 3 one sector too many in k_fit_prefilter<64>'s test), shipped like the stress build so
    that the GPU suite itself shows, under the driver's eyes, that its stage tests FAIL on each of them
    (tests/test_gpu_parity.py::test_the_suite_fails_on_wrong_builds).  Never loaded by the product path."""
    import threading
    err = []

    def _one(n):
        try:
            if force or _newer(lib_mutant(n), [os.path.join(_CSRC, "detector.hip")] + _sources(".h")):
                build_amd_variant("mut%d" % n, ["AMDAT_MUTATE=%d" % n])
        except (subprocess.CalledProcessError, OSError) as e:
            err.append(e)
    ts = [threading.Thread(target=_one, args=(n,)) for n in MUTANTS]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if err:
        raise err[0]
    return [lib_mutant(n) for n in MUTANTS]


def build_all(force=False):
    import threading
    build_synth(force)
    err = []

    def _stress():
        try:
            build_stress(force)
        except (subprocess.CalledProcessError, OSError) as e:   # only its test needs it
            err.append(e)

    def _mutants():
        try:
            build_mutants(force)
        except (subprocess.CalledProcessError, OSError) as e:   # only their test needs them
            err.append(e)
    t = threading.Thread(target=_stress)
    t.start()                      # (next to the product library: hipcc runs of the same translation unit)
    build_amd(force)
    t.join()
    tm = threading.Thread(target=_mutants)
    tm.start()
    tm.join()
    if err:
        import sys
        sys.stderr.write("isaac_ros_apriltag_amd.build: stress / mutant variants not built (%s); only their tests need them\n" % (err[0],))
    build_node(force)
    try:   # the multi-GPU example links RCCL; a machine without it still gets the product libraries
        build_host(force)
    except (subprocess.CalledProcessError, OSError) as e:
        import sys
        sys.stderr.write("isaac_ros_apriltag_amd.build: examples/multi_stream_host not built (%s); only its test needs it\n" % (e,))


if __name__ == "__main__":
    build_all(force=True)
    print("built", LIB_AMD, LIB_SYNTH)
