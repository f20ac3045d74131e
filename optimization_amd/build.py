"""Build libmi355opt.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

Usage:  python -m optimization_amd.build [--force] [-j N]
hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# MI355OPT_BUILD_TAG=x: an experiment build next to the product one (objects in _build_x/, library
# libmi355opt_x.so; load it with MI355OPT_LIB=... -- same-call A/B runs on the GPU box)
_TAG = os.environ.get("MI355OPT_BUILD_TAG", "")
OBJ = os.path.join(HERE, "_build" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(HERE, "libmi355opt" + ("_" + _TAG if _TAG else "") + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

CFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
    "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-I", os.path.join(HERE, "include"), "-I", "/opt/rocm/include",
]
CFLAGS += os.environ.get("MI355OPT_EXTRA_CFLAGS", "").split()  # experiments (e.g. -DMI_SPMM_CHUNK=8)
LDFLAGS = ["-shared", "-L/opt/rocm/lib", "-lrccl", "-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(f) > t for f in [src] + deps)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        glob.glob(os.path.join(HERE, "include", "Optimization", "LinearAlgebra", "DenseSymmetricEigen.h"))
    if not force and not _newer(src, obj, deps):
        return obj, False, ""
    cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
    return obj, True, r.stderr


# Host-only sources (csrc/*.cpp): g++ with AVX2, fused multiply-adds OFF (bit-identical to a scalar build; see
# csrc/rr_host.cpp); hipcc's host pass when there is no g++
HOST_CXX = os.environ.get("CXX", "g++")
# No -mavx2 on the command line: csrc/rr_host.cpp carries its own `target_clones("avx2","default")` on the hot solver,
# so the library also loads (and gives the same bits: no FMA, no reassociation) on a host CPU without AVX2.
# -ffp-contract=off is REQUIRED for the documented bit-for-bit agreement of the host and device Rayleigh-Ritz paths; a
# client that compiles DenseSymmetricEigen.h itself (the generic LOBPCG path) needs the same flag for that claim.
HOST_CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wall",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-I", os.path.join(HERE, "include")]


def _compile_host(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    deps = glob.glob(os.path.join(HERE, "include", "Optimization", "LinearAlgebra", "*.h"))
    if not force and not _newer(src, obj, deps):
        return obj, False, ""
    import shutil
    if shutil.which(HOST_CXX):
        cmd = [HOST_CXX] + HOST_CFLAGS
    else:  # no g++ on this machine: hipcc's host pass (clang), same flags, no device code
        cmd = [HIPCC, "-x", "c++"] + HOST_CFLAGS
    r = subprocess.run(cmd + ["-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{cmd[0]} failed on {src}:\n{r.stderr[-6000:]}")
    return obj, True, r.stderr


# Sources compiled as ONE translation unit = one code object: the four kernels of an STPCG iteration
# (k_st_spmm_gram, k_st_finish, k_cg_update, k_cg_pupdate) then sit next to each other in device memory
# instead of in separately loaded code objects at unrelated addresses.
UNITY = ["stpcg.hip", "stiefel.hip"]


def build(force=False, jobs=None, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    if os.environ.get("MI355OPT_NO_UNITY") != "1":
        unity = os.path.join(OBJ, "hot_unity.hip")
        text = "".join(f'#include "{os.path.join(CSRC, u)}"\n' for u in UNITY)
        if not os.path.exists(unity) or open(unity).read() != text:
            with open(unity, "w") as f:
                f.write(text)
        newest = max(os.path.getmtime(os.path.join(CSRC, u)) for u in UNITY)
        if os.path.getmtime(unity) < newest:
            os.utime(unity, (newest, newest))
        srcs = [s_ for s_ in srcs if os.path.basename(s_) not in UNITY] + [unity]
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    objs, rebuilt = [], False
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        for obj, did, err in ex.map(lambda s: _compile(s, force), srcs):
            objs.append(obj)
            rebuilt |= did
            if verbose and err.strip():
                print(err, file=sys.stderr)
    for src in sorted(glob.glob(os.path.join(CSRC, "*.cpp"))):
        obj, did, err = _compile_host(src, force)
        objs.append(obj)
        rebuilt |= did
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, f"--offload-arch={ARCH}"] + objs + LDFLAGS + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-6000:]}")
    return LIB


WLGEN_SRC = os.path.join(HERE, "wlgen.c")
WLGEN_LIB = os.path.join(HERE, "libmi355wl.so")


def build_wlgen(force=False):
    """optimization_amd/libmi355wl.so: the machine-independent input generators of workloads.py (mt19937_64, own sin,
    own thin QR; wlgen.c).  Plain gcc, SSE2 baseline, contraction off: the same bytes on any x86-64 host."""
    if force or _newer(WLGEN_SRC, WLGEN_LIB, []):
        r = subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra", WLGEN_SRC,
                            "-o", WLGEN_LIB, "-lm"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("wlgen build failed:\n" + r.stderr[-4000:])
    return WLGEN_LIB


def build_harness(force=False):
    """C++ test harnesses on top of the drop-in template layer (g++; plus a clang syntax check so
    the headers stay compilable by hipcc's front end):
      tests/cpp/libharness_host.so    templates on a host vector, same driver code as the real
                                      reference build (oracle/template_driver.inc)
      tests/cpp/libharness_device.so  templates on MI355::DeviceVector, linked to libmi355opt.so"""
    tdir = os.path.join(ROOT, "tests", "cpp")
    inc = ["-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "oracle")]
    common = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-type-limits"]
    hdrs = glob.glob(os.path.join(HERE, "include", "Optimization", "*", "*.h")) + \
        glob.glob(os.path.join(ROOT, "oracle", "*.inc")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    out = []
    host_src = os.path.join(tdir, "harness_host.cpp")
    host_so = os.path.join(tdir, "libharness_host.so")
    if force or _newer(host_src, host_so, hdrs):
        r = subprocess.run(common + ["-march=x86-64-v3"] + inc + [host_src, "-o", host_so],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host harness build failed:\n" + r.stderr[-6000:])
    out.append(host_so)
    dev_src = os.path.join(tdir, "harness_device.cpp")
    dev_so = os.path.join(tdir, "libharness_device.so")
    if force or _newer(dev_src, dev_so, hdrs + [LIB]):
        cmd = common + inc + ["-I", os.path.join(ROOT, "include"), dev_src, "-o", dev_so, "-L", HERE,
                              "-lmi355opt", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("device harness build failed:\n" + r.stderr[-6000:])
        # the same translation unit must also pass clang's front end (hipcc users)
        clang = "/opt/rocm/lib/llvm/bin/clang++"
        if os.path.exists(clang):
            r = subprocess.run([clang, "-std=c++17", "-fsyntax-only"] + inc +
                               ["-I", os.path.join(ROOT, "include"), dev_src], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("clang front-end check of the template layer failed:\n" + r.stderr[-6000:])
    out.append(dev_so)
    # tests/cpp/harness_sinfit.hip: the reference's TNLS sin-fit problem with HIP kernels of its own (hipcc)
    sf_src = os.path.join(tdir, "harness_sinfit.hip")
    sf_so = os.path.join(tdir, "libharness_sinfit.so")
    if force or _newer(sf_src, sf_so, hdrs + [LIB]):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall",
               "-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include"), sf_src, "-o", sf_so,
               "-L", HERE, "-lmi355opt", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("sin-fit harness build failed:\n" + r.stderr[-6000:])
    out.append(sf_so)
    # tools/bench_client.cpp: the client of the drop-in headers behind bench.py's cfg3 / cfg5 legs (no oracle includes)
    bc_src = os.path.join(ROOT, "tools", "bench_client.cpp")
    bc_so = os.path.join(ROOT, "tools", "libbench_client.so")
    if force or _newer(bc_src, bc_so, hdrs + [LIB]):
        cmd = common + ["-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include"), bc_src, "-o", bc_so,
                        "-L", HERE, "-lmi355opt", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("bench client build failed:\n" + r.stderr[-6000:])
    out.append(bc_so)
    # examples/*.cpp: stand-alone client programs of the drop-in headers (binaries under examples/bin/)
    exdir = os.path.join(ROOT, "examples")
    os.makedirs(os.path.join(exdir, "bin"), exist_ok=True)
    for src in sorted(glob.glob(os.path.join(exdir, "*.cpp"))):
        exe = os.path.join(exdir, "bin", os.path.splitext(os.path.basename(src))[0])
        if force or _newer(src, exe, hdrs + [LIB]):
            cmd = ["g++", "-std=c++17", "-O2", "-Wall"] + inc + ["-I", os.path.join(ROOT, "include"), src, "-o", exe,
                   "-L", HERE, "-lmi355opt", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"example build failed ({src}):\n" + r.stderr[-6000:])
        out.append(exe)
    # examples/*.hip: clients that bring HIP kernels of their own (a user-written Hessian-vector product): hipcc
    for src in sorted(glob.glob(os.path.join(exdir, "*.hip"))):
        exe = os.path.join(exdir, "bin", os.path.splitext(os.path.basename(src))[0])
        if force or _newer(src, exe, hdrs + [LIB]):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-std=c++17", "-O2", "-Wall"] + inc + \
                  ["-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", HERE, "-lmi355opt",
                   "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"example build failed ({src}):\n" + r.stderr[-6000:])
        out.append(exe)
    return out


class SanitizerUnavailable(RuntimeError):
    """The host has no usable ASan / UBSan toolchain or runtime: a reason to SKIP the sanitizer target, never a reason
    to refuse the GPU library."""


def _sanitizer_probe(tdir, san):
    """Can this host compile, link and RUN a trivial program under the sanitizers (libasan / libubsan installed, no
    ptrace / seccomp restriction on LeakSanitizer)?  Returns the ASAN_OPTIONS to run with, or raises."""
    import tempfile
    with tempfile.TemporaryDirectory(dir=tdir) as d:
        src, exe = os.path.join(d, "p.c"), os.path.join(d, "p")
        with open(src, "w") as f:
            f.write("#include <stdlib.h>\nint main(void){void*p=malloc(8);free(p);return 0;}\n")
        try:
            r = subprocess.run(["gcc"] + san + [src, "-o", exe], capture_output=True, text=True)
        except FileNotFoundError as e:
            raise SanitizerUnavailable(str(e))
        if r.returncode != 0:
            raise SanitizerUnavailable("gcc cannot build with -fsanitize=address,undefined here:\n" + r.stderr[-800:])
        for opts in ("detect_leaks=1:abort_on_error=0", "detect_leaks=0:abort_on_error=0"):
            r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS=opts))
            if r.returncode == 0:
                if "detect_leaks=0" in opts:
                    print("[build] warning: LeakSanitizer cannot run here (ptrace / seccomp restriction?): the sanitizer "
                          "target runs with detect_leaks=0 -- heap errors and undefined behaviour are still checked, "
                          "leaks are NOT", file=sys.stderr)
                return opts
        raise SanitizerUnavailable("a trivial sanitized program does not run here:\n" + r.stderr[-800:])


def build_sanitize(force=False, run=True):
    """tests/cpp/sanitize_host: the template layer on a host vector (the host harness's translation unit + a main that
    drives TNT, GradientDescent, LSQR, TNLS, LOBPCG once) under -fsanitize=address,undefined; run here, any REPORT is a
    build failure (the sanitizer / host-hardening target).  A host without the sanitizer runtime raises
    SanitizerUnavailable, which __graft_entry__.build() turns into a warning."""
    tdir = os.path.join(ROOT, "tests", "cpp")
    odir = os.path.join(ROOT, "oracle")
    exe = os.path.join(tdir, "sanitize_host")
    src = os.path.join(tdir, "sanitize_host.cpp")
    deps = glob.glob(os.path.join(HERE, "include", "Optimization", "*", "*.h")) + glob.glob(os.path.join(odir, "*.inc")) + \
        glob.glob(os.path.join(odir, "*.[ch]")) + [os.path.join(tdir, "harness_host.cpp")]
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1",
           "-ffp-contract=off"]
    asan_opts = _sanitizer_probe(tdir, san)
    if force or _newer(src, exe, deps):
        objs = []
        for c in ("oracle.c", "problems.c"):
            o = os.path.join(tdir, "san_" + c + ".o")
            r = subprocess.run(["gcc"] + san + ["-I", odir, "-c", os.path.join(odir, c), "-o", o],
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("sanitizer build failed:\n" + r.stderr[-4000:])
            objs.append(o)
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wno-type-limits"] + san +
                           ["-I", os.path.join(HERE, "include"), "-I", odir, "-I", tdir, src] + objs + ["-lm", "-o", exe],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("sanitizer build failed:\n" + r.stderr[-6000:])
    if run:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS=asan_opts, UBSAN_OPTIONS="print_stacktrace=1"))
        if r.returncode != 0 or "sanitize_host: ok" not in r.stdout:
            raise RuntimeError("the template layer failed under ASan/UBSan:\n" + r.stdout[-2000:] + r.stderr[-6000:])
    return exe


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build(force=force, verbose=True))
    print(build_wlgen(force=force))
    print(build_harness(force=force))
    print(build_sanitize(force=force))
