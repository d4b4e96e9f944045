"""Builds geosss_amd/libgsss_hip.so (the C-ABI + HIP kernels) with hipcc for gfx950.

    python -m geosss_amd.build [--force] [--jobs N]

One object per .hip translation unit (compiled in parallel), linked into a single shared
library that lives IN-TREE next to this file so that it travels with the repository snapshot.
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgsss_hip.so")
ARCH = "gfx950"

CXXFLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]
# experiments: GSSS_HIPCC_FLAGS="-mllvm -foo" python -m geosss_amd.build --force
CXXFLAGS += os.environ.get("GSSS_HIPCC_FLAGS", "").split()
# Per-source flags.  The group-speculative curve kernel keeps ~25 loop-invariant constants (polynomial coefficients, LDS
# offsets) in vector registers for the whole launch when MachineLICM hoists their moves out of the step loop: without it the
# kernel fits three wavefronts per SIMD without spilling (measured: 156 against 168 + 8 spilled registers).  The option is an
# internal LLVM one: it is probed once on an empty translation unit and left out if this compiler does not know it (the
# kernels then spill -- tests/test_abi.py::test_group_kernels_do_not_spill says so instead of a silent 30 GB of scratch traffic).
OPTIONAL_SOURCE_FLAGS = {"gsss_fast_curvespec.hip": ["-mllvm", "-disable-machine-licm"]}
_flag_ok = {}


def flag_supported(flags):
    key = tuple(flags)
    if key not in _flag_ok:
        r = subprocess.run([hipcc(), f"--offload-arch={ARCH}", "--cuda-device-only", *flags, "-x", "hip", "-c", "-", "-o", os.devnull],
                           input="", capture_output=True, text=True)
        _flag_ok[key] = r.returncode == 0
        if not _flag_ok[key]:
            sys.stderr.write(f"geosss_amd.build: {' '.join(flags)} is not accepted by this hipcc, building without it\n")
    return _flag_ok[key]


def source_flags(src):
    flags = OPTIONAL_SOURCE_FLAGS.get(src, [])
    return flags if flags and flag_supported(flags) else []


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "gsss.h"))
    hs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hs)


DIGEST_SOURCE = "gsss_digest.hip"  # carries source_digest() of the whole tree into the library (gsss_source_digest)


def compile_one(src, force, extra, obj_dir=OBJ):
    obj = os.path.join(obj_dir, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    defines, stamp = [], None
    if src == DIGEST_SOURCE:  # rebuilt whenever ANY kernel source moved: the digest it embeds is that of all of them
        digest = source_digest()
        defines, stamp = [f'-DGSSS_SOURCE_DIGEST="{digest}"'], obj + ".digest"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
            return obj, False
    elif not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), headers_mtime()):
        return obj, False
    cmd = [hipcc(), *CXXFLAGS, *source_flags(src), *defines, *extra, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    if stamp:
        open(stamp, "w").write(defines[0].split('"')[1])
    return obj, True


def build(force=False, jobs=None, extra=(), verbose=True, out=None):
    """`out`: write the library (and its objects) under another name -- side-by-side builds for A/B timing
    (load one with GSSS_HIP_LIB=<path>)."""
    lib, obj_dir = LIB, OBJ
    if out:
        lib = os.path.abspath(out)
        obj_dir = os.path.join(CSRC, "_obj_" + os.path.splitext(os.path.basename(lib))[0])
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sources()
    jobs = jobs or min(len(srcs), os.cpu_count() or 1)
    with cf.ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(lambda s: compile_one(s, force, list(extra), obj_dir), srcs))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(lib):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs]
        subprocess.check_call(cmd)
        if verbose:
            print(f"linked {lib}")
    elif verbose:
        print(f"{lib} is up to date")
    return lib


def source_digest():
    """sha256 over the kernels' sources (csrc/*.h, *.hip, *.inc and include/gsss.h, by name): what a profile was taken of.
    bench.py quotes the committed rocprofv3 counters (profiles/traffic.json) only for the sources they were measured on."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".inc"))]
    files.append(os.path.join(HERE, "..", "include", "gsss.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def resource_usage(src, extra=()):
    """hipcc's -Rpass-analysis=kernel-resource-usage remarks of one translation unit (device code only, nothing written):
    {mangled kernel name: {"vgprs", "sgprs", "scratch", "occupancy", "lds"}}."""
    import re
    cmd = [hipcc(), *CXXFLAGS, *source_flags(src), *extra, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(CSRC, src), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-2000:]}")
    out = {}
    for blk in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else None
        out[blk.split()[0]] = {"vgprs": g("VGPRs"), "sgprs": g("TotalSGPRs"), "scratch": g(r"ScratchSize \[bytes/lane\]"),
                               "occupancy": g(r"Occupancy \[waves/SIMD\]"), "lds": g(r"LDS Size \[bytes/block\]")}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--resource-usage", action="store_true", help="print per-kernel VGPR/SGPR/LDS usage")
    ap.add_argument("--out", default=None, help="library path for a side-by-side experimental build")
    a = ap.parse_args()
    extra = ["-Rpass-analysis=kernel-resource-usage"] if a.resource_usage else []
    build(force=a.force or a.resource_usage, jobs=a.jobs, extra=extra, out=a.out)
