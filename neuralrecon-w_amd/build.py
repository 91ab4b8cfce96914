"""Builds libneuconw_hip.so (gfx950) in-tree from csrc/*.hip with hipcc.  No torch involved.

    python -m neuralrecon_w_amd.build        (or __graft_entry__.build())
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# NCW_BUILD_TAG builds an experiment variant beside the product library (scripts/ only; lib.py loads it
# when NEUCONW_HIP_LIB points at it)
TAG = os.environ.get("NCW_BUILD_TAG", "")
OBJ = os.path.join(HERE, "csrc", "build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "libneuconw_hip%s.so" % ("_" + TAG if TAG else ""))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"]
FLAGS += os.environ.get("NCW_EXTRA_HIPCC_FLAGS", "").split()
if TAG:  # probe libraries only: unlocks scripts/probes/ncw_exp_hooks.h (csrc/ncw_common.h #errors on the hooks otherwise)
    FLAGS.append("-DNCW_PROBE_BUILD")
elif any(f.startswith("-DNCW_EXP_") for f in FLAGS):
    raise RuntimeError("NCW_EXP_* timing hooks need NCW_BUILD_TAG=<tag>: they are never compiled into the product library")
# The fine-interleaved MLP kernels (ncw_pp.hip) are VALU-issue-bound next to their MFMAs.  IEEE mode
# makes hipcc quiet every fmin/fmax input with an extra `v_max_f32 x, x, x`, and the SLP vectoriser forms v_pk_fma_f32 /
# v_pk_mul_f32, which cost ~10 issue cycles each beside MFMAs (scripts/probes/issue_probe.hip): both are switched off
# for that file only (measured on the older kernels: sdf_infer -20 %, but sdf_bwd +24 % and nerf_fwd +10 %, so they keep
# the defaults).  Nothing in it branches on NaNs.
MLP_FLAGS = ["-fno-honor-nans", "-mno-amdgpu-ieee", "-fno-slp-vectorize"]
MLP_FILES = {"ncw_pp.hip"}
if TAG and "NCW_MLP_FLAGS" in os.environ:  # A/B probe builds (scripts/): e.g. NCW_MLP_FLAGS="" NCW_BUILD_TAG=plain
    MLP_FLAGS = os.environ["NCW_MLP_FLAGS"].split()
# second flag group (A/B probe builds only): NCW_FLAGS2 applied to the files listed in NCW_FILES2
FLAGS2 = os.environ.get("NCW_FLAGS2", "").split() if TAG else []
FILES2 = set(os.environ.get("NCW_FILES2", "").split()) if TAG else set()


# The fp16 mode (NCW_PREC_F16) is the SAME source compiled a second time with the 16-bit type switched (ncw_common.h:
# ncw_h16 = _Float16, the f16 MFMA, entry points suffixed _f16, kernels in their own namespace); the bf16 objects'
# entry points forward prec == NCW_PREC_F16 to them.
F16_FILES = ["ncw_sdf.hip", "ncw_sdf8.hip", "ncw_sdf16.hip", "ncw_pp.hip", "ncw_color.hip", "ncw_nerf.hip", "ncw_wgrad.hip",
             "ncw_split.hip"]  # ncw_split.hip: fp16 only (its bf16 object is empty)


def source_hash():
    """sha256 over every file the library is built from (csrc/*.hip, csrc/*.h, include/neuconw_hip.h), name + bytes in
    sorted order.  build() compiles it into the library (`ncw_source_hash()`); lib.get_lib() recomputes it and REFUSES a
    library built from other sources -- the GPU box runs the prebuilt .so that gpurun ships, and file times do not survive
    the snapshot, so this is what proves the kernels under test are the ones in the tree."""
    import hashlib

    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "neuconw_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _hash_object():
    """csrc/build/ncw_srchash.o: `const char* ncw_source_hash(void)` of the sources as they are now (host code only)."""
    src = os.path.join(OBJ, "ncw_srchash.cpp")
    obj = os.path.join(OBJ, "ncw_srchash.o")
    text = 'extern "C" const char* ncw_source_hash(void) { return "%s"; }\n' % source_hash()
    if not (os.path.exists(src) and open(src).read() == text and os.path.exists(obj)):
        with open(src, "w") as fh:
            fh.write(text)
        r = subprocess.run(["g++", "-O1", "-fPIC", "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed for ncw_srchash.cpp:\n%s\n%s" % (r.stdout, r.stderr))
        return obj, True
    return obj, False


# ncw_rays.hip is compiled a second time for rays of more than 512 samples (-DNCW_RAYS_BIG: 1088 samples per ray in LDS, entry
# points suffixed _big; the standard object forwards to them) -- the reference's own defaults are 512 + 512 + 32 (config/defaults.py)
BIG_FILES = ["ncw_rays.hip"]


def _sources():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    return [(f, False) for f in srcs] + [(f, True) for f in F16_FILES] + [(f, "big") for f in BIG_FILES]


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "neuconw_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(job):
    src, f16 = job
    obj = os.path.join(OBJ, src[:-4] + ("_big.o" if f16 == "big" else "_f16.o" if f16 else ".o"))
    srcp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + (MLP_FLAGS if src in MLP_FILES else []) + (FLAGS2 if src in FILES2 else []) + (["-DNCW_RAYS_BIG"] if f16 == "big" else ["-DNCW_HALF_F16"] if f16 else [])
    cmd += ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s%s:\n%s\n%s" % (src, " (f16)" if f16 else "", r.stdout, r.stderr))
    return obj, True


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(srcs))) as ex:
        res = list(ex.map(_compile, srcs))
    hobj, hnew = _hash_object()
    objs = [o for o, _ in res] + [hobj]
    rebuilt = any(c for _, c in res) or hnew
    if rebuilt or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("[neuralrecon_w_amd.build] %s (%d objects, %s)" % (LIB, len(objs), "rebuilt" if rebuilt else "up to date"))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
