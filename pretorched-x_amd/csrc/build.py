"""Build libptx_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python pretorched-x_amd/csrc/build.py [--force] [--report]

Objects and the shared library are written next to this file's parent package
(`pretorched-x_amd/libptx_amd.so`), in-tree, so the built library travels with the repo
snapshot to the GPU box.  `--report` adds -Rpass-analysis=kernel-resource-usage.
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["conv_igemm.hip", "conv_chain.hip", "pack_layout.hip", "pool_head.hip", "nonlocal_attn.hip", "conv_stem_x3.hip", "conv_stem_f32.hip", "gen_stage_f16.hip"]
HEADERS = ["ptx_common.h", "conv_igemm_kernel.h", os.path.join("..", "..", "include", "ptx_amd.h")]
LIB = os.path.join(PKG, "libptx_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# per-source extras.  nonlocal_attn: the online-softmax rescale multiplies the MFMA accumulators between tiles; with
# the default AGPR accumulators hipcc copies all of them to VGPRs and back EVERY tile (128 v_accvgpr moves per 128
# MFMAs).  gfx950's register file is unified, so keep the accumulators in VGPRs (0 moves, same occupancy).
EXTRA_FLAGS = {"nonlocal_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def source_hash():
    """sha256 over every source the library is compiled from (csrc/*.hip, csrc/*.h, include/ptx_amd.h; sorted by file
    name, each as name + NUL + bytes).  Compiled into ptx_version() so a test run can prove WHICH source the loaded
    binary was built from (the .so travels prebuilt to the GPU box)."""
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(HERE) if f.endswith((".hip", ".h")))
    for path in [os.path.join(HERE, f) for f in files] + [os.path.join(HERE, "..", "..", "include", "ptx_amd.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


VERSION_SOURCE = "pack_layout.hip"       # defines ptx_version(): compiled with -DPTX_SOURCE_SHA256


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, report):
    obj = os.path.join(HERE, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    extra, stamp, sha = [], obj + ".srchash", None
    if src == VERSION_SOURCE:            # the object that carries the hash is rebuilt whenever ANY source changed
        sha = source_hash()
        extra = ['-DPTX_SOURCE_SHA256="%s"' % sha]
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if have != sha and os.path.exists(obj):
            os.remove(obj)
    if not _stale(obj, deps):
        return obj, ""
    cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + extra + (["-Rpass-analysis=kernel-resource-usage"] if report else []) + \
          ["-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
    if sha is not None:
        with open(stamp, "w") as f:
            f.write(sha + "\n")
    return obj, r.stdout


def build(force=False, report=False, verbose=True):
    if force:
        for s in SOURCES:
            o = os.path.join(HERE, os.path.splitext(s)[0] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(lambda s: _compile(s, report), SOURCES))
    objs = [o for o, _ in results]
    log = "".join(l for _, l in results)
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
    if verbose and log:
        print(log)
    return LIB


if __name__ == "__main__":
    lib = build(force="--force" in sys.argv, report="--report" in sys.argv)
    print("built", lib, os.path.getsize(lib), "bytes")
