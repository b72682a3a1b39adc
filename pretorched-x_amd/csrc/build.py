"""Build libptx_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python pretorched-x_amd/csrc/build.py [--force] [--report]

Objects and the shared library are written next to this file's parent package
(`pretorched-x_amd/libptx_amd.so`), in-tree, so the built library travels with the repo
snapshot to the GPU box.  `--report` adds -Rpass-analysis=kernel-resource-usage.

Staleness is decided by CONTENT, not by mtime: every object has a stamp file `<obj>.srchash` holding the sha256 of its
translation unit (the source, every header it can include, the compiler flags); an object is rebuilt when its stamp differs
from the tree's.  The library has a stamp too (`libptx_amd.so.srchash`: the sha256 over the object stamps), and
`ptx_version()` carries the hash over ALL sources (compiled into pack_layout.o), so `stamps_match()` proves for every
object -- not just one -- that the binary in the tree was built from the sources in the tree (tests/test_abi_and_host.py).
"""
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["conv_igemm.hip", "conv_chain.hip", "conv_program.hip", "pack_layout.hip", "pool_head.hip", "nonlocal_attn.hip",
           "conv_stem_x3.hip", "conv_stem_f32.hip", "conv_body_f32.hip", "gen_stage_f16.hip"]
HEADERS = ["ptx_common.h", "conv_igemm_kernel.h", os.path.join("..", "..", "include", "ptx_amd.h")]
LIB = os.path.join(PKG, "libptx_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# per-source extras.  nonlocal_attn: the online-softmax rescale multiplies the MFMA accumulators between tiles; with
# the default AGPR accumulators hipcc copies all of them to VGPRs and back EVERY tile (128 v_accvgpr moves per 128
# MFMAs).  gfx950's register file is unified, so keep the accumulators in VGPRs (0 moves, same occupancy).
EXTRA_FLAGS = {"nonlocal_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
VERSION_SOURCE = "pack_layout.hip"       # defines ptx_version(): compiled with -DPTX_SOURCE_SHA256


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


def obj_path(src):
    return os.path.join(HERE, os.path.splitext(src)[0] + ".o")


def tu_files(src):
    """The source and every project header it includes, transitively (`#include "..."` relative to the including file)."""
    seen, todo = [], [os.path.normpath(os.path.join(HERE, src))]
    while todo:
        path = todo.pop()
        if path in seen:
            continue
        seen.append(path)
        with open(path, "rb") as f:
            text = f.read().decode("utf-8", "replace")
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', text, re.M):
            inc = os.path.normpath(os.path.join(os.path.dirname(path), m.group(1)))
            if os.path.exists(inc):
                todo.append(inc)
    return [seen[0]] + sorted(seen[1:])


def tu_hash(src):
    """sha256 of one translation unit: the source, every project header it includes, the flags it is compiled with (and,
    for the object that carries ptx_version(), the hash over all sources)."""
    h = hashlib.sha256()
    for path in tu_files(src):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(("\0".join(FLAGS + EXTRA_FLAGS.get(src, []))).encode())
    if src == VERSION_SOURCE:
        h.update(source_hash().encode())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def lib_hash():
    """sha256 over the translation-unit hashes, in SOURCES order: what the linked library is made of."""
    return hashlib.sha256("\n".join(tu_hash(s) for s in SOURCES).encode()).hexdigest()


def stamps_match():
    """{name: bool} for every object and the library: does the artefact's stamp equal the hash of the tree's sources?"""
    out = {s: (os.path.exists(obj_path(s)) and _read(obj_path(s) + ".srchash") == tu_hash(s)) for s in SOURCES}
    out["libptx_amd.so"] = os.path.exists(LIB) and _read(LIB + ".srchash") == lib_hash()
    return out


def _compile(src, report):
    obj = obj_path(src)
    stamp, want = obj + ".srchash", tu_hash(src)
    if os.path.exists(obj) and _read(stamp) == want:
        return obj, "", False
    extra = ['-DPTX_SOURCE_SHA256="%s"' % source_hash()] if src == VERSION_SOURCE else []
    cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + extra + (["-Rpass-analysis=kernel-resource-usage"] if report else []) + \
          ["-c", os.path.join(HERE, src), "-o", obj]
    if os.path.exists(stamp):
        os.remove(stamp)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
    if tu_hash(src) != want:              # edited while it compiled: leave it unstamped, the next build redoes it
        return obj, r.stdout, True
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return obj, r.stdout, True


def build(force=False, report=False, verbose=True):
    if force:
        for s in SOURCES:
            for path in (obj_path(s), obj_path(s) + ".srchash"):
                if os.path.exists(path):
                    os.remove(path)
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(lambda s: _compile(s, report), SOURCES))
    objs = [o for o, _, _ in results]
    log = "".join(l for _, l, _ in results)
    want = lib_hash()
    if not os.path.exists(LIB) or _read(LIB + ".srchash") != want or any(c for _, _, c in results):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
        if all(_read(obj_path(s) + ".srchash") == tu_hash(s) for s in SOURCES):
            with open(LIB + ".srchash", "w") as f:
                f.write(want + "\n")
    if verbose and log:
        print(log)
    return LIB


if __name__ == "__main__":
    lib = build(force="--force" in sys.argv, report="--report" in sys.argv)
    print("built", lib, os.path.getsize(lib), "bytes")
