"""Build libcmfhip.so (gfx950 only) in-tree with hipcc.

    python -m pycmf_amd.build [--force] [--verbose] [--diag]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container.  The .so stays next to this file (git-ignored) so that it travels
with the working tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcmfhip.so")
SOURCES = ["cmf_api.hip"]
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join(ROOT, "include", "cmfhip.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libcmfhip.so cannot be built")
    return exe


def source_hash():
    """sha256 over the bytes of every source the library is compiled from (csrc/*, include/cmfhip.h), in a fixed order: compiled
    into the library (``cmf_source_hash()``) and compared by ``_lib.load()`` with the sources next to it -- a library built from
    other sources than the ones in the tree is refused instead of silently used."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


STAMP = b"cmfhip-source-sha256:"


def library_hash(path=LIB):
    """The source hash a built library carries (None: no library, or one from before the stamp).  Read from the file's BYTES:
    dlopen-ing the product library here would pin the old image in this process (glibc returns the mapped image by name, so the
    `_lib.load()` that follows a rebuild would still see the stale one) and would load a HIP runtime before
    `_lib._preload_hip_runtime()` has chosen which."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        blob = f.read()
    i = blob.find(STAMP)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(STAMP):j].decode("ascii", "replace")


def needs_build():
    return library_hash() != source_hash()


def build(force=False, verbose=False, diag=False):
    """diag: also compile the timing-only diagnostic kernel variants (row_diag / chol_diag / cmf_debug_clock: wrong results,
    their probe scripts left the tree in round 6: git history); the default product build does not carry them."""
    if not force and not diag and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-DCMF_SOURCE_HASH=\"%s\"" % source_hash(),
           "-I", os.path.join(ROOT, "include"), "-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if diag:
        cmd.append("-DCMF_DIAG_BUILD")
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)  # a new inode: an image of the old file that some process has mapped is never written over
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, diag="--diag" in sys.argv)
    print(LIB)
