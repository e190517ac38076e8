"""Builds libgpfq_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container; the
resulting .so travels to the GPU box with the source tree (it is git-ignored, not gpurun-ignored).

Every translation unit is compiled to its own object (in parallel) and the objects are linked into the
library.  Staleness is decided by CONTENT, not by time stamps (a snapshot of the tree does not keep them):
`libgpfq_hip.so.sha` holds the hash of all sources, headers and flags the library was built from, and
hip.load() refuses a library whose hash does not match the tree.
"""
import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libgpfq_hip.so")
STAMP = LIB + ".sha"
OBJDIR = os.path.join(CSRC, "build")
SOURCES = ["gpfq_capi.hip", "gpfq_onchip.hip", "gpfq_rows.hip", "gpfq_pipe.hip", "gpfq_blk.hip", "gpfq_wide.hip", "gpfq_stream.hip",
           "gpfq_gram.hip", "gpfq_gram_image.hip", "gpfq_gram_conv.hip", "gpfq_gram_s2.hip", "gpfq_gram_mfma.hip", "gpfq_misc.hip"]
HEADERS = ["gpfq_device.hpp", "gpfq_launch.hpp", "gpfq_gram_tile.hpp", "gpfq_roles.hpp", "gpfq_blk_diag.hpp", os.path.join("..", "..", "include", "gpfq.h")]

# -ffp-contract=off: the float32 products/subtraction of the residual update must round
# separately (reference numerics); float64 accumulations use explicit fma().
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall"]
# gpfq_pipe.hip writes its packed float32 operations itself; the SLP vectoriser pairs unrelated products
# through extra register moves there.
EXTRA_FLAGS = {"gpfq_pipe.hip": ["-fno-slp-vectorize"], "gpfq_blk.hip": ["-fno-slp-vectorize"]}
# diagnostic builds (never the shipped library): GPFQ_DIAG="-DGPFQ_BLK_STAMPS" adds in-kernel phase stamps to gpfq_blk.hip,
# "-DGPFQ_WIDE_STAMPS" to the several-wavefronts-per-neuron kernel of gpfq_wide.hip (printed from the kernel),
# "-DGPFQ_S2_SKIP=n" leaves a phase of gpfq_gram_s2_kernel out (wrong sums: timing experiments only)
# A diagnostic build lives beside the shipped library, in csrc/diag_<hash of the flags>/ (round 5): several variants can be built in the
# CPU container, travel to the GPU box together and be timed there without a compiler run (the box's minutes are the scarce resource);
# objects of sources that take no diagnostic flag are shared with the shipped build.
DIAG_SOURCES = ("gpfq_blk.hip", "gpfq_wide.hip", "gpfq_gram_s2.hip", "gpfq_gram_image.hip", "gpfq_misc.hip")
if os.environ.get("GPFQ_DIAG"):
    for _src in DIAG_SOURCES:
        EXTRA_FLAGS[_src] = EXTRA_FLAGS.get(_src, []) + os.environ["GPFQ_DIAG"].split()
    # (gpfq_blk.hip's switches live in gpfq_blk_diag.hpp, which only a -DGPFQ_BLK_DIAG build includes: the shipped unit compiles without them)
    if any(f.startswith("-DGPFQ_BLK_") for f in os.environ["GPFQ_DIAG"].split()) and "-DGPFQ_BLK_DIAG" not in EXTRA_FLAGS["gpfq_blk.hip"]:
        EXTRA_FLAGS["gpfq_blk.hip"] = EXTRA_FLAGS["gpfq_blk.hip"] + ["-DGPFQ_BLK_DIAG"]
    _tag = hashlib.sha256(" ".join(os.environ["GPFQ_DIAG"].split()).encode()).hexdigest()[:10]
    DIAG_DIR = os.path.join(CSRC, "diag_" + _tag)
    LIB = os.path.join(DIAG_DIR, "libgpfq_hip.so")
    STAMP = LIB + ".sha"
else:
    DIAG_DIR = None


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def tree_hash():
    """Hash of everything the library is built from (sources, headers, flags)."""
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return _sha(files, repr((FLAGS, sorted(EXTRA_FLAGS.items()))))


def built_hash():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def _stale():
    return not os.path.exists(LIB) or built_hash() != tree_hash()


def _compile(hipcc, src, verbose):
    objdir = DIAG_DIR if (DIAG_DIR and src in DIAG_SOURCES) else OBJDIR
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    flags = FLAGS + EXTRA_FLAGS.get(src, [])
    want = _sha([os.path.join(CSRC, f) for f in [src] + HEADERS], repr(flags))
    try:
        with open(stamp) as f:
            if f.read().strip() == want and os.path.exists(obj):
                return obj
    except OSError:
        pass
    cmd = [hipcc] + flags + ["-c", "-o", obj, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    with open(stamp, "w") as f:
        f.write(want)
    return obj


@contextlib.contextmanager
def _build_lock():
    """One builder at a time across processes (every rank of a torchrun job may find the library stale at once):
    an advisory lock on csrc/.build.lock; the others wait, then find the library fresh."""
    fd = os.open(os.path.join(CSRC, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def build(force=False, verbose=False):
    """Compile the HIP library if missing or built from other sources/flags; returns its path.  Safe to call from
    several processes at once: the build runs under a file lock, the library and its stamp are put in place by
    os.replace (a process that already mapped the old file keeps it; nobody ever opens a half-written one)."""
    if not force and not _stale():
        return LIB
    with _build_lock():
        if not force and not _stale():                  # another process built it while this one waited
            return LIB
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        # (the stamp is the hash of the tree as it stands NOW: a source edited while the compilers run leaves the library stale, not
        #  "fresh" with an object of the old text -- it happened in round 5)
        stamp_hash = tree_hash()
        os.makedirs(OBJDIR, exist_ok=True)
        if DIAG_DIR:
            os.makedirs(DIAG_DIR, exist_ok=True)
        if force:
            # a forced build starts from no objects: the shipped build's, or -- a diagnostic build -- the diagnostic directory's own
            # (the objects it shares with the shipped build stay: they are checked against their content stamps like any other)
            for d in ([DIAG_DIR] if DIAG_DIR else [OBJDIR]):
                for f in os.listdir(d):
                    if f.endswith(".o") or f.endswith(".o.sha"):
                        os.remove(os.path.join(d, f))
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            objs = list(ex.map(lambda s: _compile(hipcc, s, verbose), SOURCES))
        tmp_lib = LIB + ".tmp.%d" % os.getpid()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        tmp_stamp = STAMP + ".tmp.%d" % os.getpid()
        with open(tmp_stamp, "w") as f:
            f.write(stamp_hash)
        if os.path.exists(STAMP):
            os.remove(STAMP)                            # never a fresh stamp beside a stale library
        os.replace(tmp_lib, LIB)
        os.replace(tmp_stamp, STAMP)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
