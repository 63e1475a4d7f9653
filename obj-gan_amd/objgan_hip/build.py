"""Build libobjgan_hip.so (gfx950) from obj-gan_amd/csrc/*.hip with hipcc.

Replaces the reference's nvcc + torch.utils.ffi build (reference
image_generation/make.sh:1-57, models/roi_align/build.py:27-35).  hipcc cross-compiles for
gfx950 without a GPU, so this runs in the CPU-only build container; the .so is kept in-tree
(git-ignored) so that it travels to the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB_PATH = os.path.join(HERE, "libobjgan_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")

BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# OBJGAN_DEV=1: development build -- the OG_* switches of csrc/common.h (OG_KNOB) read the environment
if os.environ.get("OBJGAN_DEV") == "1":
    BASE_FLAGS.append("-DOG_DEV")
# ROIAlign index math must be bit-exact with the reference C loop: no FMA contraction.
# The Pillow-exact image resize evaluates its filter coefficients in double in Pillow's operation order.
# -munsafe-fp-atomics (hardware fp32 atomic add instead of a CAS loop) only where a float atomic is left: the
# reference-signature objgan_roi_align_backward (unordered scatter, as the CUDA original).  The convolution files carry
# no fp32 atomic any more (round 5: the first-generation split-K path is gone); the training step runs without any.
PER_FILE_FLAGS = {"roi_align.hip": ["-ffp-contract=off", "-munsafe-fp-atomics"], "resize_pil.hip": ["-ffp-contract=off"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libobjgan_hip.so")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


FLAGS_STAMP = LIB_PATH + ".flags"


def _flags_line():
    return " ".join(BASE_FLAGS) + " | " + " ".join("%s:%s" % (k, " ".join(v)) for k, v in sorted(PER_FILE_FLAGS.items()))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    try:                        # a library built with other flags (e.g. a development build) is stale
        if open(FLAGS_STAMP).read() != _flags_line():
            return True
    except OSError:
        return True
    lib_m = os.path.getmtime(LIB_PATH)
    for f in os.listdir(CSRC):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and os.path.getmtime(p) > lib_m:
            return True
    return False


def build(force=False, verbose=True):
    """Compile and link; concurrent callers (one process per GPU starting at once on a tree without the library)
    serialise on a lock file -- the first builds, the others find the library fresh."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    try:
        force = force or open(FLAGS_STAMP).read() != _flags_line()
    except OSError:
        force = True
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        srcp = os.path.join(CSRC, src)
        deps = [srcp] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps)):
            continue
        cmd = [hipcc] + BASE_FLAGS + PER_FILE_FLAGS.get(src, []) + ["-c", srcp, "-o", obj]
        if verbose:
            print("[objgan_hip.build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, cwd="/tmp" if os.path.isdir("/tmp") else None,
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % src)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print("[objgan_hip.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd="/tmp" if os.path.isdir("/tmp") else None)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(FLAGS_STAMP, "w") as f:
        f.write(_flags_line())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
