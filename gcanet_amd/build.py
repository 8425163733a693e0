"""Build libgcanet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m gcanet_amd.build [--force]

One object per csrc/*.hip (compiled in parallel), linked in-tree into
gcanet_amd/lib/libgcanet_hip.so so that the library travels with the repo snapshot.
-ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(), which
is what makes integer/index results bit-exact against oracle/gcanet_oracle.c.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
SO = os.path.join(LIBDIR, "libgcanet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


STAMP = os.path.join(OBJDIR, "flags.stamp")


def _stamp_text():
    return " ".join([HIPCC] + FLAGS)


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def stale():
    """True when libgcanet_hip.so is missing or older than any source/header (or was built with other flags)."""
    deps = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not _newer(SO, deps):
        return True
    return not os.path.exists(STAMP) or open(STAMP).read() != _stamp_text()


def _compile(src, force):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not force and _newer(obj, deps):
        return obj, False
    subprocess.check_call([HIPCC] + FLAGS + _file_flags(src) + ["-c", src, "-o", obj])
    return obj, True


def _file_flags(src):
    """Extra compiler flags a source asks for in a `// hipcc-flags: ...` line of its header comment."""
    for line in open(src).read(4096).splitlines():
        if line.startswith("// hipcc-flags:"):
            return line.split(":", 1)[1].split()
    return []


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    if not os.path.exists(STAMP) or open(STAMP).read() != _stamp_text():
        force = True                      # compiler or flags changed: every object is stale
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(SO):
        subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs)
        if verbose:
            print("linked", SO)
    elif verbose:
        print("up to date:", SO)
    with open(STAMP, "w") as f:
        f.write(_stamp_text())
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
