"""Build libdiga_hip.so (gfx950) in-tree with hipcc.

    python -m diga_amd.build [--force] [--jobs N]

Each csrc/*.hip is compiled to an object (parallel, cached on mtime) and linked into
diga_amd/libdiga_hip.so, which travels to the GPU box with the repo snapshot.
"""
import argparse
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libdiga_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-DNDEBUG", "-fvisibility=hidden"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def _compile(src, obj, extra):
    cmd = [hipcc(), *FLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    return r.stderr


def build(force=False, jobs=None, verbose=True, extra=(), variant=None):
    """variant: A/B builds of the library -- objects under build_<variant>/, result libdiga_hip_<variant>.so (load it with
    DIGA_LIB=...); `extra` = additional hipcc flags (-D switches of the diagnostic / tuning macros in csrc/)."""
    global OBJ, LIB
    if variant:
        OBJ = os.path.join(HERE, "build_" + variant)
        LIB = os.path.join(HERE, f"libdiga_hip_{variant}.so")
    os.makedirs(OBJ, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    todo = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o, headers):
            todo.append((s, o))
    if todo:
        jobs = jobs or min(len(todo), max(1, (os.cpu_count() or 2) - 1))
        with cf.ThreadPoolExecutor(jobs) as ex:
            futs = {ex.submit(_compile, s, o, list(extra)): s for s, o in todo}
            for f in cf.as_completed(futs):
                warn = f.result()
                if verbose:
                    print(f"[diga build] compiled {os.path.basename(futs[f])}")
                    if warn.strip():
                        print(warn)
    if todo or force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        # the dynamic symbol table is the C ABI and nothing else: -fvisibility=hidden leaves hipcc's kernel-handle objects
        # exported, the version script makes them local too
        vs = os.path.join(OBJ, "exports.map")
        with open(vs, "w") as fh:
            fh.write("{ global: diga_*; local: *; };\n")
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={vs}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[diga build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--variant", default=None, help="A/B build: objects in build_<variant>/, library libdiga_hip_<variant>.so")
    ap.add_argument("--define", "-D", action="append", default=[], help="extra -D macro(s) for the variant")
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs, extra=["-D" + d for d in a.define], variant=a.variant)
    sys.exit(0)
