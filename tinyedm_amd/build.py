"""Build the C-ABI HIP library (libtinyedm_hip.so) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting .so travels to the GPU box with the snapshot.  Re-builds only stale objects.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libtinyedm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-ffp-contract=fast"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = True, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(s, o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        if verbose:
            print(f"[build] compiled {s}", flush=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    stale = [o for o in os.listdir(OBJ) if o.endswith(".o") and os.path.join(OBJ, o) not in objs]
    for o in stale:                     # object of a source that no longer exists: drop it and relink
        os.remove(os.path.join(OBJ, o))
    manifest = os.path.join(OBJ, "link_manifest.txt")     # the object list the library was last linked from
    linked = open(manifest).read().split() if os.path.exists(manifest) else None
    names = [os.path.basename(o) for o in objs]
    if jobs or stale or not os.path.exists(LIB) or linked != names:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        with open(manifest, "w") as f:
            f.write("\n".join(names))
        if verbose:
            print(f"[build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
