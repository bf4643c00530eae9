"""Build a DIAGNOSTIC copy of the library with extra -D flags (in-kernel timestamps, ablations), out of tree of the product:
    python tools/build_diag_lib.py ab/lib_s_timeline.so -DEDM_S_TIMELINE
then run a tool with EDM_LIB_PATH=<that file>.  Objects go to ab/_obj_<name>/ (ab/ is git-ignored; it travels with gpurun)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1])
flags = sys.argv[2:]
csrc = os.path.join(root, "tinyedm_amd", "csrc")
obj = os.path.join(os.path.dirname(out), "_obj_" + os.path.basename(out).replace(".so", ""))
os.makedirs(obj, exist_ok=True)
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-ffp-contract=fast"]
prod = os.path.join(root, "tinyedm_amd", "_build")


def cc(src):
    # only sources that mention one of the -D symbols are rebuilt; the rest reuse the product objects
    text = open(os.path.join(csrc, src)).read()
    if not any(f[2:].split("=")[0] in text for f in flags if f.startswith("-D")):
        return os.path.join(prod, src[:-4] + ".o")
    o = os.path.join(obj, src[:-4] + ".o")
    subprocess.run(base + flags + ["-c", os.path.join(csrc, src), "-o", o], check=True)
    return o


srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
with ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(cc, srcs))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print("built", out)
