"""Register / scratch usage of every kernel in one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line
per kernel:  python tools/kernel_resources.py conv_igemm6.hip [substring]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "tinyedm_amd", "csrc", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:],
                   capture_output=True, text=True)
cur = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
    cur[k] = v
    if k.startswith("LDS Size"):
        if pat in cur["name"]:
            print(f"{cur['name'][:90]:90s} VGPR {cur.get('VGPRs')} AGPR {cur.get('AGPRs')} spill {cur.get('VGPRs Spill')} "
                  f"scratch {cur.get('ScratchSize')} occ {cur.get('Occupancy')}")
if r.returncode:
    print(r.stderr[-3000:])
