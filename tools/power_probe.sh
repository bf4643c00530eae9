#!/bin/bash
# Board power and clocks (rocm-smi, read-only) while a sustained stream of k_conv3x3_v6 launches runs (one layer shape, ~12 s),
# with one tile per workgroup and with the persistent form: is the MFMA-dense stream at the power cap, and what clock does it hold?
#   bash tools/power_probe.sh  -> gpurun_out/power_probe.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/power_probe.txt
: > $O
probe() {
  for i in 1 2 3 4 5 6; do
    sleep 1
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|junction" | tr -s ' \t' ' ' | tr '\n' ';' >> $O
    echo >> $O
  done
}
echo "== idle" >> $O
/opt/rocm/bin/rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power \(W\)|sclk|Max" | tr -s ' \t' ' ' >> $O
cat > /tmp/power_load.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["EDM_ROOT"])
from tinyedm_amd import ops
B, HW, Cin, Cout = 512, 32, 256, 256
x = torch.randn(B, HW, HW, Cin, device="cuda").to(torch.bfloat16)
wp = (torch.randn(9, Cout, Cin, device="cuda") / (Cin * 9) ** 0.5).to(torch.bfloat16)
secs = float(sys.argv[1])
for _ in range(10):
    ops.conv_igemm(x, wp, 9)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(500):
        ops.conv_igemm(x, wp, 9)
    e.record(); torch.cuda.synchronize(); n += 1
    print(f"t={time.perf_counter() - t0:5.1f} s: {s.elapsed_time(e) / 500 * 1e3:.1f} us per launch, {2.0 * B * HW * HW * Cin * Cout * 9 / (s.elapsed_time(e) / 500) / 1e9:.0f} TF/s", flush=True)
PY
for p in 0 1; do
  echo "== sustained k_conv3x3_v6, B=512 32x32 256->256, EDM_V6_PERSIST=$p" >> $O
  EDM_ROOT=$R EDM_V6_PERSIST=$p timeout -k 10 120 python3 /tmp/power_load.py 12 > $R/gpurun_out/power_probe_mb$p.txt 2>&1 &
  pid=$!
  sleep 7
  probe
  wait $pid || exit 1
  grep -v amdgpu.ids $R/gpurun_out/power_probe_mb$p.txt >> $O
done
cat $O
