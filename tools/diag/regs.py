"""register / spill table of the kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), optional name filter"""
import re, sys, subprocess, os
src = sys.argv[1]
inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'include')
out = subprocess.run(f"hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I{inc} -Wno-unused-result -Rpass-analysis=kernel-resource-usage -c {src} -o /dev/null",
                     shell=True, capture_output=True, text=True).stderr
cur = None; d = {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip(); d[cur] = {}
    for k in ("VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
        m = re.search(r"remark:\s+%s: (\d+)" % re.escape(k), l)
        if m and cur: d[cur][k] = m.group(1)
for k, v in d.items():
    if len(sys.argv) < 3 or sys.argv[2] in k:
        print(k[:100].ljust(100), 'vgpr', v.get("VGPRs"), 'agpr', v.get("AGPRs"), "spill", v.get("VGPRs Spill"), "scratch", v.get("ScratchSize [bytes/lane]"))
