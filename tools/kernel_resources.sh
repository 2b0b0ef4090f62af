#!/bin/bash
# Registers, spills, scratch and LDS of every kernel of one HIP source, as the compiler reports them for gfx950
# (-Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.sh cocons_amd/csrc/chol.hip [extra flags]
SRC=${1:-cocons_amd/csrc/chol.hip}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Rpass-analysis=kernel-resource-usage "$@" -c $SRC -o /dev/null 2>&1 | python3 -c '
import sys, re, subprocess
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: .*?\s{2,}([A-Za-z\[\] /]+): (\S+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
keep = ("VGPRs", "AGPRs", "SGPRs", "VGPR Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]")
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)[:60]
    print("%-60s %s" % (name, " ".join("%s=%s" % (a.split(" [")[0].replace(" ", ""), b) for a, b in v.items() if a in keep)))
'
