#!/bin/bash
# Registers, scratch, LDS and occupancy of every kernel of one .hip file of the library.  usage: tools/kres.sh grid_fold [filter]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$ROOT/include -I$ROOT/adhoc-queries-pointclouds_amd/csrc \
  -Rpass-analysis=kernel-resource-usage -c $ROOT/adhoc-queries-pointclouds_amd/csrc/$1.hip -o /dev/null 2>&1 | python3 -c '
import sys, re, subprocess
cur = {}
def flush():
    if cur:
        name = subprocess.run(["c++filt", cur.get("Name", "?")], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-72s vgpr %3s agpr %2s sgpr %3s scratch %5s lds %6s occ %s" % (name[:72], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("SGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("LDS Size [bytes/block]"), cur.get("Occupancy [waves/SIMD]")))
for line in sys.stdin:
    m = re.search(r"remark: [^:]+:\d+:\d+: +([A-Za-z \[\]/]+): (\S+)", line) or re.search(r":\d+:\d+: remark: +([A-Za-z \[\]/]+): (\S+)", line) or re.search(r"\d+:\d+: +([A-Za-z \[\]/]+?): (\S+) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name" or k == "Name":
        flush(); cur = {"Name": v}
    else:
        cur[k] = v
flush()
' | grep -E "${2:-.}"
