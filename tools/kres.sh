#!/bin/bash
# Per-kernel register / scratch / LDS usage of one translation unit (compile-only, no GPU).
# Usage: tools/kres.sh quflow_amd/csrc/zgemm.hip [extra hipcc flags]
src=$1; shift
cd "$(dirname "$src")"
# the flags of quflow_amd/csrc/Makefile: ozaki.hip is built WITHOUT the vgpr-form flag and with a raised unroll threshold
flags="-mllvm -amdgpu-mfma-vgpr-form=1"
[ "$(basename "$src")" = ozaki.hip ] && flags="-mllvm -pragma-unroll-threshold=400000"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$(basename "$src")" -o /tmp/kres_$$.o 2>&1 |
python3 -c '
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    k, _, v = m.group(1).partition(": ")
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k.strip()] = v.strip()
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    print("%-62s VGPR %3s AGPR %3s scratch %5s occ %s LDS %s" % (name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
rm -f /tmp/kres_$$.o
