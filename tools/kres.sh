#!/bin/bash
# kernel resource usage of one HIP source (VGPRs / scratch / occupancy per kernel), same flags as freepose_amd/build.py
#   tools/kres.sh freepose_amd/csrc/gemm_bf16.hip [grep-pattern]
src=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950:sramecc+ -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DNDEBUG \
  --cuda-device-only -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
cur = {}
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        continue
    for key, tag in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize [bytes/lane]:", "scratch"), ("Occupancy [waves/SIMD]:", "occ"), ("VGPRs Spill:", "vspill"), ("SGPRs:", "sgpr")):
        if key in ln and "remark" in ln:
            cur[tag] = ln.split(key)[1].split()[0]
    if "LDS Size" in ln and cur:
        print(cur.get("name"), "vgpr", cur.get("vgpr"), "agpr", cur.get("agpr"), "sgpr", cur.get("sgpr"), "scratch", cur.get("scratch"), "vspill", cur.get("vspill"), "occ", cur.get("occ"))
        cur = {}
' | grep -E "$pat"
