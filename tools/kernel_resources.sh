#!/bin/bash
# per-kernel register / spill / LDS summary of the library's device code (compile only, no GPU needed)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/cf_api.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-unused-value -mllvm -hoist-common-insts=false -mllvm -sink-common-insts=false $CF_EXTRA_HIPCC_FLAGS \
    -I"$ROOT/include" -I"$ROOT/clusterfusion_amd/csrc" -o "$OUT" "$ROOT/clusterfusion_amd/csrc/cf_api.hip" 2>/dev/null
python3 - "$OUT" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", body) or [None, "?"])[1]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if "fused" in dn or len(sys.argv) > 2:
        print(f"{dn[:70]:70s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s}")
PY
