#!/usr/bin/env python
"""profiles/<tag>_pmc_other.md from gpurun_out/<tag>/pmc_other/*.md (tools/pmc_other.sh): HBM read traffic per launch of the
persistent kernels other than the headline's, against their algorithmic bytes.  FETCH_SIZE is in KiB and reports half the bytes
of a wide coalesced read stream on gfx950 (guide MI355X_MICROARCH.md, HBM section; calibrated on the headline kernel: 1.006)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", tag, "pmc_other")
H, D = 4096, 128


def layer_bytes(hq, hkv, S, rows=1):
    # weights once, K and V of every cached token of every row, x / residual / out per row (fp16)
    return 2 * H * (hq + 2 * hkv) * D + 2 * H * hq * D + rows * S * 2 * hkv * D * 2 + rows * 3 * H * 2


WORK = [("plain_32_32_s1024", "Llama-2-7B, S=1024 (config 2 through the re-layout)", layer_bytes(32, 32, 1024)),
        ("gqa_32_8_s8192", "Llama-3-8B GQA, S=8192 (config 4)", layer_bytes(32, 8, 8192)),
        ("tp8_4_4_s4096", "TP-8 shard of Llama-2-7B, S=4096 (config 5, one rank)", layer_bytes(4, 4, 4096)),
        ("tp8_gqa_4_1_s8192", "TP-8 shard of Llama-3-8B (4q/1kv), S=8192", layer_bytes(4, 1, 8192)),
        ("batch2_s1024", "2 sequences x 1024", layer_bytes(32, 32, 1024, 2)),
        ("batch8_s1024", "8 sequences x 1024", layer_bytes(32, 32, 1024, 8)),
        ("batch16_s1024", "16 sequences x 1024", layer_bytes(32, 32, 1024, 16)),
        ("batch32_s1024", "32 sequences x 1024", layer_bytes(32, 32, 1024, 32))]

out = [f"<!-- bash tools/pmc_other.sh {tag}; python tools/pmc_other_md.py {tag}  (one rocprofv3 --pmc FETCH_SIZE --kernel-trace pass per workload) -->",
       "# HBM read traffic of the other persistent kernels (PMC FETCH_SIZE x 1024 x 2, per launch) against their algorithmic bytes", "",
       "| workload | kernel | dispatches | FETCH_SIZE KiB | traffic MB | algorithmic MB | traffic / algorithmic |", "|---|---|---|---|---|---|---|"]
for name, what, alg in WORK:
    p = os.path.join(src, name + ".md")
    if not os.path.exists(p):
        out.append(f"| {what} | (no data) | | | | {alg / 1e6:.2f} | |")
        continue
    rows = re.findall(r"\| `(?:void )?(cf::[^`]*)` \| FETCH_SIZE \| (\d+) \| ([\d.]+) \|", open(p).read())
    rows = [r for r in rows if "k_fused_decode" in r[0]]
    if not rows:
        out.append(f"| {what} | (no counter rows) | | | | {alg / 1e6:.2f} | |")
    for k, n, kib in rows:
        tr = float(kib) * 1024 * 2
        out.append(f"| {what} | `{k.split('(')[0]}` | {n} | {float(kib):.1f} | {tr / 1e6:.2f} | {alg / 1e6:.2f} | {tr / alg:.3f} |")
out += ["", "Algorithmic bytes: the weights once, K and V of every cached token of every row, x / residual / out per row.  The batched",
        "kernels also gather B x 8 KB of normalised rows (X0) and B x 8 KB of attention output (X3) per workgroup: those are",
        "produced on the chip a few microseconds earlier and are served by L2 / Infinity Cache where the counter shows no excess."]
dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_other.md")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out))
