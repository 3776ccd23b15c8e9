#!/usr/bin/env python
"""Turn what tools/collect_profiles.sh left under gpurun_out/<tag>/ into the tables committed under profiles/<tag>_*.

    python tools/make_profile_md.py r02b
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
O, P = os.path.join(ROOT, "gpurun_out", TAG), os.path.join(ROOT, "profiles")


def jlines(path):
    return [json.loads(l) for l in open(path) if l.strip().startswith("{")]


b = jlines(f"{O}/bench_default.json")[-1]
open(f"{P}/{TAG}_bench_default.json", "w").write(json.dumps(b, indent=1) + "\n")
us, fr = b["us_per_layer"], b["roofline"]["frac"]

kt = open(f"{O}/kt_summary.md").read()
m = re.search(r"k_fused_decode_mha<false>\(cf::FusedArgs\)` \| (\d+) \| ([\d.]+)", kt)
ktc = open(f"{O}/ktc_summary.md").read() if os.path.exists(f"{O}/ktc_summary.md") else ""
open(f"{P}/{TAG}_kernel_trace.md", "w").write(
    "<!-- rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-configs   (MI355X; hipGraph replay: 50 timed + 5 warm-up + 40 pre-warm "
    "steps, the capture and 21 eager steps (first call, per-kernel event pass) of 32 layers of the HEADLINE workload only: the kernel picks its length arm on the device, so one kernel "
    "name serves every sequence length) -->\n" + kt +
    ("\n## The other configurations of the bench line (`python bench.py --no-cpu-baseline --steps 2 --warmup 1` under the same tracer; "
     "`k_fused_decode_mha<false>` here mixes the headline's S=4096 calls with config 2's S=1024 calls)\n\n" + ktc[ktc.index("| kernel |"):] if ktc else "") +
    f"\nBench line of the un-profiled default run on the same box (`profiles/{TAG}_bench_default.json`): {us:.2f} us per layer, `roofline.frac` {fr:.4f}; "
    f"the profiled average of the headline kernel ({m.group(2)} us over {m.group(1)} launches) agrees with it.\n")

if os.path.exists(f"{O}/ktb64_summary.md"):
    parts = ["<!-- CF_NL=8 rocprofv3 --kernel-trace --stats -- python tools/batch_bench.py 1024 {64|128}  (the five-launch path of "
             "llama_decoder_layer_batch_decode_sglang beyond 32 rows: k_norm_rows, k_proj_rows_big (QKV), k_attn_split, k_proj_rows_big (O)) -->\n"]
    for bs in (64, 128):
        t = open(f"{O}/ktb{bs}_summary.md").read()
        parts.append(f"## {bs} sequences x 1024 cached tokens\n\n" + t[t.index("| kernel |"):] + "\n")
    open(f"{P}/{TAG}_batch_kernel_trace.md", "w").write("".join(parts))

pf, pw = open(f"{O}/pf_summary.md").read(), open(f"{O}/pw_summary.md").read()
fetch = float(re.search(r"FETCH_SIZE \| \d+ \| ([\d.]+)", pf).group(1))
wr = float(re.search(r"WRITE_SIZE \| \d+ \| ([\d.]+)", pw).group(1))
rb, wb = fetch * 1024 * 2, wr * 1024
open(f"{P}/{TAG}_pmc_hbm_traffic.md", "w").write(
    "<!-- rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) -- python bench.py --steps 3 --warmup 1 "
    "--layers 8 --no-graph --no-cpu-baseline --no-configs -->\n" + pf + "\n" + pw[pw.index("| kernel | counter"):] + f"""
## Reading (guide: MI355X_MICROARCH.md, HBM section)

* FETCH_SIZE is in KiB and reports 1/2 of the bytes of a wide coalesced streaming read on gfx950: read traffic per launch =
  {fetch:.1f} KiB x 1024 x 2 = **{rb / 1e6:.2f} MB**; algorithmic bytes per launch 201.38 MB -> traffic / algorithmic = **{rb / 201.38496e6:.3f}**
  (every weight and cached K/V byte is read once).  WRITE_SIZE {wr:.0f} KiB per launch (uncalibrated): granule exchanges, outputs, state words.
""")
json.dump({"_source": f"profiles/{TAG}_pmc_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x1024 x2 per the gfx950 correction)",
           "k_fused_decode_mha_bytes_per_launch": int(rb + wb), "k_fused_decode_mha_read_bytes_per_launch": int(rb),
           "k_fused_decode_mha_write_bytes_per_launch_uncalibrated": int(wb)}, open(f"{P}/hbm_traffic.json", "w"), indent=2)

out = (f"<!-- python bench.py --no-cpu-baseline --no-configs --seq S --steps 20   (one box: its headline run measured {us:.2f} us) -->\n"
       "# Headline kernel over sequence length (paged KV, page size 16, 32 distinct layers, hipGraph replay)\n\n"
       "| S | us / layer | algorithmic MB | fraction of 8 TB/s |\n|---|---|---|---|\n")
for r in jlines(f"{O}/seq.jsonl"):
    S = int(re.search(r"seq=(\d+)", r["config"]["workload"]).group(1))
    out += "| %d | %.2f | %.1f | %.3f |\n" % (S, r["us_per_layer"], r["roofline"]["bytes_per_launch"] / 1e6, r["roofline"]["frac"])
open(f"{P}/{TAG}_seq_sweep.md", "w").write(out)

t = ("<!-- CF_TL_GRAPH=1 python tools/fused_timeline.py {4096 0 | 8192 0 gqa | 4096 0 tp8 | 1024 0 b2 | 1024 0 b4 | 1024 0 b8 | 1024 0 b16}  (in-kernel 100 MHz stamps of the "
     "last launch of a replayed graph) -->\n# Phase timelines of the persistent kernels (us since the first workgroup of the launch started)\n")
for name, title in (("headline", "headline: k_fused_decode_mha<IO=false>, two-tile arm, S = 4096 paged"), ("gqa", "config 4: k_fused_decode_g<8,4>, S = 8192"),
                    ("tp8", "config 5 shard: k_fused_decode_s<4>, S = 4096 (role-split: workgroups 0..63 attention -- X1 / P2 done exist only there --, 64..255 projections; X3 resolved = records gathered and merged locally)"),
                    ("b2", "k_fused_decode_mhab<2>: 2 sequences x S = 1024"), ("b4", "k_fused_decode_mhab<4>: 4 sequences x S = 1024"),
                    ("b8", "k_fused_decode_mhaq: 8 sequences x S = 1024 (P2 done = the workgroup's last row published; 'rec published' = its first row)"),
                    ("b16", "k_fused_decode_mhaq: 16 sequences x S = 1024")):
    lines = [l for l in open(f"{O}/timeline_{name}.txt").read().splitlines() if "amdgpu.ids" not in l and not re.search(r"-\d{9,}", l)]
    t += f"\n## {title}\n```\n" + "\n".join(lines[:40]) + "\n```\n"
open(f"{P}/{TAG}_timelines.md", "w").write(t)

o = ("<!-- python tools/batch_bench.py 1024 1,2,3,4,5,8,12,16,32 ; CF_FLAGS=32 ... 1024 2,4,5,8,12,16 ; ... 4096 2,4,8,16 ; CF_FLAGS=32 ... 4096 2,4,8,16   (one box; 32 distinct "
     "layers per graph replay) -->\n# `llama_decoder_layer_batch_decode_sglang`, small batches (Llama-2-7B, paged KV page size 1, every row S cached tokens)\n\n"
     "Algorithmic MB = weights once + every row's K/V.  `k_fused_decode_mhab<NB>` = the persistent kernel with the rows sharing one weight stream\n"
     "(cf_fused_kernel_b.h, 2..4 rows); `k_fused_decode_mhaq` = one persistent launch with both projections on the matrix cores (cf_fused_kernel_q.h,\n"
     "5..32 rows; two 16-row batch tiles from 17); `stage pipeline` = the five-launch MFMA path (debug flag 32 forces it up to 32 rows).\n\n"
     "| batch | S | kernel | us / call | us / row | algorithmic MB | fraction of 8 TB/s |\n|---|---|---|---|---|---|---|\n")
for r in jlines(f"{O}/batch.jsonl"):
    o += "| %d | %d | `%s` | %.2f | %.2f | %.1f | %.3f |\n" % (r["batch"], r["S"], r["kernel"], r["us_per_call"], r["us_per_row"], r["MB"], r["frac_of_8TBs"])
o += "\nRound 2 (`profiles/r02b_batch.md`): 66.6 / 86.9 us for 8 / 16 rows at S = 1024 through the five launches.  Verdict targets: <= 50 / <= 70 us.\n"
open(f"{P}/{TAG}_batch.md", "w").write(o)
if os.path.exists(f"{O}/decode_model.jsonl"):
    open(f"{P}/{TAG}_decode_model.md", "w").write(
        "<!-- python tools/decode_bench.py 4000 64 ; ... 1024 64 ; ... 8000 32 llama3   (one MI355X) -->\n# Whole-model greedy decode, ONE hipGraph captured once and replayed while the sequence grows\n\n"
        "Random weights of the real shapes; attention block of every layer = ONE call of the fused op (paged entry, it writes the new K/V itself, reads the length on\n"
        "the device), the norms between blocks = `clusterfusion.rmsnorm` (fused add), SwiGLU FFN / LM head / argmax = plain torch matmuls (outside the reference's\n"
        "fused op as well).\n\n```\n" + "\n".join(json.dumps(r) for r in jlines(f"{O}/decode_model.jsonl")) + "\n```\n")
if os.path.exists(f"{O}/mla.jsonl"):
    rows = jlines(f"{O}/mla.jsonl")
    open(f"{P}/{TAG}_mla.md", "w").write(
        "<!-- python tools/mla_bench.py ; python tools/mla_timeline.py -->\n# deepseek_decoder_layer (MLA), S = 4096, 27 distinct layers\n\n```\n"
        + "\n".join(json.dumps(r) for r in rows) + "\n```\n\n```\n" + open(f"{O}/mla_timeline.txt").read() + "```\n")
print("wrote", [f for f in sorted(os.listdir(P)) if f.startswith(TAG)])
