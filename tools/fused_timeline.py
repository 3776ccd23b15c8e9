#!/usr/bin/env python
"""Phase timeline of the persistent fused kernel (debug stamps, 100 MHz wall clock).
Runs the bench workload (S=4096 paged, N distinct layers) and prints, per phase boundary, the
min / median / max over the 256 workgroups relative to the earliest workgroup start of a launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import clusterfusion_amd as cfa
from clusterfusion_amd import _lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
GQA = len(sys.argv) > 3 and sys.argv[3] == "gqa"
IO = len(sys.argv) > 3 and sys.argv[3] == "io"      # plain API: [in,out] weights, GPT-J RoPE, contiguous KV
TP = int(sys.argv[3][2:]) if len(sys.argv) > 3 and sys.argv[3].startswith("tp") else 0   # tp2 / tp4 / tp8: one rank's shard
GTP = int(sys.argv[3][3:]) if len(sys.argv) > 3 and sys.argv[3].startswith("gtp") else 0   # gtp2 / gtp4 / gtp8: one rank's shard of Llama-3-8B
BATCH = int(sys.argv[3][1:]) if len(sys.argv) > 3 and sys.argv[3][0] == "b" and sys.argv[3][1:].isdigit() else 0   # b2 / b3 / b4
if BATCH:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config_bench
    g = torch.Generator(device=dev).manual_seed(1)
    layers = [config_bench.make_batch(g, BATCH, S) for _ in range(int(os.environ.get("CF_TL_LAYERS", "8")))]
elif IO:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config_bench
    g = torch.Generator(device=dev).manual_seed(1)
    layers = [config_bench.make(g, hidden=4096, hq=32, hkv=32, S=S, layout="in_out", style="gptj", residual=False)
              for _ in range(8)]
elif GTP:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config_bench
    g = torch.Generator(device=dev).manual_seed(1)
    layers = [config_bench.make(g, hidden=4096, hq=32 // GTP, hkv=8 // GTP, S=S, layout="out_in", style="neox", residual=True)
              for _ in range(16)]
elif TP:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config_bench
    g = torch.Generator(device=dev).manual_seed(1)
    layers = [config_bench.make(g, hidden=4096, hq=32 // TP, hkv=32 // TP, S=S, layout="out_in", style="neox", residual=True)
              for _ in range(int(os.environ.get("CF_TL_LAYERS", "16")))]
elif GQA:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config_bench
    g = torch.Generator(device=dev).manual_seed(1)
    layers = [config_bench.make(g, hidden=4096, hq=32, hkv=8, S=S, layout="out_in", style="neox", residual=True)
              for _ in range(int(os.environ.get("CF_TL_LAYERS", "8")))]
else:
    layers = bench.build_layers(cfa, dev, 1, 0, int(os.environ.get('CF_TL_LAYERS', '8')), S, 16)[0]
if not BATCH:
    cfa.set_path("fused")
trace = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.cf_debug_set_flags(FLAGS)
torch.cuda.synchronize()
for _ in range(3):
    for p in layers:
        p.run()
torch.cuda.synchronize()
lib.cf_debug_set_trace(trace.data_ptr())
names = ["start", "P1 done", "X1 resolved", "P2 done", "rec published", "X3 resolved", "end"]
fine = {7: "q ready", 8: "tile A consumed", 9: "Wo requested", 10: "tile B consumed", 11: "wave merge done", 12: "pre-barrier(w0)"}
acc = []
fine_acc = []
raw_acc = []
BACK2BACK = os.environ.get("CF_TL_B2B", "0") == "1"   # stamp the LAST of 8 back-to-back launches
GRAPH = os.environ.get("CF_TL_GRAPH", "0") == "1"     # stamp the last launch of an 8-launch hipGraph (what bench.py replays)
if GRAPH:
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for li, p in enumerate(layers):
            p.run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for li, p in enumerate(layers):
                lib.cf_debug_set_trace(trace.data_ptr() if li == len(layers) - 1 else None)
                p.run()
        lib.cf_debug_set_trace(None)
        if os.environ.get("CF_TL_ACCT", "0") == "1":      # the SAME graph's period (HIP events over 60 replays / launches): period - span = what one launch costs outside its stamps
            for _ in range(10):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(60):
                g.replay()
            e1.record(st)
            torch.cuda.synchronize()
            PERIOD = e0.elapsed_time(e1) * 1e3 / (60 * len(layers))
        for rep in range(40):
            g.replay()
            torch.cuda.synchronize()
            raw = trace.cpu().numpy().reshape(256, 16).astype(np.float64)
            raw[raw == 0] = np.nan      # (a stamp a workgroup's role never writes: k_fused_decode_s)
            raw_acc.append(raw)
            t = raw[:, :7].copy()
            fine_acc.append((raw[:, 7:13] - raw[:, 2:3]) / 100.0)
            acc.append((t - np.nanmin(t[:, 0])) / 100.0)
for rep in range(0 if GRAPH else (5 if not BACK2BACK else 40)):
    for li, p in enumerate(layers):
        if BACK2BACK:
            lib.cf_debug_set_trace(trace.data_ptr() if li == len(layers) - 1 else None)
        p.run()
        if BACK2BACK and li != len(layers) - 1:
            continue
        torch.cuda.synchronize()
        raw = trace.cpu().numpy().reshape(256, 16).astype(np.float64)
        raw[raw == 0] = np.nan
        raw_acc.append(raw)
        t = raw[:, :7].copy()
        fine_acc.append((raw[:, 7:13] - raw[:, 2:3]) / 100.0)
        t = (t - np.nanmin(t[:, 0])) / 100.0      # us
        acc.append(t)
lib.cf_debug_set_trace(None)
t = np.stack(acc)      # [n, 256, 7]
print(f"S={S}: per-boundary time since first workgroup start, us (over {t.shape[0]} launches x 256 WGs)")
print(f"{'boundary':16s} {'min':>7s} {'p10':>7s} {'median':>7s} {'p90':>7s} {'max':>7s}")
for i, n in enumerate(names):
    v = t[:, :, i].reshape(-1)
    print(f"{n:16s} {np.nanmin(v):7.2f} {np.nanpercentile(v, 10):7.2f} {np.nanmedian(v):7.2f} {np.nanpercentile(v, 90):7.2f} {np.nanmax(v):7.2f}")
f = np.stack(fine_acc)
print("fine stamps of wavefront 0, us after X1 resolved (median / p90):")
for k, (slot, n) in enumerate(sorted(fine.items())):
    v = f[:, :, k].reshape(-1)
    print(f"  {n:18s} {np.nanmedian(v):6.2f} {np.nanpercentile(v, 90):6.2f}")
print("kernel span (max end) median over launches: %.2f us" % np.nanmedian(np.nanmax(t[:, :, 6], axis=1)))
print("per launch (median over launches): last P1 done %.2f, last X1 %.2f, last P2 done %.2f, last record %.2f, first X3 %.2f, last X3 %.2f"
      % tuple(np.nanmedian(v) for v in (np.nanmax(t[:, :, 1], axis=1), np.nanmax(t[:, :, 2], axis=1), np.nanmax(t[:, :, 3], axis=1), np.nanmax(t[:, :, 4], axis=1),
                                     np.nanmin(t[:, :, 5], axis=1), np.nanmax(t[:, :, 5], axis=1))))
print("mean over workgroups (median over launches): P1 done %.2f, X1 %.2f, P2 done %.2f"
      % tuple(np.nanmedian(v) for v in (np.nanmean(t[:, :, 1], axis=1), np.nanmean(t[:, :, 2], axis=1), np.nanmean(t[:, :, 3], axis=1))))

if os.environ.get("CF_TL_ACCT", "0") == "1" and GRAPH:
    span = np.nanmedian(np.nanmax(t[:, :, 6], axis=1))
    print(f"\naccounting (same graph, same process): period {PERIOD:.2f} us per launch, in-kernel span {span:.2f} us, outside the span {PERIOD - span:.2f} us")
    bb = np.arange(256)
    if GQA:
        jg = (bb >> 3) % 32
        lead = jg < 4
    elif TP or GTP or BATCH:
        lead = None
    else:
        lead = ((bb >> 3) & 7) == 0
    if lead is not None:
        for nm, sel in (("leaders", lead), ("the others", ~lead)):
            print(f"  {nm} ({int(sel.sum())} workgroups): " + "; ".join(
                f"{names[i]} med {np.nanmedian(t[:, sel, i]):.2f} p90 {np.nanpercentile(t[:, sel, i], 90):.2f} last {np.nanmedian(np.nanmax(t[:, sel, i], axis=1)):.2f}" for i in (3, 4, 5)))
        late = np.nanargmax(t[:, :, 4], axis=1)
        print("  the LAST 'rec published' of a launch is a leader in %d of %d launches; its XCD (b %% 8) histogram: %s" %
              (int(lead[late].sum()), len(late), np.bincount(late & 7, minlength=8).tolist()))
        nl = np.where(~lead)[0]
        late_nl = nl[np.nanargmax(t[:, nl, 4], axis=1)]
        print("  the last NON-leader record: XCD histogram %s; (b >> 3) histogram (top 6) %s" %
              (np.bincount(late_nl & 7, minlength=8).tolist(), sorted(((int(c), int(k)) for k, c in enumerate(np.bincount(late_nl >> 3, minlength=32))), reverse=True)[:6]))
        p2 = t[:, :, 3]
        print("  'P2 done' by XCD (median): " + " ".join(f"{np.nanmedian(p2[:, x::8]):.2f}" for x in range(8)) +
              " | last per launch by XCD: " + " ".join(f"{np.nanmedian(np.nanmax(p2[:, x::8], axis=1)):.2f}" for x in range(8)))
        x3 = t[:, :, 5]
        print("  'X3 resolved' by XCD (median): " + " ".join(f"{np.nanmedian(x3[:, x::8]):.2f}" for x in range(8)))

if os.environ.get("CF_TL_ROLES", "0") == "1":      # k_fused_decode_r: even b >> 3 = attention, odd = projection workgroups
    for nm, sel in (("attention (even j)", ((np.arange(256) >> 3) & 1) == 0), ("projection (odd j)", ((np.arange(256) >> 3) & 1) == 1)):
        print(f"\n{nm}:")
        for i, n in enumerate(names):
            v = t[:, sel, i].reshape(-1)
            if np.isnan(v).all():
                continue
            print(f"  {n:16s} {np.nanmin(v):7.2f} {np.nanpercentile(v, 10):7.2f} {np.nanmedian(v):7.2f} {np.nanpercentile(v, 90):7.2f} {np.nanmax(v):7.2f}")
        for k, (slot, n) in enumerate(sorted(fine.items())):
            v = (np.stack(raw_acc)[:, sel, slot] - np.stack(raw_acc)[:, sel, 0]).reshape(-1) / 100.0
            if np.isnan(v).all():
                continue
            print(f"  since own start: {n:18s} median {np.nanmedian(v):6.2f} p90 {np.nanpercentile(v, 90):6.2f}")

# ---- where does the spread come from: XCD (b % 8), head-group position j, fixed blocks? -------------
p1 = t[:, :, 1] - t[:, :, 0]          # P1 duration per WG
print("\nP1 duration by XCD (b%8): " + " ".join(f"{np.nanmedian(p1[:, x::8]):.2f}" for x in range(8)))
jj = (np.arange(256) >> 3) & 7
print("P1 duration by j:         " + " ".join(f"{np.nanmedian(p1[:, jj == k]):.2f}" for k in range(8)))
med_b = np.nanmedian(p1, axis=0)
order = np.argsort(med_b)
print("fastest blocks (b: med P1):", [(int(b), round(float(med_b[b]), 2)) for b in order[:6]])
print("slowest blocks (b: med P1):", [(int(b), round(float(med_b[b]), 2)) for b in order[-6:]])
print("per-launch spread of P1 (max-min) median: %.2f us; spread of block medians: %.2f us"
      % (np.nanmedian(np.nanmax(p1, axis=1) - np.nanmin(p1, axis=1)), np.nanmax(med_b) - np.nanmin(med_b)))
st = t[:, :, 0]
print("start offset by XCD:      " + " ".join(f"{np.nanmedian(st[:, x::8]):.2f}" for x in range(8)))
for i, n in enumerate(names[1:], 1):
    d = t[:, :, i] - t[:, :, i - 1]
    print(f"segment -> {n:14s} median {np.nanmedian(d):6.2f}  p90 {np.nanpercentile(d, 90):6.2f}")

# ---- systematic per-block structure (is the spread tied to XCD / position, i.e. fixable by a static map?)
if os.environ.get("CF_TL_ABS", "0") == "1":
    # absolute times (since the first workgroup of the launch started): what a static share table has to equalise
    for idx, nm in ((1, "P1 done"), (3, "P2 done")):
        mb = np.nanmedian(t[:, :, idx], axis=0)
        print(f"\nmedian ABSOLUTE time of '{nm}' per block, rows = b>>3, cols = b&7 (XCD):")
        for r in range(32):
            print(f"{r:2d}: " + " ".join(f"{mb[r * 8 + x]:6.2f}" for x in range(8)))
        print("col medians: " + " ".join(f"{np.nanmedian(mb[x::8]):6.2f}" for x in range(8)) +
              "   row-group (b>>6) medians: " + " ".join(f"{np.nanmedian(mb[64 * q:64 * q + 64]):6.2f}" for q in range(4)))
    np.save(os.environ.get("CF_TL_SAVE", "/tmp/tl_abs.npy"), np.nanmedian(t, axis=0))
    np.save(os.environ.get("CF_TL_SAVE", "/tmp/tl_abs.npy").replace(".npy", "_all.npy"), t)
if os.environ.get("CF_TL_MAP", "0") == "1":
    p2 = t[:, :, 3] - t[:, :, 0]          # start -> phase 2 done, per block
    mb = np.nanmedian(p2, axis=0)
    print("\nmedian (start -> P2 done) per block, rows = b>>3 (0..31), cols = b&7 (XCD):")
    for r in range(32):
        print(f"{r:2d}: " + " ".join(f"{mb[r * 8 + x]:6.2f}" for x in range(8)))
    lb = np.arange(256) ^ (64 if FLAGS & 2 else 0) ^ (1 if FLAGS & 4 else 0)    # logical block of a physical one
    print("segment medians by LOGICAL b>>6 (head slot) / logical b&1:")
    for i, n in enumerate(names[1:], 1):
        d = t[:, :, i] - t[:, :, i - 1]
        print(f"  -> {n:14s} " + " ".join(f"{np.nanmedian(d[:, (lb >> 6) == q]):6.2f}" for q in range(4)) + "   | phys even/odd XCD " +
              " ".join(f"{np.nanmedian(d[:, (np.arange(256) & 1) == q]):6.2f}" for q in range(2)))
    for slot, n in ((14, "q rows done"), (15, "k rows done")):
        d = (np.stack(raw_acc)[:, :, slot] - np.stack(raw_acc)[:, :, 0]) / 100.0
        print(f"  since start: {n:14s} " + " ".join(f"{np.nanmedian(d[:, (lb >> 6) == q]):6.2f}" for q in range(4)) + "   | phys even/odd XCD " +
              " ".join(f"{np.nanmedian(d[:, (np.arange(256) & 1) == q]):6.2f}" for q in range(2)))
    fm = np.nanmedian(f, axis=0)     # [256, 6]
    for k, (slot, n) in enumerate(sorted(fine.items())):
        print(f"  fine {n:18s} " + " ".join(f"{np.nanmedian(fm[(lb >> 6) == q, k]):6.2f}" for q in range(4)))
    hw = raw[:, 13].astype(np.int64)
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
    print("hw place of blocks (se.sh.cu) rows = b>>3, cols = XCD:")
    for r in range(32):
        print(f"{r:2d}: " + " ".join(f"{se[r*8+x]}.{sh[r*8+x]}.{cu[r*8+x]:2d}" for x in range(8)))
    print("col medians: " + " ".join(f"{np.nanmedian(mb[x::8]):6.2f}" for x in range(8)))
    print("run-to-run std of a block (median over blocks): %.2f us; std across block medians: %.2f us" %
          (np.nanmedian(p2.std(axis=0)), mb.std()))
