"""Phase timeline of the persistent MLA kernel (k_mla_fused): per-role stamp statistics over the 256 workgroups.

    python tools/mla_timeline.py [--seq 4096] [--rope-scores]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clusterfusion_amd as cfa  # noqa: E402
from clusterfusion_amd import _lib  # noqa: E402
from oracle import mla_oracle as M  # noqa: E402

ORDER = ["input", "weight_q_nope", "weight_q_pe", "weight_uk", "weight_kv_nope", "weight_k_pe", "weight_uv", "weight_o",
         "ckv_cache", "rms_input_weight", "rms_ckv_weight", "cos", "sin"]
NAMES = ["start", "tiles requested", "norm done", "A tile multiplied", "A published", "q_nope arrived (B)", "B published",
         "q arrived (C)", "attention done (C)", "C published", "(m,l) arrived (D)", "merged (D)", "D published",
         "o_h arrived (E)", "done"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--rope-scores", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfa.set_path("fused")
    layers = [[M.make_mla_inputs(50 + i, a.seq)[k].to(dev) for k in ORDER] for i in range(6)]
    trace = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    lib = _lib.load()
    acc = np.zeros((256, 16))
    n = 0
    for r in range(a.reps + 3):
        for g in layers:
            cfa.deepseek_decoder_layer(*g, rope_scores=a.rope_scores)   # cold-ish: other layers in between
        trace.zero_()
        lib.cf_debug_set_trace(C.c_void_p(trace.data_ptr()))
        cfa.deepseek_decoder_layer(*layers[0], rope_scores=a.rope_scores)
        torch.cuda.synchronize()
        lib.cf_debug_set_trace(None)
        if r < 3:
            continue
        t = trace.cpu().numpy().reshape(256, 16).astype(np.float64)
        t0 = t[:, 0].min()
        rel = np.where(t > 0, (t - t0) / 100.0, np.nan)     # 100 MHz -> us
        acc += np.nan_to_num(rel)
        n += 1
        last = rel
    rel = np.where(np.isnan(last), np.nan, acc / n)
    print(f"S={a.seq} rope_scores={a.rope_scores}: stamp (us after the first workgroup started), mean over {n} runs")
    print(f"{'stamp':28s} {'wgs':>4s} {'min':>7s} {'median':>7s} {'max':>7s}")
    for s, name in enumerate(NAMES):
        col = rel[:, s]
        col = col[~np.isnan(col)]
        if col.size:
            print(f"{name:28s} {col.size:4d} {col.min():7.2f} {np.median(col):7.2f} {col.max():7.2f}")
    cfa.check_device_errors()


if __name__ == "__main__":
    main()
