"""Shared helpers for the test-suite (CPU + GPU)."""
import json
import os

import numpy as np
import torch

from oracle import cf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Tolerances (max-abs on fp16 outputs).  north_star: output within 1e-3 of the reference on
# identical inputs at the reference test's distribution (randn*0.1).  k_new/v_new: the reference
# kernel adds 4 fp16-rounded partials in fp16 (dsm.cuh:127-139) so it is itself 1 fp16 ulp away
# from its eager definition; we hold <= 1 fp16 ulp of the value's binade.
TOL_OUT = 1e-3


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrays


def golden_inputs(meta):
    dims = O.LayerDims(*meta["dims"])
    inp = O.make_inputs(meta["seed"], meta["seq_len"], dims,
                        weight_layout=meta["weight_layout"], **meta["dist"])
    return dims, inp


def ulp16(t: torch.Tensor) -> torch.Tensor:
    """Size of one fp16 ulp at each element's magnitude."""
    a = t.float().abs().clamp_min(2.0 ** -14)
    return torch.exp2(torch.floor(torch.log2(a)) - 10)


def max_abs(a, b):
    return (a.float() - b.float()).abs().max().item()


def max_ulp(a, ref):
    return ((a.float() - ref.float()).abs() / ulp16(ref)).max().item()


def max_err_in_ulps_of_max(a, ref):
    """max-abs error measured in fp16 ulps of the LARGEST reference magnitude (cancellation-safe)."""
    return max_abs(a, ref) / ulp16(ref.float().abs().max()).item()
