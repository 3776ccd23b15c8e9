#!/usr/bin/env python
"""Extended fuzz run of the three seeded generators of tests/test_parity_gpu.py over seeds the test suite does not hold (the suite
keeps 44 so that it stays within minutes):

    python tools/fuzz_extended.py [seconds_per_generator] [first_seed]

Calls test_fuzz_paged_batch_entry_vs_oracle (1 .. 32 rows through the reference's paged / batched entry, page sizes 1 / 2 / 16 /
64) and test_fuzz_single_row_geometries_paged_vs_oracle (the eight gated geometries, lengths around the arm boundaries up to
20k) with fresh seeds until the time is used up; every case compares the HIP path with the oracle exactly as the tests do.
Prints one JSON line per generator (cases run, failures with their seeds); exit code 1 when a case failed."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import clusterfusion_amd as cfa
from clusterfusion_amd import _lib

from tests import test_parity_gpu as T

_lib.load()
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for name in ("test_fuzz_paged_batch_entry_vs_oracle", "test_fuzz_single_row_geometries_paged_vs_oracle", "test_fuzz_gqa_small_batch_vs_oracle"):
    fn = getattr(T, name)
    t0 = time.time()
    seed, fails, variants = FIRST, [], {}
    while time.time() - t0 < SECONDS:
        try:
            fn(cfa, seed)
            variants[cfa.last_variant()] = variants.get(cfa.last_variant(), 0) + 1
        except Exception as e:      # noqa: BLE001 -- a failing case is the result here
            fails.append({"seed": seed, "error": "".join(traceback.format_exception_only(type(e), e))[-400:]})
            cfa.set_path("auto")
        seed += 1
    bad += len(fails)
    print(json.dumps({"generator": name, "first_seed": FIRST, "cases": seed - FIRST, "failed": len(fails), "failures": fails[:8],
                      "kernels": variants, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
