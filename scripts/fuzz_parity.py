"""Randomised parity sweep on the GPU (the case generator and checks live in tests/_fuzz.py; the driver-run suite
runs a fixed-seed slice of it).  Usage: fuzz_parity.py [cases] [seed] [only this case, verbose]; FUZZ_LARGE=1 for
fewer, larger cases (several seconds of oracle time each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests._fuzz import FuzzFailure, run_cases  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
try:
    r = run_cases(cases, seed, torch.device("cuda:0"), large=os.environ.get("FUZZ_LARGE") == "1", only=only,
                  log=lambda m: print(m, flush=True), oracle_chunk=int(os.environ.get("FUZZ_ORACLE_CHUNK", "32")))
except FuzzFailure as e:
    print("FAIL", e, flush=True)
    sys.exit(1)
print(f"all {cases} cases ok in {r['seconds']:.0f} s ({r['near']} within 3x of the gradient bound, "
      f"{r['ties']} skipped for depth near-ties)")
