"""The reference's IK benchmark protocol on its three robots (``bench.py::ik_protocol_case``; also part of every default bench run as
``ik_reference_protocol``).    python tools/r05/ik_benchmark.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import ik_protocol_benchmark  # noqa: E402

if __name__ == "__main__":
    out = ik_protocol_benchmark(torch)
    for r in out["rows"]:
        print(json.dumps(r), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)
