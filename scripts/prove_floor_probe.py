#!/usr/bin/env python3
"""AC20 prove / verify at N = 2^k in both transcripts with the hash floor of the reference transcript (bench.prove_timing)
    python3 scripts/prove_floor_probe.py [k]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import verifiable_mpc_amd as vm

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = bench.prove_timing(vm, vm.get_context(), k, np.random.default_rng(99))
print(json.dumps({key: (round(v, 2) if isinstance(v, float) else v) for key, v in out.items()
                  if not key.startswith("roofline")}))
