#!/usr/bin/env python3
"""bench.bn256_timing alone (BASELINE config 5): python3 scripts/bn_bench_probe.py [log2n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import verifiable_mpc_amd as vm
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
out = bench.bn256_timing(vm, vm.get_context(), k)
print(json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in out.items() if "roofline" not in a}))
