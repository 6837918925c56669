"""how many host threads does the oracle's C restatement actually get on this box? (cgroup quota vs nproc)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, ed25519_ref as ed  # noqa: E402

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
rng = np.random.default_rng(1)
n = 1 << 17
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
sc[:, 31] &= 0x0F
base = np.frombuffer(ed.proj_to_bytes(ed.BASE), np.uint8)
c_oracle.set_threads(64)
_, pts = c_oracle.fixed_base(base, sc[:4096])
pts = np.tile(pts, (n // 4096, 1))
for t in (1, 8, 16, 32, 64, 128, 256):
    c_oracle.set_threads(t)
    m = n if t > 1 else n // 16
    t0 = time.perf_counter()
    c_oracle.vector_commitment(sc[:m], sc[0], pts[:m], pts[0])
    dt = time.perf_counter() - t0
    print(f"threads {t:4d}: {m / dt:10.0f} ladders/s  ({dt:.2f} s for 2^{m.bit_length() - 1})", flush=True)
