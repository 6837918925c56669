import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, hashlib
import verifiable_mpc_amd as vm
ctx = vm.get_context()
rng = np.random.default_rng(1)
for mb in (32, 64):
    data = rng.integers(0, 256, size=mb << 20, dtype=np.uint8)
    d = ctx.upload(data)
    ctx.sync()
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        got = ctx.sha256_chunks(d.ptr, len(data), 4096)
        ts.append((time.perf_counter() - t0) * 1e3)
    want = b"".join(hashlib.sha256(data[i:i + 4096].tobytes()).digest() for i in range(0, 1 << 20, 4096))
    assert got[:len(want)] == want
    print(f"{mb} MB: {min(ts):.3f} ms (min of 6, with download of the digests)")

# odd lengths and chunk sizes (tails of every kind: empty, < 56, 56..63 bytes, unaligned chunk starts)
for nbytes, chunk in ((1, 4096), (55, 64), (56, 64), (63, 64), (64, 64), (65, 64), (4096 * 3 + 119, 4096), (4096 * 2 + 4040, 4096),
                      (1000, 100), (12345, 777), (1 << 16, 4096)):
    data = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    d = ctx.upload(data)
    got = ctx.sha256_chunks(d.ptr, nbytes, chunk)
    want = b"".join(hashlib.sha256(data[i:i + chunk].tobytes()).digest() for i in range(0, nbytes, chunk))
    assert got == want, (nbytes, chunk)
print("tails ok")
