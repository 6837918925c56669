#!/usr/bin/env python3
"""Generate tests/golden/ed25519_openssl.json with the OpenSSL 3 command-line tool.

Runs only in the build container (needs the `openssl` binary; the GPU box only reads the JSON):
    python3 tests/golden/make_openssl_vectors.py [count]

An implementation of Ed25519 that is NOT this repository's (OpenSSL's) produces, per vector:
    seed   32-byte private key            (openssl genpkey -algorithm ed25519)
    pub    32-byte RFC 8032 public key    = compress(clamp(SHA-512(seed)[:32]) * B)
    msg    the signed message
    sig    64-byte signature R || S       (openssl pkeyutl -sign -rawin), with
           S * B == R + SHA-512(R || pub || msg) * A   for A = decompress(pub)
The public keys pin the oracle's (and the kernels') fixed-base scalar multiplication on 255-bit
scalars; the signature equation pins VARIABLE-base scalar multiplication and point addition on
points this repository did not produce (tests/test_oracle_openssl.py, tests/test_gpu_openssl.py).
The fixture is data; no OpenSSL source is involved.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def _hex_field(text, name):
    m = re.search(name + r":\s*((?:[0-9a-f]{2}:?\s*)+)", text)
    return re.sub(r"[^0-9a-f]", "", m.group(1))


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    version = subprocess.check_output(["openssl", "version"], text=True).strip()
    out = {"generator": version, "vectors": []}
    with tempfile.TemporaryDirectory() as tmp:
        key, msgf, sigf = (os.path.join(tmp, n) for n in ("k.pem", "m.bin", "s.bin"))
        for i in range(count):
            subprocess.check_call(["openssl", "genpkey", "-algorithm", "ed25519", "-out", key])
            text = subprocess.check_output(["openssl", "pkey", "-in", key, "-text", "-noout"], text=True)
            seed, pub = _hex_field(text, "priv"), _hex_field(text, "pub")
            assert len(seed) == 64 and len(pub) == 64
            # messages of varying length, derived from the index (nothing secret about them)
            msg = hashlib.sha256(b"vmpc openssl vector %d" % i).digest() * (1 + i % 3)
            msg = msg[:1 + (7 * i) % len(msg)]
            with open(msgf, "wb") as f:
                f.write(msg)
            subprocess.check_call(["openssl", "pkeyutl", "-sign", "-inkey", key, "-rawin", "-in", msgf,
                                   "-out", sigf])
            with open(sigf, "rb") as f:
                sig = f.read()
            assert len(sig) == 64
            out["vectors"].append({"seed": seed, "pub": pub, "msg": msg.hex(), "sig": sig.hex()})
    with open(os.path.join(HERE, "ed25519_openssl.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(len(out["vectors"]), "vectors from", version)


if __name__ == "__main__":
    main()
