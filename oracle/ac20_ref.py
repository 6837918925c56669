"""ORACLE (test infrastructure, NOT the product): AC20 pivot protocols, big-int Python.

CPU restatement of the reference's hot path, function by function:
    verifiable_mpc/ac20/pivot.py              (vector_commitment, fiat_shamir_hash,
                                               forms, Pi_s prover/verifier)
    verifiable_mpc/ac20/compressed_pivot.py   (Protocol 4 / Protocol 5)
    verifiable_mpc/ac20/circuit_sat_r1cs.py:47-93 (create_generators)
written in a flat, explicit style (lists of residues and coordinate tuples, explicit
transcript strings) instead of the reference's operator-overloaded objects.  All
randomness is passed in explicitly (the reference draws it from module-level
`prng`s: compressed_pivot.py:105-106, circuit_sat_r1cs.py:64,81).

Scalars are ints mod l standing for MPyC GF(l) elements; points are (X, Y, Z) tuples
from oracle/ed25519_ref.py.  Two transcript modes exist:
  "reference": byte-for-byte the reference's pre-image str(input_list)
               (pivot.py:131-136) under the [mpyc-recall] repr formats;
  "compact":   the build's own opt-in byte transcript, which
               hashes O(1) bytes per round.

PARITY STATUS: protocol logic pinned by tests/golden/*.json, produced by running the
reference's own pivot.py / compressed_pivot.py over a build-written mpyc shim
(tests/golden/make_fixtures.py).  Real-MPyC byte formats: parity unpinned (see
oracle/ed25519_ref.py header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import hashlib

from . import ed25519_ref as ed

ELL = ed.ELL


# ----------------------------------------------------------------------------------
# pivot.py
# ----------------------------------------------------------------------------------

def fiat_shamir_hash_text(text, order=ELL):
    """pivot.py:131-136 on an already-stringified input list."""
    digest = hashlib.sha256(text.encode("utf-8")).digest()
    return int.from_bytes(digest, "little") % order


def form_eval(coeffs, constant, values):
    """AffineForm.eval, pivot.py:84-92."""
    assert len(values) == len(coeffs), \
        "Length of inputs to be equal to coefficients of linear form."
    return (sum(c * v for c, v in zip(coeffs, values)) + constant) % ELL


def form_repr(coeffs, constant, raw=None):
    """AffineForm.__repr__, pivot.py:81-82: '[c0, c1, ...], const'.

    `raw[i]` (optional) is the exact Python int of a coefficient that the caller passed
    as a plain int rather than a GF(l) element: the reference never reduces those
    (pivot.py:62-70 multiplies ints as ints), so their repr is the unreduced integer."""
    if raw is None:
        items = [ed.scalar_repr(c) for c in coeffs]
    else:
        items = [ed.scalar_repr(c) if raw[i] is None else str(raw[i])
                 for i, c in enumerate(coeffs)]
    return "[" + ", ".join(items) + "], " + ed.scalar_repr(constant)


def affine_to_linear(coeffs, constant, y):
    """pivot.py:148-153: drop the constant term from the form and from y."""
    return list(coeffs), 0, (y - constant) % ELL


def vector_commitment(x, gamma, g, h, signed_exponents=True):
    """pivot.py:139-145: h^gamma * prod g_i^{x_i}; per-term double-and-add
    (pivot.py:143) then the reduce tree of pivot.list_mul (pivot.py:26-28).

    `signed_exponents`: field-element exponents pass through pivot._int
    (pivot.py:119-128), i.e. the signed residue; plain Python ints are used as is."""
    assert len(g) >= len(x), "Not enough generators."
    if hasattr(g, "commit"):        # a c_oracle.PointArray: the same ladders + tree in C (threaded)
        return g.commit(x, gamma, h, signed_exponents)
    conv = ed.scalar_int if signed_exponents else (lambda v: v)
    terms = [ed.pt_repeat(g[i], conv(x_i)) for i, x_i in enumerate(x)]
    prod = ed.tree_reduce(ed.pt_add, terms, initial=ed.IDENTITY)
    return ed.pt_add(ed.pt_repeat(h, gamma), prod)


def create_generators(exponents_g, exponent_k=None, base=ed.BASE):
    """circuit_sat_r1cs.py:47-93 (PivotChoice.pivot / .compressed branches):
    g_i = h ** r_i (:64-70), k = h ** r (:81); dict key order g, h, k (:82)."""
    h = base
    gens = {"g": [ed.pt_repeat(h, r) for r in exponents_g], "h": h}
    if exponent_k is not None:
        gens["k"] = ed.pt_repeat(h, exponent_k)
    return gens


def generators_repr(generators):
    """str() of the generators dict as it appears in compressed_pivot.py:118."""
    parts = []
    for key in ("g", "h", "k"):
        if key not in generators:
            continue
        v = generators[key]
        if key == "g":
            parts.append("'g': [" + ", ".join(ed.pt_repr(p) for p in v) + "]")
        else:
            parts.append(f"'{key}': " + ed.pt_repr(v))
    return "{" + ", ".join(parts) + "}"


def pis_hash_text(t, A, g, h, P, coeffs, constant, y):
    """Pre-image of pivot.py:169-174 / :194-201."""
    return ("[" + ed.scalar_repr(t) + ", " + ed.pt_repr(ed.pt_normalize(A)) + ", ["
            + ", ".join(ed.pt_repr(p) for p in g) + "], " + ed.pt_repr(h) + ", "
            + ed.pt_repr(ed.pt_normalize(P)) + ", " + form_repr(coeffs, constant) + ", "
            + ed.scalar_repr(y) + "]")


def prove_linear_form_eval(g, h, P, coeffs, constant, y, x, gamma, r, rho):
    """Pi_s prover, pivot.py:156-181.  Returns (z, phi, c)."""
    coeffs, constant, y = affine_to_linear(coeffs, constant, y)
    t = form_eval(coeffs, constant, r)
    A = vector_commitment(r, rho, g, h)
    c = fiat_shamir_hash_text(pis_hash_text(t, A, g, h, P, coeffs, constant, y))
    z = [(c * x_i + r[i]) % ELL for i, x_i in enumerate(x)]
    phi = (c * gamma + rho) % ELL
    return z, phi, c


def verify_linear_form_proof(g, h, P, coeffs, constant, y, z, phi, c):
    """Pi_s verifier, pivot.py:184-205."""
    coeffs, constant, y = affine_to_linear(coeffs, constant, y)
    # pivot.py:187: vector_commitment(z, phi, g, h) * ((P ** c) ** (-1))
    A_check = ed.pt_add(vector_commitment(z, phi, g, h),
                        ed.pt_repeat(ed.pt_repeat(P, c), -1))
    t_check = (form_eval(coeffs, constant, z) - c * y) % ELL
    return c == fiat_shamir_hash_text(
        pis_hash_text(t_check, A_check, g, h, P, coeffs, constant, y))


# ----------------------------------------------------------------------------------
# compact transcript (build-defined, opt-in)
# ----------------------------------------------------------------------------------

CHUNK = 4096


def chunked_digest(tag, data):
    """Two-level SHA-256: leaves over 4096-byte chunks, root over tag|len|leaves.
    Chosen so the leaves can be hashed in parallel on the device."""
    leaves = b"".join(hashlib.sha256(data[o:o + CHUNK]).digest()
                      for o in range(0, len(data), CHUNK))
    return hashlib.sha256(tag + len(data).to_bytes(8, "little") + leaves).digest()


def _sc(v):
    return int(v % ELL).to_bytes(32, "little")


def compact_generators_digest(generators):
    g = generators["g"]
    data = g.affine().tobytes() if hasattr(g, "affine") else b"".join(ed.affine_to_bytes(p) for p in g)
    data += ed.affine_to_bytes(generators["h"]) + ed.affine_to_bytes(generators["k"])
    return chunked_digest(b"vmpc-ac20/gens/v1", data)


def compact_form_digest(coeffs):
    return chunked_digest(b"vmpc-ac20/form/v1", b"".join(_sc(c) for c in coeffs))


def compact_p5_seed(generators, P, coeffs, y, t, A, gens_digest=None):
    gd = gens_digest or compact_generators_digest(generators)
    return hashlib.sha256(b"vmpc-ac20/p5/v1" + gd + compact_form_digest(coeffs)
                          + ed.affine_to_bytes(P) + _sc(y) + _sc(t)
                          + ed.affine_to_bytes(A)).digest()


def compact_challenge(digest):
    return int.from_bytes(digest, "little") % ELL


# ----------------------------------------------------------------------------------
# compressed_pivot.py
# ----------------------------------------------------------------------------------

def p4_hash_text(A, B, g_hat, k, Q, Lt, Lt_raw=None):
    """Pre-image of compressed_pivot.py:51-59 / :166-173: normalised A, B, Q; the
    current generator vector and k as they are (un-normalised representatives)."""
    return ("[" + ed.pt_repr(ed.pt_normalize(A)) + ", " + ed.pt_repr(ed.pt_normalize(B))
            + ", [" + ", ".join(ed.pt_repr(p) for p in g_hat) + "], " + ed.pt_repr(k)
            + ", " + ed.pt_repr(ed.pt_normalize(Q)) + ", " + form_repr(Lt, 0, Lt_raw) + "]")


def p5_hash_texts(t, A, generators, P, coeffs, constant, y, raw=None):
    """Both pre-images of compressed_pivot.py:117-130 (c0 with tag 0, c1 with tag 1)."""
    head = ("[" + ed.scalar_repr(t) + ", " + ed.pt_repr(ed.pt_normalize(A)) + ", "
            + generators_repr(generators) + ", " + ed.pt_repr(ed.pt_normalize(P)) + ", "
            + form_repr(coeffs, constant, raw) + ", " + ed.scalar_repr(y) + ", ")
    tail = ", 'First hash of compressed pivot']"
    return head + "0" + tail, head + "1" + tail


def _dot(a, b):
    return sum(x * y for x, y in zip(a, b)) % ELL


def fold_generators(g_l, g_r, c):
    """compressed_pivot.py:64 / :178: g'_i = (g_l[i] ** c) * g_r[i]."""
    if hasattr(g_l, "fold"):        # c_oracle.PointArray
        return g_l.fold(g_r, c)
    return [ed.pt_add(ed.pt_repeat(g_l[i], c), g_r[i]) for i in range(len(g_l))]


def protocol_4_prover(g_hat, k, Q, Lt, z_hat, proof, mode="reference", state=None,
                      trace=None, Lt_raw=None):
    """compressed_pivot.py:29-86, recursion unrolled into a loop.  `Lt_raw`: unreduced
    Python-int coefficients of the FIRST round's form (see form_repr); from the second
    round on every coefficient has been multiplied by gf(c) (:70) and is a field element."""
    round_i = 0
    while True:
        half = len(g_hat) // 2
        g_l, g_r = g_hat[:half], g_hat[half:]
        z_l, z_r = z_hat[:half], z_hat[half:]
        # :41-42  L_tilde([0]*half + z_l) = <Lt_r, z_l>,  L_tilde(z_r + [0]*half) = <Lt_l, z_r>
        A = vector_commitment(z_l, ed.scalar_int(_dot(Lt[half:], z_l)), g_r, k)
        B = vector_commitment(z_r, ed.scalar_int(_dot(Lt[:half], z_r)), g_l, k)
        proof["A" + str(round_i)] = A
        proof["B" + str(round_i)] = B
        if mode == "reference":
            c = fiat_shamir_hash_text(
                p4_hash_text(A, B, g_hat, k, Q, Lt, Lt_raw if round_i == 0 else None))  # :51-59
        else:
            state = hashlib.sha256(state + round_i.to_bytes(4, "little")
                                   + ed.affine_to_bytes(A) + ed.affine_to_bytes(B)).digest()
            c = compact_challenge(state)
        if trace is not None:
            trace.setdefault("c", []).append(c)
        g_prime = fold_generators(g_l, g_r, c)                                   # :64
        Q = ed.pt_add(ed.pt_add(A, ed.pt_repeat(Q, c)), ed.pt_repeat(B, c * c))  # :66
        Lt = [(Lt[i] * c + Lt[half + i]) % ELL for i in range(half)]             # :70-73
        z_prime = [(z_l[i] + c * z_r[i]) % ELL for i in range(half)]            # :76
        if trace is not None:
            trace.setdefault("g_hat", []).append(g_prime)
            trace.setdefault("Q", []).append(Q)
        if len(z_prime) <= 2:                                                    # :77-79
            proof["z_prime"] = z_prime
            return proof
        g_hat, z_hat = g_prime, z_prime
        round_i += 1


def _lt_raw(raw, c1):
    """(L.coeffs + [0]) * c1 for plain-int coefficients (compressed_pivot.py:141)."""
    if raw is None:
        return None
    return [None if v is None else v * c1 for v in raw] + [None]


def protocol_5_prover(generators, P, coeffs, constant, y, x, gamma, r, rho,
                      mode="reference", trace=None, coeffs_raw=None):
    """compressed_pivot.py:89-145.  `r`, `rho` are the prover's masks (:105-106).
    `coeffs_raw`: see form_repr (only changes the "reference" transcript text)."""
    g, h, k = generators["g"], generators["h"], generators["k"]
    proof = {}
    n = len(x)
    coeffs, constant, y = affine_to_linear(coeffs, constant, y)
    assert bin(n + 1).count("1") == 1, \
        "This implementation requires n+1 to be power of 2 (else, use padding with zeros)."
    t = form_eval(coeffs, constant, r)                                           # :108
    A = vector_commitment(r, rho, g, h, signed_exponents=False)                  # :110
    proof["t"] = t
    proof["A"] = A
    state = None
    if mode == "reference":
        t0, t1 = p5_hash_texts(t, A, generators, P, coeffs, constant, y, coeffs_raw)
        c0, c1 = fiat_shamir_hash_text(t0), fiat_shamir_hash_text(t1)            # :125-130
    else:
        seed = compact_p5_seed(generators, P, coeffs, y, t, A)
        c0 = compact_challenge(hashlib.sha256(seed + b"\x00").digest())
        c1 = compact_challenge(hashlib.sha256(seed + b"\x01").digest())
    z = [(c0 * x_i + r[i]) % ELL for i, x_i in enumerate(x)]                     # :134
    phi = (c0 * gamma + rho) % ELL                                               # :135
    z_hat = z + [phi]
    g_hat = g.appended(h) if hasattr(g, "appended") else list(g) + [h]                                                        # :138
    Q = ed.pt_add(ed.pt_add(A, ed.pt_repeat(P, c0)),
                  ed.pt_repeat(k, ed.scalar_int(c1 * (c0 * y + t))))             # :140
    Lt = [(c * c1) % ELL for c in coeffs] + [0]                                  # :141
    assert form_eval(coeffs, 0, z) * c1 % ELL == _dot(Lt, z_hat)                 # :142
    if mode != "reference":
        # Q is a function of values already bound by `seed`; it is not hashed again
        state = hashlib.sha256(b"vmpc-ac20/p4/v2" + seed + ed.affine_to_bytes(k)).digest()
    if trace is not None:
        trace.update({"c0": c0, "c1": c1, "Q0": Q})
    return protocol_4_prover(g_hat, k, Q, Lt, z_hat, proof, mode, state, trace,
                             _lt_raw(coeffs_raw, c1))


def protocol_4_verifier(g_hat, k, Q, Lt, proof, mode="reference", state=None, Lt_raw=None):
    """compressed_pivot.py:148-202."""
    round_i = 0
    while True:
        half = len(g_hat) // 2
        g_l, g_r = g_hat[:half], g_hat[half:]
        A = proof["A" + str(round_i)]
        B = proof["B" + str(round_i)]
        if mode == "reference":
            c = fiat_shamir_hash_text(
                p4_hash_text(A, B, g_hat, k, Q, Lt, Lt_raw if round_i == 0 else None))
        else:
            state = hashlib.sha256(state + round_i.to_bytes(4, "little")
                                   + ed.affine_to_bytes(A) + ed.affine_to_bytes(B)).digest()
            c = compact_challenge(state)
        g_prime = fold_generators(g_l, g_r, c)                                   # :178
        Q = ed.pt_add(ed.pt_add(A, ed.pt_repeat(Q, c)), ed.pt_repeat(B, c * c))  # :180
        Lt = [(Lt[i] * c + Lt[half + i]) % ELL for i in range(half)]             # :185-188
        if len(g_prime) <= 2:                                                    # :191-198
            z_prime = proof["z_prime"]
            Q_check = vector_commitment(z_prime, ed.scalar_int(_dot(Lt, z_prime)), g_prime, k)
            return ed.pt_eq(Q_check, Q)
        g_hat = g_prime
        round_i += 1


def protocol_5_verifier(generators, P, coeffs, constant, y, proof, mode="reference",
                        coeffs_raw=None):
    """compressed_pivot.py:205-239."""
    g, h, k = generators["g"], generators["h"], generators["k"]
    coeffs, constant, y = affine_to_linear(coeffs, constant, y)
    t, A = proof["t"], proof["A"]
    state = None
    if mode == "reference":
        t0, t1 = p5_hash_texts(t, A, generators, P, coeffs, constant, y, coeffs_raw)
        c0, c1 = fiat_shamir_hash_text(t0), fiat_shamir_hash_text(t1)
    else:
        seed = compact_p5_seed(generators, P, coeffs, y, t, A)
        c0 = compact_challenge(hashlib.sha256(seed + b"\x00").digest())
        c1 = compact_challenge(hashlib.sha256(seed + b"\x01").digest())
    g_hat = g.appended(h) if hasattr(g, "appended") else list(g) + [h]
    Q = ed.pt_add(ed.pt_add(A, ed.pt_repeat(P, c0)),
                  ed.pt_repeat(k, ed.scalar_int(c1 * (c0 * y + t))))
    Lt = [(c * c1) % ELL for c in coeffs] + [0]
    if mode != "reference":
        state = hashlib.sha256(b"vmpc-ac20/p4/v2" + seed + ed.affine_to_bytes(k)).digest()
    return protocol_4_verifier(g_hat, k, Q, Lt, proof, mode, state, _lt_raw(coeffs_raw, c1))
