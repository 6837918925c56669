"""Pins oracle/bn256_ref.py: curve constants by internal known answers, and the prover's sums by
the fixture produced from the reference's own trinocchio/pynocchio.py (keygen + compute_proof run
over the mpyc shim, tests/golden/make_fixtures.py::pynocchio_case).  CPU only."""
import json
import os

import pytest

from oracle import bn256_ref as bn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pynocchio_bn256.json")
h2i = lambda s: int(s, 16)


@pytest.fixture(scope="module")
def case():
    return json.load(open(GOLDEN))


def dec(v):
    if v is None:
        return None
    vals = [h2i(x) for x in v]
    return (vals[0], vals[1]) if len(vals) == 2 else ((vals[0], vals[1]), (vals[2], vals[3]))


def test_curve_constants():
    v = 1868033                                   # verifiable_mpc/ac20/pairing.py:49
    u = v ** 3
    assert bn.P == 36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1 and bn.P.bit_length() == 256
    assert bn.N == bn.P - 6 * u**2                # trace of Frobenius t = 6u^2 + 1
    assert bn.E1.on_curve(bn.G1) and bn.E2.on_curve(bn.G2)
    assert bn.E1.mul(bn.N, bn.G1) is None and bn.E2.mul(bn.N, bn.G2) is None
    assert bn.E1.mul(-3, bn.G1) == bn.E1.neg(bn.E1.mul(3, bn.G1))
    assert bn.g1_from_bytes(bn.g1_to_bytes(bn.G1)) == bn.G1 and bn.g2_from_bytes(bn.g2_to_bytes(bn.G2)) == bn.G2
    assert bn.g1_from_bytes(bytes(64)) is None


def expected_elements(case):
    """compute_proof restated (pynocchio.py:228-273) with the oracle's group law."""
    mid, c, ek = case["indices_mid"], [h2i(v) for v in case["c"]], case["evalkey"]
    dv, dw, dy = (h2i(v) for v in case["deltas"])
    hc = [h2i(v) for v in case["h"]]

    def elem(E, fmt, zk):
        acc = E.msm([c[i] for i in mid], [dec(ek[fmt(i)]) for i in mid])
        for d, name in zk:
            acc = E.add(acc, E.mul(d, dec(ek[name])))
        return acc
    return {
        "r_v*v_mid*g1": elem(bn.E1, lambda i: f"r_v*v{i}*g1", [(dv, "r_v*t*g1")]),
        "r_w*w_mid*g2": elem(bn.E2, lambda i: f"r_w*w{i}*g2", [(dw, "r_w*t*g2")]),
        "r_y*y_mid*g1": elem(bn.E1, lambda i: f"r_y*y{i}*g1", [(dy, "r_y*t*g1")]),
        "r_v*alpha_v*v_mid*g1": elem(bn.E1, lambda i: f"r_v*alpha_v*v{i}*g1", [(dv, "r_v*alpha_v*t*g1")]),
        "r_w*alpha_w*w_mid*g1": elem(bn.E1, lambda i: f"r_w*alpha_w*w{i}*g1", [(dw, "r_w*alpha_w*t*g1")]),
        "r_y*alpha_y*y_mid*g1": elem(bn.E1, lambda i: f"r_y*alpha_y*y{i}*g1", [(dy, "r_y*alpha_y*t*g1")]),
        "r_v*beta*v_mid+r_w*beta*w_mid+r_y*beta*y_mid*g1": elem(
            bn.E1, lambda i: f"r_v*beta*v+r_w*beta*w+r_y*beta*y{i}_g1",
            [(dv, "r_v*beta*t*g1"), (dw, "r_w*beta*t*g1"), (dy, "r_y*beta*t*g1")]),
        "h*g1": bn.E1.msm(hc, [dec(ek[f"s^{i}*g1"]) for i in range(len(hc))]),
    }


def test_pynocchio_fixture(case):
    want = {k: dec(v) for k, v in case["proof"].items()}
    assert expected_elements(case) == want
    for name, pt in case["evalkey"].items():
        E = bn.E2 if name.endswith("g2") else bn.E1
        assert E.on_curve(dec(pt)), name
