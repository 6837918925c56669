"""Data stand-ins for the reference's circuit front end (`circuit`, `x` of circuit_sat_cb.py:255): what
circuit_builder + calculate_fgh_polys would hand to Protocol 8, either replayed from the fixture of the seeded
demos/demo_zkp_ac20.py --elliptic run (tests/golden/demo_zkp_ac20_elliptic.json, made by importing the reference) or
synthesised for a group the fixture does not cover."""
from .ac20 import pivot


def typed_value(gf, s):
    """"i:<decimal>" -> int, "f:<hex residue>" -> field element: the demo mixes both and the pre-image shows it"""
    kind, v = s.split(":")
    return int(v) if kind == "i" else gf(int(v, 16))


def _form(gf, rec):
    cls = pivot.LinearForm if rec["linear"] else pivot.AffineForm
    return cls([typed_value(gf, c) for c in rec["coeffs"]], typed_value(gf, rec["constant"]))


class FixtureCircuit:
    """the demo's circuit as recorded: N = 128, 80 inputs"""

    def __init__(self, case, gf):
        self.p8, self.gf = case["protocol8"], gf
        self.g_length = case["n"]
        self.first_challenges = []

    def __str__(self):
        return self.p8["circuit_str"]

    def inputs(self):
        # x is a prefix of z (circuit_sat_cb.py:91); its length is not recorded and does not matter to the path
        return [typed_value(self.gf, v) for v in self.p8["z_typed"]]

    def witness(self, x):
        return list(x)

    def responses(self, c, z):
        self.first_challenges.append(c)
        assert format(c, "x") == self.p8["hashes"][0]["c"], "first Protocol-8 challenge differs from the reference's"
        return [typed_value(self.gf, v) for v in self.p8["y_typed"]]

    def forms(self, c, y):
        assert format(c, "x") == self.p8["hashes"][0]["c"]
        outputs = [typed_value(self.gf, v) for v in self.p8["outputs_typed"]]
        return outputs, [_form(self.gf, f) for f in self.p8["circuit_forms"]], \
            [_form(self.gf, f) for f in self.p8["lin_forms"]]


class SyntheticCircuit:
    """Any field: `n_forms` random affine forms F_j with outputs F_j(z); the three challenge-dependent forms are
    random forms seeded by the challenge, their values on z the responses (y3 is made the product by shifting the
    third form's constant).  lin_forms vanish on z, which is all the pivot proves."""

    def __init__(self, gf, g_length, n_forms, seed):
        import random
        self.gf, self.g_length, self.n_forms, self.seed = gf, g_length, n_forms, seed
        rng = random.Random(seed)
        self.z = [gf(rng.randrange(gf.order)) for _ in range(g_length)]
        self.static = [pivot.AffineForm([gf(rng.randrange(gf.order)) for _ in range(g_length)],
                                        gf(rng.randrange(gf.order))) for _ in range(n_forms)]

    def __str__(self):
        return f"synthetic circuit {self.g_length}/{self.n_forms}/{self.seed}"

    def inputs(self):
        return list(self.z)

    def witness(self, x):
        return list(x)

    def _fgh(self, c):
        import random
        rng = random.Random(c)
        gf, n = self.gf, self.g_length
        f, g, h = (pivot.AffineForm([gf(rng.randrange(gf.order)) for _ in range(n)], gf(rng.randrange(gf.order)))
                   for _ in range(3))
        return f, g, h + (f(self.z) * g(self.z) - h(self.z))

    def responses(self, c, z):
        return [form(z) for form in self._fgh(c)]

    def forms(self, c, y):
        outputs = [form(self.z) for form in self.static]
        lin_forms = [form - out for form, out in zip(self.static, outputs)] + \
            [form - yi for form, yi in zip(self._fgh(c), y)]
        return outputs, list(self.static), lin_forms
