"""install(): put the GPU hot path behind an importable copy of the reference.

The reference has no plugin interface; its seam is a handful of module-level functions that its own callers reach
through the module object (`pivot.vector_commitment(...)`, `compressed_pivot.protocol_5_prover(...)`:
verifiable_mpc/ac20/circuit_sat_cb.py:103,263-266,294-299) or through a name imported into the caller
(`create_generators`: circuit_sat_cb.py:24-30, called as `cs.create_generators` by demos/demo_zkp_ac20.py:78).
install() rebinds exactly those names.  Each rebound name is a DISPATCHER that keeps the reference's original:

    Ed25519 in projective coordinates  ->  this package (HIP kernels; raises if the GPU or the library is missing)
    anything else                      ->  the reference's own function, untouched

"Anything else" is what the reference's demo uses by default - QuadraticResidues(l=1024), demos/demo_zkp_ac20.py:51 -
and the BN256 groups of the KoE pivot (`--koe`, demo_zkp_ac20.py:34-43): SURVEY.md section 8b requires that they keep
running on the reference's CPU path.  That is not a CPU fallback OF the accelerated path: an Ed25519 call never
reaches the original, whatever happens.

The group is read off the call's own arguments (the commitment base `h` / `k`, `generators["h"]`, the `group` handed
to create_generators), so one process can mix groups freely.
"""
import functools
import importlib

from . import circuit_sat, compressed_pivot, pivot
from .groups import is_ed25519_element, is_ed25519_group

_MISSING = object()


def _base_at(index, name):
    """selector: the positional / keyword argument `name` (position `index`) is an Ed25519 element"""
    def select(*args, **kwargs):
        obj = args[index] if len(args) > index else kwargs.get(name)
        return is_ed25519_element(obj)
    return select


def _generators_at(index, name):
    def select(*args, **kwargs):
        gens = args[index] if len(args) > index else kwargs.get(name)
        try:
            return is_ed25519_element(gens["h"])
        except (TypeError, KeyError, IndexError):
            return False
    return select


def _select_create_generators(g_length=None, pivot_choice=None, group=None, progress_bar=False):
    # koe hands over a LIST of two BN256 groups (circuit_sat_r1cs.py:83-89): never ours
    return circuit_sat.choice_name(pivot_choice) in ("pivot", "compressed") and not isinstance(group, (list, tuple)) \
        and is_ed25519_group(group)


def _always(*args, **kwargs):
    return True


# (module of the reference, name, this package's function, "is this call ours?")
_SEAM = [
    ("pivot", "vector_commitment", pivot.vector_commitment, _base_at(3, "h")),
    # str(input_list) of ANY objects: lists / dicts are walked, everything else contributes its own repr()
    # (pivot._feed), so the one function serves every group; it has to be ours because generator vectors that live
    # in HBM are formatted there
    ("pivot", "fiat_shamir_hash", pivot.fiat_shamir_hash, _always),
    ("pivot", "prove_linear_form_eval", pivot.prove_linear_form_eval, _base_at(1, "h")),
    ("pivot", "verify_linear_form_proof", pivot.verify_linear_form_proof, _base_at(1, "h")),
    ("compressed_pivot", "protocol_5_prover", compressed_pivot.protocol_5_prover, _generators_at(0, "generators")),
    ("compressed_pivot", "protocol_5_verifier", compressed_pivot.protocol_5_verifier, _generators_at(0, "generators")),
    ("compressed_pivot", "protocol_4_prover", compressed_pivot.protocol_4_prover, _base_at(1, "k")),
    ("compressed_pivot", "protocol_4_verifier", compressed_pivot.protocol_4_verifier, _base_at(1, "k")),
    ("circuit_sat_r1cs", "create_generators", circuit_sat.create_generators, _select_create_generators),
    # the name `from circuit_sat_r1cs import create_generators` left in the caller's namespace
    ("circuit_sat_cb", "create_generators", circuit_sat.create_generators, _select_create_generators),
]


def _dispatcher(ours, original, is_ours):
    @functools.wraps(ours)
    def call(*args, **kwargs):
        if original is None or is_ours(*args, **kwargs):
            return ours(*args, **kwargs)
        return original(*args, **kwargs)
    call.__vmpc_original__ = original
    call.__vmpc_accelerated__ = ours
    return call


class _ReferencePrng:
    """`prng` of one of this package's modules after install(): draws come from the reference module's own `prng`,
    looked up at every draw - a caller who replaces `verifiable_mpc.ac20.compressed_pivot.prng` with a seeded
    generator to make a run reproducible keeps getting that effect (draw order is the reference's:
    compressed_pivot.py:105-106, circuit_sat_r1cs.py:64,81, pivot.py:163-164)."""

    def __init__(self, module):
        self._module = module

    def __getattr__(self, name):
        return getattr(self._module.prng, name)


def install(reference_package="verifiable_mpc.ac20"):
    """Rebind the hot-path names inside the reference's modules (idempotent).  Returns the patched names."""
    patched = []
    mods = {}
    for modname, name, ours, is_ours in _SEAM:
        if modname not in mods:
            try:
                mods[modname] = importlib.import_module(f"{reference_package}.{modname}")
            except ImportError:
                if modname == "circuit_sat_cb":       # needs more of MPyC than the path itself; optional
                    mods[modname] = None
                else:
                    raise
        mod = mods[modname]
        if mod is None:
            continue
        current = getattr(mod, name, None)
        if getattr(current, "__vmpc_accelerated__", None) is not ours:
            setattr(mod, name, _dispatcher(ours, current, is_ours))
        patched.append(f"{mod.__name__}.{name}")
    for ours_mod, modname in ((pivot, "pivot"), (compressed_pivot, "compressed_pivot"),
                              (circuit_sat, "circuit_sat_r1cs")):
        if hasattr(mods.get(modname), "prng"):
            ours_mod.prng = _ReferencePrng(mods[modname])
    return patched


def uninstall(reference_package="verifiable_mpc.ac20"):
    """Put the reference's own functions back (tests; a process that wants to compare the two)."""
    import sys
    from random import SystemRandom
    restored = []
    for modname, name, ours, _ in _SEAM:
        mod = sys.modules.get(f"{reference_package}.{modname}")
        current = getattr(mod, name, None)
        if getattr(current, "__vmpc_accelerated__", None) is ours:
            original = current.__vmpc_original__
            if original is None:
                delattr(mod, name)
            else:
                setattr(mod, name, original)
            restored.append(f"{mod.__name__}.{name}")
    for ours_mod in (pivot, compressed_pivot, circuit_sat):
        if isinstance(ours_mod.prng, _ReferencePrng):
            ours_mod.prng = SystemRandom()
    return restored
