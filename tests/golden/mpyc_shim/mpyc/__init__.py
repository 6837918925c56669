"""Build-written minimal stand-in for the `mpyc` package (NOT MPyC source).

Purpose: let tests/golden/make_fixtures.py import the reference's own
verifiable_mpc.ac20 modules IN THE BUILD CONTAINER ONLY and run them on seeded
inputs, so that the committed fixtures pin the reference's protocol logic.  Only the
names the hot path touches exist.  The group/field arithmetic
is the shim's OWN statement of the [mpyc-recall] formulas (it imports nothing from oracle/), so the
fixtures are a second opinion that the oracle must match (tests/test_shim_independent.py);
formats marked [mpyc-recall] are recalled, not verified.  Never imported by the product, the GPU tests, smoke() or bench.py.
"""
__version__ = "0.0-shim"
