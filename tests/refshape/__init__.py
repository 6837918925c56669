"""A builder-written stand-in for an importable copy of the reference (test infrastructure only).

The reference (toonsegers/verifiable_mpc) never travels to the GPU box, and its circuit front end needs more of MPyC than
tests/golden/mpyc_shim provides there.  What a drop-in has to survive, though, is not the reference's code but its
CALL PATTERN: module-level functions reached through the module object, a PivotChoice enum that is not this package's,
field elements and curve points that are MPyC's (here: the shim's), a form class that is the caller's own.  The modules
under refshape/ac20 have the reference's module and function names and follow its calling conventions
(circuit_sat_cb.py:255-318, demos/demo_zkp_ac20.py:69-90), with generic-group CPU code behind them that serves as
"the reference's own function" for the groups the GPU path leaves alone.  `verifiable_mpc_amd.install("tests.refshape.ac20")`
then does to them what it does to the real thing.
"""
