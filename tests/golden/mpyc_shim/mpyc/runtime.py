"""Shim of mpyc.runtime: `logging` and a SINGLE-PARTY `mpc` (m = 1, threshold 0, no sockets) with the
few members verifiable_mpc/ac20/mpc_ac20.py touches.  Randomness comes from `mpc._rng` so that the
fixture writer can seed it."""
import asyncio
import logging  # noqa: F401  (pivot.py:16 does `from mpyc.runtime import logging`)
import random


class _Mpc:
    def __init__(self):
        self._rng = random.SystemRandom()
        self.parties = [object()]
        self.pid = 0
        self.threshold = 0

    def if_else(self, c, x, y):
        return x if c else y

    def SecFld(self, *a, **k):
        from .sectypes import SecFld
        return SecFld(*a, **k)

    def _random(self, sectype, bound=None):
        return sectype(sectype.field(self._rng.randrange(sectype.field.order)))

    async def output(self, x, receivers=None, threshold=None, raw=False):
        from .sectypes import SecureObject
        one = lambda v: v.share if isinstance(v, SecureObject) else v
        return [one(v) for v in x] if isinstance(x, (list, tuple)) else one(x)

    async def gather(self, *aws):
        if len(aws) == 1 and isinstance(aws[0], (list, tuple)):
            return [await a for a in aws[0]]
        return [await a for a in aws]

    async def start(self):
        pass

    async def shutdown(self):
        pass

    def run(self, coro):
        return asyncio.get_event_loop_policy().new_event_loop().run_until_complete(coro)

    def __getattr__(self, name):
        raise NotImplementedError(f"mpyc.runtime.mpc.{name} is outside the shim's scope")


mpc = _Mpc()
