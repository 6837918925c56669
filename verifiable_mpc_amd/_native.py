"""ctypes binding of libvmpc_hip.so (include/vmpc.h).

There is deliberately NO CPU fallback: if the HIP library is missing or no GPU is
visible, every entry point raises.  The Python modules above this file (pivot.py,
compressed_pivot.py, ...) mirror the reference's call signatures and send all O(N)
group / vector work through here.
"""
import ctypes
import time
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VMPC_LIB_PATH") or os.path.join(_HERE, "libvmpc_hip.so")   # override: developer A/B builds

OK = 0
E_INVAL, E_NONCANON, E_NOTONCURVE, E_NOMEM, E_HIP, E_NODEV, E_AGAIN = -22, -34, -33, -12, -5, -19, -11
_ERR_NAMES = {E_AGAIN: "VMPC_E_AGAIN", E_INVAL: "VMPC_E_INVAL", E_NONCANON: "VMPC_E_NONCANON",
              E_NOTONCURVE: "VMPC_E_NOTONCURVE", E_NOMEM: "VMPC_E_NOMEM", E_HIP: "VMPC_E_HIP",
              E_NODEV: "VMPC_E_NODEV"}

SCALAR_BYTES, AFFINE_BYTES, PROJ_BYTES, EXT_BYTES = 32, 64, 96, 128

# every symbol include/vmpc.h declares (tests/test_cabi_symbols.py checks the list
# against the header and against the built library)
SYMBOLS = [
    "vmpc_backend_info", "vmpc_last_error", "vmpc_ctx_create", "vmpc_ctx_destroy",
    "vmpc_ctx_set_stream", "vmpc_ctx_sync", "vmpc_ctx_set_short_path", "vmpc_ctx_get_short_path", "vmpc_ctx_debug_hold_wait", "vmpc_ctx_query", "vmpc_ctx_wait_for", "vmpc_malloc", "vmpc_free", "vmpc_memcpy_h2d",
    "vmpc_memcpy_d2h", "vmpc_memcpy_d2d", "vmpc_ctx_profile", "vmpc_ctx_profile_read",
    "vmpc_ctx_set_window", "vmpc_ed25519_msm_plan", "vmpc_ed25519_madd_rate", "vmpc_ed25519_msm", "vmpc_ed25519_fold",
    "vmpc_ed25519_fixed_base_batch", "vmpc_fr_axpy", "vmpc_fr_dot", "vmpc_points_validate_dev",
    "vmpc_msm_dev", "vmpc_msm_table_bytes", "vmpc_msm_table_build_dev", "vmpc_msm_table_dev", "vmpc_msm_table_batch_dev", "vmpc_points_sum_dev", "vmpc_points_sum_many_dev", "vmpc_fixed_base_dev", "vmpc_repeat_dev", "vmpc_fold_dev",
    "vmpc_tree_reduce_dev", "vmpc_normalize_dev", "vmpc_affine_to_proj_dev", "vmpc_fr_axpy_dev",
    "vmpc_fr_scale_dev", "vmpc_fr_axpy_tail_dev", "vmpc_fr_dot_dev", "vmpc_fr_dot_to_dev", "vmpc_format_points_dev", "vmpc_format_scalars_dev",
    "vmpc_format_points_chunked_dev", "vmpc_format_scalars_chunked_dev", "vmpc_ed25519_fold_commitment_host", "vmpc_ed25519_lincomb_host",
    "vmpc_format_points_async_dev", "vmpc_format_scalars_async_dev", "vmpc_host_alloc", "vmpc_host_free",
    "vmpc_sha256_chunks_dev", "vmpc_fr_challenge_products_dev", "vmpc_fr_tail_scalars_dev", "vmpc_fr_tail_scalars_inc_dev", "vmpc_fr_tail_scalars_block_dev",
    "vmpc_bn256_g1_msm", "vmpc_bn256_g2_msm", "vmpc_bn256_g1_msm_dev", "vmpc_bn256_g2_msm_dev",
    "vmpc_bn256_validate_dev", "vmpc_bn256_fixed_base_dev",
    "vmpc_msm_table_fold_dev", "vmpc_msm_table_fold_table_dev", "vmpc_p4_create", "vmpc_p4_create_opts", "vmpc_p4_prefold", "vmpc_p4_set_commit_table", "vmpc_p4_round", "vmpc_p4_round_begin", "vmpc_p4_round_end", "vmpc_p4_finish", "vmpc_p4_run_compact", "vmpc_p4_destroy", "vmpc_bn256_table_bytes", "vmpc_bn256_table_build_dev", "vmpc_bn256_table_msm_dev", "vmpc_bn256_table_msm_multi_dev",
    "vmpc_comm_unique_id", "vmpc_comm_create_rccl", "vmpc_comm_create_callback", "vmpc_comm_destroy", "vmpc_comm_info",
    "vmpc_comm_allgather_dev", "vmpc_comm_points_allsum_dev", "vmpc_p4_create_sharded", "vmpc_gather_probe_dev", "vmpc_bn256_madd_rate",
    "vmpc_stream_create", "vmpc_stream_destroy", "vmpc_ctx_set_bucket_stream",
    "vmpc_set_reference_format", "vmpc_get_reference_format",
]


class VmpcError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {_ERR_NAMES.get(code, code)} {detail}".strip())


_lib = None
# vmpc_exchange_fn (include/vmpc.h): int fn(void *user, const void *mine, void *gathered, size_t bytes_per_rank)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


def load_library():
    """dlopen the in-tree HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m verifiable_mpc_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the AC20 hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, i32, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)
    cp = ctypes.c_char_p
    sig = {
        "vmpc_backend_info": (i32, [cp, sz]),
        "vmpc_last_error": (cp, []),
        "vmpc_ctx_create": (i32, [i32, ctypes.POINTER(vp)]),
        "vmpc_ctx_destroy": (i32, [vp]),
        "vmpc_ctx_set_stream": (i32, [vp, vp]),
        "vmpc_ctx_sync": (i32, [vp]),
        "vmpc_ctx_set_short_path": (i32, [vp, i32]),
        "vmpc_ctx_get_short_path": (i32, [vp, ctypes.POINTER(ctypes.c_int)]),
        "vmpc_ctx_debug_hold_wait": (i32, [vp, i32]),
        "vmpc_ctx_query": (i32, [vp, ctypes.POINTER(i32)]),
        "vmpc_ctx_wait_for": (i32, [vp, vp]),
        "vmpc_malloc": (i32, [vp, sz, ctypes.POINTER(vp)]),
        "vmpc_free": (i32, [vp, vp]),
        "vmpc_memcpy_h2d": (i32, [vp, vp, vp, sz]),
        "vmpc_memcpy_d2h": (i32, [vp, vp, vp, sz]),
        "vmpc_memcpy_d2d": (i32, [vp, vp, vp, sz]),
        "vmpc_ctx_profile": (i32, [vp, i32]),
        "vmpc_ctx_profile_read": (i32, [vp, cp, sz, ctypes.POINTER(ctypes.c_double), u64p, i32, i32]),
        "vmpc_ctx_set_window": (i32, [vp, i32]),
        "vmpc_set_reference_format": (i32, [ctypes.c_char, ctypes.c_char, i32]),
        "vmpc_get_reference_format": (i32, [cp, cp, ctypes.POINTER(i32)]),
        "vmpc_stream_create": (i32, [i32, i32, ctypes.POINTER(vp)]),
        "vmpc_stream_destroy": (i32, [vp]),
        "vmpc_ctx_set_bucket_stream": (i32, [vp, vp, i32]),
        "vmpc_ed25519_msm_plan": (i32, [vp, sz, vp, vp]),
        "vmpc_ed25519_madd_rate": (i32, [vp, i32, vp]),
        "vmpc_ed25519_msm": (i32, [vp, vp, sz, vp]),
        "vmpc_ed25519_fold": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_ed25519_fixed_base_batch": (i32, [vp, vp, sz, vp]),
        "vmpc_fr_axpy": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_fr_dot": (i32, [vp, vp, sz, vp]),
        "vmpc_points_validate_dev": (i32, [vp, vp, sz, u64p]),
        "vmpc_msm_dev": (i32, [vp, vp, vp, sz, vp, vp, sz, vp, vp]),
        "vmpc_msm_table_bytes": (i32, [sz, sz, i32, vp]),
        "vmpc_msm_table_build_dev": (i32, [vp, vp, sz, vp, sz, i32, vp]),
        "vmpc_msm_table_dev": (i32, [vp, vp, sz, sz, i32, vp, sz, vp, vp, vp]),
        "vmpc_msm_table_batch_dev": (i32, [vp, vp, sz, sz, i32, vp, sz, vp, i32, vp, vp]),
        "vmpc_points_sum_dev": (i32, [vp, vp, sz, vp, vp]),
        "vmpc_points_sum_many_dev": (i32, [vp, vp, sz, sz, vp, vp]),
        "vmpc_repeat_dev": (i32, [vp, vp, sz, i32, vp, sz, i32, vp, vp]),
        "vmpc_fixed_base_dev": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_fold_dev": (i32, [vp, vp, vp, i32, vp, sz, vp, vp]),
        "vmpc_tree_reduce_dev": (i32, [vp, vp, sz, i32, vp]),
        "vmpc_normalize_dev": (i32, [vp, vp, sz, vp]),
        "vmpc_affine_to_proj_dev": (i32, [vp, vp, sz, vp]),
        "vmpc_fr_axpy_dev": (i32, [vp, vp, vp, vp, sz, vp]),
        "vmpc_fr_scale_dev": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_fr_axpy_tail_dev": (i32, [vp, vp, vp, vp, sz, vp, vp]),
        "vmpc_fr_dot_dev": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_fr_dot_to_dev": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_format_points_dev": (i32, [vp, vp, sz, vp, sz, u64p]),
        "vmpc_format_scalars_dev": (i32, [vp, vp, sz, i32, vp, sz, u64p]),
        "vmpc_format_points_async_dev": (i32, [vp, vp, sz, vp, sz, vp, vp]),
        "vmpc_format_scalars_async_dev": (i32, [vp, vp, sz, i32, vp, sz, vp, vp]),
        "vmpc_ed25519_fold_commitment_host": (i32, [vp, vp, vp, vp, vp]),
        "vmpc_ed25519_lincomb_host": (i32, [vp, vp, sz, vp]),
        "vmpc_format_points_chunked_dev": (i32, [vp, vp, sz, vp, sz, vp, vp, sz]),
        "vmpc_format_scalars_chunked_dev": (i32, [vp, vp, sz, i32, vp, sz, vp, vp, sz]),
        "vmpc_host_alloc": (i32, [sz, ctypes.POINTER(vp)]),
        "vmpc_host_free": (i32, [vp]),
        "vmpc_sha256_chunks_dev": (i32, [vp, vp, sz, sz, vp]),
        "vmpc_fr_challenge_products_dev": (i32, [vp, vp, i32, i32, vp, sz, vp]),
        "vmpc_fr_tail_scalars_dev": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "vmpc_fr_tail_scalars_inc_dev": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
        "vmpc_fr_tail_scalars_block_dev": (i32, [vp, vp, i32, i32, vp, sz, sz, vp, vp, vp]),
        "vmpc_bn256_g1_msm": (i32, [vp, vp, sz, vp]),
        "vmpc_bn256_g2_msm": (i32, [vp, vp, sz, vp]),
        "vmpc_bn256_g1_msm_dev": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_bn256_g2_msm_dev": (i32, [vp, vp, vp, sz, vp]),
        "vmpc_bn256_validate_dev": (i32, [vp, i32, vp, sz, u64p]),
        "vmpc_bn256_fixed_base_dev": (i32, [vp, i32, vp, vp, sz, vp]),
        "vmpc_msm_table_fold_dev": (i32, [vp, vp, sz, sz, i32, sz, i32, vp, vp]),
        "vmpc_msm_table_fold_table_dev": (i32, [vp, vp, sz, sz, i32, sz, i32, vp, vp, sz, i32, vp]),
        "vmpc_p4_create": (i32, [vp, vp, sz, sz, i32, i32, i32, vp, vp, vp, ctypes.POINTER(vp)]),
        "vmpc_p4_set_commit_table": (i32, [vp, vp, i32]),
        "vmpc_p4_create_opts": (i32, [vp, vp, sz, sz, i32, i32, i32, vp, vp, vp, i32, i32, ctypes.POINTER(vp)]),
        "vmpc_p4_prefold": (i32, [vp]),
        "vmpc_p4_round": (i32, [vp, vp, vp, vp]),
        "vmpc_p4_round_begin": (i32, [vp, vp]),
        "vmpc_p4_round_end": (i32, [vp, vp, vp]),
        "vmpc_p4_finish": (i32, [vp, vp, vp]),
        "vmpc_p4_run_compact": (i32, [vp, vp, i32, vp, vp]),
        "vmpc_p4_destroy": (i32, [vp]),
        "vmpc_bn256_table_bytes": (i32, [i32, sz, vp]),
        "vmpc_bn256_table_build_dev": (i32, [vp, i32, vp, sz, vp]),
        "vmpc_bn256_table_msm_dev": (i32, [vp, i32, vp, sz, vp, sz, vp, vp]),
        "vmpc_bn256_table_msm_multi_dev": (i32, [vp, i32, vp, i32, sz, vp, sz, vp]),
        "vmpc_comm_unique_id": (i32, [vp]),
        "vmpc_comm_create_rccl": (i32, [vp, vp, i32, i32, ctypes.POINTER(vp)]),
        "vmpc_comm_create_callback": (i32, [i32, i32, EXCHANGE_FN, vp, ctypes.POINTER(vp)]),
        "vmpc_comm_destroy": (i32, [vp]),
        "vmpc_comm_info": (i32, [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]),
        "vmpc_comm_allgather_dev": (i32, [vp, vp, vp, vp, sz]),
        "vmpc_comm_points_allsum_dev": (i32, [vp, vp, vp, sz, vp, vp, vp]),
        "vmpc_p4_create_sharded": (i32, [vp, vp, vp, sz, sz, i32, i32, vp, vp, vp, ctypes.POINTER(vp)]),
        "vmpc_bn256_madd_rate": (i32, [vp, i32, i32, vp]),
        "vmpc_gather_probe_dev": (i32, [vp, vp, sz, sz, i32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]),
    }
    for name in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype, fn.argtypes = sig[name]
    _lib = lib
    # a reference format chosen before the library was first needed (formats.set_reference_format) takes effect now
    from . import formats
    f = formats.get_reference_format()
    if f != formats._DEFAULT:
        set_reference_format(f["point_brackets"][0], f["point_brackets"][1], f["coord_signed"])
    return lib


def library_loaded():
    return _lib is not None


def _check(rc, where):
    if rc != OK:
        detail = load_library().vmpc_last_error().decode(errors="replace") if rc in (E_HIP, E_NONCANON, E_NOMEM, E_NODEV, E_AGAIN) else ""
        raise VmpcError(rc, where, detail)


def backend_info():
    lib = load_library()
    buf = ctypes.create_string_buffer(256)
    n = lib.vmpc_backend_info(buf, 256)
    return n, buf.value.decode()


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def as_bytes_array(a, width):
    """C-contiguous uint8 array of shape (n, width)."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 1:
        a = a.reshape(-1, width)
    assert a.ndim == 2 and a.shape[1] == width, (a.shape, width)
    return a


def scalar_to_bytes(v):
    return int(v).to_bytes(32, "little")


def ints_to_array(vals, width=32):
    """list of non-negative ints -> (n, width) uint8 little-endian."""
    return np.frombuffer(b"".join(int(v).to_bytes(width, "little") for v in vals),
                         dtype=np.uint8).reshape(-1, width).copy() if len(vals) else \
        np.zeros((0, width), dtype=np.uint8)


def array_to_ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    w = a.shape[-1]
    raw = a.tobytes()
    return [int.from_bytes(raw[i:i + w], "little") for i in range(0, len(raw), w)]


def _size_class(nbytes):
    """power-of-two size classes (>= 256 B) so that the halving rounds reuse blocks"""
    n = max(int(nbytes), 256)
    return 1 << (n - 1).bit_length()


class DeviceBuffer:
    """Owned device allocation - returned to the context's block cache with the object.

    hipMalloc/hipFree cost 0.1-1 ms and hipFree synchronises the device; a Protocol-5 proof
    allocates a few vectors per round, so blocks are recycled per size class.  Reuse is safe
    in stream order: a block handed out again is only touched by work enqueued later on the
    same context's stream."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.cap = _size_class(nbytes)
        self.ptr = ctx._take_block(self.cap)

    def at(self, byte_offset):
        return ctypes.c_void_p(self.ptr + int(byte_offset))

    def free(self):
        if self.ptr and self.ctx.handle:
            self.ctx._give_block(self.cap, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory (vmpc_host_alloc) viewed as a uint8 numpy array."""

    def __init__(self, nbytes):
        lib = load_library()
        p = ctypes.c_void_p()
        _check(lib.vmpc_host_alloc(int(nbytes), ctypes.byref(p)), "vmpc_host_alloc")
        self.ptr, self.nbytes, self._lib = p.value, int(nbytes), lib
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * self.nbytes).from_address(self.ptr))

    def free(self):
        if self.ptr:
            self.array = None
            self._lib.vmpc_host_free(ctypes.c_void_p(self.ptr))
            self.ptr = None


class PendingText:
    """Transcript text being formatted and copied on a context's stream (PointVector /
    ScalarVector .text_begin()); result() synchronises that stream."""

    def __init__(self, ctx, pinned, dev, cap, keepalive=None, chunk_bytes=0):
        self.ctx, self.pinned, self.dev, self.cap = ctx, pinned, dev, cap
        self.chunk_bytes = chunk_bytes
        self.n_chunks = -(-(cap - 16) // chunk_bytes) if chunk_bytes else 0      # the library copies cap - 16 bytes
        # the SOURCE vector's device block belongs to another context's block cache: hold it until this
        # stream has been synchronised, so that it cannot be recycled under the formatter's reads
        self.keepalive = keepalive

    def result(self):
        if self.dev is not None:
            self.ctx.sync()
            n = int(np.frombuffer(self.pinned.array[:8], dtype=np.uint64)[0])
            self._view = self.pinned.array[16:16 + n]
            self.dev = None
            self.keepalive = None
        return self._view

    def chunks(self, trim=0):
        """the text minus its last `trim` bytes as consecutive memory views, each handed out as soon as its piece
        has landed on the host (vmpc_format_*_chunked_dev): the caller hashes one while the next is on the link"""
        if self.dev is None or not self.chunk_bytes:
            view = self.result()
            yield view[:len(view) - trim] if trim else view
            return
        words = np.frombuffer(self.pinned.array[:16], dtype=np.uint32)      # length (2 words), pieces landed
        total, k, step = None, 0, self.chunk_bytes
        deadline = time.monotonic() + 120.0
        while True:
            spins = 0
            while int(words[2]) <= k:
                spins += 1
                if spins & 0xfff == 0 and time.monotonic() > deadline:
                    self.ctx.sync()                  # reports the stream's error, if that is why nothing lands
                    raise VmpcError(E_HIP, "transcript text did not arrive")
            if total is None:
                total = int(np.frombuffer(self.pinned.array[:8], dtype=np.uint64)[0]) - trim
            lo, hi = k * step, min((k + 1) * step, total)
            if lo >= total:
                return
            yield self.pinned.array[16 + lo:16 + hi]
            if hi >= total:
                return
            k += 1

    def _landed(self):
        """every byte this text's launches write has arrived (nothing of it is still queued on the stream)"""
        if self.dev is None:
            return True                               # result() synchronised the stream behind the copy
        if self.chunk_bytes:
            return int(np.frombuffer(self.pinned.array[:16], dtype=np.uint32)[2]) >= self.n_chunks
        return False

    def __del__(self):
        # the pinned block goes back to the pool only when nobody can still read the text - and only when the copies
        # into it are over.  Synchronising the stream for that used to cost 30 ms per reference-transcript proof: a
        # round's text dies when the next round's vector replaces it, i.e. right after the NEXT text's launches were
        # queued on the same stream, and the wait covered those too.
        try:
            if self.pinned is not None and self.ctx is not None and self.ctx.handle:
                if not self._landed():
                    self.ctx.sync()
                self._view = None
                self.keepalive = None
                self.ctx._give_pinned(self.cap + 16, self.pinned)
                self.pinned = None
        except Exception:
            pass


class TextSequence:
    """Several PendingTexts that are one text (a vector formatted slice by slice, PointVector.fold)."""

    def __init__(self, parts):
        self.parts = parts

    def result(self):
        views = [p.result() for p in self.parts]
        return views[0] if len(views) == 1 else np.concatenate(views)

    def chunks(self, trim=0):
        for i, p in enumerate(self.parts):
            yield from p.chunks(trim if i == len(self.parts) - 1 else 0)


class Context:
    """One GPU, one stream (vmpc_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        _check(self.lib.vmpc_ctx_create(device, ctypes.byref(h)), "vmpc_ctx_create")
        self.handle = h
        self.device = device
        self._cache = {}          # size class -> [device pointers]
        self._cached_bytes = 0
        self.cache_limit = 16 << 30
        self._pinned = {}         # size class -> [PinnedBuffer]

    def _take_pinned(self, nbytes):
        cap = _size_class(nbytes)
        lst = self._pinned.get(cap)
        return lst.pop() if lst else PinnedBuffer(cap)

    def _give_pinned(self, nbytes, buf):
        # NB: the array handed to the caller stays valid until the buffer is taken again, i.e.
        # until the NEXT text_begin of the same size class on this context
        self._pinned.setdefault(_size_class(nbytes), []).append(buf)

    TEXT_CHUNK_BYTES = 8 << 20      # pieces of the transcript text on their way to the host (PendingText.chunks)

    def format_begin(self, kind, src_ptr, n, is_signed=True, keepalive=None, own_signal=False):
        """enqueue formatting + D2H of a vector's transcript text; returns a PendingText.
        `keepalive`: the owner of `src_ptr` when it lives in another context's block cache.
        `own_signal`: deliver in pieces with a landed count whatever the size, so that a reader of THIS text waits
        for this text only - not, through a stream synchronisation, for everything queued behind it (the slices of
        PointVector.fold(stream_text=True) share one stream)"""
        per = (3 * 79 + 8) if kind == "points" else (78 + 3)       # 78 digits + a sign per coordinate
        cap = n * per + 16
        pinned = self._take_pinned(cap + 16)
        dev = DeviceBuffer(self, cap)
        host_len = ctypes.c_void_p(pinned.ptr)
        host_text = ctypes.c_void_p(pinned.ptr + 16)
        chunk = self.TEXT_CHUNK_BYTES if (own_signal or cap > 2 * self.TEXT_CHUNK_BYTES) else 0
        if chunk:
            pinned.array[:16] = 0                                   # length, pieces landed
            if kind == "points":
                rc = self.lib.vmpc_format_points_chunked_dev(self.handle, ctypes.c_void_p(src_ptr), n,
                                                             ctypes.c_void_p(dev.ptr), cap, host_text, host_len, chunk)
            else:
                rc = self.lib.vmpc_format_scalars_chunked_dev(self.handle, ctypes.c_void_p(src_ptr), n,
                                                              1 if is_signed else 0, ctypes.c_void_p(dev.ptr),
                                                              cap, host_text, host_len, chunk)
        elif kind == "points":
            rc = self.lib.vmpc_format_points_async_dev(self.handle, ctypes.c_void_p(src_ptr), n,
                                                       ctypes.c_void_p(dev.ptr), cap, host_text, host_len)
        else:
            rc = self.lib.vmpc_format_scalars_async_dev(self.handle, ctypes.c_void_p(src_ptr), n,
                                                        1 if is_signed else 0, ctypes.c_void_p(dev.ptr),
                                                        cap, host_text, host_len)
        _check(rc, "vmpc_format_async")
        return PendingText(self, pinned, dev, cap, keepalive, chunk)

    def _take_block(self, cap):
        lst = self._cache.get(cap)
        if lst:
            self._cached_bytes -= cap
            return lst.pop()
        p = ctypes.c_void_p()
        rc = self.lib.vmpc_malloc(self.handle, cap, ctypes.byref(p))
        if rc == E_NOMEM and self._cached_bytes:
            self.trim()
            rc = self.lib.vmpc_malloc(self.handle, cap, ctypes.byref(p))
        _check(rc, "vmpc_malloc")
        return p.value

    def _give_block(self, cap, ptr):
        if self._cached_bytes + cap > self.cache_limit:
            self.lib.vmpc_free(self.handle, ctypes.c_void_p(ptr))
            return
        self._cache.setdefault(cap, []).append(ptr)
        self._cached_bytes += cap

    def trim(self):
        """release every cached block back to the HIP allocator"""
        for lst in self._cache.values():
            for ptr in lst:
                self.lib.vmpc_free(self.handle, ctypes.c_void_p(ptr))
        self._cache = {}
        self._cached_bytes = 0

    def close(self):
        if self.handle:
            for lst in self._pinned.values():
                for b in lst:
                    b.free()
            self._pinned = {}
            self.trim()
            # VMPC_E_INVAL while a round context (vmpc_p4) of this context is alive: keep the handle, so that the
            # stream, workspace, event pool and arena are released by a later close() instead of leaking
            _check(self.lib.vmpc_ctx_destroy(self.handle), "vmpc_ctx_destroy")
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory -------------------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, max(arr.nbytes, 1))
        if arr.nbytes:
            _check(self.lib.vmpc_memcpy_h2d(self.handle, ctypes.c_void_p(buf.ptr), _np_ptr(arr),
                                            arr.nbytes), "vmpc_memcpy_h2d")
        return buf

    def upload_into(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            _check(self.lib.vmpc_memcpy_h2d(self.handle, ctypes.c_void_p(ptr), _np_ptr(arr),
                                            arr.nbytes), "vmpc_memcpy_h2d")

    def download(self, ptr, nbytes, shape=None):
        out = np.empty(int(nbytes), dtype=np.uint8)
        if nbytes:
            _check(self.lib.vmpc_memcpy_d2h(self.handle, _np_ptr(out), ctypes.c_void_p(ptr),
                                            int(nbytes)), "vmpc_memcpy_d2h")
        return out.reshape(shape) if shape is not None else out

    def copy(self, dst_ptr, src_ptr, nbytes):
        _check(self.lib.vmpc_memcpy_d2d(self.handle, ctypes.c_void_p(dst_ptr),
                                        ctypes.c_void_p(src_ptr), int(nbytes)), "vmpc_memcpy_d2d")

    def sync(self):
        _check(self.lib.vmpc_ctx_sync(self.handle), "vmpc_ctx_sync")

    def set_short_path(self, on, forget_overflow=False):
        """commitments over short 16-row tables: the fused three-launch path (csrc/msm_short.hip) on / off;
        forget_overflow: also end the back-off (64 eligible calls on the general path) that follows an overflow"""
        _check(self.lib.vmpc_ctx_set_short_path(self.handle, (2 if forget_overflow else 1) if on else 0),
               "vmpc_ctx_set_short_path")

    def get_short_path(self):
        on = ctypes.c_int(0)
        _check(self.lib.vmpc_ctx_get_short_path(self.handle, ctypes.byref(on)), "vmpc_ctx_get_short_path")
        return bool(on.value)

    def on_general_path(self, fn):
        """fn() with the short path off, the PREVIOUS setting restored afterwards (a user who switched the path off
        keeps it off): what a caller does after sync() raised VMPC_E_AGAIN (a commitment's scalars were skewed beyond
        the short path's fixed capacities; its result is void), and what callers do whose consumer cannot act on
        that answer (a collective that has already summed the partial)"""
        prev = self.get_short_path()
        if prev:
            self.set_short_path(False)
        try:
            return fn()
        finally:
            if prev:
                self.set_short_path(True)

    def done(self):
        """True when all work enqueued on this context's stream has completed (does not block)"""
        d = ctypes.c_int(0)
        _check(self.lib.vmpc_ctx_query(self.handle, ctypes.byref(d)), "vmpc_ctx_query")
        return bool(d.value)

    def wait_for(self, other):
        """device-side ordering: this context's stream waits for `other`'s work so far"""
        _check(self.lib.vmpc_ctx_wait_for(self.handle, other.handle), "vmpc_ctx_wait_for")

    def set_stream(self, stream_ptr):
        _check(self.lib.vmpc_ctx_set_stream(self.handle, ctypes.c_void_p(stream_ptr)),
               "vmpc_ctx_set_stream")

    def set_bucket_stream(self, stream, wgs_per_cu=0):
        """phase pipelining: bucket stages of this context run on the shared `stream` (a SharedStream or None)"""
        _check(self.lib.vmpc_ctx_set_bucket_stream(self.handle, stream.handle if stream is not None else None,
                                                   int(wgs_per_cu)), "vmpc_ctx_set_bucket_stream")
        self._bucket_stream = stream        # keep it alive as long as this context points at it

    def set_window(self, c_bits):
        _check(self.lib.vmpc_ctx_set_window(self.handle, int(c_bits)), "vmpc_ctx_set_window")

    def msm_plan(self, n):
        """(window width c, window count W) the planner picks for an n-term Ed25519 MSM"""
        c, w = ctypes.c_int(0), ctypes.c_int(0)
        _check(self.lib.vmpc_ed25519_msm_plan(self.handle, int(n), ctypes.byref(c), ctypes.byref(w)),
               "vmpc_ed25519_msm_plan")
        return c.value, w.value

    def madd_rate(self, iters=400):
        """mixed additions per second of the register-resident ALU ceiling probe"""
        r = ctypes.c_double(0.0)
        _check(self.lib.vmpc_ed25519_madd_rate(self.handle, int(iters), ctypes.byref(r)),
               "vmpc_ed25519_madd_rate")
        return r.value

    def bn256_madd_rate(self, group, iters=200):
        """Jacobian mixed additions per second (group 1: F_p, 2: F_p^2) of the register-resident probe"""
        r = ctypes.c_double(0.0)
        _check(self.lib.vmpc_bn256_madd_rate(self.handle, int(group), int(iters), ctypes.byref(r)),
               "vmpc_bn256_madd_rate")
        return r.value

    def profile(self, enable=True):
        _check(self.lib.vmpc_ctx_profile(self.handle, 1 if enable else 0), "vmpc_ctx_profile")

    def profile_read(self, reset=True):
        names = ctypes.create_string_buffer(2048)
        ms = (ctypes.c_double * 64)()
        cnt = (ctypes.c_uint64 * 64)()
        k = self.lib.vmpc_ctx_profile_read(self.handle, names, 2048, ms, cnt, 64, 1 if reset else 0)
        if k < 0:
            _check(k, "vmpc_ctx_profile_read")
        nm = names.value.decode().split(";") if k else []
        return {nm[i]: (ms[i], int(cnt[i])) for i in range(k)}

    # ---- device entry points (pointers are ints / c_void_p) --------------------------------
    def validate_points(self, affine_ptr, n):
        bad = ctypes.c_uint64()
        _check(self.lib.vmpc_points_validate_dev(self.handle, ctypes.c_void_p(affine_ptr), n,
                                                 ctypes.byref(bad)), "vmpc_points_validate_dev")
        return bad.value

    def msm(self, scalars_ptr, points_ptr, n, extra_scalars_ptr=None, extra_points_ptr=None,
            n_extra=0, out_ext_ptr=None, out_affine_ptr=None):
        _check(self.lib.vmpc_msm_dev(self.handle, ctypes.c_void_p(scalars_ptr),
                                     ctypes.c_void_p(points_ptr), n,
                                     ctypes.c_void_p(extra_scalars_ptr),
                                     ctypes.c_void_p(extra_points_ptr), n_extra,
                                     ctypes.c_void_p(out_ext_ptr), ctypes.c_void_p(out_affine_ptr)),
               "vmpc_msm_dev")

    def msm_table_build(self, points_ptr, n, extra_points_ptr=None, n_extra=0, rows=16):
        """Fixed-base table (DeviceBuffer) of `rows` rows over n points followed by n_extra extras."""
        nbytes = ctypes.c_size_t(0)
        _check(self.lib.vmpc_msm_table_bytes(n, n_extra, rows, ctypes.byref(nbytes)), "vmpc_msm_table_bytes")
        table = self.alloc(nbytes.value)
        _check(self.lib.vmpc_msm_table_build_dev(self.handle, ctypes.c_void_p(points_ptr), n,
                                                 ctypes.c_void_p(extra_points_ptr), n_extra, rows,
                                                 ctypes.c_void_p(table.ptr)), "vmpc_msm_table_build_dev")
        return table

    def msm_table(self, table_ptr, table_n, table_extra, scalars_ptr, m, extra_scalars_ptr=None,
                  out_ext_ptr=None, out_affine_ptr=None, rows=16):
        _check(self.lib.vmpc_msm_table_dev(self.handle, ctypes.c_void_p(table_ptr), table_n, table_extra, rows,
                                           ctypes.c_void_p(scalars_ptr), m,
                                           ctypes.c_void_p(extra_scalars_ptr),
                                           ctypes.c_void_p(out_ext_ptr), ctypes.c_void_p(out_affine_ptr)),
               "vmpc_msm_table_dev")

    def msm_table_batch(self, table_ptr, table_n, table_extra, scalar_ptrs, m, extra_scalar_ptrs=None,
                        out_ext_ptr=None, out_affine_ptr=None, rows=16):
        """len(scalar_ptrs) commitments over one table in one pass; outputs consecutive (128 / 64 bytes each)"""
        k = len(scalar_ptrs)
        sc = (ctypes.c_void_p * k)(*[ctypes.c_void_p(p) for p in scalar_ptrs])
        ex = None
        if extra_scalar_ptrs is not None:
            ex = (ctypes.c_void_p * k)(*[ctypes.c_void_p(p) for p in extra_scalar_ptrs])
        _check(self.lib.vmpc_msm_table_batch_dev(self.handle, ctypes.c_void_p(table_ptr), table_n, table_extra, rows,
                                                 sc, m, ex, k, ctypes.c_void_p(out_ext_ptr),
                                                 ctypes.c_void_p(out_affine_ptr)), "vmpc_msm_table_batch_dev")

    def msm_table_fold(self, table_ptr, table_n, table_extra, rows, n_cols, scalars, out_affine_ptr):
        """out[j] = sum_b scalars[b] * P[j + b * (n_cols / len(scalars))]: log2(len(scalars)) folds in one pass"""
        k = len(scalars).bit_length() - 1
        assert len(scalars) == 1 << k
        raw = ctypes.create_string_buffer(b"".join(scalar_to_bytes(v) for v in scalars), 32 << k)
        _check(self.lib.vmpc_msm_table_fold_dev(self.handle, ctypes.c_void_p(table_ptr), table_n, table_extra, rows,
                                                n_cols, k, raw, ctypes.c_void_p(out_affine_ptr)),
               "vmpc_msm_table_fold_dev")

    def msm_table_fold_table(self, table_ptr, table_n, table_extra, rows, n_cols, scalars, extras_ptr, n_extra, out_rows):
        """msm_table_fold, leaving the folded vector's fixed-base table (a DeviceBuffer) instead of the vector"""
        k = len(scalars).bit_length() - 1
        assert len(scalars) == 1 << k
        nbytes = ctypes.c_size_t()
        _check(self.lib.vmpc_msm_table_bytes(n_cols >> k, n_extra, out_rows, ctypes.byref(nbytes)), "vmpc_msm_table_bytes")
        out = DeviceBuffer(self, nbytes.value)
        raw = ctypes.create_string_buffer(b"".join(scalar_to_bytes(v) for v in scalars), 32 << k)
        _check(self.lib.vmpc_msm_table_fold_table_dev(self.handle, ctypes.c_void_p(table_ptr), table_n, table_extra, rows,
                                                      n_cols, k, raw, ctypes.c_void_p(extras_ptr), n_extra, out_rows,
                                                      ctypes.c_void_p(out.ptr)), "vmpc_msm_table_fold_table_dev")
        return out

    def points_sum(self, ext_ptr, m, out_ext_ptr=None, out_affine_ptr=None):
        _check(self.lib.vmpc_points_sum_dev(self.handle, ctypes.c_void_p(ext_ptr), m,
                                            ctypes.c_void_p(out_ext_ptr),
                                            ctypes.c_void_p(out_affine_ptr)), "vmpc_points_sum_dev")

    def gather_probe(self, table_ptr, table_lines, n_gathers, mode=0, seed=1, timed=True):
        """vmpc_gather_probe_dev: n_gathers 128-byte line reads in the bucket stage's pattern; returns ms"""
        ms = ctypes.c_double()
        _check(self.lib.vmpc_gather_probe_dev(self.handle, ctypes.c_void_p(table_ptr), table_lines, n_gathers, mode,
                                              seed, ctypes.byref(ms) if timed else None), "vmpc_gather_probe_dev")
        return ms.value

    def points_sum_many(self, ext_ptr, m, k, out_ext_ptr=None, out_affine_ptr=None):
        _check(self.lib.vmpc_points_sum_many_dev(self.handle, ctypes.c_void_p(ext_ptr), m, k,
                                                 ctypes.c_void_p(out_ext_ptr),
                                                 ctypes.c_void_p(out_affine_ptr)), "vmpc_points_sum_many_dev")

    def repeat(self, bases_ptr, n_bases, bases_affine, scalars_ptr, n, signed_scalars,
               out_proj_ptr=None, out_affine_ptr=None):
        _check(self.lib.vmpc_repeat_dev(self.handle, ctypes.c_void_p(bases_ptr), n_bases,
                                        1 if bases_affine else 0, ctypes.c_void_p(scalars_ptr), n,
                                        int(signed_scalars), ctypes.c_void_p(out_proj_ptr),
                                        ctypes.c_void_p(out_affine_ptr)), "vmpc_repeat_dev")

    def fixed_base(self, base_affine_ptr, scalars_ptr, n, out_affine_ptr):
        _check(self.lib.vmpc_fixed_base_dev(self.handle, ctypes.c_void_p(base_affine_ptr),
                                            ctypes.c_void_p(scalars_ptr), n, ctypes.c_void_p(out_affine_ptr)),
               "vmpc_fixed_base_dev")

    def fold(self, gl_ptr, gr_ptr, in_affine, c, half, out_proj_ptr=None, out_affine_ptr=None):
        cb = ctypes.create_string_buffer(scalar_to_bytes(c), 32)
        _check(self.lib.vmpc_fold_dev(self.handle, ctypes.c_void_p(gl_ptr), ctypes.c_void_p(gr_ptr),
                                      1 if in_affine else 0, cb, half, ctypes.c_void_p(out_proj_ptr),
                                      ctypes.c_void_p(out_affine_ptr)), "vmpc_fold_dev")

    def tree_reduce(self, proj_ptr, n, append_identity, out_proj_ptr):
        _check(self.lib.vmpc_tree_reduce_dev(self.handle, ctypes.c_void_p(proj_ptr), n,
                                             1 if append_identity else 0,
                                             ctypes.c_void_p(out_proj_ptr)), "vmpc_tree_reduce_dev")

    def normalize(self, proj_ptr, n, out_affine_ptr):
        _check(self.lib.vmpc_normalize_dev(self.handle, ctypes.c_void_p(proj_ptr), n,
                                           ctypes.c_void_p(out_affine_ptr)), "vmpc_normalize_dev")

    def affine_to_proj(self, affine_ptr, n, out_proj_ptr):
        _check(self.lib.vmpc_affine_to_proj_dev(self.handle, ctypes.c_void_p(affine_ptr), n,
                                                ctypes.c_void_p(out_proj_ptr)),
               "vmpc_affine_to_proj_dev")

    def fr_axpy(self, c, x_ptr, y_ptr, n, out_ptr):
        cb = ctypes.create_string_buffer(scalar_to_bytes(c), 32)
        _check(self.lib.vmpc_fr_axpy_dev(self.handle, cb, ctypes.c_void_p(x_ptr),
                                         ctypes.c_void_p(y_ptr), n, ctypes.c_void_p(out_ptr)),
               "vmpc_fr_axpy_dev")

    def fr_axpy_tail(self, c, x_ptr, y_ptr, n, tail, out_ptr):
        """out[0..n) = c * x + y (y_ptr None: c * x), out[n] = tail"""
        cb = ctypes.create_string_buffer(scalar_to_bytes(c), 32)
        tb = ctypes.create_string_buffer(scalar_to_bytes(tail), 32)
        _check(self.lib.vmpc_fr_axpy_tail_dev(self.handle, cb, ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), n, tb,
                                              ctypes.c_void_p(out_ptr)), "vmpc_fr_axpy_tail_dev")

    def fr_scale(self, c, x_ptr, n, out_ptr):
        cb = ctypes.create_string_buffer(scalar_to_bytes(c), 32)
        _check(self.lib.vmpc_fr_scale_dev(self.handle, cb, ctypes.c_void_p(x_ptr), n,
                                          ctypes.c_void_p(out_ptr)), "vmpc_fr_scale_dev")

    def fr_challenge_products(self, challenges, low_bits, z_ptr, out_ptr):
        rounds = len(challenges)
        buf = ctypes.create_string_buffer(b"".join(scalar_to_bytes(c) for c in challenges), 32 * max(rounds, 1))
        n = 1 << (rounds + low_bits)
        _check(self.lib.vmpc_fr_challenge_products_dev(self.handle, buf, rounds, low_bits,
                                                       ctypes.c_void_p(z_ptr), n,
                                                       ctypes.c_void_p(out_ptr)),
               "vmpc_fr_challenge_products_dev")

    def fr_tail_scalars(self, challenges, log2_m0, z_ptr, out_a_ptr, out_b_ptr):
        t = len(challenges)
        buf = ctypes.create_string_buffer(b"".join(scalar_to_bytes(c) for c in challenges), 32 * max(t, 1))
        _check(self.lib.vmpc_fr_tail_scalars_dev(self.handle, buf, t, log2_m0, ctypes.c_void_p(z_ptr),
                                                 ctypes.c_void_p(out_a_ptr), ctypes.c_void_p(out_b_ptr)),
               "vmpc_fr_tail_scalars_dev")

    def fr_tail_scalars_inc(self, newest_challenge, t, log2_m0, z_ptr, products_ptr, out_a_ptr, out_b_ptr):
        buf = ctypes.create_string_buffer(scalar_to_bytes(newest_challenge) if t else bytes(32), 32)
        _check(self.lib.vmpc_fr_tail_scalars_inc_dev(self.handle, buf, t, log2_m0, ctypes.c_void_p(z_ptr),
                                                     ctypes.c_void_p(products_ptr), ctypes.c_void_p(out_a_ptr),
                                                     ctypes.c_void_p(out_b_ptr)), "vmpc_fr_tail_scalars_inc_dev")

    def fr_tail_scalars_block(self, newest_challenge, t, log2_m0, z_ptr, j0, count, products_ptr, out_a_ptr,
                              out_b_ptr):
        buf = ctypes.create_string_buffer(scalar_to_bytes(newest_challenge) if t else bytes(32), 32)
        _check(self.lib.vmpc_fr_tail_scalars_block_dev(self.handle, buf, t, log2_m0, ctypes.c_void_p(z_ptr), j0, count,
                                                       ctypes.c_void_p(products_ptr), ctypes.c_void_p(out_a_ptr),
                                                       ctypes.c_void_p(out_b_ptr)), "vmpc_fr_tail_scalars_block_dev")

    def fr_dot(self, a_ptr, b_ptr, n):
        out = ctypes.create_string_buffer(32)
        _check(self.lib.vmpc_fr_dot_dev(self.handle, ctypes.c_void_p(a_ptr), ctypes.c_void_p(b_ptr),
                                        n, out), "vmpc_fr_dot_dev")
        return int.from_bytes(out.raw, "little")

    def fr_dot_to_dev(self, a_ptr, b_ptr, n):
        """inner product left on the device: a 32-byte DeviceBuffer, no synchronisation"""
        out = self.alloc(32)
        _check(self.lib.vmpc_fr_dot_to_dev(self.handle, ctypes.c_void_p(a_ptr), ctypes.c_void_p(b_ptr), n,
                                           ctypes.c_void_p(out.ptr)), "vmpc_fr_dot_to_dev")
        return out

    def _format(self, fn, name, src_ptr, n, per_item_cap, *extra):
        cap = n * per_item_cap + 16
        buf = DeviceBuffer(self, cap)
        ln = ctypes.c_uint64()
        _check(fn(self.handle, ctypes.c_void_p(src_ptr), n, *extra, ctypes.c_void_p(buf.ptr), cap,
                  ctypes.byref(ln)), name)
        out = self.download(buf.ptr, ln.value)
        buf.free()
        return out

    def bn256_msm(self, group, scalars_ptr, points_ptr, n, out_ptr):
        fn = self.lib.vmpc_bn256_g1_msm_dev if group == 1 else self.lib.vmpc_bn256_g2_msm_dev
        _check(fn(self.handle, ctypes.c_void_p(scalars_ptr), ctypes.c_void_p(points_ptr), n,
                  ctypes.c_void_p(out_ptr)), f"vmpc_bn256_g{group}_msm_dev")

    def bn256_table_build(self, group, points_ptr, n):
        nbytes = ctypes.c_size_t(0)
        _check(self.lib.vmpc_bn256_table_bytes(group, n, ctypes.byref(nbytes)), "vmpc_bn256_table_bytes")
        table = self.alloc(nbytes.value)
        _check(self.lib.vmpc_bn256_table_build_dev(self.handle, group, ctypes.c_void_p(points_ptr), n,
                                                   ctypes.c_void_p(table.ptr)), "vmpc_bn256_table_build_dev")
        return table

    def bn256_table_msm(self, group, table_ptr, table_n, scalars_ptr, m, out_ptr=None, out_jac_ptr=None):
        _check(self.lib.vmpc_bn256_table_msm_dev(self.handle, group, ctypes.c_void_p(table_ptr), table_n,
                                                 ctypes.c_void_p(scalars_ptr), m, ctypes.c_void_p(out_ptr),
                                                 ctypes.c_void_p(out_jac_ptr)),
               "vmpc_bn256_table_msm_dev")

    def bn256_table_msm_multi(self, group, table_ptrs, table_n, scalars_ptr, m, out_jac_ptr):
        """len(table_ptrs) prepared keys of the same length, one scalar vector: Jacobian sums, consecutive"""
        k = len(table_ptrs)
        tabs = (ctypes.c_void_p * k)(*[ctypes.c_void_p(p) for p in table_ptrs])
        _check(self.lib.vmpc_bn256_table_msm_multi_dev(self.handle, group, tabs, k, table_n,
                                                       ctypes.c_void_p(scalars_ptr), m, ctypes.c_void_p(out_jac_ptr)),
               "vmpc_bn256_table_msm_multi_dev")

    def bn256_fixed_base(self, group, base_ptr, scalars_ptr, n, out_ptr):
        _check(self.lib.vmpc_bn256_fixed_base_dev(self.handle, group, ctypes.c_void_p(base_ptr),
                                                  ctypes.c_void_p(scalars_ptr), n, ctypes.c_void_p(out_ptr)),
               "vmpc_bn256_fixed_base_dev")

    def bn256_validate(self, group, points_ptr, n):
        bad = ctypes.c_uint64()
        _check(self.lib.vmpc_bn256_validate_dev(self.handle, group, ctypes.c_void_p(points_ptr), n,
                                                ctypes.byref(bad)), "vmpc_bn256_validate_dev")
        return bad.value

    def sha256_chunks(self, data_ptr, nbytes, chunk_bytes=4096):
        """bytes object: concatenated 32-byte SHA-256 digests of the chunks of a device buffer"""
        n_chunks = (nbytes + chunk_bytes - 1) // chunk_bytes
        if n_chunks == 0:
            return b""
        out = DeviceBuffer(self, 32 * n_chunks)
        _check(self.lib.vmpc_sha256_chunks_dev(self.handle, ctypes.c_void_p(data_ptr), nbytes,
                                               chunk_bytes, ctypes.c_void_p(out.ptr)),
               "vmpc_sha256_chunks_dev")
        res = self.download(out.ptr, 32 * n_chunks).tobytes()
        out.free()
        return res

    def sha256_chunks_begin(self, data_ptr, nbytes, chunk_bytes=4096, keepalive=None):
        """enqueue the chunk digests on this context; .result() synchronises it and returns the bytes"""
        n_chunks = (nbytes + chunk_bytes - 1) // chunk_bytes
        out = DeviceBuffer(self, 32 * max(1, n_chunks))
        if n_chunks:
            _check(self.lib.vmpc_sha256_chunks_dev(self.handle, ctypes.c_void_p(data_ptr), nbytes,
                                                   chunk_bytes, ctypes.c_void_p(out.ptr)),
                   "vmpc_sha256_chunks_dev")
        return PendingDigests(self, out, n_chunks, keepalive)

    def format_points(self, proj_ptr, n):
        """uint8 array 'item0, item1, ..., ' for n projective points."""
        return self._format(self.lib.vmpc_format_points_dev, "vmpc_format_points_dev", proj_ptr, n,
                            3 * 79 + 8)

    def format_scalars(self, sc_ptr, n, is_signed=True):
        return self._format(self.lib.vmpc_format_scalars_dev, "vmpc_format_scalars_dev", sc_ptr, n,
                            78 + 3, 1 if is_signed else 0)


class PendingDigests:
    def __init__(self, ctx, out, n_chunks, keepalive):
        self.ctx, self.out, self.n_chunks, self.keepalive = ctx, out, n_chunks, keepalive

    def result(self):
        res = self.ctx.download(self.out.ptr, 32 * self.n_chunks).tobytes() if self.n_chunks else b""
        self.out.free()
        self.keepalive = None
        return res


def set_reference_format(point_open, point_close, coord_signed):
    """how csrc/format.hip prints a curve point (process-wide; formats.set_reference_format is the caller)"""
    _check(load_library().vmpc_set_reference_format(point_open.encode(), point_close.encode(), 1 if coord_signed else 0),
           "vmpc_set_reference_format")


def get_reference_format():
    o, c, s = ctypes.create_string_buffer(1), ctypes.create_string_buffer(1), ctypes.c_int()
    _check(load_library().vmpc_get_reference_format(o, c, ctypes.byref(s)), "vmpc_get_reference_format")
    return o.raw.decode(), c.raw.decode(), bool(s.value)


class SharedStream:
    """a HIP stream several contexts enqueue their bucket stages on (vmpc_stream_create; priority < 0: lowest)"""

    def __init__(self, device=0, priority=-1):
        self.lib = load_library()
        h = ctypes.c_void_p()
        _check(self.lib.vmpc_stream_create(int(device), int(priority), ctypes.byref(h)), "vmpc_stream_create")
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.vmpc_stream_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """vmpc_comm: the exchange step of the multi-GPU path (all-gather of partial points + rank-ordered add).

    Comm.rccl(ctx, unique_id, world, rank)      ncclAllGather on the context's stream (one process per GPU)
    Comm.callback(world, rank, fn)               fn(mine_ptr, gathered_ptr, bytes_per_rank) moves the bytes
    Comm.solo()                                  world = 1"""

    KINDS = {0: "self", 1: "rccl", 2: "callback"}

    def __init__(self, handle, world, rank, keep=None):
        self.lib = load_library()
        self.handle, self.world, self.rank, self._keep = handle, world, rank, keep

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _check(load_library().vmpc_comm_unique_id(buf), "vmpc_comm_unique_id")
        return buf.raw

    @classmethod
    def rccl(cls, ctx, unique_id, world, rank):
        h = ctypes.c_void_p()
        _check(ctx.lib.vmpc_comm_create_rccl(ctx.handle, ctypes.create_string_buffer(bytes(unique_id), 128), world, rank,
                                             ctypes.byref(h)), "vmpc_comm_create_rccl")
        return cls(h, world, rank)

    @classmethod
    def callback(cls, world, rank, fn):
        def trampoline(_user, mine, gathered, nbytes):
            try:
                fn(mine, gathered, nbytes)
                return 0
            except BaseException:       # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        cfn = EXCHANGE_FN(trampoline)
        h = ctypes.c_void_p()
        _check(load_library().vmpc_comm_create_callback(world, rank, cfn, None, ctypes.byref(h)),
               "vmpc_comm_create_callback")
        return cls(h, world, rank, keep=cfn)

    @classmethod
    def solo(cls):
        h = ctypes.c_void_p()
        _check(load_library().vmpc_comm_create_callback(1, 0, EXCHANGE_FN(0), None, ctypes.byref(h)),
               "vmpc_comm_create_callback")
        return cls(h, 1, 0)

    @property
    def kind(self):
        k = ctypes.c_int()
        _check(self.lib.vmpc_comm_info(self.handle, None, None, ctypes.byref(k)), "vmpc_comm_info")
        return self.KINDS[k.value]

    def info(self):
        """{"kind", "world", "rank"} as the library reports them (an RCCL communicator asks RCCL itself)"""
        w, r, k = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(self.lib.vmpc_comm_info(self.handle, ctypes.byref(w), ctypes.byref(r), ctypes.byref(k)), "vmpc_comm_info")
        return {"kind": self.KINDS[k.value], "world": w.value, "rank": r.value}

    def allgather(self, ctx, mine_ptr, gathered_ptr, bytes_per_rank):
        _check(self.lib.vmpc_comm_allgather_dev(self.handle, ctx.handle, ctypes.c_void_p(mine_ptr),
                                                ctypes.c_void_p(gathered_ptr), bytes_per_rank),
               "vmpc_comm_allgather_dev")

    def points_allsum(self, ctx, mine_ptr, k, scratch_ptr, out_ext_ptr, out_affine_ptr=None):
        """k rank-ordered sums of the ranks' partial points (enqueued on ctx's stream)"""
        _check(self.lib.vmpc_comm_points_allsum_dev(self.handle, ctx.handle, ctypes.c_void_p(mine_ptr), k,
                                                    ctypes.c_void_p(scratch_ptr), ctypes.c_void_p(out_ext_ptr),
                                                    ctypes.c_void_p(out_affine_ptr) if out_affine_ptr else None),
               "vmpc_comm_points_allsum_dev")

    def close(self):
        if self.handle:
            self.lib.vmpc_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class P4Rounds:
    """vmpc_p4_*: the Protocol-4 prover's rounds with z_hat, L~ and the challenge products resident in HBM."""

    def __init__(self, ctx, table, h_slots, k_slot, z_ptr, l_ptr, n_total=None, comm=None, commit_table=None,
                 jump_k=None, lazy_fold=False):
        """comm (a Comm): `table` holds this rank's block of g_hat (vmpc_p4_create_sharded), z / L~ all N scalars.
        commit_table: a second table over the same generators and extras for the pairs of the rounds before the
        fold (the 13-row wide-window table, vmpc_p4_set_commit_table)
        jump_k, lazy_fold: vmpc_p4_create_opts (rounds before the generator fold; the fold waits for prefold())"""
        self.ctx, self.table, self.comm, self.commit_table = ctx, table, comm, commit_table
        # the C side takes N = world * (table.n + h_slots) and reads that many scalars from z_hat and L~
        world = comm.world if comm is not None else 1
        assert n_total is None or n_total == world * (table.n + h_slots), \
            "round context needs the whole tabulated vector"
        h = ctypes.c_void_p()
        k_aff = ctypes.create_string_buffer(table.extra_bytes[k_slot], 64)
        if comm is not None:
            assert h_slots == 0, "a sharded CRS keeps h as the last generator of the last block"
            _check(ctx.lib.vmpc_p4_create_sharded(ctx.handle, comm.handle, ctypes.c_void_p(table.ptr), table.n,
                                                  len(table.extra_bytes), table.rows, k_slot, k_aff,
                                                  ctypes.c_void_p(z_ptr), ctypes.c_void_p(l_ptr), ctypes.byref(h)),
                   "vmpc_p4_create_sharded")
        elif jump_k is not None or lazy_fold:
            _check(ctx.lib.vmpc_p4_create_opts(ctx.handle, ctypes.c_void_p(table.ptr), table.n, len(table.extra_bytes),
                                               table.rows, h_slots, k_slot, k_aff, ctypes.c_void_p(z_ptr),
                                               ctypes.c_void_p(l_ptr), -1 if jump_k is None else int(jump_k),
                                               1 if lazy_fold else 0, ctypes.byref(h)),
                   "vmpc_p4_create_opts")
        else:
            _check(ctx.lib.vmpc_p4_create(ctx.handle, ctypes.c_void_p(table.ptr), table.n, len(table.extra_bytes),
                                          table.rows, h_slots, k_slot, k_aff, ctypes.c_void_p(z_ptr),
                                          ctypes.c_void_p(l_ptr), ctypes.byref(h)), "vmpc_p4_create")
        self.handle = h
        if commit_table is not None and comm is None:
            assert commit_table.n == table.n and commit_table.extra_bytes == table.extra_bytes
            _check(ctx.lib.vmpc_p4_set_commit_table(h, ctypes.c_void_p(commit_table.ptr), commit_table.rows),
                   "vmpc_p4_set_commit_table")

    def round(self, prev_challenge=None):
        a, b = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        c = ctypes.create_string_buffer(scalar_to_bytes(prev_challenge), 32) if prev_challenge is not None else None
        _check(self.ctx.lib.vmpc_p4_round(self.handle, c, a, b), "vmpc_p4_round")
        return a.raw, b.raw

    def round_begin(self, prev_challenge=None):
        """round() without the wait: the pair is enqueued on the context's stream; round_end() collects it"""
        c = ctypes.create_string_buffer(scalar_to_bytes(prev_challenge), 32) if prev_challenge is not None else None
        _check(self.ctx.lib.vmpc_p4_round_begin(self.handle, c), "vmpc_p4_round_begin")

    def round_end(self):
        a, b = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        _check(self.ctx.lib.vmpc_p4_round_end(self.handle, a, b), "vmpc_p4_round_end")
        return a.raw, b.raw

    def prefold(self):
        """a generator fold that is due (lazy_fold context, jump_k challenges fed): enqueue it now, wait for nothing"""
        _check(self.ctx.lib.vmpc_p4_prefold(self.handle), "vmpc_p4_prefold")

    def finish(self, last_challenge):
        z = ctypes.create_string_buffer(64)
        c = ctypes.create_string_buffer(scalar_to_bytes(last_challenge), 32)
        _check(self.ctx.lib.vmpc_p4_finish(self.handle, c, z), "vmpc_p4_finish")
        return int.from_bytes(z.raw[:32], "little"), int.from_bytes(z.raw[32:], "little")

    def run_compact(self, state, first_round_index, rounds):
        """all rounds + finish with the compact challenge chain: (new state, [(A, B) bytes], (z0, z1))"""
        st = ctypes.create_string_buffer(bytes(state), 32)
        ab = ctypes.create_string_buffer(128 * rounds)
        z = ctypes.create_string_buffer(64)
        _check(self.ctx.lib.vmpc_p4_run_compact(self.handle, st, first_round_index, ab, z), "vmpc_p4_run_compact")
        raw = ab.raw
        return (st.raw, [(raw[128 * i:128 * i + 64], raw[128 * i + 64:128 * i + 128]) for i in range(rounds)],
                (int.from_bytes(z.raw[:32], "little"), int.from_bytes(z.raw[32:], "little")))

    def close(self):
        if self.handle:
            self.ctx.lib.vmpc_p4_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host-buffer one-shots ------------------------------------------------------------------
def ed25519_msm(scalars, points):
    lib = load_library()
    s = as_bytes_array(scalars, 32)
    p = as_bytes_array(points, 64)
    assert len(s) == len(p)
    out = np.zeros(64, dtype=np.uint8)
    _check(lib.vmpc_ed25519_msm(_np_ptr(s), _np_ptr(p), len(s), _np_ptr(out)), "vmpc_ed25519_msm")
    return out


def ed25519_fold(pts_l, pts_r, c):
    lib = load_library()
    l = as_bytes_array(pts_l, 64)
    r = as_bytes_array(pts_r, 64)
    assert len(l) == len(r)
    out = np.zeros((len(l), 64), dtype=np.uint8)
    cb = ctypes.create_string_buffer(scalar_to_bytes(c), 32)
    _check(lib.vmpc_ed25519_fold(_np_ptr(l), _np_ptr(r), cb, len(l), _np_ptr(out)),
           "vmpc_ed25519_fold")
    return out


def ed25519_fixed_base_batch(base, scalars):
    lib = load_library()
    b = as_bytes_array(base, 64)
    s = as_bytes_array(scalars, 32)
    out = np.zeros((len(s), 64), dtype=np.uint8)
    _check(lib.vmpc_ed25519_fixed_base_batch(_np_ptr(b), _np_ptr(s), len(s), _np_ptr(out)),
           "vmpc_ed25519_fixed_base_batch")
    return out


def lincomb_host(points_affine, scalars):
    """sum_i scalars[i] * points[i] (64-byte affine encodings, ints mod l) on the host: vmpc_ed25519_lincomb_host"""
    lib = load_library()
    n = len(points_affine)
    out = ctypes.create_string_buffer(64)
    sc = b"".join(scalar_to_bytes(v) for v in scalars)
    _check(lib.vmpc_ed25519_lincomb_host(ctypes.c_char_p(b"".join(points_affine)), ctypes.c_char_p(sc), n, out),
           "vmpc_ed25519_lincomb_host")
    return out.raw


def fold_commitment_host(a_affine, q_affine, b_affine, c):
    """A + c Q + c^2 B (64-byte affine encodings in and out) on the host: vmpc_ed25519_fold_commitment_host"""
    lib = load_library()
    out = ctypes.create_string_buffer(64)
    _check(lib.vmpc_ed25519_fold_commitment_host(ctypes.c_char_p(a_affine), ctypes.c_char_p(q_affine),
                                                 ctypes.c_char_p(b_affine), ctypes.c_char_p(scalar_to_bytes(c)), out),
           "vmpc_ed25519_fold_commitment_host")
    return out.raw


def fr_axpy(c, x, y):
    lib = load_library()
    xa, ya = as_bytes_array(x, 32), as_bytes_array(y, 32)
    out = np.zeros_like(xa)
    cb = ctypes.create_string_buffer(scalar_to_bytes(c), 32)
    _check(lib.vmpc_fr_axpy(cb, _np_ptr(xa), _np_ptr(ya), len(xa), _np_ptr(out)), "vmpc_fr_axpy")
    return out


def fr_dot(a, b):
    lib = load_library()
    aa, ba = as_bytes_array(a, 32), as_bytes_array(b, 32)
    out = ctypes.create_string_buffer(32)
    _check(lib.vmpc_fr_dot(_np_ptr(aa), _np_ptr(ba), len(aa), out), "vmpc_fr_dot")
    return int.from_bytes(out.raw, "little")


def bn256_msm(group, scalars, points):
    """host-buffer one-shot: (n,32) scalars, (n,64|128) points -> affine bytes"""
    lib = load_library()
    width = 64 if group == 1 else 128
    s = as_bytes_array(scalars, 32)
    p = as_bytes_array(points, width)
    assert len(s) == len(p)
    out = np.zeros(width, dtype=np.uint8)
    fn = lib.vmpc_bn256_g1_msm if group == 1 else lib.vmpc_bn256_g2_msm
    _check(fn(_np_ptr(s), _np_ptr(p), len(s), _np_ptr(out)), f"vmpc_bn256_g{group}_msm")
    return out
