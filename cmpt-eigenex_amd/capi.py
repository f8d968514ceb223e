"""ctypes binding of include/eigenex_hip.h (the C ABI of libeigenex_hip.so).

This is plumbing for tests/ and bench.py; the product's host side is the
header-only C++ in cmpt-eigenex_amd/include/cmpt/eigen_ex/.  Nothing here
computes: every function forwards to the HIP library and raises EigenexError
when the library reports a failure (there is no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libeigenex_hip.so")

ORTHO_BATCHED = 0
ORTHO_SEQUENTIAL = 1
ORTHO_BATCHED_TWICE = 2
ORTHO_BATCHED_ADAPTIVE = 3
VEC_V = -1
VEC_W = -2
VEC_START = -3
K_SPMV, K_DOTS, K_UPDATE, K_SMALL, K_COMM, K_RITZ = range(6)


def VEC_COL(c: int) -> int:
    return int(c)


def VEC_ORTHO(q: int) -> int:
    return -16 - int(q)


class EigenexError(RuntimeError):
    pass


class State(C.Structure):
    _fields_ = [("nvec", C.c_int32), ("iterations", C.c_int32), ("nalpha", C.c_int32), ("nbeta", C.c_int32),
                ("stopped", C.c_int32), ("calls_true", C.c_int32), ("residue", C.c_double)]


MATVEC_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/eigenex_hip.h declares
SIGNATURES = {
    "eigenex_version": (C.c_int, []),
    "eigenex_last_error": (C.c_char_p, []),
    "eigenex_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "eigenex_partition": (C.c_int, [C.c_int64, C.c_int, C.c_int, _lp, _lp]),
    "eigenex_halo_plan": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int64, _ip, _lp, _ip, _lp]),
    "eigenex_rccl_unique_id": (C.c_int, [_vp]),
    "eigenex_context_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "eigenex_context_create_loopback": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "eigenex_context_destroy": (C.c_int, [_vp]),
    "eigenex_context_selftest": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "eigenex_context_sync": (C.c_int, [_vp]),
    "eigenex_context_info": (C.c_int, [_vp] + [C.POINTER(C.c_int)] * 4),
    "eigenex_context_comm_info": (C.c_int, [_vp] + [C.POINTER(C.c_int)] * 3),
    "eigenex_context_trace": (C.c_int, [_vp, C.c_int]),
    "eigenex_context_set_halo_overlap": (C.c_int, [_vp, C.c_int]),
    "eigenex_context_trace_get": (C.c_int, [_vp, _ip, _ip, C.c_int, C.POINTER(C.c_int)]),
    "eigenex_plan_create": (C.c_int, [C.c_int64, C.c_int, C.c_int, _ip, _ip, C.POINTER(_vp)]),
    "eigenex_plan_destroy": (C.c_int, [_vp]),
    "eigenex_plan_tiles": (C.c_int, [_vp, _ip, C.POINTER(C.c_int64), _ip, C.POINTER(C.c_int64)]),
    "eigenex_plan_sizes": (C.c_int, [_vp] + [C.POINTER(C.c_int64)] * 4 + [C.POINTER(C.c_int)] * 2 + [C.POINTER(C.c_int64)]),
    "eigenex_plan_local_columns": (C.c_int, [_vp, _ip]),
    "eigenex_plan_halo_columns": (C.c_int, [_vp, _ip]),
    "eigenex_plan_recv_segments": (C.c_int, [_vp, _ip, _lp, _lp]),
    "eigenex_plan_add_request": (C.c_int, [_vp, C.c_int, _ip, C.c_int64]),
    "eigenex_plan_send_segments": (C.c_int, [_vp, _ip, _lp, _lp, _lp]),
    "eigenex_plan_send_rows": (C.c_int, [_vp, _ip]),
    "eigenex_lanczos_collectives": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                              _ip, _ip, C.c_int, C.POINTER(C.c_int)]),
    "eigenex_context_stream": (_vp, [_vp]),
    "eigenex_profile_enable": (C.c_int, [_vp, C.c_int]),
    "eigenex_profile_reset": (C.c_int, [_vp]),
    "eigenex_profile_get": (C.c_int, [_vp, C.c_int, _lp, _dp, _dp]),
    "eigenex_csr_upload": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _ip, _ip, _dp, C.POINTER(_vp)]),
    "eigenex_csr_upload_z": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _ip, _ip, _dp, C.POINTER(_vp)]),
    "eigenex_csr_upload64": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), _ip, _dp, C.POINTER(_vp)]),
    "eigenex_csr_laplacian3d": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "eigenex_csr_destroy": (C.c_int, [_vp]),
    "eigenex_csr_upload_ex": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _ip, _ip, _dp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "eigenex_csr_column_blocks": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "eigenex_csr_layout": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "eigenex_csr_upload_device": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "eigenex_block_upload": (C.c_int, [_vp, C.c_int64, C.c_int, _lp, C.c_int, _lp, C.c_int64, _lp, _lp, C.POINTER(C.c_void_p), C.POINTER(_vp)]),
    "eigenex_block_upload_z": (C.c_int, [_vp, C.c_int64, C.c_int, _lp, C.c_int, _lp, C.c_int64, _lp, _lp, C.POINTER(C.c_void_p), C.POINTER(_vp)]),
    "eigenex_csr_info": (C.c_int, [_vp, _lp, _lp, _lp, _lp]),
    "eigenex_basis_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "eigenex_basis_create_ex": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "eigenex_basis_is_complex": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "eigenex_basis_configure_z": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int]),
    "eigenex_basis_destroy": (C.c_int, [_vp]),
    "eigenex_basis_set_host_operator": (C.c_int, [_vp, MATVEC_FN, _vp]),
    "eigenex_basis_configure": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int64, C.c_int]),
    "eigenex_basis_clear": (C.c_int, [_vp]),
    "eigenex_vec_upload": (C.c_int, [_vp, C.c_int, _dp]),
    "eigenex_vec_download": (C.c_int, [_vp, C.c_int, _dp]),
    "eigenex_vec_copy": (C.c_int, [_vp, C.c_int, C.c_int]),
    "eigenex_basis_reserve": (C.c_int, [_vp, C.c_int]),
    "eigenex_basis_capacity": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "eigenex_basis_tune": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "eigenex_ritz_vectors_complex": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, C.c_int64]),
    "eigenex_krylov_combine": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, C.c_int64]),
    "eigenex_apply": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, _dp]),
    "eigenex_dots": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "eigenex_update": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "eigenex_axpy2": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int]),
    "eigenex_scale": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double]),
    "eigenex_lanczos_enqueue": (C.c_int, [_vp, C.c_int]),
    "eigenex_basis_set_alpha_fusion": (C.c_int, [_vp, C.c_int]),
    "eigenex_basis_clone": (C.c_int, [_vp, C.POINTER(_vp)]),
    "eigenex_basis_graph_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "eigenex_arnoldi_enqueue": (C.c_int, [_vp, C.c_int]),
    "eigenex_lanczos_restart": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double]),
    "eigenex_lanczos_state": (C.c_int, [_vp, C.POINTER(State), _dp, _dp]),
    "eigenex_arnoldi_state": (C.c_int, [_vp, C.POINTER(State), _dp, C.c_int]),
    "eigenex_ritz_vectors": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int64]),
}

_LIB = None


def _preload_torch_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / librccl.so.1 (ROCm 7.0 here) under the same SONAMEs as
    the system ROCm this library is linked against.  Whichever copy is loaded first serves both: torch first is
    fine (this library then runs on torch's runtime; that is how bench.py and the tests run), this library first makes
    torch mix runtimes and abort at exit ("double free or corruption").  So when torch is installed it is imported
    before the library is loaded.  EIGENEX_NO_TORCH_PRELOAD=1 skips this for processes that never import torch."""
    if "torch" in sys.modules or os.environ.get("EIGENEX_NO_TORCH_PRELOAD"):
        return
    import importlib.util

    if importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:  # a broken torch install must not keep the library from loading
            pass


def _hip_runtimes_mapped():
    """distinct libamdhip64 / librccl files mapped into this process (from /proc/self/maps)"""
    found = {"libamdhip64": set(), "librccl": set()}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                base = os.path.basename(path)
                for k in found:
                    if base.startswith(k + ".so"):
                        found[k].add(os.path.realpath(path))
    except OSError:
        pass
    return found


def _refuse_mixed_runtimes():
    """Two copies of the HIP (or RCCL) runtime in one process -- the system one this library links against and the
    one a PyTorch-ROCm wheel bundles under the same SONAME -- end in an abort at interpreter exit ("double free or
    corruption") long after the cause.  That can only happen when something loaded one copy by path before the other
    was resolved by name; refuse right here, where the order can still be fixed."""
    for name, paths in _hip_runtimes_mapped().items():
        if len(paths) > 1:
            raise EigenexError(
                f"two copies of {name} are loaded in this process ({', '.join(sorted(paths))}): import torch BEFORE anything "
                "loads the system ROCm runtime (cmpt_eigenex_amd.capi does so by itself unless EIGENEX_NO_TORCH_PRELOAD is set), "
                "or do not import torch in this process at all")


def lib():
    """Load libeigenex_hip.so (built in-tree by cmpt_eigenex_amd.build).  Fails loudly if missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EigenexError(
                f"{LIB_PATH} is missing: build it with `python -m cmpt_eigenex_amd.build` "
                "(there is no CPU fallback for the Krylov hot path)")
        _preload_torch_runtime()
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _refuse_mixed_runtimes()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def _chk(rc: int):
    if rc != 0:
        raise EigenexError(f"eigenex error {rc}: {lib().eigenex_last_error().decode(errors='replace')}")


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().eigenex_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def partition(n_global: int, nshards: int, shard: int):
    b, e = C.c_int64(), C.c_int64()
    _chk(lib().eigenex_partition(n_global, nshards, shard, C.byref(b), C.byref(e)))
    return b.value, e.value


def halo_plan(n_global: int, nshards: int, shard: int, col_global: np.ndarray):
    col = np.ascontiguousarray(col_global, np.int32)
    nh = C.c_int64()
    _chk(lib().eigenex_halo_plan(n_global, nshards, shard, col.size, _i(col), C.byref(nh), None, None))
    cols = np.empty(nh.value, np.int32)
    per = np.zeros(nshards, np.int64)
    _chk(lib().eigenex_halo_plan(n_global, nshards, shard, col.size, _i(col), C.byref(nh), _i(cols),
                                 per.ctypes.data_as(_lp)))
    return cols, per


COLL_ALLREDUCE, COLL_HALO = 1, 2


def lanczos_collectives(call_index: int, last_in_batch: bool, alpha_pending: bool, interval=1, n_ortho=0, ortho_mode=ORTHO_BATCHED,
                        alpha_fusion=True, is_complex=False):
    """[(op, count), ...] of one Lanczos step call between shards, and the alpha-fusion state after it"""
    pend = C.c_int(int(alpha_pending))
    n = C.c_int()
    ops, cnt = np.zeros(4096, np.int32), np.zeros(4096, np.int32)
    _chk(lib().eigenex_lanczos_collectives(call_index, int(last_in_batch), C.byref(pend), interval, n_ortho, ortho_mode, int(alpha_fusion),
                                           int(is_complex), _i(ops), _i(cnt), ops.size, C.byref(n)))
    return [(int(ops[i]), int(cnt[i])) for i in range(n.value)], bool(pend.value)


class ShardPlan:
    """eigenex_plan_*: the host-side plan of one CSR row shard (no GPU).  rowptr relative to the shard's first row,
    col with GLOBAL column indices."""

    def __init__(self, n_global: int, nshards: int, shard: int, rowptr, col_global):
        self.h = _vp()
        rp = np.ascontiguousarray(rowptr, np.int32)
        cl = np.ascontiguousarray(col_global, np.int32)
        _chk(lib().eigenex_plan_create(n_global, nshards, shard, _i(rp), _i(cl), C.byref(self.h)))

    def sizes(self):
        a = [C.c_int64() for _ in range(4)]
        b = [C.c_int() for _ in range(2)]
        c = C.c_int64()
        _chk(lib().eigenex_plan_sizes(self.h, *[C.byref(x) for x in a], *[C.byref(x) for x in b], C.byref(c)))
        return dict(n_local=a[0].value, n_pad=a[1].value, nnz=a[2].value, n_halo=a[3].value, n_recv=b[0].value, n_send=b[1].value,
                    n_send_rows=c.value)

    def local_columns(self):
        out = np.zeros(max(self.sizes()["nnz"], 1), np.int32)
        _chk(lib().eigenex_plan_local_columns(self.h, _i(out)))
        return out[: self.sizes()["nnz"]]

    def tiles(self):
        """(interior, boundary): the 256-row tiles that read no halo column / at least one (eigenex_plan_tiles)"""
        ni, nb = C.c_int64(), C.c_int64()
        _chk(lib().eigenex_plan_tiles(self.h, None, C.byref(ni), None, C.byref(nb)))
        ti, tb = np.zeros(max(ni.value, 1), np.int32), np.zeros(max(nb.value, 1), np.int32)
        _chk(lib().eigenex_plan_tiles(self.h, _i(ti), C.byref(ni), _i(tb), C.byref(nb)))
        return ti[:ni.value], tb[:nb.value]

    def halo_columns(self):
        out = np.zeros(max(self.sizes()["n_halo"], 1), np.int32)
        _chk(lib().eigenex_plan_halo_columns(self.h, _i(out)))
        return out[: self.sizes()["n_halo"]]

    def recv_segments(self):
        n = self.sizes()["n_recv"]
        peer, off, cnt = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64)
        _chk(lib().eigenex_plan_recv_segments(self.h, _i(peer), off.ctypes.data_as(_lp), cnt.ctypes.data_as(_lp)))
        return [(int(peer[i]), int(off[i]), int(cnt[i])) for i in range(n)]

    def add_request(self, from_shard: int, rows_global):
        r = np.ascontiguousarray(rows_global, np.int32)
        _chk(lib().eigenex_plan_add_request(self.h, from_shard, _i(r), r.size))

    def send_segments(self):
        n = self.sizes()["n_send"]
        peer = np.zeros(max(n, 1), np.int32)
        off, cnt, cs = (np.zeros(max(n, 1), np.int64) for _ in range(3))
        _chk(lib().eigenex_plan_send_segments(self.h, _i(peer), *[x.ctypes.data_as(_lp) for x in (off, cnt, cs)]))
        return [(int(peer[i]), int(off[i]), int(cnt[i]), int(cs[i])) for i in range(n)]

    def send_rows(self):
        n = self.sizes()["n_send_rows"]
        out = np.zeros(max(n, 1), np.int32)
        _chk(lib().eigenex_plan_send_rows(self.h, _i(out)))
        return out[:n]

    def close(self):
        if self.h:
            lib().eigenex_plan_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _chk(lib().eigenex_rccl_unique_id(buf))
    return buf.raw


class Context:
    def __init__(self, device=0, rank=0, world_size=1, rccl_id: bytes | None = None, loopback_shards: int = 0):
        self.h = _vp()
        if loopback_shards:
            _chk(lib().eigenex_context_create_loopback(device, loopback_shards, C.byref(self.h)))
        else:
            idbuf = C.create_string_buffer(rccl_id, 128) if rccl_id else None
            _chk(lib().eigenex_context_create(device, rank, world_size, idbuf, C.byref(self.h)))

    def info(self):
        v = [C.c_int() for _ in range(4)]
        _chk(lib().eigenex_context_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("rank", "world_size", "nshards_total", "nshards_local"), (x.value for x in v)))

    def trace(self, on=True):
        _chk(lib().eigenex_context_trace(self.h, int(on)))

    def set_halo_overlap(self, on=True) -> bool:
        """neighbour exchange beside the interior rows (True) or in front of the operator (False); returns what is in force"""
        rc = lib().eigenex_context_set_halo_overlap(self.h, int(on))
        if rc < 0:
            _chk(rc)
        return rc == 1

    def trace_get(self):
        n = C.c_int()
        _chk(lib().eigenex_context_trace_get(self.h, None, None, 0, C.byref(n)))
        ops, cnt = np.zeros(max(n.value, 1), np.int32), np.zeros(max(n.value, 1), np.int32)
        _chk(lib().eigenex_context_trace_get(self.h, _i(ops), _i(cnt), ops.size, C.byref(n)))
        return [(int(ops[i]), int(cnt[i])) for i in range(n.value)]

    def comm_info(self):
        """(ranks, rank, device) as the RCCL communicator reports them; ranks = 0 without a communicator"""
        v = [C.c_int() for _ in range(3)]
        _chk(lib().eigenex_context_comm_info(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def sync(self):
        _chk(lib().eigenex_context_sync(self.h))

    def rccl_selftest(self) -> bool:
        ok = C.c_int(0)
        _chk(lib().eigenex_context_selftest(self.h, C.byref(ok)))
        return bool(ok.value)

    def profile_enable(self, on=True):
        _chk(lib().eigenex_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        _chk(lib().eigenex_profile_reset(self.h))

    def profile_get(self, kind: int):
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        _chk(lib().eigenex_profile_get(self.h, kind, C.byref(n), C.byref(ms), C.byref(by)))
        return n.value, ms.value, by.value

    def close(self):
        if self.h:
            lib().eigenex_context_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Csr:
    def __init__(self, ctx: Context, handle):
        self.ctx, self.h = ctx, handle

    @classmethod
    def upload(cls, ctx: Context, n_global: int, rowptr, col, val, row_begin: int = 0, column_blocks: int | None = None):
        """val real (float64) or complex (complex128: uploaded as interleaved pairs, eigenex_csr_upload_z).
        column_blocks: None = the plain entry points (automatic choice), else eigenex_csr_upload_ex's argument."""
        rp = np.ascontiguousarray(rowptr, np.int32)
        cl = np.ascontiguousarray(col, np.int32)
        h = _vp()
        cplx = bool(np.iscomplexobj(val))
        vl = np.ascontiguousarray(val, np.complex128 if cplx else np.float64)
        vp = _d(vl.view(np.float64))
        if column_blocks is not None:
            _chk(lib().eigenex_csr_upload_ex(ctx.h, n_global, row_begin, rp.size - 1, _i(rp), _i(cl), vp, int(cplx), int(column_blocks), C.byref(h)))
        elif cplx:
            _chk(lib().eigenex_csr_upload_z(ctx.h, n_global, row_begin, rp.size - 1, _i(rp), _i(cl), vp, C.byref(h)))
        else:
            _chk(lib().eigenex_csr_upload(ctx.h, n_global, row_begin, rp.size - 1, _i(rp), _i(cl), vp, C.byref(h)))
        obj = cls(ctx, h)
        obj.is_complex = bool(np.iscomplexobj(val))
        return obj

    @classmethod
    def upload64(cls, ctx: Context, n_global: int, rowptr, col, val, row_begin: int = 0):
        """eigenex_csr_upload64: real CSR with 64-bit row pointers (shards of >= 2^31 stored entries are kept as plain CSR with
        64-bit row pointers on the device, smaller ones exactly as `upload` stores them)."""
        rp = np.ascontiguousarray(rowptr, np.int64)
        cl = np.ascontiguousarray(col, np.int32)
        vl = np.ascontiguousarray(val, np.float64)
        h = _vp()
        _chk(lib().eigenex_csr_upload64(ctx.h, n_global, row_begin, rp.size - 1, rp.ctypes.data_as(C.POINTER(C.c_int64)), _i(cl), _d(vl), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device(cls, ctx: Context, n: int, rowptr_ptr: int, col_ptr: int, val_ptr: int, is_complex: bool = False):
        """eigenex_csr_upload_device: CSR arrays already on this GPU, given as raw device addresses
        (e.g. torch tensors' .data_ptr(): int32 rowptr[n+1], int32 col[nnz], float64/complex128 val[nnz])."""
        h = _vp()
        _chk(lib().eigenex_csr_upload_device(ctx.h, n, _vp(rowptr_ptr), _vp(col_ptr), _vp(val_ptr), int(is_complex), C.byref(h)))
        obj = cls(ctx, h)
        obj.is_complex = bool(is_complex)
        return obj

    @classmethod
    def upload_blocks(cls, ctx: Context, row_sizes, col_sizes, blocks):
        """eigenex_block_upload[_z]: blocks = {(qr, qc): 2-D array of shape (row_sizes[qr], col_sizes[qc])}, all real
        (float64) or, if any is complex, all uploaded as complex128."""
        rs = np.ascontiguousarray(row_sizes, np.int64)
        cs = np.ascontiguousarray(col_sizes, np.int64)
        keys = list(blocks.keys())
        cplx = any(np.iscomplexobj(blocks[k]) for k in keys)
        dt = np.complex128 if cplx else np.float64
        mats = []
        for (r, c) in keys:
            m = np.asfortranarray(blocks[(r, c)], dt)
            if not (0 <= r < rs.size and 0 <= c < cs.size) or m.shape != (rs[r], cs[c]):
                raise ValueError(f"block ({r}, {c}): shape {m.shape} does not match the partition")
            mats.append(m)
        qr = np.array([k[0] for k in keys], np.int64)
        qc = np.array([k[1] for k in keys], np.int64)
        ptrs = (C.c_void_p * max(len(mats), 1))(*[m.ctypes.data for m in mats])
        h = _vp()
        fn = lib().eigenex_block_upload_z if cplx else lib().eigenex_block_upload
        _chk(fn(ctx.h, int(rs.sum()), rs.size, rs.ctypes.data_as(_lp), cs.size, cs.ctypes.data_as(_lp),
                len(mats), qr.ctypes.data_as(_lp), qc.ctypes.data_as(_lp), ptrs, C.byref(h)))
        obj = cls(ctx, h)
        obj.is_complex = cplx
        return obj

    @classmethod
    def upload_blocks_raw(cls, ctx: Context, row_sizes, col_sizes, qr, qc, values, offsets):
        """eigenex_block_upload for many blocks without a Python object per block: block k is the column-major
        array starting at values[offsets[k]] (values: one contiguous float64 array)."""
        rs = np.ascontiguousarray(row_sizes, np.int64)
        cs = np.ascontiguousarray(col_sizes, np.int64)
        qr = np.ascontiguousarray(qr, np.int64)
        qc = np.ascontiguousarray(qc, np.int64)
        values = np.ascontiguousarray(values, np.float64)
        offsets = np.ascontiguousarray(offsets, np.int64)
        sizes = rs[qr] * cs[qc]
        if qr.size and (offsets.min() < 0 or (offsets + sizes).max() > values.size):
            raise ValueError("a block reaches outside the value array")
        ptrs = (values.ctypes.data + 8 * offsets).astype(np.uint64)
        h = _vp()
        _chk(lib().eigenex_block_upload(ctx.h, int(rs.sum()), rs.size, rs.ctypes.data_as(_lp), cs.size, cs.ctypes.data_as(_lp),
                                        qr.size, qr.ctypes.data_as(_lp), qc.ctypes.data_as(_lp),
                                        C.cast(ptrs.ctypes.data, C.POINTER(C.c_void_p)), C.byref(h)))
        obj = cls(ctx, h)
        obj.is_complex = False
        return obj

    @classmethod
    def laplacian3d(cls, ctx: Context, n: int):
        h = _vp()
        _chk(lib().eigenex_csr_laplacian3d(ctx.h, n, C.byref(h)))
        return cls(ctx, h)

    def layout(self) -> str:
        v = C.c_int()
        _chk(lib().eigenex_csr_layout(self.h, C.byref(v)))
        return ("csr", "column_blocked", "sorted_tiles", "dense_blocks", "split_tiles")[v.value]

    def column_blocks(self) -> int:
        k = C.c_int()
        _chk(lib().eigenex_csr_column_blocks(self.h, C.byref(k)))
        return k.value

    def info(self):
        v = [C.c_int64() for _ in range(4)]
        _chk(lib().eigenex_csr_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("n_global", "n_local", "nnz_local", "n_halo_local"), (x.value for x in v)))

    def close(self):
        if self.h:
            lib().eigenex_csr_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Basis:
    """Krylov state on the device (basis slab, work vectors, coefficient arrays).
    dtype float64 or complex128 (taken from the CSR operator when one is given)."""

    def __init__(self, ctx: Context, csr: Csr | None, n_global: int, capacity: int, n_ortho: int = 0, dtype=None):
        self.ctx, self.csr, self.n_global, self.capacity, self.n_ortho = ctx, csr, n_global, capacity, n_ortho
        if dtype is None:
            dtype = np.complex128 if (csr is not None and getattr(csr, "is_complex", False)) else np.float64
        self.dtype = np.dtype(dtype)
        self.is_complex = self.dtype.kind == "c"
        self.h = _vp()
        _chk(lib().eigenex_basis_create_ex(ctx.h, csr.h if csr else None, n_global, capacity, n_ortho,
                                           1 if self.is_complex else 0, C.byref(self.h)))
        info = ctx.info()
        if info["nshards_local"] == info["nshards_total"]:
            self.n_rows = n_global
        else:
            b, e = partition(n_global, info["world_size"], info["rank"])
            self.n_rows = e - b
        self._cb = None

    def _vec(self, x):
        x = np.ascontiguousarray(x, self.dtype)
        return x, _d(x.view(np.float64))

    def configure(self, shift=0.0, threshold=1e-12, interval=1, ortho_mode=ORTHO_BATCHED):
        sh = complex(shift)
        _chk(lib().eigenex_basis_configure_z(self.h, sh.real, sh.imag, threshold, interval, ortho_mode))

    def set_host_operator(self, fn):
        """fn(x: ndarray) -> ndarray  (the reference's MatMulFunction, lanczos.hpp:116)."""
        n, es, dt = self.n_rows, (2 if self.is_complex else 1), self.dtype

        def tramp(pin, pout, _user):
            x = np.ctypeslib.as_array(pin, shape=(n * es,)).view(dt)
            y = np.ctypeslib.as_array(pout, shape=(n * es,)).view(dt)
            y[:] = fn(x)

        self._cb = MATVEC_FN(tramp)
        _chk(lib().eigenex_basis_set_host_operator(self.h, self._cb, None))

    def tune(self, vec_blocks_per_cu=2, spmv_blocks_per_cu=4, flags=0):
        _chk(lib().eigenex_basis_tune(self.h, vec_blocks_per_cu, spmv_blocks_per_cu, flags))

    def graph_info(self):
        n, t, lim = C.c_int(), C.c_int64(), C.c_int64()
        _chk(lib().eigenex_basis_graph_info(self.h, C.byref(n), C.byref(t), C.byref(lim)))
        return dict(graphs=n.value, nodes=t.value, node_limit=lim.value)

    def set_alpha_fusion(self, on: bool):
        _chk(lib().eigenex_basis_set_alpha_fusion(self.h, int(bool(on))))

    def clear(self):
        _chk(lib().eigenex_basis_clear(self.h))

    def upload(self, ref: int, x):
        x, p = self._vec(x)
        assert x.size == self.n_rows
        _chk(lib().eigenex_vec_upload(self.h, ref, p))

    def copy(self, dst_ref: int, src_ref: int):
        _chk(lib().eigenex_vec_copy(self.h, dst_ref, src_ref))

    def reserve(self, capacity: int):
        _chk(lib().eigenex_basis_reserve(self.h, capacity))
        self.capacity = max(self.capacity, capacity)

    def download(self, ref: int):
        x = np.empty(self.n_rows, self.dtype)
        _chk(lib().eigenex_vec_download(self.h, ref, _d(x.view(np.float64))))
        return x

    # -- primitives
    def apply(self, x_ref, y_ref, shift=0.0, want_dot=False):
        d = np.zeros(1, self.dtype)
        _chk(lib().eigenex_apply(self.h, x_ref, y_ref, shift, _d(d.view(np.float64)) if want_dot else None))
        return d[0] if want_dot else None

    def dots(self, w_ref, first, stride, count, n_ortho_used=0):
        h = np.zeros(count + n_ortho_used, self.dtype)
        _chk(lib().eigenex_dots(self.h, w_ref, first, stride, count, n_ortho_used, _d(h.view(np.float64))))
        return h

    def update(self, w_ref, first, stride, count, h, n_ortho_used=0):
        h, p = self._vec(h)
        nrm2 = C.c_double()
        _chk(lib().eigenex_update(self.h, w_ref, first, stride, count, n_ortho_used, p if h.size else None, C.byref(nrm2)))
        return nrm2.value

    def axpy2(self, z_ref, x_ref, a, p_ref, b, q_ref):
        _chk(lib().eigenex_axpy2(self.h, z_ref, x_ref, a, p_ref, b, q_ref))

    def scale(self, dst_ref, src_ref, s):
        _chk(lib().eigenex_scale(self.h, dst_ref, src_ref, s))

    # -- fused steps
    def lanczos_enqueue(self, ncalls: int):
        _chk(lib().eigenex_lanczos_enqueue(self.h, ncalls))

    def lanczos_restart(self, S, coupling_last: float):
        """thick restart: keep the Ritz vectors V_m S (S: m x nkeep, real)"""
        S = np.asfortranarray(S, np.float64)
        _chk(lib().eigenex_lanczos_restart(self.h, S.shape[1], _d(S), S.shape[0], float(coupling_last)))

    def arnoldi_enqueue(self, ncalls: int):
        _chk(lib().eigenex_arnoldi_enqueue(self.h, ncalls))

    def lanczos_state(self):
        st = State()
        a = np.zeros(self.capacity + 2)
        b = np.zeros(self.capacity + 2)
        _chk(lib().eigenex_lanczos_state(self.h, C.byref(st), _d(a), _d(b)))
        return st, a[: st.nalpha].copy(), b[: st.nbeta].copy()

    def arnoldi_state(self):
        st = State()
        ldh = self.capacity + 2
        H = np.zeros((self.capacity + 1, ldh), self.dtype)  # row c = column c of H
        _chk(lib().eigenex_arnoldi_state(self.h, C.byref(st), _d(H.view(np.float64)), ldh))
        m = min(st.nalpha, self.n_global)
        return st, H[:m, :m].T.copy()

    def ritz_vectors(self, nvec: int, S):
        """X = V S, normalised and phase-fixed.  Real S: X has the basis dtype; complex S: X is complex."""
        nev = S.shape[1]
        if np.iscomplexobj(S):
            S = np.asfortranarray(S, np.complex128)
            Sr, Si = np.asfortranarray(S.real), np.asfortranarray(S.imag)
            X = np.zeros((self.n_rows, nev), np.complex128, order="F")
            if nev:
                _chk(lib().eigenex_ritz_vectors_complex(self.h, nvec, nev, _d(Sr), _d(Si), S.shape[0],
                                                        X.ctypes.data_as(_dp), self.n_rows))
            return X
        S = np.asfortranarray(S, np.float64)
        X = np.zeros((self.n_rows, nev), self.dtype, order="F")
        if nev:
            _chk(lib().eigenex_ritz_vectors(self.h, nvec, nev, _d(S), S.shape[0], X.ctypes.data_as(_dp), self.n_rows))
        return X

    def close(self):
        if self.h:
            lib().eigenex_basis_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
