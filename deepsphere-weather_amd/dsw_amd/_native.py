"""ctypes binding of ``libdsw_hip.so`` (C ABI declared in ``include/dsw_hip.h``).

The library is built in-tree (``__graft_entry__.build()`` or ``python -m dsw_amd.build``) next to
this file.  There is NO fallback: if the shared object is missing or a symbol cannot be
resolved, importing the compute path raises - a GPU box must run the HIP kernels or fail loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Environment read by the PACKAGE (csrc/ reads none in the product build), each announced on stderr when in effect:
#   DSW_HIP_LIB            another build of the library (A/B runs of tools/build_variant*.sh)      [_native.py]
#   DSW_DIST_BACKEND       process-group backend instead of nccl / gloo by device                    [parallel.py]
#   DSW_FORCE_GRAD_SYNC=1  run the gradient exchange in a one-rank world (tests of the N > 1 path)   [parallel.py]
LIB_PATH = os.environ.get("DSW_HIP_LIB") or os.path.join(_HERE, "libdsw_hip.so")

DSW_F32 = 0
DSW_BF16 = 1

_i32p = ctypes.c_void_p
_f32p = ctypes.c_void_p
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_f32 = ctypes.c_float
_int = ctypes.c_int

# name -> (restype, argtypes): exactly the declarations of include/dsw_hip.h
SIGNATURES = {
    "dsw_version": (_int, []),
    "dsw_strerror": (ctypes.c_char_p, [_int]),
    "dsw_build_flags": (_int, []),
    "dsw_trace_begin": (_int, [_int]),
    "dsw_trace_end": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int]),
    "dsw_spmm_csr": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _f32, _vp, _f32, _vp, _f32, _int, _vp],
    ),
    "dsw_spmm_csr_ld": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _f32, _vp, _i64, _f32, _vp, _f32, _int, _vp],
    ),
    "dsw_remap_csr": (
        _int,
        [_vp, _i32p, _i32p, _f32p, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _f32, _int, _vp],
    ),
    "dsw_spmm2_fused": (
        _int,
        [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _int, _vp],
    ),
    "dsw_spmm2_supported": (_int, [_vp, _i64, _int]),
    "dsw_spmm_staged": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _f32, _int, _vp, _int]),
    "dsw_spmm_staged_supported": (_int, [_vp, _i64, _int]),
    "dsw_cheb_basis_fwd": (_int, [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp]),
    "dsw_cheb_basis_adj": (
        _int, [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp]
    ),
    "dsw_cheb_mix_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp]),
    "dsw_cheb_mix_first": (_int, [_i64, _i64, _i64]),
    "dsw_cheb_bwd_needs_basis": (_int, [_vp, _i64, _i64, _i64, _i64, _int]),
    "dsw_cheb_fwd_path": (_int, [_vp, _i64, _i64, _i64, _int]),
    "dsw_rezero_residual_workspace_bytes": (_i64, []),
    "dsw_rezero_residual_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "dsw_rezero_residual_fwd_ld": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "dsw_rezero_residual_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "dsw_maxval_pool_fwd": (_int, [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "dsw_maxval_pool_bwd": (_int, [_i32p, _i32p, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "dsw_maxval_unpool_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "dsw_maxval_unpool_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp]),
    "dsw_maxval_unpool_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp]),
    "dsw_cheb_fwd": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp],
    ),
    "dsw_cheb_fwd_act": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int],
    ),
    "dsw_relu_bwd": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "dsw_cheb_fwd_res": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int,
         _vp, _vp, _i64],
    ),
    "dsw_cheb_bwd_res": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
         _i64, _int, _vp, _vp, _vp, _vp, _i64, _int],
    ),
    "dsw_cheb_fwd_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64, _i64, _int]),
    "dsw_cheb_fwd_ws": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int,
         _vp, _vp, _i64, _vp, _i64],
    ),
    "dsw_rezero_param_grads_workspace_bytes": (_i64, []),
    "dsw_rezero_param_grads": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _int, _vp]),
    "dsw_cheb_bwd_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64, _i64, _int]),
    "dsw_cheb_bwd": (
        _int,
        [_i32p, _i32p, _f32p, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
         _i64, _int, _vp, _vp],
    ),
}

_lib = None


class DswNativeError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DswNativeError(
            f"{LIB_PATH} not found: the HIP library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root). "
            "There is no CPU/PyTorch fallback for the dsw hot path."
        )
    if os.environ.get("DSW_HIP_LIB"):
        import sys

        print("dsw_amd: DSW_HIP_LIB is set - loading %s instead of the in-tree library" % LIB_PATH, file=sys.stderr, flush=True)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    flags = int(lib.dsw_build_flags())
    if flags and not os.environ.get("DSW_HIP_LIB"):
        # a diagnostics build (environment overrides of the kernel selection) or an ablation build (kernels with phases cut
        # out: wrong results by design) must never stand in for the product library by accident
        raise DswNativeError(
            f"{LIB_PATH} is a diagnostics build (dsw_build_flags() = {flags}: bit 0 DSW_DIAG, bit 1 ablation switches); "
            "rebuild the product library (`python -m dsw_amd.build --force`) or name the variant explicitly with DSW_HIP_LIB")
    _lib = lib
    return lib


ROLE_NAMES = {1: "spmm", 2: "spmm2", 3: "spmm_staged", 4: "basis_fwd", 5: "basis_adj", 6: "mix_fwd", 7: "fwd_one_launch",
              8: "bwd_gemm_fused", 9: "bwd_dgrad", 10: "bwd_wgrad", 11: "zmix", 12: "clenshaw_fwd", 13: "elementwise",
              14: "bwd_fused", 15: "basis_dual", 16: "fwd_hop2_mix", 17: "bwd_dual"}


class LaunchTrace:
    """``with LaunchTrace(capacity) as tr: ...steps...`` then ``tr.kernels``: one record per kernel the library launched on
    the way, in launch order: (call index, role name, aux0, aux1, aux2, us, launch-site name) - ``call index`` numbers the role
    calls (all kernels of one role of one entry-point call share it), ``us`` is the kernel's duration from the start / stop
    events attached to its dispatch.  ``tr.intervals`` sums the kernels per role call: (role, aux0, aux1, aux2, us)."""

    def __init__(self, capacity=16384):
        self.capacity = int(capacity)
        self.kernels = []
        self.intervals = []

    def __enter__(self):
        check(load().dsw_trace_begin(self.capacity), "dsw_trace_begin")
        return self

    def __exit__(self, *exc):
        import numpy as np

        cap, stride = self.capacity, 96
        call, roles, a0, a1, a2 = (np.zeros(cap, dtype=np.int32) for _ in range(5))
        us = np.zeros(cap, dtype=np.float32)
        names = np.zeros(cap * stride, dtype=np.uint8)
        n = int(load().dsw_trace_end(call.ctypes.data, roles.ctypes.data, a0.ctypes.data, a1.ctypes.data, a2.ctypes.data,
                                     us.ctypes.data, names.ctypes.data, stride, cap))
        if n < 0:
            if exc[0] is None:
                check(n, "dsw_trace_end")
            return False
        n = min(n, cap)
        self.kernels = [(int(call[i]), ROLE_NAMES.get(int(roles[i]), str(int(roles[i]))), int(a0[i]), int(a1[i]), int(a2[i]),
                         float(us[i]), bytes(names[i * stride:(i + 1) * stride]).split(b"\0", 1)[0].decode(errors="replace"))
                        for i in range(n)]
        self.intervals = []
        last = None
        for c, role, x0, x1, x2, t, _name in self.kernels:
            if c != last:
                self.intervals.append([role, x0, x1, x2, 0.0])
                last = c
            self.intervals[-1][4] += t
        self.intervals = [tuple(v) for v in self.intervals]
        return False


def check(rc: int, what: str):
    if rc != 0:
        msg = load().dsw_strerror(int(rc)).decode()
        raise DswNativeError(f"{what} failed: {msg} (code {rc})")
