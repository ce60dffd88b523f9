"""ctypes wrapper of the plain-C oracle (``oracle/cheb_oracle.c``).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np
from scipy import sparse

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcheb_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "cheb_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _csr(rowptr, colind, values):
    return (np.ascontiguousarray(rowptr, dtype=np.int32), np.ascontiguousarray(colind, dtype=np.int32),
            np.ascontiguousarray(values, dtype=np.float32))


def cheb_forward(rowptr, colind, values, x, w, bias=None):
    rp, ci, va = _csr(rowptr, colind, values)
    B, V, Fin = x.shape
    _, K, Fout = w.shape
    xd, wd = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(w, dtype=np.float64)
    bd = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
    y = np.empty((B, V, Fout)); basis = np.empty((K, B, V, Fin))
    rc = lib().oracle_cheb_forward(_p(rp), _p(ci), _p(va), ctypes.c_int64(V), _p(xd), _p(wd),
                                   None if bd is None else _p(bd), _p(y), _p(basis), ctypes.c_int64(B),
                                   ctypes.c_int64(Fin), ctypes.c_int64(Fout), ctypes.c_int64(K))
    assert rc == 0
    return y, basis


def cheb_backward(rowptr, colind, values, basis, w, dy, has_bias=True):
    K, B, V, Fin = basis.shape
    Fout = w.shape[2]
    m = sparse.csr_matrix((np.asarray(values, dtype=np.float32), colind, rowptr), shape=(V, V)).T.tocsr()
    m.sort_indices()
    rp, ci, va = _csr(m.indptr, m.indices, m.data)
    wd, gd = np.ascontiguousarray(w, dtype=np.float64), np.ascontiguousarray(dy, dtype=np.float64)
    dx = np.empty((B, V, Fin)); dw = np.empty((Fin, K, Fout)); db = np.empty(Fout)
    rc = lib().oracle_cheb_backward(_p(rp), _p(ci), _p(va), ctypes.c_int64(V), _p(np.ascontiguousarray(basis)), _p(wd),
                                    _p(gd), _p(dx), _p(dw), _p(db) if has_bias else None, ctypes.c_int64(B),
                                    ctypes.c_int64(Fin), ctypes.c_int64(Fout), ctypes.c_int64(K))
    assert rc == 0
    return dx, dw, (db if has_bias else None)


def remap(rowptr, colind, values, shape, x):
    rp, ci, va = _csr(rowptr, colind, values)
    B, V, C = x.shape
    xd = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty((B, shape[0], C))
    rc = lib().oracle_remap(_p(rp), _p(ci), _p(va), ctypes.c_int64(shape[0]), ctypes.c_int64(shape[1]), _p(xd), _p(y),
                            ctypes.c_int64(B), ctypes.c_int64(C))
    assert rc == 0
    return y
