"""CPU oracle for the ConvCheb / RemapBlock hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement of the reference algorithm; it is the *checker*, never
the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The shipped path (``deepsphere-weather_amd/``) never
does and raises if its HIP library is missing.

Pinning: the reference has no tests / golden vectors of its own, so
this oracle is pinned against outputs of the reference itself, imported in the build
container by ``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks every fixture).

Two independent restatements are kept:

* ``conv_cheb_torch`` / ``remap_torch`` - the same torch op sequence as the reference
  (``torch.sparse.mm`` + ``matmul`` on a ``[V, Fin*B]`` layout), so it executes the same
  ATen CPU kernels.  This is the ``cpu_baseline`` ("port") that ``bench.py`` times.
* ``cheb_forward_f64`` / ``cheb_backward_f64`` - closed form in the native ``[B, V, C]``
  layout, float64 numpy + scipy CSR, with the hand-derived backward
  (SURVEY.md section 8a1), independent of autograd.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy import sparse


# --------------------------------------------------------------------------------------
# Operator preparation  (reference: modules/layers.py:72-106, 584-594)
# --------------------------------------------------------------------------------------
def scale_operator(laplacian, lmax, scale=1.0):
    """``L * 2*scale/lmax - I``  (reference ``modules/layers.py:72-79``)."""
    n = laplacian.shape[0]
    ident = sparse.identity(n, format=laplacian.format, dtype=laplacian.dtype)
    return laplacian * (2.0 * scale / lmax) - ident


def prepare_laplacian_fixed_lmax(laplacian, lmax):
    """``prepare_torch_laplacian`` (reference ``modules/layers.py:82-106``) with the
    ARPACK eigenvalue estimate replaced by a given ``lmax`` (ARPACK is nondeterministic)."""
    lap = laplacian.astype(np.float32)
    lap = scale_operator(lap, lmax)
    lap = sparse.coo_matrix(lap, dtype=lap.dtype)
    idx = np.stack((lap.row, lap.col), axis=0).astype(np.int64)
    t = torch.sparse_coo_tensor(
        torch.from_numpy(idx), torch.from_numpy(lap.data), lap.shape, dtype=torch.float32
    )
    return t.coalesce()


def coo_from_scipy(mat):
    """``convert_to_torch_sparse`` (reference ``modules/layers.py:584-594``)."""
    mat = sparse.coo_matrix(mat)
    idx = np.stack((mat.row, mat.col), axis=0).astype(np.int64)
    t = torch.sparse_coo_tensor(
        torch.from_numpy(idx), torch.from_numpy(mat.data), mat.shape, dtype=torch.get_default_dtype()
    )
    return t.coalesce()


def coo_from_csr_arrays(rowptr, colind, values, shape, dtype=torch.float32):
    rowptr = np.asarray(rowptr, dtype=np.int64)
    rows = np.repeat(np.arange(shape[0], dtype=np.int64), np.diff(rowptr))
    idx = np.stack((rows, np.asarray(colind, dtype=np.int64)), axis=0)
    t = torch.sparse_coo_tensor(
        torch.from_numpy(idx), torch.as_tensor(np.asarray(values)), tuple(shape), dtype=dtype
    )
    return t.coalesce()


def csr_arrays_from_coo(t):
    """Coalesced torch COO -> (rowptr int32, colind int32, values) numpy arrays."""
    t = t.coalesce()
    idx = t.indices().numpy()
    vals = t.values().float().numpy()
    n = t.shape[0]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, idx[0] + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), idx[1].astype(np.int32), vals


# --------------------------------------------------------------------------------------
# Restatement 1: the reference's torch op sequence
# --------------------------------------------------------------------------------------
def conv_cheb_torch(laplacian, inputs, weight):
    """Chebyshev convolution, same op sequence as reference ``modules/layers.py:113-180``."""
    B, V, Fin1 = inputs.shape
    Fin, K, Fout = weight.shape
    if Fin1 != Fin:
        raise ValueError("Input tensor shape does not match the expected shape")
    x0 = inputs.permute(1, 2, 0).contiguous().view(V, Fin * B)  # layers.py:158-159
    stack = [x0]
    if K > 1:
        x1 = torch.sparse.mm(laplacian, x0)  # layers.py:164
        stack.append(x1)
    for _ in range(2, K):
        x2 = 2 * torch.sparse.mm(laplacian, x1) - x0  # layers.py:167
        stack.append(x2)
        x0, x1 = x1, x2
    x = torch.stack(stack, 0).view(K, V, Fin, B)
    x = x.permute(3, 1, 2, 0).contiguous().view(B * V, Fin * K)  # layers.py:171-173
    x = x.matmul(weight.view(Fin * K, Fout))  # layers.py:176-177
    return x.view(B, V, Fout)


def conv_cheb_layer_torch(laplacian, inputs, weight, bias):
    """``ConvCheb.forward`` (reference ``modules/layers.py:365-376``)."""
    out = conv_cheb_torch(laplacian, inputs, weight)
    if bias is not None:
        out = out + bias
    return out


def remap_torch(matrix, x):
    """``RemapBlock.forward`` (reference ``modules/layers.py:956-964``)."""
    n_batch, n_nodes, n_val = x.shape
    new_nodes = matrix.shape[0]
    y = x.permute(1, 2, 0).reshape(n_nodes, n_batch * n_val)
    y = torch.sparse.mm(matrix, y)
    return y.reshape(new_nodes, n_val, n_batch).permute(2, 0, 1)


def conv_cheb_fwd_bwd_torch(laplacian, x, weight, bias, grad_out):
    """Forward + autograd backward through the reference op sequence.

    Returns ``(y, dx, dw, db)`` (``db`` is None if ``bias`` is None)."""
    x = x.detach().clone().requires_grad_(True)
    w = weight.detach().clone().requires_grad_(True)
    b = None if bias is None else bias.detach().clone().requires_grad_(True)
    y = conv_cheb_layer_torch(laplacian, x, w, b)
    y.backward(grad_out)
    return y.detach(), x.grad, w.grad, (None if b is None else b.grad)


# --------------------------------------------------------------------------------------
# Restatement 2: closed form, float64, native [B, V, C] layout, explicit backward
# --------------------------------------------------------------------------------------
def _csr64(rowptr, colind, values, shape):
    return sparse.csr_matrix(
        (np.asarray(values, dtype=np.float64), np.asarray(colind), np.asarray(rowptr)), shape=shape
    )


def cheb_basis_f64(L, x, K):
    """``T_0 = x, T_1 = L x, T_k = 2 L T_{k-1} - T_{k-2}`` for every sample; list of [B,V,F]."""
    B, V, F = x.shape
    xt = np.ascontiguousarray(np.transpose(x.astype(np.float64), (1, 0, 2))).reshape(V, B * F)
    T = [xt]
    if K > 1:
        T.append(L @ xt)
    for k in range(2, K):
        T.append(2.0 * (L @ T[k - 1]) - T[k - 2])
    return [np.transpose(t.reshape(V, B, F), (1, 0, 2)) for t in T]


def cheb_forward_f64(rowptr, colind, values, x, weight, bias=None):
    """``Y[b,v,o] = sum_{f,k} T_k[b,v,f] W[f,k,o] (+ bias)`` in float64."""
    B, V, Fin = x.shape
    Fin_w, K, Fout = weight.shape
    assert Fin == Fin_w
    L = _csr64(rowptr, colind, values, (V, V))
    T = cheb_basis_f64(L, x, K)
    w = weight.astype(np.float64)
    y = np.zeros((B, V, Fout), dtype=np.float64)
    for k in range(K):
        y += T[k] @ w[:, k, :]
    if bias is not None:
        y += bias.astype(np.float64)
    return y


def cheb_backward_f64(rowptr, colind, values, x, weight, grad_out, has_bias=True):
    """Hand-derived backward (valid for non-symmetric L):

    ``dW[f,k,o] = sum_{b,v} T_k[b,v,f] dY[b,v,o]``;  ``G_k = dY W[:,k,:]^T``;
    for k = K-1..2: ``G_{k-1} += 2 L^T G_k; G_{k-2} -= G_k``; if K>1 ``G_0 += L^T G_1``;
    ``dX = G_0``; ``db = sum_{b,v} dY``.
    """
    B, V, Fin = x.shape
    _, K, Fout = weight.shape
    L = _csr64(rowptr, colind, values, (V, V))
    Lt = L.T.tocsr()
    T = cheb_basis_f64(L, x, K)
    gy = grad_out.astype(np.float64)
    w = weight.astype(np.float64)
    dw = np.zeros((Fin, K, Fout), dtype=np.float64)
    G = []
    for k in range(K):
        dw[:, k, :] = T[k].reshape(B * V, Fin).T @ gy.reshape(B * V, Fout)   # = einsum("bvf,bvo->fo") via BLAS
        G.append(gy @ w[:, k, :].T)

    def apply_t(g):
        gt = np.ascontiguousarray(np.transpose(g, (1, 0, 2))).reshape(V, B * Fin)
        return np.transpose((Lt @ gt).reshape(V, B, Fin), (1, 0, 2))

    for k in range(K - 1, 1, -1):
        G[k - 1] = G[k - 1] + 2.0 * apply_t(G[k])
        G[k - 2] = G[k - 2] - G[k]
    if K > 1:
        G[0] = G[0] + apply_t(G[1])
    db = gy.sum(axis=(0, 1)) if has_bias else None
    return G[0], dw, db


def remap_f64(rowptr, colind, values, shape, x):
    """``Y[b,d,f] = sum_v M[d,v] X[b,v,f]`` in float64."""
    M = _csr64(rowptr, colind, values, shape)
    B, V, F = x.shape
    xt = np.ascontiguousarray(np.transpose(x.astype(np.float64), (1, 0, 2))).reshape(V, B * F)
    return np.transpose((M @ xt).reshape(shape[0], B, F), (1, 0, 2))


def remap_backward_f64(rowptr, colind, values, shape, grad_out):
    """``dX[b,v,f] = sum_d M[d,v] dY[b,d,f]``."""
    M = _csr64(rowptr, colind, values, shape).T.tocsr()
    B, D, F = grad_out.shape
    gt = np.ascontiguousarray(np.transpose(grad_out.astype(np.float64), (1, 0, 2))).reshape(D, B * F)
    return np.transpose((M @ gt).reshape(shape[1], B, F), (1, 0, 2))


def _to_f64(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().to("cpu", torch.float64).numpy()
    return np.asarray(a, dtype=np.float64)


def max_rel_err(a, ref):
    """max |a - ref| / max |ref|  (the normalisation used for the tolerances in SURVEY 8c)."""
    a = _to_f64(a)
    ref = _to_f64(ref)
    denom = np.max(np.abs(ref))
    if denom == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - ref)) / denom)


# --------------------------------------------------------------------------------------
# Max-value pooling / unpooling  (reference: modules/layers.py:1040-1103)
# --------------------------------------------------------------------------------------
def maxval_pool_torch(matrix, x):
    """``GeneralMaxValPool.forward`` (reference ``modules/layers.py:1043-1079``), same torch op sequence; the Python
    ``Counter`` over the row indices (``:1056-1057``) is replaced by ``torch.bincount`` (same per-row counts).
    Returns ``(x_pooled [B, Vd, F] (permuted view), nnz_ind [2, B*F*Vd])``."""
    n_batch, n_nodes, n_val = x.shape
    new_nodes, old_nodes = matrix.shape
    assert n_nodes == old_nodes, "remap_matrix.shape[1] != x.shape[1]"
    x = x.permute(1, 2, 0).reshape(n_nodes, n_batch * n_val)                       # :1049
    row, col = matrix.indices()
    weights = matrix.values()
    counts = torch.bincount(row, minlength=new_nodes)
    kernel_sizes = [int(c) for c in counts if int(c) > 0]                          # :1056-1057
    col = col.repeat(n_batch * n_val, 1).T                                         # :1059
    val = torch.gather(x, dim=0, index=col).detach()                               # :1061
    weighted_val = weights.view(-1, 1) * val                                       # :1063
    start_row, max_val_index = 0, []
    for k in kernel_sizes:                                                         # :1065-1070
        curr = weighted_val[start_row:start_row + k]
        max_val_index.append(torch.argmax(curr, dim=0) + start_row)
        start_row += k
    max_val_index = torch.stack(max_val_index)
    nnz_row = torch.gather(col, dim=0, index=max_val_index)                        # :1071
    x_pooled = torch.gather(x, dim=0, index=nnz_row)                               # :1073
    nnz_col = torch.arange(x_pooled.shape[1]).expand(x_pooled.shape[0], -1)        # np.indices(...)[1], :1075
    nnz_ind = torch.stack([nnz_row, nnz_col], dim=2).permute(1, 0, 2).reshape(-1, 2).T   # :1077-1078
    return x_pooled.reshape(new_nodes, n_val, n_batch).permute(2, 0, 1), nnz_ind


def maxval_unpool_torch(new_nodes, x, index):
    """``GeneralMaxValUnpool.forward`` (reference ``modules/layers.py:1085-1103``)."""
    n_batch, _, n_val = x.shape
    flat = x.permute(2, 0, 1).flatten()
    out = torch.zeros([new_nodes, n_batch * n_val], dtype=x.dtype, device=x.device)
    row, col = index
    out = torch.index_put(out, (row, col), flat)
    return out.reshape(new_nodes, n_val, n_batch).permute(2, 0, 1)


def maxval_pool_np(rowptr, colind, values, x):
    """Independent numpy statement of the same selection in the native ``[B, V, F]`` layout: returns
    ``(y [B, Vd, F], sel int32 [B, Vd, F])`` - first maximum of ``w * x`` over the row's non-zeros."""
    B, V, F = x.shape
    D = len(rowptr) - 1
    y = np.zeros((B, D, F), dtype=x.dtype)
    sel = np.full((B, D, F), -1, dtype=np.int32)
    for d in range(D):
        cols = np.asarray(colind[rowptr[d]:rowptr[d + 1]])
        if cols.size == 0:
            continue
        cand = x[:, cols, :]                                                      # [B, k, F]
        score = cand.astype(np.float32) * np.asarray(values[rowptr[d]:rowptr[d + 1]], dtype=np.float32)[None, :, None]
        arg = np.argmax(score, axis=1)                                            # first maximum
        y[:, d, :] = np.take_along_axis(cand, arg[:, None, :], axis=1)[:, 0, :]
        sel[:, d, :] = cols[arg]
    return y, sel


def maxval_pool_backward_np(sel, v_fine, gy):
    """``dx[b, sel[b,d,f], f] += gy[b,d,f]`` (autograd of the gather)."""
    B, D, F = gy.shape
    dx = np.zeros((B, v_fine, F), dtype=np.float64)
    b, f = np.meshgrid(np.arange(B), np.arange(F), indexing="ij")
    for d in range(D):
        np.add.at(dx, (b, sel[:, d, :], f), gy[:, d, :])
    return dx


def maxval_unpool_np(sel, v_fine, x):
    """``y[b, sel[b,d,f], f] = x[b,d,f]``, increasing d (the last write wins), zeros elsewhere."""
    B, D, F = x.shape
    y = np.zeros((B, v_fine, F), dtype=x.dtype)
    b, f = np.meshgrid(np.arange(B), np.arange(F), indexing="ij")
    for d in range(D):
        y[b, sel[:, d, :], f] = x[:, d, :]
    return y


def maxval_unpool_backward_np(sel, gy):
    """``dx[b,d,f] = gy[b, sel[b,d,f], f]``."""
    B, D, F = sel.shape
    b, f = np.meshgrid(np.arange(B), np.arange(F), indexing="ij")
    return np.stack([gy[b, sel[:, d, :], f] for d in range(D)], axis=1)
