"""Deterministic data recipes shared by ``make_golden.py`` (generation, build container only)
and the parity tests (replay, everywhere).  Pure numpy; nothing here comes from the reference."""
import numpy as np


def rand(seed, shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def unet_param_fill(index, name, shape):
    """Seeded value for parameter ``name`` (``index`` = position in name-sorted order)."""
    rng = np.random.default_rng(5000 + index)
    if name.endswith("rezero_weight") or name.endswith("res_increment"):
        return np.full(shape, 0.5 + 0.05 * (index % 5), dtype=np.float32)
    if len(shape) == 1:
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    fan = int(np.prod(shape[:-1])) if not name.endswith("res_connection.weight") else shape[-1]
    return (rng.standard_normal(shape) * np.sqrt(1.0 / fan)).astype(np.float32)


def grad_probe(index, g):
    """Compact fingerprint of a gradient tensor: (l2, dot with a seeded vector, first 32 values)."""
    g = np.asarray(g, dtype=np.float64).ravel()
    r = np.random.default_rng(9000 + index).standard_normal(g.size)
    head = np.zeros(32, dtype=np.float64)
    head[: min(32, g.size)] = g[:32]
    return np.concatenate([[np.sqrt((g * g).sum()), (g * r).sum()], head])


def irregular_operator(n, seed, min_deg=5, max_deg=200, symmetric=False):
    """Random sparse operator with row lengths in [min_deg, max_deg] (CSR arrays, float32)."""
    rng = np.random.default_rng(seed)
    deg = rng.integers(min_deg, min(max_deg, n) + 1, size=n)
    # a few heavy rows, many light rows (pole-like distribution)
    heavy = rng.random(n) < 0.05
    deg = np.where(heavy, deg, np.minimum(deg, min_deg + rng.integers(0, 12, size=n)))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    colind = np.empty(rowptr[-1], dtype=np.int64)
    for i in range(n):
        colind[rowptr[i] : rowptr[i + 1]] = np.sort(rng.choice(n, size=deg[i], replace=False))
    vals = (rng.standard_normal(rowptr[-1]) / np.sqrt(np.repeat(deg, deg))).astype(np.float32)
    if symmetric:
        from scipy import sparse

        m = sparse.csr_matrix((vals, colind, rowptr), shape=(n, n))
        m = ((m + m.T) * 0.5).tocsr()
        m.sort_indices()
        return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)
    return rowptr.astype(np.int32), colind.astype(np.int32), vals


def ar_area_weights(n):
    """Seeded, non-uniform positive node weights summing to 1 (stand-in for loss.AreaWeights, which needs CDO)."""
    w = 0.5 + np.random.default_rng(4242).random(n)
    return (w / w.sum()).astype(np.float32)
