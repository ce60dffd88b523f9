"""First-order conservative remapping weights between two spherical samplings, self-contained.

The reference obtains its pooling / unpooling matrices from ``xsphere.compute_interpolation_weights(method=
"conservative", normalization="fracarea")`` (``/root/reference/modules/layers.py:531-581``): xsphere builds the
spherical Voronoi mesh of each sampling's nodes and lets the CDO binary (``gencon``) integrate the overlap areas of the
cells.  Neither xsphere nor CDO is available, so this module restates the published algorithm:

1. cells = spherical Voronoi regions of the node positions (``scipy.spatial.SphericalVoronoi``) - convex geodesic
   polygons;
2. overlap area of a (destination, source) cell pair = area of the intersection polygon, obtained by clipping the
   source polygon against the great-circle half-spaces of the destination polygon's edges (Sutherland-Hodgman on the
   sphere; exact for convex geodesic polygons) and summing signed spherical triangle areas (Van Oosterom-Strackee);
3. ``remap_matrix[d, s] = overlap[d, s] / area[d]`` ("fracarea": destination rows sum to 1).

All pairs are clipped at once in padded numpy arrays (no Python loop over cells).  PARITY UNPINNED against CDO's
numbers (absent); what IS checked (tests/test_sphere_conservative.py) are the invariants the reference itself asserts
on CDO's output (layers.py:540-571): rows sum to 1, ``W^T dst_area = src_area``, cell areas tile the sphere.
"""
from __future__ import annotations

import numpy as np
from scipy import sparse
from scipy.spatial import SphericalVoronoi, cKDTree

__all__ = ["voronoi_cells", "polygon_areas", "overlap_areas", "conservative_weights", "RemapWeights"]


def _normalise(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def polygon_areas(P, n):
    """Signed areas of geodesic polygons ``P [m, M, 3]`` (unit vectors, ``n [m]`` valid vertices each, counter-clockwise
    seen from outside = positive) as a fan of spherical triangles (Van Oosterom & Strackee 1983)."""
    m, M, _ = P.shape
    area = np.zeros(m)
    a = P[:, 0]
    for i in range(1, M - 1):
        b, c = P[:, i], P[:, i + 1]
        num = np.einsum("ij,ij->i", a, np.cross(b, c))
        den = 1.0 + np.einsum("ij,ij->i", a, b) + np.einsum("ij,ij->i", b, c) + np.einsum("ij,ij->i", c, a)
        tri = 2.0 * np.arctan2(num, den)
        area += np.where(i + 1 < n, tri, 0.0)
    return area


def voronoi_cells(coords):
    """Spherical Voronoi cells of unit vectors ``coords [V, 3]``: padded vertex array ``[V, M, 3]`` (counter-clockwise),
    vertex counts ``[V]`` and cell areas ``[V]`` (sum = 4 pi)."""
    sv = SphericalVoronoi(np.asarray(coords, dtype=np.float64), radius=1.0, center=np.zeros(3))
    sv.sort_vertices_of_regions()
    # ragged region lists -> padded array; coincident consecutive vertices are merged first (at a pole of an equiangular
    # grid hundreds of Voronoi vertices coincide: the ring of cells around it would otherwise drag the padded width of
    # EVERY cell to its vertex count)
    regions = []
    for r in sv.regions:
        v = sv.vertices[r]
        keep = np.linalg.norm(v - np.roll(v, -1, axis=0), axis=1) > 1e-11
        if keep.sum() >= 3:
            r = [ri for ri, k in zip(r, keep) if k]
        regions.append(r)
    counts = np.array([len(r) for r in regions])
    M = int(counts.max())
    idx = np.zeros((len(regions), M), dtype=np.int64)
    for i, r in enumerate(regions):
        idx[i, : len(r)] = r
        idx[i, len(r):] = r[0]
    P = sv.vertices[idx]
    area = polygon_areas(P, counts)
    flip = area < 0                         # sort_vertices_of_regions leaves the orientation free: make all CCW
    if flip.any():
        for i in np.nonzero(flip)[0]:
            k = counts[i]
            P[i, :k] = P[i, :k][::-1]
        area = np.abs(area)
    return P, counts, area


def _clip(P, n, nrm):
    """Clip polygons ``P [m, M, 3]`` (``n`` valid vertices) by the half-spaces ``x . nrm >= 0`` (``nrm [m, 3]``).
    Returns the clipped polygons padded to ``M + 1`` vertices and their vertex counts."""
    m, M, _ = P.shape
    ar = np.arange(M)[None, :]
    valid = ar < n[:, None]
    nxt = np.where(ar + 1 < n[:, None], ar + 1, 0)
    Pn = np.take_along_axis(P, nxt[:, :, None], axis=1)
    d = np.einsum("ijk,ik->ij", P, nrm)
    dn = np.take_along_axis(d, nxt, axis=1)
    inside = (d >= 0) & valid
    cross = ((d >= 0) != (dn >= 0)) & valid
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(cross, d / (d - dn), 0.0)
    X = P + t[:, :, None] * (Pn - P)        # the chord point with x . nrm = 0; normalised -> on the geodesic edge
    X = X / np.maximum(np.linalg.norm(X, axis=2, keepdims=True), 1e-300)
    emit = inside.astype(np.int64) + cross.astype(np.int64)
    pos = np.cumsum(emit, axis=1) - emit    # output slot of each vertex' first emitted point
    n_out = emit.sum(axis=1)
    out = np.zeros((m, M + 1, 3))
    rows = np.repeat(np.arange(m)[:, None], M, axis=1)
    out[rows[inside], pos[inside]] = P[inside]
    pc = pos + inside.astype(np.int64)
    out[rows[cross], pc[cross]] = X[cross]
    # pad the tail with the first vertex (keeps the triangle fan of polygon_areas degenerate there)
    tail = np.arange(M + 1)[None, :] >= n_out[:, None]
    out = np.where(tail[:, :, None], out[:, :1], out)
    return out, n_out


def overlap_areas(src_P, src_n, dst_P, dst_n, pairs_d, pairs_s):
    """Area of ``dst cell d  intersect  src cell s`` for every listed pair (vectorised over the pairs)."""
    P = src_P[pairs_s]
    n = src_n[pairs_s].copy()
    Q, qn = dst_P[pairs_d], dst_n[pairs_d]
    Mq = Q.shape[1]
    for j in range(Mq):
        active = j < qn
        if not active.any():
            break
        a = Q[:, j]
        b = np.take_along_axis(Q, np.where(j + 1 < qn, j + 1, 0)[:, None, None].repeat(3, axis=2), axis=1)[:, 0]
        nrm = np.cross(a, b)                # inward normal of edge j of a counter-clockwise polygon
        nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
        Pc, nc = _clip(P, n, nrm)
        # pairs whose destination polygon has fewer than j + 1 edges keep their polygon
        keep = ~active
        if keep.any():
            Pk = np.concatenate([P, P[:, :1]], axis=1)
            Pc = np.where(keep[:, None, None], Pk, Pc)
            nc = np.where(keep, n, nc)
        P, n = Pc, nc
    area = polygon_areas(P, n)
    return np.where(n >= 3, np.maximum(area, 0.0), 0.0)


class RemapWeights:
    """What the reference reads from xsphere's dataset (layers.py:546-553): addresses, fracarea weights, cell areas."""

    def __init__(self, dst_address, src_address, remap_matrix, src_grid_area, dst_grid_area):
        self.dst_address = dst_address
        self.src_address = src_address
        self.remap_matrix = remap_matrix
        self.src_grid_area = src_grid_area
        self.dst_grid_area = dst_grid_area


def conservative_weights(src_coords, dst_coords, min_frac=1e-12) -> RemapWeights:
    """First-order conservative remap weights from the sampling ``src`` to ``dst`` (0-based addresses)."""
    src_P, src_n, src_area = voronoi_cells(src_coords)
    dst_P, dst_n, dst_area = voronoi_cells(dst_coords)
    # candidate pairs: centres closer than the sum of the two circumradii
    def radius(P, n, c):
        dots = np.einsum("ijk,ik->ij", P, c)
        ang = np.arccos(np.clip(dots, -1.0, 1.0))
        return np.where(np.arange(P.shape[1])[None, :] < n[:, None], ang, 0.0).max(axis=1)

    src_c, dst_c = _normalise(np.asarray(src_coords, float)), _normalise(np.asarray(dst_coords, float))
    r_src, r_dst = radius(src_P, src_n, src_c), radius(dst_P, dst_n, dst_c)
    tree = cKDTree(src_c)
    chord = lambda ang: 2.0 * np.sin(np.minimum(ang, np.pi) / 2.0)
    lists = tree.query_ball_point(dst_c, chord(r_dst + r_src.max()) * (1 + 1e-9))
    pd_ = np.repeat(np.arange(len(lists)), [len(l) for l in lists])
    ps_ = np.fromiter((s for l in lists for s in l), dtype=np.int64, count=pd_.size)
    sep = np.arccos(np.clip(np.einsum("ij,ij->i", dst_c[pd_], src_c[ps_]), -1.0, 1.0))
    close = sep <= (r_dst[pd_] + r_src[ps_]) * (1 + 1e-9)
    pd_, ps_ = pd_[close], ps_[close]
    area = np.empty(pd_.size)
    # pairs are clipped in groups of similar vertex counts: the padded work arrays are as wide as the widest polygon
    # of the group (a handful of cells - around poles, at face corners - have many more vertices than the rest)
    key = np.maximum(src_n[ps_], 8) * 1000 + np.maximum(dst_n[pd_], 8)
    order = np.argsort(key, kind="stable")
    bounds = np.flatnonzero(np.diff(key[order])) + 1
    step = 200_000                           # bound the padded work arrays
    for lo, hi in zip(np.concatenate([[0], bounds]), np.concatenate([bounds, [order.size]])):
        grp = order[lo:hi]
        ms, md = int(src_n[ps_[grp]].max()), int(dst_n[pd_[grp]].max())
        for i in range(0, grp.size, step):
            sel = grp[i:i + step]
            area[sel] = overlap_areas(src_P[:, :ms], src_n, dst_P[:, :md], dst_n, pd_[sel], ps_[sel])
    keep = area > min_frac * dst_area[pd_]
    pd_, ps_, area = pd_[keep], ps_[keep], area[keep]
    return RemapWeights(pd_, ps_, area / dst_area[pd_], src_area, dst_area)


def interpolation_matrix(src_coords, dst_coords):
    """The unnormalised overlap matrix ``[V_dst, V_src]`` of ``_build_interpolation_matrix`` (layers.py:529-573),
    with the reference's own sanity checks on the weights (loosened from assert_allclose's 1e-7 only where the
    polygon arithmetic needs it)."""
    ds = conservative_weights(src_coords, dst_coords)
    weights = sparse.csr_matrix((ds.remap_matrix, (ds.dst_address, ds.src_address)),
                                shape=(len(ds.dst_grid_area), len(ds.src_grid_area)))
    np.testing.assert_allclose(np.asarray(weights.sum(axis=1)).ravel(), 1, rtol=1e-9)              # :557
    np.testing.assert_allclose(weights.T @ ds.dst_grid_area, ds.src_grid_area, rtol=1e-7)          # :559
    weights = weights.multiply(ds.dst_grid_area[:, np.newaxis])                                    # :562
    np.testing.assert_allclose(np.asarray(weights.sum(1)).squeeze(), ds.dst_grid_area, rtol=1e-9)  # :565
    np.testing.assert_allclose(np.asarray(weights.sum(0)).squeeze(), ds.src_grid_area, rtol=1e-7)  # :566
    return sparse.csr_matrix(weights), ds
