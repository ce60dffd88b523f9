"""Spherical samplings -> k-NN graphs -> normalized Laplacians and pooling matrices.

Host-side (numpy / scipy) builder of the *operators* the hot path consumes.  The
reference gets these from the un-vendored ``pygsp@sphere-graphs`` package
(``/root/reference/modules/models.py:43-46``, ``modules/utils_models.py:11-20``) and,
for interpolation pooling, from ``xsphere`` + the CDO binary
(``/root/reference/modules/layers.py:531-581``).  None of those are available, so this
module provides self-contained stand-ins written from the public HEALPix geometry
(Gorski et al. 2005) and a plain symmetrised k-NN Gaussian graph.

PARITY UNPINNED: the edge weights are *not* claimed to equal pygsp's (which uses a
per-(k, nside) optimal kernel width table), and the conservative remap weights
(``dsw_amd.conservative``: overlap areas of the spherical Voronoi meshes) are checked
against the invariants the reference asserts on CDO's output (``modules/layers.py:540-571``),
not against CDO's numbers (CDO is absent).
Everything downstream (ConvCheb, RemapBlock) takes the operator as an input, so parity
tests always feed the *same prepared operator* to reference and build.
"""
from __future__ import annotations

import numpy as np
from scipy import sparse
from scipy.spatial import cKDTree

__all__ = [
    "healpix_pix2vec",
    "healpix_nest2ring",
    "equiangular_vec",
    "knn_graph_laplacian",
    "SphereHealpix",
    "SphereEquiangular",
    "healpix_pool_matrices",
    "equiangular_pool_matrices",
    "knn_interp_pool_matrices",
    "conservative_pool_matrices",
    "cell_areas",
    "build_pooling_matrices",
]

_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)


def _compress_bits(v: np.ndarray) -> np.ndarray:
    """Keep the even bits of ``v`` and pack them (inverse of Morton spreading)."""
    v = v & 0x5555555555555555
    v = (v | (v >> 1)) & 0x3333333333333333
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
    v = (v | (v >> 16)) & 0x00000000FFFFFFFF
    return v


def _nest2xyf(nside: int, pix: np.ndarray):
    npface = nside * nside
    face = pix // npface
    p = pix % npface
    ix = _compress_bits(p)
    iy = _compress_bits(p >> 1)
    return ix, iy, face


def _xyf2loc(nside: int, ix, iy, face):
    """(x, y, face) -> (z, phi) following the HEALPix base-pixel geometry."""
    npix = 12 * nside * nside
    fact2 = 4.0 / npix
    fact1 = (2 * nside) * fact2
    jr = _JRLL[face] * nside - ix - iy - 1
    north = jr < nside
    south = jr > 3 * nside
    nr = np.where(north, jr, np.where(south, 4 * nside - jr, nside))
    z = np.where(
        north,
        1.0 - nr.astype(np.float64) ** 2 * fact2,
        np.where(south, nr.astype(np.float64) ** 2 * fact2 - 1.0, (2 * nside - jr) * fact1),
    )
    kshift = np.where(north | south, 0, (jr - nside) & 1)
    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > 4 * nr, jp - 4 * nr, jp)
    jp = np.where(jp < 1, jp + 4 * nr, jp)
    phi = (jp - (kshift + 1) * 0.5) * (np.pi / 2.0 / nr)
    return z, phi, jr, jp, nr


def _ring2loc(nside: int, pix: np.ndarray):
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    fact2 = 4.0 / npix
    fact1 = (2 * nside) * fact2
    z = np.empty(pix.shape, dtype=np.float64)
    phi = np.empty(pix.shape, dtype=np.float64)
    # north polar cap
    m = pix < ncap
    p = pix[m]
    iring = (1 + np.floor(np.sqrt(1 + 2 * p)).astype(np.int64)) >> 1
    iphi = p + 1 - 2 * iring * (iring - 1)
    z[m] = 1.0 - iring.astype(np.float64) ** 2 * fact2
    phi[m] = (iphi - 0.5) * (np.pi / 2.0) / iring
    # equatorial belt
    m = (pix >= ncap) & (pix < npix - ncap)
    ip = pix[m] - ncap
    tmp = ip // (4 * nside)
    iring = tmp + nside
    iphi = ip - 4 * nside * tmp + 1
    fodd = np.where(((iring + nside) & 1) == 1, 1.0, 0.5)
    z[m] = (2 * nside - iring) * fact1
    phi[m] = (iphi - fodd) * (np.pi / 2.0) / nside
    # south polar cap
    m = pix >= npix - ncap
    ip = npix - pix[m]
    iring = (1 + np.floor(np.sqrt(2 * ip - 1)).astype(np.int64)) >> 1
    iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1))
    z[m] = -1.0 + iring.astype(np.float64) ** 2 * fact2
    phi[m] = (iphi - 0.5) * (np.pi / 2.0) / iring
    return z, phi


def healpix_pix2vec(nside: int, nest: bool = True) -> np.ndarray:
    """Unit vectors ``[12*nside**2, 3]`` of all HEALPix pixel centres."""
    if nside < 1 or (nside & (nside - 1)) != 0:
        raise ValueError("nside must be a power of two")
    pix = np.arange(12 * nside * nside, dtype=np.int64)
    if nest:
        ix, iy, face = _nest2xyf(nside, pix)
        z, phi, *_ = _xyf2loc(nside, ix, iy, face)
    else:
        z, phi = _ring2loc(nside, pix)
    st = np.sqrt(np.clip(1.0 - z * z, 0.0, None))
    return np.stack([st * np.cos(phi), st * np.sin(phi), z], axis=1)


def healpix_nest2ring(nside: int) -> np.ndarray:
    """``ring_index[nest_index]`` permutation for all pixels."""
    pix = np.arange(12 * nside * nside, dtype=np.int64)
    ix, iy, face = _nest2xyf(nside, pix)
    _, _, jr, jp, nr = _xyf2loc(nside, ix, iy, face)
    ncap = 2 * nside * (nside - 1)
    npix = 12 * nside * nside
    north = jr < nside
    south = jr > 3 * nside
    n_before = np.where(
        north,
        2 * nr * (nr - 1),
        np.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - nside) * 4 * nside),
    )
    return n_before + jp - 1


def equiangular_vec(nlat: int, nlon: int):
    """Cell-centre unit vectors of an equiangular grid, row-major (lat, lon)."""
    lat = np.pi / 2.0 - (np.arange(nlat) + 0.5) * np.pi / nlat
    lon = np.arange(nlon) * 2.0 * np.pi / nlon
    lat2, lon2 = np.meshgrid(lat, lon, indexing="ij")
    lat2 = lat2.ravel()
    lon2 = lon2.ravel()
    xyz = np.stack(
        [np.cos(lat2) * np.cos(lon2), np.cos(lat2) * np.sin(lon2), np.sin(lat2)], axis=1
    )
    return xyz, lat2, lon2


def icosahedral_vec(m: int, dual: bool = False) -> np.ndarray:
    """Vertices of the icosahedron with every edge cut into ``m`` segments and every face into ``m^2`` triangles,
    projected onto the sphere (public geometry; pygsp's ``SphereIcosahedral(subdivisions=m)``): ``10 m^2 + 2`` points,
    or the ``20 m^2`` face centres with ``dual``.  Order: face by face (points shared with an earlier face are not
    repeated), i.e. consecutive rows are neighbours on the sphere."""
    if m < 1:
        raise ValueError("subdivisions must be >= 1")
    phi = (1.0 + np.sqrt(5.0)) / 2.0
    base = np.array([[-1, phi, 0], [1, phi, 0], [-1, -phi, 0], [1, -phi, 0], [0, -1, phi], [0, 1, phi],
                     [0, -1, -phi], [0, 1, -phi], [phi, 0, -1], [phi, 0, 1], [-phi, 0, -1], [-phi, 0, 1]], dtype=np.float64)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
             (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
             (9, 8, 1)]
    pts = []
    for a, b, c in faces:
        A, Bv, C = base[a], base[b], base[c]
        if dual:     # centres of the m^2 small triangles
            for i in range(m):
                for j in range(m - i):
                    p0 = (A * (m - i - j) + Bv * i + C * j)
                    p1 = (A * (m - i - j - 1) + Bv * (i + 1) + C * j)
                    p2 = (A * (m - i - j - 1) + Bv * i + C * (j + 1))
                    pts.append((p0 + p1 + p2) / 3.0)
                    if i + j < m - 1:
                        p3 = (A * (m - i - j - 2) + Bv * (i + 1) + C * (j + 1))
                        pts.append((p1 + p2 + p3) / 3.0)
        else:
            for i in range(m + 1):
                for j in range(m + 1 - i):
                    pts.append(A * (m - i - j) + Bv * i + C * j)
    pts = np.asarray(pts)
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    if not dual:     # vertices on shared edges / corners appear once per face: keep the first occurrence
        key = np.round(pts * 1e9).astype(np.int64)
        _, first = np.unique(key, axis=0, return_index=True)
        pts = pts[np.sort(first)]
    return pts


def cubed_vec(m: int, spacing: str = "equiangular") -> np.ndarray:
    """Cell centres of the cubed sphere with ``m x m`` cells per face (``6 m^2`` points; pygsp's ``SphereCubed``):
    ``equiangular`` spacing cuts every face into equal angles (tan of a uniform angle in [-pi/4, pi/4]), ``equidistant``
    into equal steps on the cube face.  Order: face by face, row-major inside a face."""
    if m < 1:
        raise ValueError("subdivisions must be >= 1")
    if spacing == "equiangular":
        t = np.tan(-np.pi / 4 + (np.arange(m) + 0.5) * (np.pi / 2) / m)
    elif spacing == "equidistant":
        t = -1.0 + (np.arange(m) + 0.5) * 2.0 / m
    else:
        raise ValueError("spacing must be 'equiangular' or 'equidistant'")
    a, b = np.meshgrid(t, t, indexing="ij")
    a, b, one = a.ravel(), b.ravel(), np.ones(m * m)
    faces = [np.stack(f, axis=1) for f in ((one, a, b), (-a, one, b), (-one, -a, b), (a, -one, b), (-b, a, one), (b, a, -one))]
    pts = np.concatenate(faces)
    return pts / np.linalg.norm(pts, axis=1, keepdims=True)


def gauss_legendre_vec(nlat: int, nlon="ecmwf-octahedral"):
    """Gauss-Legendre sampling (pygsp's ``SphereGaussLegendre``): latitudes at the roots of the Legendre polynomial of
    degree ``nlat``; ``nlon`` an int (the same number of longitudes on every ring; default of pygsp: 2 nlat) or
    ``'ecmwf-octahedral'`` - the reduced grid of ECMWF's O-grids: ``4 i + 16`` longitudes on the i-th ring from a pole
    (O24 = ``nlat`` 48: 20, 24, .., 112, .., 24, 20 = 3168 points).  Order: ring by ring from north to south."""
    if nlat < 2 or nlat % 2:
        raise ValueError("nlat must be an even number >= 2")
    x, _ = np.polynomial.legendre.leggauss(nlat)          # roots in ascending order: sin(latitude) from south to north
    lat = np.arcsin(x[::-1])
    if isinstance(nlon, str):
        if nlon != "ecmwf-octahedral":
            raise ValueError("nlon must be an int or 'ecmwf-octahedral'")
        half = 4 * np.arange(1, nlat // 2 + 1) + 16
        per_ring = np.concatenate([half, half[::-1]])
    else:
        per_ring = np.full(nlat, int(nlon))
    lat2 = np.repeat(lat, per_ring)
    lon2 = np.concatenate([np.arange(n) * 2.0 * np.pi / n for n in per_ring])
    xyz = np.stack([np.cos(lat2) * np.cos(lon2), np.cos(lat2) * np.sin(lon2), np.sin(lat2)], axis=1)
    return xyz, lat2, lon2


# Published kernel widths of the HEALPix k-NN graphs (DeepSphere, Defferrard et al. 2020, table of "optimal" widths for
# equivariance): they scale with the pixel size, ~ c_k / nside; c_k below are the nside = 32 values times 32.  pygsp's
# sphere-graphs branch keeps the full per-(k, nside) table (`_OPTIMAL_KERNEL_WIDTHS`), which is not vendored with the
# reference: these figures are quoted from the paper, the 1 / nside scaling is an approximation (the table's own entries
# drift from it by ~2 %) - parity with pygsp's graphs stays UNPINNED.
HEALPIX_KERNEL_WIDTH_TIMES_NSIDE = {8: 0.02500 * 32, 20: 0.03185 * 32, 40: 0.042432 * 32, 60: 0.051720 * 32}


def knn_graph_laplacian(coords: np.ndarray, k: int, lap_type: str = "normalized", kernel_width=None):
    """Symmetrised k-NN Gaussian graph and its Laplacian (scipy CSR, float64).

    Weight ``exp(-d^2 / (2 s^2))``.  ``kernel_width``: ``None`` - ``s^2`` = the mean SQUARED k-NN distance (this
    module's rule since round 1); ``"mean"`` - ``s`` = the mean k-NN distance (the default rule of pygsp's ``NNGraph``);
    a float - that width (e.g. ``healpix_kernel_width``).  Symmetrisation keeps an edge if either endpoint selected it
    (so degrees are >= k and irregular where the sampling is anisotropic, e.g. equiangular poles).
    """
    n = coords.shape[0]
    if k >= n:
        raise ValueError(f"a {k}-nearest-neighbour graph needs more than {k} vertices (the sampling has {n})")
    tree = cKDTree(coords)
    dist, idx = tree.query(coords, k=k + 1)
    dist = dist[:, 1:]
    idx = idx[:, 1:]
    if kernel_width is None:
        s2 = float(np.mean(dist**2))
    elif isinstance(kernel_width, str):
        if kernel_width != "mean":
            raise ValueError("kernel_width must be None, 'mean' or a positive number")
        s2 = float(np.mean(dist)) ** 2
    else:
        if not kernel_width > 0:
            raise ValueError("kernel_width must be None, 'mean' or a positive number")
        s2 = float(kernel_width) ** 2
    w = np.exp(-(dist**2) / (2.0 * s2))
    rows = np.repeat(np.arange(n), k)
    W = sparse.csr_matrix((w.ravel(), (rows, idx.ravel())), shape=(n, n))
    W = W.maximum(W.T).tocsr()
    W.setdiag(0)
    W.eliminate_zeros()
    d = np.asarray(W.sum(axis=1)).ravel()
    if lap_type == "combinatorial":
        L = sparse.diags(d) - W
    elif lap_type == "normalized":
        dinv = 1.0 / np.sqrt(d)
        L = sparse.identity(n) - sparse.diags(dinv) @ W @ sparse.diags(dinv)
    else:
        raise ValueError("unknown lap_type")
    L = sparse.csr_matrix(L)
    L.sort_indices()
    return W, L


class _SphereGraph:
    """Minimal stand-in for the pygsp graph objects the reference touches:
    ``.L``, ``.W``, ``.n_vertices``, ``.coords``, ``.signals['lat'|'lon']``
    (``/root/reference/modules/models.py:54``, ``modules/layers.py:540-543``)."""

    def __init__(self, coords, k, lap_type, kernel_width=None):
        self.coords = coords
        self.n_vertices = coords.shape[0]
        self.k = k
        self.lap_type = lap_type
        self.kernel_width = kernel_width
        self.W, self.L = knn_graph_laplacian(coords, k, lap_type, kernel_width)
        lat = np.degrees(np.arcsin(np.clip(coords[:, 2], -1, 1)))
        lon = np.degrees(np.arctan2(coords[:, 1], coords[:, 0])) % 360.0
        self.signals = {"lat": lat, "lon": lon}


def healpix_kernel_width(k: int, nside: int) -> float:
    """Kernel width pygsp's ``SphereHealpix`` would use (its per-(k, nside) table, approximated: see
    ``HEALPIX_KERNEL_WIDTH_TIMES_NSIDE``); raises for a ``k`` the published table does not have, as pygsp does."""
    if k not in HEALPIX_KERNEL_WIDTH_TIMES_NSIDE:
        raise ValueError("No known optimal kernel width for {} neighbors and nside={}.".format(k, nside))
    return HEALPIX_KERNEL_WIDTH_TIMES_NSIDE[k] / float(nside)


class SphereHealpix(_SphereGraph):
    """HEALPix k-NN graph (``subdivisions`` = nside), nested or ring order.  ``kernel_width="optimal"`` takes the
    published width for (k, nside) - the rule of pygsp's class -, a float that width, the default this module's own rule."""

    def __init__(self, subdivisions=2, nest=False, k=20, lap_type="normalized", kernel_width=None, **kwargs):
        self.subdivisions = int(subdivisions)
        self.nest = bool(nest)
        if isinstance(kernel_width, str) and kernel_width == "optimal":
            kernel_width = healpix_kernel_width(k, self.subdivisions)
        super().__init__(healpix_pix2vec(self.subdivisions, self.nest), k, lap_type, kernel_width)


class SphereEquiangular(_SphereGraph):
    """Equiangular (nlat x nlon) k-NN graph; irregular degree near the poles."""

    def __init__(self, nlat=36, nlon=72, poles=0, k=20, lap_type="normalized", kernel_width=None, **kwargs):
        self.nlat = int(nlat)
        self.nlon = int(nlon)
        coords, _, _ = equiangular_vec(self.nlat, self.nlon)
        super().__init__(coords, k, lap_type, kernel_width)


class SphereIcosahedral(_SphereGraph):
    """Subdivided-icosahedron k-NN graph (``10 subdivisions^2 + 2`` vertices, or the ``20 subdivisions^2`` face centres
    with ``dual``); configs/UNetSpherical/Icosahedral_400km uses ``subdivisions`` 16 -> 8 -> 4."""

    def __init__(self, subdivisions=2, dual=False, k=20, lap_type="normalized", kernel_width=None, **kwargs):
        self.subdivisions = int(subdivisions)
        self.dual = bool(dual)
        super().__init__(icosahedral_vec(self.subdivisions, self.dual), k, lap_type, kernel_width)


class SphereCubed(_SphereGraph):
    """Cubed-sphere k-NN graph (``6 subdivisions^2`` cell centres); configs/UNetSpherical/Cubed_400km: 24 -> 12 -> 6."""

    def __init__(self, subdivisions=3, spacing="equiangular", k=20, lap_type="normalized", kernel_width=None, **kwargs):
        self.subdivisions = int(subdivisions)
        self.spacing = spacing
        super().__init__(cubed_vec(self.subdivisions, spacing), k, lap_type, kernel_width)


class SphereGaussLegendre(_SphereGraph):
    """Gauss-Legendre k-NN graph, regular (``nlon`` an int, default 2 nlat) or reduced (``'ecmwf-octahedral'``);
    configs/UNetSpherical/O24: ``nlat`` 48 -> 24 -> 12, octahedral."""

    def __init__(self, nlat=4, nlon="ecmwf-octahedral", k=20, lap_type="normalized", kernel_width=None, **kwargs):
        self.nlat = int(nlat)
        self.nlon = nlon
        coords, _, _ = gauss_legendre_vec(self.nlat, nlon)
        super().__init__(coords, k, lap_type, kernel_width)


def healpix_pool_matrices(nside_fine: int, nest: bool = True):
    """Exact-hierarchy HEALPix pooling (4 children x 0.25) / unpooling (x 1.0).

    Returns scipy COO ``(pool [V/4, V], unpool [V, V/4])`` in the requested ordering.
    """
    nside_c = nside_fine // 2
    vf = 12 * nside_fine**2
    vc = 12 * nside_c**2
    child = np.arange(vf, dtype=np.int64)
    parent = child // 4
    if not nest:
        child = healpix_nest2ring(nside_fine)[child]
        parent = healpix_nest2ring(nside_c)[parent]
    pool = sparse.coo_matrix((np.full(vf, 0.25), (parent, child)), shape=(vc, vf))
    unpool = sparse.coo_matrix((np.ones(vf), (child, parent)), shape=(vf, vc))
    return pool, unpool


def equiangular_pool_matrices(nlat: int, nlon: int, c: int = 2):
    """Area-weighted (cos-latitude band) conservative c x c block pooling."""
    nlat_c, nlon_c = nlat // c, nlon // c
    edges = np.pi / 2.0 - np.arange(nlat + 1) * np.pi / nlat
    band = np.sin(edges[:-1]) - np.sin(edges[1:])  # cell area ~ band area / nlon
    i, j = np.meshgrid(np.arange(nlat), np.arange(nlon), indexing="ij")
    src = (i * nlon + j).ravel()
    dst = ((i // c) * nlon_c + (j // c)).ravel()
    area = band[i.ravel()]
    weights = sparse.csr_matrix((area, (dst, src)), shape=(nlat_c * nlon_c, nlat * nlon))
    return _normalise_pool_unpool(weights)


def knn_interp_pool_matrices(src_coords, dst_coords, k: int = 6):
    """Generic overlap-like interpolation weights between two samplings.

    Each coarse (dst) cell collects its ``k`` nearest fine (src) cells with a
    compact kernel weight; used where no exact hierarchy exists.  Not CDO-conservative
    (parity unpinned) but rectangular, irregular and row-normalised like the real thing.
    """
    tree = cKDTree(src_coords)
    dist, idx = tree.query(dst_coords, k=k)
    h = dist[:, -1:] * 1.0001
    w = np.clip(1.0 - (dist / h) ** 2, 1e-6, None)
    rows = np.repeat(np.arange(dst_coords.shape[0]), k)
    weights = sparse.csr_matrix(
        (w.ravel(), (rows, idx.ravel())), shape=(dst_coords.shape[0], src_coords.shape[0])
    )
    # make sure every source cell is covered (column sums > 0) for the unpool normalisation
    covered = np.asarray(weights.sum(axis=0)).ravel() > 0
    if not covered.all():
        miss = np.nonzero(~covered)[0]
        _, near = cKDTree(dst_coords).query(src_coords[miss], k=1)
        weights = weights + sparse.csr_matrix(
            (np.full(miss.size, 0.5), (near, miss)), shape=weights.shape
        )
    return _normalise_pool_unpool(sparse.csr_matrix(weights))


def _normalise_pool_unpool(weights):
    """Same normalisation as ``/root/reference/modules/layers.py:576-581``."""
    pool = weights.multiply(1.0 / weights.sum(1))
    unpool = weights.multiply(1.0 / weights.sum(0)).T
    return sparse.coo_matrix(pool), sparse.coo_matrix(unpool)


def cell_areas(graph) -> np.ndarray:
    """Area of every node's cell (steradians): HEALPix pixels are equal-area by construction, any other sampling gets
    the areas of its spherical Voronoi cells (what xsphere hands to CDO, loss.py:60-68)."""
    if isinstance(graph, SphereHealpix):
        return np.full(graph.n_vertices, 4.0 * np.pi / graph.n_vertices)
    from . import conservative

    return conservative.voronoi_cells(graph.coords)[2]


def conservative_pool_matrices(src_coords, dst_coords):
    """(pool, unpool) from first-order conservative remapping weights between the spherical Voronoi meshes of two
    samplings - the construction of ``layers.py:529-581`` with xsphere + CDO replaced by ``dsw_amd.conservative``."""
    from . import conservative

    weights, _ = conservative.interpolation_matrix(src_coords, dst_coords)
    return _normalise_pool_unpool(weights)


def build_pooling_matrices(src_graph, dst_graph, method="auto"):
    """Build (pool, unpool) between two graphs of this module (src = finer).

    ``auto``: exact shortcuts where the two samplings nest (HEALPix parent / children, equiangular c x c blocks: the
    conservative weights of the true pixels are known in closed form), otherwise conservative overlap areas of the
    spherical Voronoi meshes (``conservative``), which is what the reference computes for every pair.
    ``knn``: the cheap inverse-distance stand-in (not area-conserving; kept for huge irregular pairs)."""
    if method not in ("auto", "conservative", "knn"):
        raise ValueError("method must be 'auto', 'conservative' or 'knn'")
    if method == "auto":
        if isinstance(src_graph, SphereHealpix) and isinstance(dst_graph, SphereHealpix):
            if src_graph.subdivisions == 2 * dst_graph.subdivisions and src_graph.nest == dst_graph.nest:
                return healpix_pool_matrices(src_graph.subdivisions, src_graph.nest)
        if isinstance(src_graph, SphereEquiangular) and isinstance(dst_graph, SphereEquiangular):
            c = src_graph.nlat // dst_graph.nlat
            if c >= 1 and dst_graph.nlat * c == src_graph.nlat and dst_graph.nlon * c == src_graph.nlon:
                return equiangular_pool_matrices(src_graph.nlat, src_graph.nlon, c)
    if method == "knn":
        return knn_interp_pool_matrices(src_graph.coords, dst_graph.coords)
    return conservative_pool_matrices(src_graph.coords, dst_graph.coords)
