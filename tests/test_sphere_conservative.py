"""Conservative interpolation-matrix construction (dsw_amd.conservative / sphere.build_pooling_matrices), CPU only.

The reference gets these matrices from xsphere + the CDO binary and asserts a list of invariants on CDO's output
(/root/reference/modules/layers.py:540-571).  CDO is absent, so the VALUES are parity-unpinned; every invariant of that
list that concerns the weights is asserted here on the build's own construction, plus the geometric facts they rest on."""
import numpy as np
import pytest
from scipy import sparse

from dsw_amd import conservative as cv
from dsw_amd import sphere


def _rot(coords, seed):
    q, _ = np.linalg.qr(np.random.default_rng(seed).standard_normal((3, 3)))
    return coords @ q.T


def test_voronoi_cells_tile_the_sphere():
    for coords in (sphere.healpix_pix2vec(4, True), sphere.equiangular_vec(12, 24)[0]):
        P, n, area = cv.voronoi_cells(coords)
        assert n.min() >= 3
        np.testing.assert_allclose(area.sum(), 4 * np.pi, rtol=1e-12)
        assert (area > 0).all()
    # HEALPix nodes: Voronoi cells are NOT the equal-area HEALPix pixels (layers.py comment in SURVEY 8a4), but close
    area = cv.voronoi_cells(sphere.healpix_pix2vec(8, True))[2]
    assert 0.9 < area.min() / area.mean() < 1.0 < area.max() / area.mean() < 1.1


def test_polygon_clipping_against_closed_forms():
    # a cell with itself, with a disjoint cell, and the octant triangle (area pi/2) with a half-space through its middle
    coords = sphere.healpix_pix2vec(2, True)
    P, n, area = cv.voronoi_cells(coords)
    same = cv.overlap_areas(P, n, P, n, np.arange(48), np.arange(48))
    np.testing.assert_allclose(same, area, rtol=1e-12)
    far = np.argmin(coords @ coords[0])
    assert cv.overlap_areas(P, n, P, n, np.array([0]), np.array([far]))[0] == 0.0
    tri = np.eye(3)[None]                                        # x, y, z axes: counter-clockwise octant
    np.testing.assert_allclose(cv.polygon_areas(tri, np.array([3])), np.pi / 2)
    nrm = np.array([[1.0, -1.0, 0.0]]) / np.sqrt(2)              # the meridian plane x = y halves it
    half, nh = cv._clip(tri, np.array([3]), nrm)
    np.testing.assert_allclose(cv.polygon_areas(half, nh), np.pi / 4, rtol=1e-12)


@pytest.mark.parametrize("pair", ["healpix8->4", "healpix8->rotated4", "equi24x48->healpix4", "equi18x36->equi9x18"])
def test_conservative_matrices_satisfy_the_reference_invariants(pair):
    if pair == "healpix8->4":
        src, dst = sphere.healpix_pix2vec(8, True), sphere.healpix_pix2vec(4, True)
    elif pair == "healpix8->rotated4":                           # no nesting at all between the two meshes
        src, dst = sphere.healpix_pix2vec(8, True), _rot(sphere.healpix_pix2vec(4, False), 3)
    elif pair == "equi24x48->healpix4":
        src, dst = sphere.equiangular_vec(24, 48)[0], sphere.healpix_pix2vec(4, True)
    else:
        src, dst = sphere.equiangular_vec(18, 36)[0], sphere.equiangular_vec(9, 18)[0]
    ds = cv.conservative_weights(src, dst)
    W = sparse.csr_matrix((ds.remap_matrix, (ds.dst_address, ds.src_address)), shape=(len(dst), len(src)))
    # layers.py:555  shape;  :557 destination rows sum to 1 (fracarea);  :559 conservation  W^T dst_area = src_area
    assert W.shape == (len(dst), len(src))
    np.testing.assert_allclose(np.asarray(W.sum(axis=1)).ravel(), 1, rtol=1e-9)
    np.testing.assert_allclose(W.T @ ds.dst_grid_area, ds.src_grid_area, rtol=1e-7)
    # :562-566 unnormalised weights: row sums = destination areas, column sums = source areas
    A = W.multiply(ds.dst_grid_area[:, None])
    np.testing.assert_allclose(np.asarray(A.sum(1)).ravel(), ds.dst_grid_area, rtol=1e-9)
    np.testing.assert_allclose(np.asarray(A.sum(0)).ravel(), ds.src_grid_area, rtol=1e-7)
    # :544-545 no masked / partially covered cells: both meshes tile the sphere
    np.testing.assert_allclose(ds.src_grid_area.sum(), 4 * np.pi, rtol=1e-12)
    np.testing.assert_allclose(ds.dst_grid_area.sum(), 4 * np.pi, rtol=1e-12)
    assert (ds.remap_matrix > 0).all()
    # pool / unpool (layers.py:576-581): both row-stochastic; pooling a constant field keeps it, and
    # pooling conserves the area integral of any field
    pool, unpool = sphere._normalise_pool_unpool(sparse.csr_matrix(A))
    np.testing.assert_allclose(np.asarray(pool.sum(1)).ravel(), 1, rtol=1e-9)
    np.testing.assert_allclose(np.asarray(unpool.sum(1)).ravel(), 1, rtol=1e-9)
    f = np.random.default_rng(0).standard_normal(len(src))
    np.testing.assert_allclose(ds.dst_grid_area @ (pool @ f), ds.src_grid_area @ f, rtol=1e-7, atol=1e-9)


def test_build_pooling_matrices_methods_and_loss_weights():
    gs, gd = sphere.SphereHealpix(4, nest=True, k=8), sphere.SphereHealpix(2, nest=True, k=8)
    pool_h, _ = sphere.build_pooling_matrices(gs, gd)                       # exact hierarchy: 4 children x 0.25
    assert pool_h.nnz == 192 and set(np.unique(pool_h.data)) == {0.25}
    pool_c, unpool_c = sphere.build_pooling_matrices(gs, gd, method="conservative")
    assert pool_c.shape == (48, 192) and unpool_c.shape == (192, 48)
    assert np.diff(sparse.csr_matrix(pool_c).indptr).min() >= 4             # children + partial overlaps
    # the four children carry most of a Voronoi parent too
    main = np.sort(sparse.csr_matrix(pool_c).toarray(), axis=1)[:, -4:].sum(1)
    assert main.min() > 0.8
    ge = sphere.SphereEquiangular(nlat=12, nlon=24, k=8)
    pool_x, unpool_x = sphere.build_pooling_matrices(ge, gd)                # no shortcut: conservative by default
    np.testing.assert_allclose(np.asarray(pool_x.sum(1)).ravel(), 1, rtol=1e-9)
    with pytest.raises(ValueError):
        sphere.build_pooling_matrices(gs, gd, method="nearest")
    a = sphere.cell_areas(ge)
    np.testing.assert_allclose(a.sum(), 4 * np.pi, rtol=1e-12)
    assert a[0] < a[len(a) // 2]                                            # polar cells are smaller than equatorial ones
    from modules.loss import AreaWeights, WeightedMSELoss
    import torch

    w = AreaWeights(ge)
    assert w.dtype == torch.float32 and abs(float(w.sum()) - 1) < 1e-6
    assert torch.allclose(AreaWeights(gs), torch.full((192,), 1 / 192))
    pred, obs = torch.randn(3, 288, 2), torch.randn(3, 288, 2)
    loss = WeightedMSELoss(weights=w)(pred, obs)
    ref = (((pred - obs) ** 2) * w.view(1, -1, 1)).sum() / w.sum() / 3 / 2
    assert torch.allclose(loss, ref)
    assert torch.allclose(WeightedMSELoss()(pred, obs), ((pred - obs) ** 2).mean())
    assert WeightedMSELoss(reduction="none", weights=w)(pred, obs).shape == pred.shape
    with pytest.raises(ValueError, match="does not match"):
        WeightedMSELoss(weights=w[:-1])(pred, obs)
    with pytest.raises(TypeError):
        WeightedMSELoss(weights=w.numpy())


@pytest.mark.parametrize("sampling,kwargs,n_expected", [
    ("healpix", {"subdivisions": 8, "nest": True}, (768, 192, 48)),
    ("equiangular", {"nlat": 36, "nlon": 72, "poles": 0}, (2592, 648, 162)),
    ("icosahedral", {"subdivisions": 16}, (2562, 642, 162)),               # configs/UNetSpherical/Icosahedral_400km
    ("cubed", {"subdivisions": 24}, (3456, 864, 216)),                      # configs/UNetSpherical/Cubed_400km
    ("gauss", {"nlat": 48, "nlon": "ecmwf-octahedral"}, (3168, 1008, 360)), # configs/UNetSpherical/O24
])
def test_unet_builds_on_all_five_samplings(sampling, kwargs, n_expected):
    """f2: every sampling of the reference's table (utils_models.py:11-20) constructs - graphs per level, scaled
    Laplacians, conservative pooling matrices between the levels (their invariants: rows of pool and of unpool^T sum to 1,
    non-negative) - and a forward on the CPU wiring (the oracle behind the host logic) has the right shape."""
    import torch
    import modules.my_models_graph as arch
    from dsw_amd import functional
    from _oracle_backend import OracleBackend

    V = n_expected[0]
    tensor_info = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 2, "output_n_feature": 1, "input_n_time": 2, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
    }
    torch.manual_seed(0)
    model = arch.UNetSpherical(tensor_info, sampling=sampling, sampling_kwargs=dict(kwargs), kernel_size_conv=3,
                               conv_type="graph", graph_type="knn", knn=20, pool_method="interp")
    assert tuple(g.n_vertices for g in model.graphs) == n_expected
    for lvl in (1, 2):
        pool = getattr(model, f"pool{lvl}").remap_matrix.to_dense().double()
        unpool = getattr(model, f"unpool{lvl}").remap_matrix.to_dense().double()
        assert pool.shape == (n_expected[lvl], n_expected[lvl - 1]) and unpool.shape == pool.t().shape
        assert float(pool.min()) >= 0 and float(unpool.min()) >= 0
        np.testing.assert_allclose(pool.sum(1).numpy(), 1, rtol=1e-5)
        np.testing.assert_allclose(unpool.sum(1).numpy(), 1, rtol=1e-5)
    functional.set_test_backend(OracleBackend())
    try:
        with torch.no_grad():
            y = model(torch.randn(1, 2, V, 2))
    finally:
        functional.set_test_backend(None)
    assert y.shape == (1, 1, V, 1) and bool(torch.isfinite(y).all())


def test_sampling_geometry_and_kernel_widths():
    """Vertex counts / symmetries of the new samplings, and the kernel-width options of the k-NN graphs."""
    ico = sphere.icosahedral_vec(3)
    assert ico.shape == (92, 3) and sphere.icosahedral_vec(4, dual=True).shape == (320, 3)
    np.testing.assert_allclose(ico.sum(0), 0, atol=1e-12)                     # centrally symmetric point set
    cub = sphere.cubed_vec(5)
    assert cub.shape == (150, 3)
    np.testing.assert_allclose(cub.sum(0), 0, atol=1e-12)
    np.testing.assert_allclose(np.abs(cub).max(1).min(), np.abs(cub).max(1).min())
    xyz, lat, lon = sphere.gauss_legendre_vec(48)
    assert xyz.shape == (3168, 3) and np.isclose(lat[0], -lat[-1]) and lat[0] > 0
    counts = np.unique(np.round(lat, 12), return_counts=True)[1]
    assert counts.min() == 20 and counts.max() == 112 and (counts == counts[::-1]).all()
    assert sphere.gauss_legendre_vec(8, nlon=16)[0].shape == (128, 3)
    for bad in (lambda: sphere.gauss_legendre_vec(7), lambda: sphere.gauss_legendre_vec(8, nlon="reduced"),
                lambda: sphere.cubed_vec(4, spacing="x"), lambda: sphere.icosahedral_vec(0)):
        with pytest.raises(ValueError):
            bad()
    g0 = sphere.SphereHealpix(8, nest=True, k=20)
    g1 = sphere.SphereHealpix(8, nest=True, k=20, kernel_width="optimal")
    g2 = sphere.SphereHealpix(8, nest=True, k=20, kernel_width="mean")
    assert g1.kernel_width == pytest.approx(0.03185 * 32 / 8)
    assert abs(g0.L - g1.L).max() > 1e-3 and abs(g0.L - g2.L).max() > 1e-6   # three different weightings of the same edges
    assert (g0.L != 0).nnz == (g1.L != 0).nnz
    with pytest.raises(ValueError, match="No known optimal kernel width"):
        sphere.SphereHealpix(8, k=12, kernel_width="optimal")
    with pytest.raises(ValueError):
        sphere.SphereHealpix(8, k=8, kernel_width="median")
