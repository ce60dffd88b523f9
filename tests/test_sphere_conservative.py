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
