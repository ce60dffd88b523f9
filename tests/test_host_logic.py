"""Host logic on CPU: C-ABI surface, operator caches, module API / state_dict, U-Net wiring (G5)."""
import ctypes
import re

import numpy as np
import pytest
import torch
from scipy import sparse

from conftest import REPO, load_golden
from oracle import cheb_oracle as orc
import recipes


def _header_symbols():
    text = open(f"{REPO}/include/dsw_hip.h").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsw_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from dsw_amd import _native

    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _header_symbols()
    assert set(names) == set(_native.SIGNATURES), (names, sorted(_native.SIGNATURES))
    for n in names:
        assert hasattr(lib, n), n
    lib = _native.load()
    assert lib.dsw_version() >= 101
    assert lib.dsw_strerror(0) == b"ok" and b"workspace" in lib.dsw_strerror(-3)
    # argument validation needs no GPU: negative sizes / bad dtype are rejected before any launch
    assert lib.dsw_spmm_csr(None, None, None, -1, 1, 0, None, None, 1, 1, 1.0, None, 0.0, None, 0.0, 0, None) == -1
    assert lib.dsw_cheb_bwd_workspace_bytes(16, 49152, 32, 64, 3, 0) > 2 * 16 * 49152 * 32 * 4
    assert lib.dsw_cheb_bwd_workspace_bytes(1, 1, 0, 1, 1, 0) < 0


def test_plan_struct_of_another_header_version_is_refused():
    """dsw_hop2_plan grew between versions; a caller built against an older header passes a shorter struct.  The plan
    now carries sizeof(dsw_hop2_plan) as its caller knows it, and every entry point that takes a plan refuses another size
    before reading anything else (the predicates say "not supported")."""
    from dsw_amd import _native
    from dsw_amd.hop2 import Hop2PlanStruct

    lib = _native.load()
    assert lib.dsw_version() >= 101
    st = Hop2PlanStruct()
    st.n_tiles, st.tile_rows, st.max_n1, st.max_n2, st.reserved, st.hops = 4, 64, 100, 150, 9, 2
    st.struct_bytes = ctypes.sizeof(Hop2PlanStruct) - 8          # what a build against the previous header would say: garbage / short
    p = ctypes.byref(st)
    assert lib.dsw_cheb_fwd_path(p, 32, 64, 3, 0) == -1          # DSW_ERR_BAD_ARG
    assert lib.dsw_cheb_bwd_needs_basis(p, 4096, 32, 64, 3, 0) == -1
    assert lib.dsw_spmm2_supported(p, 32, 0) == 0 and lib.dsw_spmm_staged_supported(p, 32, 0) == 0
    assert lib.dsw_cheb_fwd(None, None, None, 64, 0, None, None, None, None, None, 1, 32, 64, 3, 0, None, p) == -1
    assert lib.dsw_cheb_bwd(None, None, None, 64, 0, None, None, None, None, None, None, None, None, 0, 1, 32, 64, 3, 0, None, p) == -1
    st.struct_bytes = ctypes.sizeof(Hop2PlanStruct)
    assert lib.dsw_spmm2_supported(ctypes.byref(st), 32, 0) == 1


def test_cpu_tensors_fail_loudly():
    from modules.layers import ConvCheb, prepare_torch_laplacian
    from dsw_amd import sphere

    lap = prepare_torch_laplacian(sphere.SphereHealpix(2, nest=True, k=8).L, lmax=2.0)
    layer = ConvCheb(3, 4, 3, laplacian=lap)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(torch.zeros(1, 48, 3))


def test_csr_operator_and_transpose():
    from dsw_amd.functional import CsrOperator, get_operator

    rp, ci, va = recipes.irregular_operator(97, seed=3, min_deg=1, max_deg=30)
    coo = orc.coo_from_csr_arrays(rp, ci, va, (97, 97))
    op = get_operator(coo)
    assert get_operator(coo) is op  # cached
    np.testing.assert_array_equal(op.rowptr.numpy(), rp)
    np.testing.assert_array_equal(op.colind.numpy(), ci)
    assert op.rowptr.dtype == torch.int32 and op.colind.dtype == torch.int32 and op.values.dtype == torch.float32
    ref_t = sparse.csr_matrix((va, ci, rp), shape=(97, 97)).T.tocsr()
    ref_t.sort_indices()
    t = op.transpose()
    np.testing.assert_array_equal(t.rowptr.numpy(), ref_t.indptr)
    np.testing.assert_array_equal(t.colind.numpy(), ref_t.indices)
    np.testing.assert_array_equal(t.values.numpy(), ref_t.data)
    assert t.transpose() is op
    # rectangular
    m = sparse.random(13, 40, density=0.2, random_state=1, format="coo", dtype=np.float32)
    opm = CsrOperator.from_sparse_coo(orc.coo_from_scipy(m).float())
    assert opm.shape == (13, 40) and opm.transpose().shape == (40, 13)
    # in-place update of the buffer invalidates the cache
    coo2 = orc.coo_from_csr_arrays(rp, ci, va * 2, (97, 97))
    coo.copy_(coo2)
    assert get_operator(coo) is not op
    np.testing.assert_allclose(get_operator(coo).values.numpy(), va * 2)


def test_prepare_laplacian_matches_golden():
    from modules.layers import convert_to_torch_sparse, prepare_torch_laplacian

    g = load_golden("G4_prepare")
    n = len(g["in_rowptr"]) - 1
    L = sparse.csr_matrix((g["in_values"], g["in_colind"], g["in_rowptr"]), shape=(n, n))
    t = prepare_torch_laplacian(L, lmax=float(g["lmax"][0]))
    assert t.is_coalesced() and t.dtype == torch.float32 and t.indices().dtype == torch.int64
    np.testing.assert_array_equal(t.indices().numpy(), g["out_indices"])
    np.testing.assert_allclose(t.values().numpy(), g["out_values"], rtol=0, atol=1e-7)
    # default path (ARPACK) differs only by the lmax estimate: spectrum must end up inside [-1, 1]
    t2 = prepare_torch_laplacian(L)
    dense = t2.to_dense().double().numpy()
    ev = np.linalg.eigvalsh((dense + dense.T) / 2)
    assert ev.min() >= -1.0 - 1e-4 and ev.max() <= 1.0
    c = convert_to_torch_sparse(sparse.coo_matrix(L))
    assert c.is_coalesced() and c.shape == (n, n)


def test_convcheb_module_contract(oracle_backend):
    from modules.layers import ConvCheb, GeneralConvBlock, conv_cheb, get_conv_fun

    g = load_golden("G2_conv_K_sweep")
    p = "ns_K3_"
    B, V, Fin, Fout, K, has_bias, _ = [int(v) for v in g[p + "meta"]]
    lap = orc.coo_from_csr_arrays(g[p + "rowptr"], g[p + "colind"], g[p + "values"], (V, V))
    torch.manual_seed(0)
    layer = GeneralConvBlock.getConvLayer(Fin, Fout, K, conv_type="graph", laplacian=lap, bias=True,
                                          lonlat_ratio=None, periodic_padding=True)
    assert isinstance(layer, ConvCheb) and get_conv_fun("graph") is ConvCheb
    assert sorted(layer.state_dict().keys()) == ["bias", "laplacian", "weight"]
    assert layer.state_dict()["laplacian"].is_sparse
    assert tuple(layer.weight.shape) == (Fin, K, Fout) and float(layer.bias.detach().abs().sum()) == 0.0
    assert "kernel_size=3" in repr(layer) and "bias=True" in repr(layer)
    # Kaiming-normal fan-in init (statistical check on a wide layer)
    wide = ConvCheb(64, 256, 3, laplacian=lap)
    assert abs(float(wide.weight.detach().std()) - np.sqrt(2.0 / (64 * 3))) < 0.004
    with pytest.raises(ValueError, match="unknown fan"):
        wide.reset_parameters(fan="sideways")
    # forward/backward through the autograd wiring (oracle stands in for the kernels on CPU)
    layer.set_parameters(torch.from_numpy(g[p + "w"]), torch.from_numpy(g[p + "b"]))
    x = torch.from_numpy(g[p + "x"]).permute(1, 0, 2).contiguous().permute(1, 0, 2)  # non-contiguous view
    x.requires_grad_(True)
    y = layer(x)
    y += 0.0  # callers mutate the output in place
    y.backward(torch.from_numpy(g[p + "gy"]))
    assert orc.max_rel_err(y.detach(), g[p + "y"]) < 1e-5
    assert orc.max_rel_err(x.grad, g[p + "dx"]) < 1e-5
    assert orc.max_rel_err(layer.weight.grad, g[p + "dw"]) < 1e-5
    assert orc.max_rel_err(layer.bias.grad, g[p + "db"]) < 1e-5
    # functional form, error text, state_dict round trip, SWAG-style plain-tensor reassignment
    y2 = conv_cheb(lap, x.detach(), layer.weight.detach())
    assert orc.max_rel_err(y2 + layer.bias.detach(), g[p + "y"]) < 1e-5
    err = str(load_golden("G7_errors")["fin_mismatch"])
    with pytest.raises(ValueError) as ei:
        layer(torch.zeros(1, V, Fin + 1))
    assert str(ei.value).split(":")[0] == err.split(":")[0]
    with pytest.raises(ValueError, match="conv_type is not supported"):
        GeneralConvBlock.getConvLayer(4, 4, 3, conv_type="mesh", laplacian=lap)
    other = ConvCheb(Fin, Fout, K, laplacian=lap.clone())
    other.load_state_dict(layer.state_dict(), strict=True)
    assert torch.equal(other.weight, layer.weight)
    w_plain = layer._parameters.pop("weight").detach() * 2
    layer.__setattr__("weight", w_plain)
    y3 = layer(x.detach())
    assert orc.max_rel_err(y3 - layer.bias.detach(), 2 * (torch.from_numpy(g[p + "y"]) - layer.bias.detach())) < 1e-5
    assert layer.double().laplacian.dtype == torch.float64


def test_remap_modules(oracle_backend):
    from modules.layers import (GeneralAvgPool, GeneralAvgUnpool, GeneralMaxAreaPool, GeneralMaxAreaUnpool,
                                PoolUnpoolBlock)
    from dsw_amd import sphere

    g = load_golden("G3_remap")
    for tag in ("hier", "interp"):
        pm = sparse.csr_matrix((g[f"{tag}_pool_values"], g[f"{tag}_pool_colind"], g[f"{tag}_pool_rowptr"]), shape=(192, 768))
        um = sparse.csr_matrix((g[f"{tag}_unpool_values"], g[f"{tag}_unpool_colind"], g[f"{tag}_unpool_rowptr"]), shape=(768, 192))
        pool, unpool = GeneralAvgPool(sparse.coo_matrix(pm)), GeneralAvgUnpool(sparse.coo_matrix(um))
        assert list(pool.state_dict()) == ["remap_matrix"] and pool.remap_matrix.is_sparse
        x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
        y, idx = pool(x)
        assert idx is None and y.shape == (2, 192, 6)
        y.backward(torch.from_numpy(g[f"{tag}_gyp"]))
        assert orc.max_rel_err(y.detach(), g[f"{tag}_yp"]) < 1e-5
        assert orc.max_rel_err(x.grad, g[f"{tag}_dxp"]) < 1e-5
        xu = torch.from_numpy(g[f"{tag}_xu"]).requires_grad_(True)
        yu = unpool(xu, None)  # decode() passes the (None) pool indices positionally
        yu.backward(torch.from_numpy(g[f"{tag}_gyu"]))
        assert orc.max_rel_err(yu.detach(), g[f"{tag}_yu"]) < 1e-5
        assert orc.max_rel_err(xu.grad, g[f"{tag}_dxu"]) < 1e-5
    gs, gd = sphere.SphereHealpix(4, nest=True, k=8), sphere.SphereHealpix(2, nest=True, k=8)
    pool, unpool = PoolUnpoolBlock.getGeneralPoolUnpoolLayer(gd, gs, "interp")  # swapped on purpose
    assert pool.remap_matrix.shape == (48, 192) and unpool.remap_matrix.shape == (192, 48)
    pool, unpool = PoolUnpoolBlock.getGeneralPoolUnpoolLayer(gs, gd, "maxarea")
    assert isinstance(pool, GeneralMaxAreaPool) and isinstance(unpool, GeneralMaxAreaUnpool)
    assert float(pool.remap_matrix.values().sum()) == 48.0 and float(unpool.remap_matrix.values().sum()) == 48.0  # one fine cell per coarse cell (layers.py:1019-1036)
    with pytest.raises(NotImplementedError):
        PoolUnpoolBlock.getGeneralPoolUnpoolLayer(gs, gd, "learn")
    with pytest.raises(ValueError, match="not supoorted"):
        PoolUnpoolBlock.getGeneralPoolUnpoolLayer(gs, gd, "median")


def build_g5_model(device="cpu"):
    """UNetSpherical nside=8 with the fixture's operators and seeded parameters."""
    import modules.my_models_graph as arch

    g = load_golden("G5_unet_nside8")
    V = 768
    tensor_info = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
    }
    model = arch.UNetSpherical(tensor_info, sampling="healpix", sampling_kwargs={"subdivisions": 8, "nest": True},
                               kernel_size_conv=3, conv_type="graph", graph_type="knn", knn=20, pool_method="interp")
    # identical prepared operators on both sides (ARPACK lmax is nondeterministic)
    laps = [orc.coo_from_csr_arrays(g[f"lap{i}_rowptr"], g[f"lap{i}_colind"], g[f"lap{i}_values"],
                                    (len(g[f"lap{i}_rowptr"]) - 1,) * 2) for i in range(3)]
    sizes = {lap.shape[0]: lap for lap in laps}
    sd = model.state_dict()
    for key in sd:
        if key.endswith("laplacian"):
            sd[key] = sizes[sd[key].shape[0]].clone()
        elif key.endswith("remap_matrix"):
            nm = key.split(".")[0]
            sd[key] = orc.coo_from_csr_arrays(g[f"{nm}_rowptr"], g[f"{nm}_colind"], g[f"{nm}_values"], tuple(g[f"{nm}_shape"]))
    names = sorted(n for n, _ in model.named_parameters())
    for i, n in enumerate(names):
        sd[n] = torch.from_numpy(recipes.unet_param_fill(i, n, tuple(sd[n].shape)))
    model.load_state_dict(sd, strict=True)
    return model.to(device), g, names


def check_g5(model, g, names, device="cpu", tol=2e-5):
    """Returns the measured errors {y, loss, grad_l2, grad_head, grad_dot} (all relative) after asserting the bounds:
    ``tol`` on output and loss, ``10 * tol`` on the gradient fingerprints."""
    x = torch.from_numpy(recipes.rand(501, (2, 3, 768, 6))).to(device)
    target = torch.from_numpy(recipes.rand(502, (2, 1, 768, 2))).to(device)
    y = model(x)
    loss = ((y - target) ** 2).mean()
    loss.backward()
    assert y.shape == (2, 1, 768, 2)
    params = dict(model.named_parameters())
    probes = np.stack([recipes.grad_probe(i, params[n].grad.detach().cpu().numpy()) for i, n in enumerate(names)])
    ref = g["grad_probes"]
    scale = np.abs(ref[:, :1]) + 1e-12  # per-tensor gradient l2 norm
    # the dot-probe sums ~1e5 random-signed terms: compare against |g|*|r| ~ l2 * sqrt(n)
    n_el = np.array([params[n].numel() for n in names], dtype=np.float64)
    errs = {
        "y": orc.max_rel_err(y.detach().cpu(), g["y"]),
        "loss": abs(loss.item() - float(g["loss"][0])) / max(1.0, float(g["loss"][0])),
        "grad_l2": float(np.max(np.abs(probes[:, 0] - ref[:, 0]) / scale[:, 0])),
        "grad_head": float(np.max(np.abs(probes[:, 2:] - ref[:, 2:]) / scale)),
        "grad_dot": float(np.max(np.abs(probes[:, 1] - ref[:, 1]) / (scale[:, 0] * np.sqrt(n_el)))),
    }
    assert errs["y"] < tol and errs["loss"] < tol, errs
    assert errs["grad_l2"] < 10 * tol and errs["grad_head"] < 10 * tol and errs["grad_dot"] < 10 * tol, errs
    return errs


def test_unet_matches_reference_fixture_on_cpu_wiring(oracle_backend):
    model, g, names = build_g5_model()
    assert list(model.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    assert names == [str(n) for n in g["param_names"]]
    shapes = [str(tuple(p.shape)) for _, p in sorted(model.named_parameters())]
    assert shapes == [str(s) for s in g["param_shapes"]]
    assert sum(p.numel() for p in model.parameters()) == 1770122
    check_g5(model, g, names)


def test_config_driven_construction():
    """Same ``architecture_name`` + inspect-filtered kwargs convention as utils_config.get_pytorch_model."""
    import inspect
    import modules.my_models_graph as arch

    settings = {
        "pretrained_model_name": None, "kernel_size_conv": 3, "bias": True, "batch_norm": False,
        "batch_norm_before_activation": False, "activation": True, "activation_fun": "relu",
        "pool_method": "Interp", "kernel_size_pooling": 4, "conv_type": "graph", "graph_type": "knn", "knn": 8,
        "periodic_padding": "True", "sampling_name": "Healpix_x", "sampling": "healpix",
        "sampling_kwargs": {"subdivisions": 4, "nest": True}, "architecture_name": "UNetSpherical",
    }
    settings["tensor_info"] = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": 192}}, "output_shape_info": {"dynamic": {"node": 192}},
    }
    cls = getattr(arch, settings["architecture_name"])
    args = inspect.getfullargspec(cls.__init__).args
    model = cls(**{k: v for k, v in settings.items() if k in args})
    assert len([k for k in model.state_dict() if k.endswith("laplacian")]) == 11
    assert len([k for k in model.state_dict() if k.endswith("remap_matrix")]) == 4
    assert model.pool1.remap_matrix.shape == (48, 192)


def test_hop2_plan_matches_two_plain_hops():
    """The tile plan of the fused two-hop kernel, emulated in numpy, equals two plain operator
    applications (HEALPix nested / ring order and an irregular non-symmetric operator)."""
    from dsw_amd import hop2, sphere

    cases = {
        "nest_k8": sphere.SphereHealpix(4, nest=True, k=8).L,
        "ring_k20": sphere.SphereHealpix(4, nest=False, k=20).L,
    }
    ops = {k: (m.indptr, m.indices, m.data.astype(np.float32)) for k, m in cases.items()}
    ops["irregular"] = recipes.irregular_operator(300, seed=4, min_deg=0, max_deg=40)
    rng = np.random.default_rng(0)
    for name, (rp, ci, va) in ops.items():
        n = len(rp) - 1
        L = sparse.csr_matrix((np.asarray(va, dtype=np.float64), ci, rp), shape=(n, n))
        for rows in (64, 100):
            plan = hop2.build_hop2_plan(rp, ci, va, rows)
            assert plan.n_tiles == -(-n // rows) and plan.lds_bytes(128) > 0
            assert (plan.tile_meta[:, 1] <= plan.max_n1).all() and (plan.tile_meta[:, 2] <= plan.max_n2).all()
            U, Z1, Z1b, Z2 = (rng.standard_normal((n, 3)) for _ in range(4))
            y1, y2 = hop2.emulate_hop2(plan, U, Z1, Z1b, Z2, 2.0, 1.0, -1.0, 1.5, -1.0, 0.5)
            r1 = 2.0 * (L @ U) + Z1 - Z1b
            r2 = 1.5 * (L @ r1) - U + 0.5 * Z2
            np.testing.assert_allclose(y1, r1, atol=1e-12)
            np.testing.assert_allclose(y2, r2, atol=1e-12)
        # tiles clustered from the graph (any row sets): a partition into <= rows rows each, same two hops
        for rows in (64, 48):
            tiles = hop2.cluster_tiles(rp, ci, rows)
            cover = np.sort(np.concatenate(tiles))
            assert np.array_equal(cover, np.arange(n)) and max(len(t) for t in tiles) <= rows
            plan = hop2.build_hop2_plan(rp, ci, va, rows, tiles=tiles)
            assert plan.explicit_tiles and plan.n_tiles == len(tiles)
            assert [int(m[5]) for m in plan.tile_meta] == [len(t) for t in tiles]
            U, Z1, Z1b, Z2 = (rng.standard_normal((n, 3)) for _ in range(4))
            y1, y2 = hop2.emulate_hop2(plan, U, Z1, Z1b, Z2, 2.0, 1.0, -1.0, 1.5, -1.0, 0.5)
            r1 = 2.0 * (L @ U) + Z1 - Z1b
            np.testing.assert_allclose(y1, r1, atol=1e-12)
            np.testing.assert_allclose(y2, 1.5 * (L @ r1) - U + 0.5 * Z2, atol=1e-12)
    with pytest.raises(ValueError):
        hop2.build_hop2_plan(rp, ci, va, 64, tiles=[np.arange(10)])     # not a partition of the rows


def test_hop1_plan_matches_one_plain_hop():
    """Plans of the staged ONE-hop kernel (dense stencils): local CSR of the tile rows only, gather list = tile + 1-ring;
    emulated in numpy they equal one operator application with its axpby epilogue - consecutive and clustered tiles."""
    from dsw_amd import hop2, sphere

    ops = {"nest_k20": sphere.SphereHealpix(4, nest=True, k=20).L, "ring_k20": sphere.SphereHealpix(4, nest=False, k=20).L}
    ops = {k: (m.indptr, m.indices, m.data.astype(np.float32)) for k, m in ops.items()}
    ops["irregular"] = recipes.irregular_operator(300, seed=5, min_deg=0, max_deg=40)
    rng = np.random.default_rng(1)
    for name, (rp, ci, va) in ops.items():
        n = len(rp) - 1
        L = sparse.csr_matrix((np.asarray(va, dtype=np.float64), ci, rp), shape=(n, n))
        for rows, clustered in ((128, False), (64, False), (64, True)):
            tiles = hop2.cluster_tiles(rp, ci, rows, max_n1=150) if clustered else None
            plan = hop2.build_hop2_plan(rp, ci, va, rows, tiles=tiles, hops=1)
            assert plan.hops == 1 and plan.max_n1 <= rows and plan.lds_bytes(128) > 0
            assert (plan.tile_meta[:, 1] <= rows).all() and (plan.tile_meta[:, 2] >= plan.tile_meta[:, 1]).all()
            if clustered:
                assert (plan.tile_meta[:, 2] <= 150).all()
            U, Z, Z2 = (rng.standard_normal((n, 3)) for _ in range(3))
            y = hop2.emulate_hop1(plan, U, Z, Z2, 2.0, -1.0, 0.5)
            np.testing.assert_allclose(y, 2.0 * (L @ U) - Z + 0.5 * Z2, atol=1e-12)
            np.testing.assert_allclose(hop2.emulate_hop1(plan, U, None, None, 1.0, 0.0, 0.0), L @ U, atol=1e-12)
    with pytest.raises(ValueError):
        hop2.build_hop2_plan(rp, ci, va, 64, hops=3)


def test_clustered_tiles_make_non_local_row_orders_compact():
    """HEALPix RING order and equiangular row-major order: a strip of 64 consecutive rows has a 2-ring several times the
    tile; the graph clustering brings it down to what a square patch has, so the operator takes the fused two-hop path."""
    from dsw_amd import hop2, sphere

    for name, L in (("ring", sphere.SphereHealpix(16, nest=False, k=20).L),
                    ("equiangular", sphere.SphereEquiangular(nlat=48, nlon=96, k=20).L)):
        L = L.tocsr()
        rp, ci, va = L.indptr, L.indices, L.data.astype(np.float32)
        strips = hop2.build_hop2_plan(rp, ci, va, 64)
        patches = hop2.build_hop2_plan(rp, ci, va, 64, tiles=hop2.cluster_tiles(rp, ci, 64))
        assert patches.n_tiles <= 1.1 * strips.n_tiles + 2, name
        assert patches.tile_meta[:, 2].mean() < 0.7 * strips.tile_meta[:, 2].mean(), name
        assert patches.lds_bytes(128, True) <= 156 * 1024, name

def test_training_driver_config_and_ar_logic():
    """scripts_training/train_synthetic_state.py host logic: config schema, tensor_info, model factory convention and the
    autoregressive window update (CPU; a stand-in model, no kernels involved)."""
    import json
    import os
    import sys
    import types

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts_training"))
    import train_synthetic_state as drv

    cfg = drv.read_config(os.path.join(root, "configs/UNetSpherical/Healpix_400km/InterpPool-Graph_knn.synthetic.json"))
    info = drv.synthetic_tensor_info(cfg)
    assert info["input_n_feature"] == 6 and info["output_n_feature"] == 2 and info["input_n_time"] == 3
    assert info["input_shape_info"]["dynamic"]["node"] == 12 * 16 ** 2

    class Toy(torch.nn.Module):
        def __init__(self, tensor_info, knn=3, not_in_config=7):
            super().__init__()
            self.knn, self.extra = knn, not_in_config
            self.scale = torch.nn.Parameter(torch.ones(1))

        def forward(self, x):                      # predicts the dynamic features of the last step, scaled
            return self.scale * x[:, -1:, :, -2:]

    mod = types.ModuleType("toy_arch")
    mod.Toy = Toy
    ms = dict(cfg["model_settings"], architecture_name="Toy", tensor_info=info)
    model = drv.get_pytorch_model(mod, ms)
    assert model.knn == 20 and model.extra == 7     # config keys the ctor declares are passed, the rest filtered out
    with pytest.raises(TypeError):
        drv.get_pytorch_model("not a module", ms)

    x = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6)
    targets = [torch.zeros(2, 1, 4, 2) for _ in range(3)]
    loss = drv.ar_training_step(model, x, targets, n_dyn=2)
    # iteration 2 sees a window whose last step carries the previous prediction in its dynamic features
    y0 = x[:, -1:, :, -2:]
    expect = 3 * torch.mean(y0 ** 2)               # the toy model reproduces the same prediction every iteration
    assert torch.allclose(loss, expect)
    loss.backward()
    assert model.scale.grad is not None


def _g8_layers(g, tag, kind):
    """(pool, unpool) layers of the build on the fixture's matrices (kind: 'maxval' | 'maxarea')."""
    from modules.layers import GeneralMaxAreaPool, GeneralMaxAreaUnpool, GeneralMaxValPool, GeneralMaxValUnpool

    pm = sparse.csr_matrix((g[f"{tag}_pool_values"], g[f"{tag}_pool_colind"], g[f"{tag}_pool_rowptr"]), shape=(192, 768))
    um = sparse.csr_matrix((g[f"{tag}_unpool_values"], g[f"{tag}_unpool_colind"], g[f"{tag}_unpool_rowptr"]), shape=(768, 192))
    if kind == "maxval":
        return GeneralMaxValPool(sparse.coo_matrix(pm)), GeneralMaxValUnpool(sparse.coo_matrix(um))
    # max-area selection compares the matrix entries as given (fp64): rebuild the generator's matrices (the fixture
    # stores the fp32 buffers, whose rounding can flip near-ties of the arg-max)
    from dsw_amd import sphere

    if tag == "hier":
        pm64, _ = sphere.healpix_pool_matrices(8, nest=True)
    else:
        gs, gd = sphere.SphereHealpix(8, nest=True, k=8), sphere.SphereHealpix(4, nest=True, k=8)
        pm64, _ = sphere.knn_interp_pool_matrices(gs.coords, gd.coords, k=7)
    return GeneralMaxAreaPool(pm64), GeneralMaxAreaUnpool(pm64.T)


def check_g8(tag, device="cpu", tol=1e-6):
    """Max-value and max-area pooling layers against the reference's outputs (fixture G8), forward and backward."""
    g = load_golden("G8_maxpool")
    dev = lambda a: torch.from_numpy(a).to(device)
    pool, unpool = (m.to(device) for m in _g8_layers(g, tag, "maxval"))
    x = dev(g[f"{tag}_mv_x"]).requires_grad_(True)
    yp, idx_ref = pool(x)                                 # default: the reference's [2, B*F*Vd] int64 index tensor
    assert idx_ref.dtype == torch.int64 and torch.equal(idx_ref.cpu(), torch.from_numpy(g[f"{tag}_mv_index"]))
    pool.index_format = "compact"
    yp, idx = pool(x)
    assert idx.dtype == torch.int32 and idx.shape == yp.shape
    assert torch.equal(pool.reference_index(idx).cpu(), torch.from_numpy(g[f"{tag}_mv_index"]))
    assert torch.equal(yp.detach().cpu(), torch.from_numpy(g[f"{tag}_mv_yp"]))           # selection: exact
    yp.backward(dev(g[f"{tag}_mv_gyp"]))
    assert orc.max_rel_err(x.grad, g[f"{tag}_mv_dxp"]) <= tol
    for index in (idx, dev(g[f"{tag}_mv_index"])):                                       # compact and reference form
        xu = dev(g[f"{tag}_mv_xu"]).requires_grad_(True)
        yu = unpool(xu, index)
        assert torch.equal(yu.detach().cpu(), torch.from_numpy(g[f"{tag}_mv_yu"]))
        yu.backward(dev(g[f"{tag}_mv_gyu"]))
        assert torch.equal(xu.grad.cpu(), torch.from_numpy(g[f"{tag}_mv_dxu"]))
    pool, unpool = (m.to(device) for m in _g8_layers(g, tag, "maxarea"))
    x = dev(g[f"{tag}_ma_x"]).requires_grad_(True)
    yp, none_idx = pool(x)
    assert none_idx is None
    yp.backward(dev(g[f"{tag}_mv_gyp"]))
    assert orc.max_rel_err(yp, g[f"{tag}_ma_yp"]) <= tol and orc.max_rel_err(x.grad, g[f"{tag}_ma_dxp"]) <= tol
    xu = dev(g[f"{tag}_ma_xu"]).requires_grad_(True)
    yu = unpool(xu, None)
    yu.backward(dev(g[f"{tag}_mv_gyu"]))
    assert orc.max_rel_err(yu, g[f"{tag}_ma_yu"]) <= tol and orc.max_rel_err(xu.grad, g[f"{tag}_ma_dxu"]) <= tol
    # the 0/1 matrices themselves equal the reference's
    np.testing.assert_array_equal(orc.csr_arrays_from_coo(pool.remap_matrix.cpu())[1], g[f"{tag}_ma_pool_colind"])
    np.testing.assert_array_equal(orc.csr_arrays_from_coo(unpool.remap_matrix.cpu())[1], g[f"{tag}_ma_unpool_colind"])


@pytest.mark.parametrize("tag", ["hier", "interp"])
def test_max_pooling_layers_match_reference_fixture_on_cpu_wiring(oracle_backend, tag):
    check_g8(tag)


def test_maxval_factory_and_unet_wiring(oracle_backend):
    from dsw_amd import sphere
    from modules.layers import GeneralMaxValPool, GeneralMaxValUnpool, PoolUnpoolBlock

    gs, gd = sphere.SphereHealpix(4, nest=True, k=8), sphere.SphereHealpix(2, nest=True, k=8)
    pool, unpool = PoolUnpoolBlock.getGeneralPoolUnpoolLayer(gs, gd, "maxval")
    assert isinstance(pool, GeneralMaxValPool) and isinstance(unpool, GeneralMaxValUnpool)
    x = torch.randn(2, 192, 5)
    y, idx = pool(x)
    back = unpool(y, idx)
    # every coarse value lands on the fine cell it came from: pooling the unpooled field again is the identity
    y2, _ = pool(back * 1e3 + torch.where(back != 0, 0.0, -1e9))
    assert torch.equal(y2, y * 1e3) or orc.max_rel_err(y2, (y * 1e3).numpy()) < 1e-6


def build_g9_trainer(device="cpu", use_graph=False):
    """The training driver's Trainer on the state of fixture G9 (reference UNetSpherical nside=8 + reference
    WeightedMSELoss + Adam(eps=1e-7), three AR optimisation steps of two forwards each)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(REPO, "scripts_training"))
    import train_synthetic_state as drv
    import modules.my_models_graph as arch

    g = load_golden("G9_ar_steps")
    V = 768
    tensor_info = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
    }
    model = arch.UNetSpherical(tensor_info, sampling="healpix", sampling_kwargs={"subdivisions": 8, "nest": True},
                               kernel_size_conv=3, conv_type="graph", graph_type="knn", knn=20, pool_method="interp")
    laps = {}
    for i in range(3):
        rp = g[f"lap{i}_rowptr"]
        laps[len(rp) - 1] = orc.coo_from_csr_arrays(rp, g[f"lap{i}_colind"], g[f"lap{i}_values"], (len(rp) - 1,) * 2)
    sd = model.state_dict()
    for key in sd:
        if key.endswith("laplacian"):
            sd[key] = laps[sd[key].shape[0]].clone()
        elif key.endswith("remap_matrix"):
            nm = key.split(".")[0]
            sd[key] = orc.coo_from_csr_arrays(g[f"{nm}_rowptr"], g[f"{nm}_colind"], g[f"{nm}_values"], tuple(g[f"{nm}_shape"]))
    names = [str(n) for n in g["param_names"]]
    for i, n in enumerate(names):
        sd[n] = torch.from_numpy(recipes.unet_param_fill(i, n, tuple(sd[n].shape)))
    model.load_state_dict(sd, strict=True)
    model = model.to(device)
    x = torch.from_numpy(recipes.rand(901, (2, 3, V, 6))).to(device)
    targets = [torch.from_numpy(recipes.rand(902 + i, (2, 1, V, 2))).to(device) for i in range(2)]
    trainer = drv.Trainer(model, x, targets, n_dyn=2, lr=float(g["lr"][0]), weights=torch.from_numpy(g["weights"]),
                          use_graph=use_graph)
    return trainer, g, names


def _relu_output_modules(model):
    """(name, module whose forward output IS a ReLU result) for every ConvBlock with an activation: the conv itself when
    conv + bias + relu run fused (its output is what backward takes the mask from), the block when the activation is a
    separate op after a batch norm."""
    out = []
    for name, block in model.named_modules():
        if not (hasattr(block, "act_fun") and hasattr(block, "conv") and block.act):
            continue
        if block.norm and block.bn_before_act:
            out.append((name, block))
        else:
            out.append((name + ".conv", block.conv))
    return out


def record_relu_masks(model, tiny=1e-4):
    """Forward hooks that keep, per module and call, the sign pattern of every ReLU output AND which outputs are below
    `tiny` (for a positive output the ReLU result IS the pre-activation: a second run that puts such an element at zero
    is off by at least that much).  Returns (masks, handles); masks[name][call] = (positive, small)."""
    masks, handles = {}, []
    for name, mod in _relu_output_modules(model):
        def hook(_m, _a, out, name=name):
            o = out.detach()
            masks.setdefault(name, []).append(((o > 0).cpu(), (o < tiny).cpu()))
        handles.append(mod.register_forward_hook(hook))
    return masks, handles


def pin_relu_masks(model, masks, tiny=1e-4):
    """Give a second run of the same model the ReLU decisions of a first one (`record_relu_masks`).  Two correct fp32
    evaluations of a network agree to ~1e-6 per element, which still leaves the odd pre-activation with |z| ~ 1e-7 on
    different sides of zero; on a 48- or 192-node level ONE such element moves a weight gradient by 1e-2 (measured:
    uconv2.convblock1, |z| = 7e-7 -> 6e-2), so an element-wise gradient comparison would test the dice, not the kernels.
    Only elements whose two evaluations disagree in sign are touched, and BOTH evaluations must be smaller than `tiny`
    there, else the runs really differ and the hook raises: where this run is positive and the recorded one was not, its
    own value is checked; where this run is at zero and the recorded one was positive - this run's PRE-activation was
    negative by an unknown amount, invisible in its ReLU output - the recorded value must have been below
    `tiny`.  Returns a counter of the decisions that differed (callers bound "n" against "total")."""
    flipped = {"n": 0, "max_abs": 0.0, "total": 0}
    handles = []
    for name, mod in _relu_output_modules(model):
        state = {"call": 0}

        def hook(_m, _a, out, name=name, state=state):
            want, small = (t.to(out.device) for t in masks[name][state["call"]])
            state["call"] += 1
            y = out.data                          # .data: not an autograd-visible edit of the saved ReLU output
            dis = (y > 0) != want
            flipped["total"] += y.numel()
            n = int(dis.sum())
            if n:
                mag = float(y[dis].abs().max())
                assert mag < tiny, (name, n, mag)
                # this run at zero, the recorded run positive: the recorded value bounds how negative this run may have been
                assert bool(small[dis & want].all()), (name, "a pre-activation is negative here and was >= %g in the checker run" % tiny)
                flipped["n"] += n
                flipped["max_abs"] = max(flipped["max_abs"], mag)
                y[dis & want] = 1e-30
                y[dis & ~want] = 0.0
        handles.append(mod.register_forward_hook(hook))
    return flipped, handles


def pin_g9_ties(model, g):
    """Fixture G9 records, for every pre-activation the reference found within `tie_threshold` (1e-4) of zero, on which
    side of zero its fp32 sum fell (block, forward number, flat index, sign).  A kernel that sums a row in another order
    may land on the other side for the handful of elements with |z| ~ 1e-7 - a legitimate fp32 result, but on the 48-node
    level ONE flipped mask moves a weight-gradient fingerprint by ~1e-2.  The hooks below give exactly these elements
    the reference's mask (forward values change by < |z| <= 1e-4 only where the sign disagrees, i.e. by ~1e-7) and touch
    nothing else; returns a counter of how many decisions actually differed."""
    ties, blocks = g["ties"], [str(b) for b in g["tie_blocks"]]
    flipped = {"n": 0}
    handles = []
    for bi, name in enumerate(blocks):
        block = model.get_submodule(name)
        mine = ties[ties[:, 0] == bi]
        state = {"call": 0}

        def hook(_m, _a, out, mine=mine, state=state):
            rows = mine[mine[:, 1] == state["call"]]
            state["call"] += 1
            if len(rows):
                flat = out.data.view(-1)                     # .data: not an autograd-visible edit of the saved ReLU output
                idx = torch.from_numpy(rows[:, 2].copy()).to(out.device)
                pos = torch.from_numpy(rows[:, 3].astype(bool)).to(out.device)
                cur = flat[idx]
                flipped["n"] += int(((cur > 0) != pos).sum())
                flipped["ties"] = flipped.get("ties", 0) + len(rows)
                # the fixture lists an element only because the reference's |z| was < 1e-4: whatever this run computed there
                # must be just as small when it is positive (a zero here can hide a negative pre-activation of any size only
                # if the reference itself was within 1e-4 of zero - which is what makes it a listed tie)
                assert float(cur.abs().max()) < 1e-3, float(cur.abs().max())
                flat[idx] = torch.where(pos, cur.clamp_min(1e-30), torch.zeros_like(cur))
        handles.append(block.register_forward_hook(hook))
    return flipped, handles


def check_g9(trainer, g, names, tol=1e-5):
    """Losses of three optimisation steps, gradient fingerprints of step 0 and the first Adam update against the
    reference run.  Adam's first update is lr * g / (|g| + 1e-7): elements whose gradient is ~1e-7 amplify rounding
    differences by up to lr / eps, so the update is compared through its per-tensor l2 norm (dominated by the
    well-conditioned elements) and the later losses get 20x the tolerance of the first."""
    model = trainer.model
    params = dict(model.named_parameters())
    before = {n: params[n].detach().clone() for n in names}
    losses = [float(trainer.step())]
    grads = {n: params[n].grad.detach().cpu().numpy().copy() for n in names} if trainer.graph is None else None
    after1 = {n: params[n].detach().clone() for n in names}
    losses += [float(trainer.step()), float(trainer.step())]
    ref = g["losses"]
    assert abs(losses[0] - ref[0]) <= tol * abs(ref[0]), (losses, ref)
    assert abs(losses[1] - ref[1]) <= 20 * tol * abs(ref[1]) and abs(losses[2] - ref[2]) <= 20 * tol * abs(ref[2]), (losses, ref)
    if grads is not None:
        probes = np.stack([recipes.grad_probe(i, grads[n]) for i, n in enumerate(names)])
        gp = g["grad_probes0"]
        scale = np.abs(gp[:, :1]) + 1e-12
        assert np.max(np.abs(probes[:, 0] - gp[:, 0]) / scale[:, 0]) <= 20 * tol
        assert np.max(np.abs(probes[:, 2:] - gp[:, 2:]) / scale) <= 20 * tol
    upd = np.array([float((after1[n] - before[n]).double().norm()) for n in names])
    assert np.max(np.abs(upd - g["update_l2"]) / (g["update_l2"] + 1e-12)) <= 1e-3
    lr = float(g["lr"][0])
    heads = np.stack([np.resize(after1[n].detach().cpu().numpy().ravel()[:32], 32) for n in names])
    # element-wise: every parameter moved by at most lr, and almost all of them exactly as in the reference
    close = np.abs(heads - g["param_heads1"]) <= 2e-2 * lr
    assert close.mean() >= 0.99, close.mean()
    return losses


def test_ar_training_steps_match_reference_fixture_on_cpu_wiring(oracle_backend):
    trainer, g, names = build_g9_trainer()
    assert trainer.launch == "eager"
    flipped, _handles = pin_g9_ties(trainer.model, g)
    check_g9(trainer, g, names)
    assert flipped["n"] == 0, flipped     # the oracle backend IS the reference's op sequence: every recorded mask agrees


def check_concat_in_place(device, dtype=torch.float32, tol=2e-6):
    """Decoder concatenation without the copy: an encoder-side `rezero_residual(..., out=slot)` and an
    unpooling `sparse_remap(..., out=left)` fill the two channel slices of one buffer, a pooling reads the strided
    slice; values and every gradient against the plain `torch.cat` formulation in fp64."""
    import scipy.sparse as sp
    from dsw_amd import functional as F_

    rng = np.random.default_rng(31)
    B, Vf, Vc, c_up, c_skip = 3, 48, 12, 8, 16
    up = sp.random(Vf, Vc, density=0.3, random_state=5, format="csr", dtype=np.float64) + sp.eye(Vf, Vc, format="csr")
    down = sp.random(Vc, Vf, density=0.2, random_state=6, format="csr", dtype=np.float64) + sp.eye(Vc, Vf, format="csr")

    def op_of(m):
        m = m.tocoo()
        t = torch.sparse_coo_tensor(np.stack([m.row, m.col]), m.data.astype(np.float32), m.shape).coalesce()
        return F_.get_operator(t.to(device))

    op_up, op_down = op_of(up), op_of(down)
    leaves64 = {k: torch.from_numpy(rng.standard_normal(s)) for k, s in
                (("c", (B, Vf, c_skip)), ("r", (B, Vf, c_skip)), ("xc", (B, Vc, c_up)))}
    w64 = torch.tensor([0.4], dtype=torch.float64)
    gcat64 = torch.from_numpy(rng.standard_normal((B, Vf, c_up + c_skip)))
    gpool64 = torch.from_numpy(rng.standard_normal((B, Vc, c_skip)))

    def run(t_dtype, dev, in_place):
        lv = {k: v.to(t_dtype).to(dev).requires_grad_(True) for k, v in leaves64.items()}
        w = w64.to(t_dtype).to(dev).requires_grad_(True)
        if in_place:
            slot = F_.skip_slot(lv["c"], Vf, c_up, c_skip)
            assert slot is not None and F_.row_stride(slot) == c_up + c_skip
            skip = F_.rezero_residual(lv["c"], lv["r"], w, out=slot)
            pooled = F_.sparse_remap(op_down, skip)                       # reads the strided slice
            buf = F_.skip_buffer(skip, c_up)
            assert buf is not None and buf.shape == (B, Vf, c_up + c_skip)
            left = F_.sparse_remap(op_up, lv["xc"], out=F_.left_slot(buf, c_up))
            cat = F_.concat_in_place(left, skip, buf)
            assert cat.data_ptr() == buf.data_ptr() and F_.skip_buffer(skip, c_up) is None   # one completion only
        else:
            skip = w * lv["c"] + lv["r"]
            U = torch.from_numpy(up.toarray()).to(t_dtype).to(dev)
            D = torch.from_numpy(down.toarray()).to(t_dtype).to(dev)
            pooled = torch.einsum("dv,bvf->bdf", D, skip)
            cat = torch.cat((torch.einsum("dv,bvf->bdf", U, lv["xc"]), skip), dim=2)
        loss = (cat * gcat64.to(t_dtype).to(dev)).sum() + (pooled * gpool64.to(t_dtype).to(dev)).sum()
        loss.backward()
        return cat.detach(), pooled.detach(), {k: v.grad for k, v in lv.items()}, w.grad

    cat, pooled, grads, gw = run(dtype, device, True)
    cat_r, pooled_r, grads_r, gw_r = run(torch.float64, "cpu", False)
    assert orc.max_rel_err(cat, cat_r.numpy()) <= tol
    assert orc.max_rel_err(pooled, pooled_r.numpy()) <= tol
    for k in grads:
        assert orc.max_rel_err(grads[k], grads_r[k].numpy()) <= tol, k
    assert abs(float(gw) - float(gw_r)) <= 50 * tol * max(1.0, abs(float(gw_r)))


def test_concat_in_place_cpu_wiring(oracle_backend):
    check_concat_in_place("cpu")


def test_skip_slot_is_used_by_the_unet_and_only_once(oracle_backend):
    """encode() hands out skip tensors that live in the concatenation buffers; a SECOND decode of the same encodings must
    not overwrite the first one's buffer (it takes the copying path) and gives the same values."""
    from dsw_amd import functional as F_
    model, g, names = build_g5_model("cpu")
    x = torch.from_numpy(recipes.rand(501, (2, 3, 768, 6)))
    enc = model.encode(x)
    x_enc2, x_enc1 = enc[1], enc[2]
    assert F_.skip_buffer(x_enc1, model.uconv1.convblock1.conv.in_channels - x_enc1.shape[2]) is not None
    assert F_.skip_buffer(x_enc2, model.uconv2.convblock1.conv.in_channels - x_enc2.shape[2]) is not None
    y1 = model.decode(*enc)
    keep = y1.detach().clone()
    assert F_.skip_buffer(x_enc1, model.uconv1.convblock1.conv.in_channels - x_enc1.shape[2]) is None
    y2 = model.decode(*enc)
    assert torch.equal(y1.detach(), keep)
    assert orc.max_rel_err(y2.detach(), keep.numpy()) <= 1e-6


def test_pooling_fork_gives_the_gradients_of_the_plain_graph(oracle_backend):
    """`forward_fork`: same outputs, and the input's gradient equals what autograd accumulates for the plain graph
    (pooling + a second consumer), for the index-less pooling layers; max-value pooling declares no fork."""
    from modules.layers import GeneralAvgPool, GeneralMaxAreaPool, GeneralMaxValPool
    from scipy import sparse as sp

    rng = np.random.default_rng(2)
    dense = (rng.random((24, 96)) < 0.1) * rng.random((24, 96))
    dense[np.arange(24), rng.integers(0, 96, 24)] += 0.5
    mat = sp.csr_matrix(dense.astype(np.float32))
    assert GeneralMaxValPool(mat).forward_fork is None
    for cls in (GeneralAvgPool, GeneralMaxAreaPool):
        pool = cls(mat)
        x0 = torch.from_numpy(recipes.rand(5, (2, 96, 8)))
        other_w = torch.from_numpy(recipes.rand(6, (2, 96, 8)))
        gy = torch.from_numpy(recipes.rand(7, (2, 24, 8)))
        xa = x0.clone().requires_grad_(True)
        ya, none_a = pool(xa)
        ((xa * other_w).sum() + (ya * gy).sum()).backward()
        xb = x0.clone().requires_grad_(True)
        x_again, (yb, none_b) = pool.forward_fork(xb)
        assert none_a is None and none_b is None and x_again.data_ptr() == xb.data_ptr()
        ((x_again * other_w).sum() + (yb * gy).sum()).backward()
        assert torch.equal(ya, yb)
        np.testing.assert_allclose(xb.grad.numpy(), xa.grad.numpy(), rtol=0, atol=1e-6)


def test_residual_branch_linear_is_a_drop_in_for_nn_linear():
    """`_NodeLinear` keeps torch.nn.Linear's parameter names, shapes, initial values and state_dict round trip; only the
    weight's memory order differs (column-major, so that neither pass needs a transpose copy)."""
    from modules.my_models_graph import _NodeLinear
    torch.manual_seed(3)
    mine = _NodeLinear(6, 4)
    torch.manual_seed(3)
    ref = torch.nn.Linear(6, 4)
    assert [k for k, _ in mine.named_parameters()] == [k for k, _ in ref.named_parameters()]
    assert torch.equal(mine.weight, ref.weight) and torch.equal(mine.bias, ref.bias)
    assert mine.weight.shape == ref.weight.shape and mine.weight.t().is_contiguous() and mine.weight.is_leaf
    ref2 = torch.nn.Linear(6, 4)
    mine.load_state_dict(ref2.state_dict())
    assert torch.equal(mine.weight, ref2.weight) and mine.weight.t().is_contiguous()
    ref.load_state_dict(mine.state_dict())
    assert torch.equal(ref.weight, ref2.weight) and ref.weight.is_contiguous()
    half = mine.to(torch.bfloat16)
    assert half.weight.t().is_contiguous() and half.weight.dtype == torch.bfloat16
    mine = mine.float()
    (torch.randn(5, 6) @ mine.weight.t()).sum().backward()
    assert mine.weight.grad.stride() == mine.weight.stride()        # what autograd hands the optimizer needs no re-layout


def test_cluster_tiles_random_graphs_property():
    """Random sparse graphs (isolated nodes, hubs, several components, asymmetric patterns): `cluster_tiles` always returns
    a partition into tiles of 1..R rows, with or without neighbourhood caps, and the plan built on it reproduces two
    plain hops."""
    from hypothesis import given, settings, strategies as st
    from dsw_amd import hop2

    @settings(max_examples=25, deadline=None)
    @given(n=st.integers(5, 400), deg=st.integers(0, 12), seed=st.integers(0, 10_000), R=st.sampled_from([8, 48, 64]),
           capped=st.booleans())
    def check(n, deg, seed, R, capped):
        rng = np.random.default_rng(seed)
        rows = np.repeat(np.arange(n), rng.integers(0, deg + 1, size=n))
        cols = rng.integers(0, n, size=rows.size)
        vals = rng.standard_normal(rows.size)
        A = sparse.csr_matrix((vals, (rows, cols)), shape=(n, n))
        A.sum_duplicates()
        rp, ci, va = A.indptr, A.indices, A.data.astype(np.float32)
        tiles = hop2.cluster_tiles(rp, ci, R, max_n1=40 if capped else 0, max_n2=90 if capped else 0)
        assert np.array_equal(np.sort(np.concatenate(tiles)), np.arange(n))
        assert all(1 <= len(t) <= R for t in tiles)
        plan = hop2.build_hop2_plan(rp, ci, va, R, tiles=tiles)
        U = rng.standard_normal((n, 2))
        y1, y2 = hop2.emulate_hop2(plan, U, None, None, None, 1.0, 0.0, 0.0, 2.0, -1.0, 0.0)
        L = sparse.csr_matrix((va.astype(np.float64), ci, rp), shape=(n, n))
        np.testing.assert_allclose(y1, L @ U, atol=1e-10)
        np.testing.assert_allclose(y2, 2.0 * (L @ (L @ U)) - U, atol=1e-10)

    check()


def test_row_stride_helper():
    from dsw_amd import functional as F_

    buf = torch.zeros(3, 10, 24)
    assert F_.row_stride(buf) == 24
    assert F_.row_stride(buf[..., 8:]) == 24 and F_.row_stride(buf[..., :8]) == 24
    assert F_.row_stride(buf[:, ::2, :]) == 48             # every other row: still rows at a fixed stride, samples back to back
    assert F_.row_stride(buf[:, :4, :]) is None            # a row range: the sample stride is no longer V * ld
    assert F_.row_stride(buf.transpose(1, 2)) is None      # channels not contiguous
    assert F_.row_stride(buf[0]) is None                   # not [B, V, C]
    assert F_.row_stride(torch.zeros(1, 10, 24)[..., 4:12]) == 24
    assert F_.row_stride(torch.zeros(2, 1, 16)) == 16
    slot = F_.skip_slot(torch.zeros(2, 5, 4), 5, 8, 16)
    assert slot.shape == (2, 5, 16) and F_.row_stride(slot) == 24 and slot.storage_offset() == 8
    assert F_.skip_slot(torch.zeros(2, 5, 4), 5, 3, 16) is None        # 12-byte left slice: not 16-byte aligned


def test_in_tree_library_is_the_product_build():
    """The library that travels to the GPU box reads no environment variable: the diagnostic switches of csrc/ exist only
    in a `DSW_BUILD_DIAG=1` build, which must never be what is left in the tree."""
    from dsw_amd import _native

    blob = open(_native.LIB_PATH, "rb").read()
    for name in (b"DSW_GEMM_X3", b"DSW_FWD_FUSED", b"DSW_H2_CHUNKS", b"DSW_SPMM_XCD", b"DSW_MIX_FIRST"):
        assert name not in blob, "diagnostics build in the tree: rebuild with `python -m dsw_amd.build --force`"


# ----------------------------------------------------------------------------------------------
# ADVICE round 2
# ----------------------------------------------------------------------------------------------
def test_node_linear_swag_flow(oracle_backend):
    """The reference's SWAG (modules/swag.py:33-48, utils_config.py:399) pops every entry of `_parameters`, calls
    `.to(device)`, then assigns plain tensors by attribute: `_NodeLinear` must survive `.to()` without a registered
    weight, must not re-register one afterwards, and must give the same values from a row-major plain tensor."""
    from modules.my_models_graph import _NodeLinear

    torch.manual_seed(3)
    lin = _NodeLinear(6, 4)
    assert lin.weight.stride() == (1, 4)
    x = torch.randn(2, 5, 6)
    want = torch.nn.functional.linear(x, lin.weight.detach(), lin.bias.detach())
    assert torch.allclose(lin(x), want, atol=1e-6)
    w_mean, b_mean = lin.weight.detach().clone().contiguous(), lin.bias.detach().clone()
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    ident = id(lin.weight)
    lin.float()                                        # an _apply with a registered parameter: same Parameter object
    assert id(lin.weight) == ident and opt.param_groups[0]["params"][0] is lin.weight
    for name in list(lin._parameters):                 # SWAG: parameters become plain attributes
        lin._parameters.pop(name)
    lin.to("cpu").float()                              # used to raise AttributeError: no attribute 'weight'
    lin.weight, lin.bias = w_mean, b_mean              # SWAG.sample(): plain (row-major) tensors
    lin.double().float()
    assert "weight" not in lin._parameters and not isinstance(lin.weight, torch.nn.Parameter)
    assert torch.allclose(lin(x), want, atol=1e-6)


def test_skip_buffer_ignores_lookalike_tensors():
    """A caller's tensor that merely HAS the strides of a skip slice (right-hand channel slice of its own dense wider
    tensor) is not a concatenation buffer: only storages allocated by skip_slot are ever completed in place."""
    from dsw_amd import functional as F_

    wide = torch.arange(2 * 5 * 24, dtype=torch.float32).reshape(2, 5, 24)
    lookalike = wide[..., 8:]
    assert lookalike.stride() == (5 * 24, 24, 1) and lookalike.storage_offset() == 8
    assert F_.skip_buffer(lookalike, 8) is None
    slot = F_.skip_slot(torch.zeros(2, 5, 4), 5, 8, 16)
    assert F_.skip_buffer(slot, 8) is not None
    assert F_.skip_buffer(slot, 4) is None and F_.skip_buffer(slot[:, :, :8], 8) is None   # other geometry: no
    view = slot.view_as(slot)                                                             # what autograd nodes return
    assert F_.skip_buffer(view, 8) is not None


def test_fused_activation_path_fires_module_hooks(oracle_backend):
    """ConvBlock's conv + relu runs through `ConvCheb.__call__` (activation kwarg): forward hooks / pre-hooks registered
    on the conv fire on the fused path exactly as on the plain one."""
    from modules.my_models_graph import ConvBlock

    lap = tiny_laplacian() if "tiny_laplacian" in globals() else None
    if lap is None:
        from dsw_amd import sphere
        from modules.layers import prepare_torch_laplacian
        lap = prepare_torch_laplacian(sphere.SphereHealpix(2, nest=True, k=8).L, lmax=1.9)
    block = ConvBlock(3, 5, lap, kernel_size=3)
    seen = []
    block.conv.register_forward_pre_hook(lambda m, a: seen.append("pre"))
    block.conv.register_forward_hook(lambda m, a, o: seen.append(("post", tuple(o.shape), bool((o >= 0).all()))))
    x = torch.randn(2, lap.shape[0], 3)
    y = block(x)
    assert seen == ["pre", ("post", (2, lap.shape[0], 5), True)]
    assert torch.equal(y, torch.relu(block.conv(x)))


def test_weighted_mse_keeps_fp32_weights_under_bf16():
    from modules.loss import WeightedMSELoss

    torch.manual_seed(0)
    w = torch.rand(7) + 0.5
    w = w / w.sum()
    crit = WeightedMSELoss(weights=w)
    p, t = torch.randn(3, 7, 2), torch.randn(3, 7, 2)
    loss = crit(p.bfloat16(), t.bfloat16())
    assert loss.dtype == torch.float32 and crit.weights is w and crit.weights.dtype == torch.float32
    err2 = (p.bfloat16() - t.bfloat16()) ** 2                 # the reference: bf16 element-wise error x fp32 weights
    want = (err2 * w.view(1, -1, 1)).sum() / w.sum() / 3 / 2
    assert torch.allclose(loss, want, rtol=1e-6)
    assert torch.allclose(crit(p, t), ((p - t) ** 2 * w.view(1, -1, 1)).sum() / w.sum() / 3 / 2, rtol=1e-6)


def test_bench_gpus_flag_is_not_ignored():
    import os
    """VERDICT r2: `--gpus N` used to be parsed and dropped.  Without devices the self-launcher must refuse loudly (no
    silent 1-rank run), and a torchrun environment whose WORLD_SIZE differs from --gpus must be rejected before any
    process group is created."""
    import subprocess
    import sys

    bench = os.path.join(REPO, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSW_DIST_BACKEND")}
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode != 0 and "--gpus 2" in r.stderr and "device" in r.stderr, (r.returncode, r.stderr[-400:])
        assert '"metric"' not in r.stdout
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "2"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "2 rank" in r.stderr, (r.returncode, r.stderr[-400:])


def check_resblock_tail(device, dtype=torch.float32, tol=2e-6, shapes=None):
    """ResBlock with its tail (ReZero scale + residual add) in the last convolution's epilogue and the block input's
    gradient collected inside the residual map's backward GEMM, against the plain sequence of the reference
    (my_models_graph.py:205-216) evaluated in fp64: output, input gradient, every parameter gradient."""
    from dsw_amd import sphere
    from modules.layers import prepare_torch_laplacian
    from modules.my_models_graph import ResBlock

    lap = prepare_torch_laplacian(sphere.SphereHealpix(4, nest=True, k=8).L, lmax=1.9)
    V = lap.shape[0]
    opts = dict(kernel_size=3, conv_type="graph", bias=True, batch_norm=False, batch_norm_before_activation=False,
                activation=True, activation_fun="relu", periodic_padding=True, lonlat_ratio=None)
    worst = 0.0
    for B, cin, widths in shapes or [(2, 18, (64, 128)), (2, 256, (128, 64)), (3, 64, (2,)), (2, 32, (32,)), (2, 7, (12, 5))]:
        torch.manual_seed(5)
        block = ResBlock(cin, widths if len(widths) > 1 else widths[0], laplacian=lap, convblock_kwargs=opts)
        with torch.no_grad():
            block.rezero_weight.fill_(0.37)
            for p in block.parameters():
                if p.dim() == 1 and p.numel() > 1:
                    p.normal_(0, 0.1)
        x64 = torch.from_numpy(recipes.rand(900 + cin, (B, V, cin))).double()
        g64 = torch.from_numpy(recipes.rand(901 + cin, (B, V, widths[-1]))).double()

        def reference():     # fp64, dense operator, the reference's op order
            L = lap.to_dense().double()
            P = {n: p.detach().double().requires_grad_(True) for n, p in block.named_parameters()}
            x = x64.clone().requires_grad_(True)

            def conv(h, w, b):
                t = [h, torch.einsum("vu,buf->bvf", L, h)]
                t.append(2 * torch.einsum("vu,buf->bvf", L, t[1]) - t[0])
                return sum(torch.einsum("bvf,fo->bvo", t[k], w[:, k, :]) for k in range(3)) + b
            h = x
            for i, name in enumerate(block.conv_names_list):
                h = conv(h, P[f"{name}.conv.weight"], P[f"{name}.conv.bias"])
                if i + 1 < len(block.conv_names_list):
                    h = torch.relu(h)
            res = x if cin == widths[-1] else x @ P["res_connection.weight"].t() + P["res_connection.bias"]
            out = h * P["rezero_weight"] + res
            out.backward(g64)
            return out.detach(), x.grad, {n: p.grad for n, p in P.items()}

        out_r, dx_r, gp_r = reference()
        for fused in (True, False):
            blk = block.to(device).to(dtype)
            blk.fuse_tail = fused
            blk.zero_grad(set_to_none=True)
            x = x64.to(dtype).to(device).requires_grad_(True)
            out = blk(x)
            out.backward(g64.to(dtype).to(device))
            errs = {"out": orc.max_rel_err(out.float(), out_r.numpy()), "dx": orc.max_rel_err(x.grad.float(), dx_r.numpy())}
            for n, p in blk.named_parameters():
                ref = gp_r[n].numpy()
                errs[n] = orc.max_rel_err(p.grad.float().reshape(ref.shape), ref)
            bad = {k: v for k, v in errs.items() if v > (tol if "rezero" not in k else 20 * tol)}
            assert not bad, (cin, widths, fused, bad)
            worst = max(worst, max(errs.values()))
    return worst


def test_resblock_fused_tail_cpu_wiring(oracle_backend):
    check_resblock_tail("cpu")


def test_resblock_tail_respects_hooks_and_switch(oracle_backend):
    """A forward hook on the last ConvBlock / its conv / the residual map sends the block down the plain path (the
    fused tail does not call those modules), and so does `fuse_tail = False`."""
    from dsw_amd import sphere
    from modules.layers import prepare_torch_laplacian
    from modules.my_models_graph import ResBlock

    lap = prepare_torch_laplacian(sphere.SphereHealpix(2, nest=True, k=8).L, lmax=1.9)
    opts = dict(kernel_size=3, conv_type="graph", bias=True, batch_norm=False, batch_norm_before_activation=False,
                activation=True, activation_fun="relu", periodic_padding=True, lonlat_ratio=None)
    block = ResBlock(6, (8, 12), laplacian=lap, convblock_kwargs=opts)
    x = torch.randn(2, lap.shape[0], 6)
    assert not block._fusable_tail(x)            # off by default (measured slower at the U-Net's sizes, see ResBlock)
    block.fuse_tail = True
    assert block._fusable_tail(x)
    narrow = ResBlock(6, (8, 4), laplacian=lap, convblock_kwargs=opts)
    narrow.fuse_tail = True
    assert not narrow._fusable_tail(x)           # narrow output layer: plain path
    seen = []
    h = block.convblock2.conv.register_forward_hook(lambda m, a, o: seen.append(tuple(o.shape)))
    assert not block._fusable_tail(x)
    y = block(x)
    assert seen == [(2, lap.shape[0], 12)]
    h.remove()
    assert block._fusable_tail(x) and torch.allclose(block(x), y, atol=1e-6)
    block.fuse_tail = False
    assert not block._fusable_tail(x)
    bn = ResBlock(6, (8, 12), laplacian=lap, convblock_kwargs=dict(opts, batch_norm=True))
    bn.fuse_tail = True
    assert not bn._fusable_tail(x)
