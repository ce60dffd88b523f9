"""The oracle against the golden vectors recorded from the reference itself (CPU only)."""
import numpy as np
import pytest
import torch
from scipy import sparse

from conftest import load_golden
from oracle import cheb_oracle as orc

TOL32 = 1e-5   # fp32 restatement vs reference fp32
TOL64 = 2e-6   # fp64 closed form vs reference fp32 (reference's own fp32 error is < 1e-6)


def _conv_case(g, prefix=""):
    B, V, Fin, Fout, K, has_bias, _ = [int(v) for v in g[prefix + "meta"]]
    rp, ci, va = g[prefix + "rowptr"], g[prefix + "colind"], g[prefix + "values"]
    x, w, gy = g[prefix + "x"], g[prefix + "w"], g[prefix + "gy"]
    b = g[prefix + "b"] if has_bias else None
    return (B, V, Fin, Fout, K, has_bias), (rp, ci, va), (x, w, b, gy)


def _check_conv(g, prefix=""):
    (B, V, Fin, Fout, K, has_bias), (rp, ci, va), (x, w, b, gy) = _conv_case(g, prefix)
    # restatement 1: same torch op sequence
    lap = orc.coo_from_csr_arrays(rp, ci, va, (V, V))
    y, dx, dw, db = orc.conv_cheb_fwd_bwd_torch(
        lap, torch.from_numpy(x), torch.from_numpy(w), None if b is None else torch.from_numpy(b),
        torch.from_numpy(gy),
    )
    assert orc.max_rel_err(y, g[prefix + "y"]) <= TOL32
    assert orc.max_rel_err(dx, g[prefix + "dx"]) <= TOL32
    assert orc.max_rel_err(dw, g[prefix + "dw"]) <= TOL32
    if has_bias:
        assert orc.max_rel_err(db, g[prefix + "db"]) <= TOL32
    # restatement 2: fp64 closed form with hand-derived backward
    y64 = orc.cheb_forward_f64(rp, ci, va, x, w, b)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, x, w, gy, has_bias=bool(has_bias))
    assert orc.max_rel_err(g[prefix + "y"], y64) <= TOL64
    assert orc.max_rel_err(g[prefix + "dx"], dx64) <= TOL64
    assert orc.max_rel_err(g[prefix + "dw"], dw64) <= TOL64
    if has_bias:
        assert orc.max_rel_err(g[prefix + "db"], db64) <= TOL64


@pytest.mark.parametrize("name", ["G1_conv_c1_k8", "G1_conv_c1_k20", "G6_conv_irregular"])
def test_conv_fixtures(name):
    _check_conv(load_golden(name))


@pytest.mark.parametrize("kind", ["sym", "ns"])
@pytest.mark.parametrize("K", [1, 2, 3, 5])
def test_conv_K_sweep(kind, K):
    _check_conv(load_golden("G2_conv_K_sweep"), f"{kind}_K{K}_")


@pytest.mark.parametrize("tag", ["hier", "interp"])
def test_remap_fixture(tag):
    g = load_golden("G3_remap")
    for which, xin, yout, gyin, dxout in (
        ("pool", "x", "yp", "gyp", "dxp"),
        ("unpool", "xu", "yu", "gyu", "dxu"),
    ):
        rp, ci, va = (g[f"{tag}_{which}_{k}"] for k in ("rowptr", "colind", "values"))
        x, y_ref, gy, dx_ref = g[f"{tag}_{xin}"], g[f"{tag}_{yout}"], g[f"{tag}_{gyin}"], g[f"{tag}_{dxout}"]
        shape = (y_ref.shape[1], x.shape[1])
        m = orc.coo_from_csr_arrays(rp, ci, va, shape)
        y = orc.remap_torch(m, torch.from_numpy(x))
        assert orc.max_rel_err(y, y_ref) <= TOL32
        # the reference returns a PERMUTED VIEW (layers.py:963); the restatement reproduces its recorded strides
        assert tuple(y.stride()) == tuple(int(v) for v in g[f"{tag}_{yout}_strides"])
        assert orc.max_rel_err(y_ref, orc.remap_f64(rp, ci, va, shape, x)) <= TOL64
        assert orc.max_rel_err(dx_ref, orc.remap_backward_f64(rp, ci, va, shape, gy)) <= TOL64
        # invariants the reference asserts on its pooling matrices (layers.py:557-571): rows sum to 1
        rows = np.add.reduceat(va, rp[:-1][np.diff(rp) > 0])
        np.testing.assert_allclose(rows, 1.0, rtol=1e-5)


def test_prepare_fixture():
    g = load_golden("G4_prepare")
    n = len(g["in_rowptr"]) - 1
    L = sparse.csr_matrix((g["in_values"], g["in_colind"], g["in_rowptr"]), shape=(n, n))
    t = orc.prepare_laplacian_fixed_lmax(L, float(g["lmax"][0]))
    assert t.indices().dtype == torch.int64
    np.testing.assert_array_equal(t.indices().numpy(), g["out_indices"])
    np.testing.assert_allclose(t.values().numpy(), g["out_values"], rtol=0, atol=1e-7)
    assert str(g["full_indices_dtype"]) == "torch.int64" and bool(g["full_is_sorted"])
    # CSR round trip of the coalesced COO
    rp, ci, va = orc.csr_arrays_from_coo(t)
    back = orc.coo_from_csr_arrays(rp, ci, va, (n, n))
    np.testing.assert_array_equal(back.indices().numpy(), t.indices().numpy())


def test_error_fixture():
    g = load_golden("G7_errors")
    assert "Input tensor shape does not match the expected shape" in str(g["fin_mismatch"])
    assert "conv_type is not supported" in str(g["bad_conv_type"])
    lap = orc.coo_from_csr_arrays(np.array([0, 1, 2]), np.array([0, 1]), np.array([1.0, 1.0], dtype=np.float32), (2, 2))
    with pytest.raises(ValueError, match="does not match the expected shape"):
        orc.conv_cheb_torch(lap, torch.zeros(1, 2, 3), torch.zeros(4, 3, 2))


# ---------------------------------------------------------------------------------------------
# plain-C restatement (oracle/cheb_oracle.c) against the same fixtures
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,prefix", [("G1_conv_c1_k20", ""), ("G6_conv_irregular", ""),
                                         ("G2_conv_K_sweep", "ns_K5_"), ("G2_conv_K_sweep", "sym_K1_"),
                                         ("G2_conv_K_sweep", "ns_K2_")])
def test_c_oracle_conv(name, prefix):
    from oracle import c_oracle

    g = load_golden(name)
    (B, V, Fin, Fout, K, has_bias), (rp, ci, va), (x, w, b, gy) = _conv_case(g, prefix)
    y, basis = c_oracle.cheb_forward(rp, ci, va, x, w, b)
    dx, dw, db = c_oracle.cheb_backward(rp, ci, va, basis, w, gy, bool(has_bias))
    assert orc.max_rel_err(g[prefix + "y"], y) <= TOL64
    assert orc.max_rel_err(g[prefix + "dx"], dx) <= TOL64
    assert orc.max_rel_err(g[prefix + "dw"], dw) <= TOL64
    if has_bias:
        assert orc.max_rel_err(g[prefix + "db"], db) <= TOL64


def test_c_oracle_remap():
    from oracle import c_oracle

    g = load_golden("G3_remap")
    rp, ci, va = (g[f"interp_pool_{k}"] for k in ("rowptr", "colind", "values"))
    y = c_oracle.remap(rp, ci, va, (192, 768), g["interp_x"])
    assert orc.max_rel_err(g["interp_yp"], y) <= TOL64


# ---------------------------------------------------------------------------------------------
# UNet-level restatement (oracle/unet_oracle.py) against fixture G5 (outputs of the reference model itself)
# ---------------------------------------------------------------------------------------------
def _g5_state(g):
    """The reference model's state_dict of fixture G5, rebuilt from the fixture's operators and the seeded recipe."""
    import recipes

    names = [str(n) for n in g["param_names"]]
    shapes = [eval(str(s)) for s in g["param_shapes"]]          # "(64, 3, 128)" strings written by make_golden.py
    sd = {}
    for i, (n, shp) in enumerate(zip(names, shapes)):
        sd[n] = torch.from_numpy(recipes.unet_param_fill(i, n, tuple(shp))).requires_grad_(True)
    laps = {}
    for i in range(3):
        rp = g[f"lap{i}_rowptr"]
        laps[len(rp) - 1] = orc.coo_from_csr_arrays(rp, g[f"lap{i}_colind"], g[f"lap{i}_values"], (len(rp) - 1,) * 2)
    level = {"conv1": 768, "conv2": 192, "conv3": 48, "uconv2": 192, "uconv1": 768, "uconv1_final": 768}
    for key in (str(k) for k in g["state_keys"]):
        if key.endswith("laplacian"):
            sd[key] = laps[level[key.split(".")[0]]]
        elif key.endswith("remap_matrix"):
            nm = key.split(".")[0]
            sd[key] = orc.coo_from_csr_arrays(g[f"{nm}_rowptr"], g[f"{nm}_colind"], g[f"{nm}_values"], tuple(g[f"{nm}_shape"]))
    return sd, names


def test_unet_oracle_matches_reference_fixture():
    import recipes
    from oracle import unet_oracle

    g = load_golden("G5_unet_nside8")
    sd, names = _g5_state(g)
    x = torch.from_numpy(recipes.rand(501, (2, 3, 768, 6)))
    target = torch.from_numpy(recipes.rand(502, (2, 1, 768, 2)))
    y, loss, grads = unet_oracle.unet_fwd_bwd(sd, x, target)
    assert orc.max_rel_err(y, g["y"]) <= TOL32
    assert abs(loss - float(g["loss"][0])) <= TOL32 * max(1.0, float(g["loss"][0]))
    probes = np.stack([recipes.grad_probe(i, grads[n].numpy()) for i, n in enumerate(names)])
    ref = g["grad_probes"]
    scale = np.abs(ref[:, :1]) + 1e-12
    assert np.max(np.abs(probes[:, 0] - ref[:, 0]) / scale[:, 0]) <= 1e-4
    assert np.max(np.abs(probes[:, 2:] - ref[:, 2:]) / scale) <= 1e-4


# ---------------------------------------------------------------------------------------------
# max-value / max-area pooling restatements against fixture G8 (outputs of the reference's classes)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["hier", "interp"])
def test_maxval_oracle_matches_reference_fixture(tag):
    g = load_golden("G8_maxpool")
    rp, ci, va = (g[f"{tag}_pool_{k}"] for k in ("rowptr", "colind", "values"))
    m = orc.coo_from_csr_arrays(rp, ci, va, (192, 768))
    x = torch.from_numpy(g[f"{tag}_mv_x"]).requires_grad_(True)
    # restatement 1: the reference's torch op sequence
    yp, idx = orc.maxval_pool_torch(m, x)
    assert torch.equal(idx, torch.from_numpy(g[f"{tag}_mv_index"]))
    assert torch.equal(yp.detach().contiguous(), torch.from_numpy(g[f"{tag}_mv_yp"]))
    yp.backward(torch.from_numpy(g[f"{tag}_mv_gyp"]))
    assert orc.max_rel_err(x.grad, g[f"{tag}_mv_dxp"]) <= 1e-6
    xu = torch.from_numpy(g[f"{tag}_mv_xu"]).requires_grad_(True)
    yu = orc.maxval_unpool_torch(768, xu, idx)
    assert torch.equal(yu.detach().contiguous(), torch.from_numpy(g[f"{tag}_mv_yu"]))
    yu.backward(torch.from_numpy(g[f"{tag}_mv_gyu"]))
    assert torch.equal(xu.grad, torch.from_numpy(g[f"{tag}_mv_dxu"]))
    # restatement 2: numpy, native layout, compact int32 selection; and the index conversion both ways
    y2, sel = orc.maxval_pool_np(rp, ci, va, g[f"{tag}_mv_x"])
    np.testing.assert_array_equal(y2, g[f"{tag}_mv_yp"])
    B, D, F = sel.shape
    ref_row = g[f"{tag}_mv_index"][0].reshape(F, B, D).transpose(1, 2, 0)       # column c = f*B + b, column-major list
    np.testing.assert_array_equal(sel, ref_row)
    np.testing.assert_array_equal(g[f"{tag}_mv_index"][1], np.repeat(np.arange(F * B), D))
    assert orc.max_rel_err(orc.maxval_pool_backward_np(sel, 768, g[f"{tag}_mv_gyp"]), g[f"{tag}_mv_dxp"]) <= 1e-6
    np.testing.assert_array_equal(orc.maxval_unpool_np(sel, 768, g[f"{tag}_mv_xu"]), g[f"{tag}_mv_yu"])
    np.testing.assert_array_equal(orc.maxval_unpool_backward_np(sel, g[f"{tag}_mv_gyu"]), g[f"{tag}_mv_dxu"])
    # max-area pooling is a remap with a 0/1 selection matrix
    for which, xin, yout, dxout, gy, shape in (("pool", "x", "yp", "dxp", "gyp", (192, 768)),
                                               ("unpool", "xu", "yu", "dxu", "gyu", (768, 192))):
        arp, aci, ava = (g[f"{tag}_ma_{which}_{k}"] for k in ("rowptr", "colind", "values"))
        assert set(np.unique(ava)) == {1.0}
        assert orc.max_rel_err(g[f"{tag}_ma_{yout}"], orc.remap_f64(arp, aci, ava, shape, g[f"{tag}_ma_{xin}"])) <= 1e-6
        assert orc.max_rel_err(g[f"{tag}_ma_{dxout}"], orc.remap_backward_f64(arp, aci, ava, shape, g[f"{tag}_mv_{gy}"])) <= 1e-6
