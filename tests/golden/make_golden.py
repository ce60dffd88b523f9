#!/usr/bin/env python3
"""Generate the golden fixtures ``tests/golden/G*.npz`` by running the REFERENCE itself.

Build-container only: imports ``/root/reference/modules/{layers,my_models_graph}.py``
(needs two stub modules) and records its CPU fp32 outputs.
The reference never travels to the GPU box; only the ``.npz`` data written here does.

    python tests/golden/make_golden.py

Every fixture stores the *already prepared* operator as CSR arrays (rowptr/colind int32,
values fp32) so that both sides of a parity test see bit-identical operators (ARPACK's
``estimate_lmax`` is nondeterministic).
"""
import os
import sys
import types

import numpy as np
import torch
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# -- stubs needed to import the reference (modules/layers.py:16, modules/models.py:8) -------
xs = types.ModuleType("xsphere")
xsr = types.ModuleType("xsphere.remapping")


def _no_cdo(*a, **k):
    raise NotImplementedError("xsphere/CDO not available")


xsr.compute_interpolation_weights = _no_cdo
xs.remapping = xsr
sys.modules["xsphere"] = xs
sys.modules["xsphere.remapping"] = xsr

sys.path.insert(0, REF)  # reference `modules` package wins
sys.path.append(os.path.join(REPO, "deepsphere-weather_amd"))  # for dsw_amd.sphere only
sys.path.append(HERE)
sys.path.append(REPO)

from dsw_amd import sphere  # noqa: E402  (graph/operator generator; not reference code)

pg = types.ModuleType("pygsp")
pgg = types.ModuleType("pygsp.graphs")
pgg.SphereHealpix = sphere.SphereHealpix
pgg.SphereEquiangular = sphere.SphereEquiangular
pgg.SphereIcosahedral = pgg.SphereCubed = pgg.SphereGaussLegendre = None
pg.graphs = pgg
sys.modules["pygsp"] = pg
sys.modules["pygsp.graphs"] = pgg

import modules.layers as ref_layers  # noqa: E402  -> /root/reference/modules/layers.py
import modules.my_models_graph as ref_models  # noqa: E402

assert ref_layers.__file__.startswith(REF), ref_layers.__file__

import recipes  # noqa: E402
from oracle import cheb_oracle as orc  # noqa: E402


def csr_of(t):
    return orc.csr_arrays_from_coo(t)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, keys={sorted(arrays)}")


def run_ref_conv(lap, x, w, b, gy):
    """Reference ConvCheb forward + autograd backward (modules/layers.py:183-376)."""
    layer = ref_layers.ConvCheb(w.shape[0], w.shape[2], w.shape[1], laplacian=lap, bias=b is not None)
    layer.set_parameters(w.clone(), None if b is None else b.clone())
    xin = x.clone().requires_grad_(True)
    y = layer(xin)
    y.backward(gy)
    return (
        y.detach().numpy(),
        xin.grad.numpy(),
        layer.weight.grad.numpy(),
        None if b is None else layer.bias.grad.numpy(),
    )


def conv_case(prefix, lap, B, Fin, Fout, K, seed, bias=True, permuted_input=False):
    V = lap.shape[0]
    if permuted_input:
        x = torch.from_numpy(recipes.rand(seed, (V, B, Fin))).permute(1, 0, 2)  # non-contiguous view
    else:
        x = torch.from_numpy(recipes.rand(seed, (B, V, Fin)))
    w = torch.from_numpy(recipes.rand(seed + 1, (Fin, K, Fout), np.sqrt(2.0 / (Fin * K))))
    b = torch.from_numpy(recipes.rand(seed + 2, (Fout,), 0.1)) if bias else None
    gy = torch.from_numpy(recipes.rand(seed + 3, (B, V, Fout)))
    y, dx, dw, db = run_ref_conv(lap, x, w, b, gy)
    rp, ci, va = csr_of(lap)
    out = {
        prefix + "rowptr": rp,
        prefix + "colind": ci,
        prefix + "values": va,
        prefix + "x": x.contiguous().numpy(),
        prefix + "w": w.numpy(),
        prefix + "gy": gy.numpy(),
        prefix + "y": y,
        prefix + "dx": dx,
        prefix + "dw": dw,
        prefix + "meta": np.array([B, V, Fin, Fout, K, int(bias), int(permuted_input)], dtype=np.int64),
    }
    if bias:
        out[prefix + "b"] = b.numpy()
        out[prefix + "db"] = db
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)

    # ---- G4: operator preparation with fixed lmax -----------------------------------------
    g = sphere.SphereHealpix(4, nest=True, k=8)
    L = g.L.copy()
    lmax = 1.93
    Ls = ref_layers.scale_operator(L.astype(np.float32).copy(), lmax)  # layers.py:72-79
    Lcoo = sparse.coo_matrix(Ls)
    # the tail of prepare_torch_laplacian (layers.py:93-106) on the scaled matrix
    idx = torch.from_numpy(np.stack((Lcoo.row, Lcoo.col)).astype(np.int64))
    t = torch.sparse_coo_tensor(idx, Lcoo.data, Lcoo.shape, dtype=torch.float32).coalesce()
    full = ref_layers.prepare_torch_laplacian(g.L.copy())  # with ARPACK lmax
    Lin = sparse.csr_matrix(L)
    save(
        "G4_prepare",
        in_rowptr=Lin.indptr.astype(np.int32),
        in_colind=Lin.indices.astype(np.int32),
        in_values=Lin.data.astype(np.float64),
        lmax=np.array([lmax]),
        out_indices=t.indices().numpy(),
        out_values=t.values().numpy(),
        full_indices_dtype=np.array(str(full.indices().dtype)),
        full_is_sorted=np.array(bool(((full.indices()[0][1:] * full.shape[1] + full.indices()[1][1:])
                                       > (full.indices()[0][:-1] * full.shape[1] + full.indices()[1][:-1])).all())),
        full_nnz=np.array(full._nnz()),
    )

    # ---- G1: C1 shape, HEALPix nside=16 nested, k=8 and k=20 --------------------------------
    for k in (8, 20):
        g = sphere.SphereHealpix(16, nest=True, k=k)
        lap = ref_layers.prepare_torch_laplacian(g.L.copy())
        save(f"G1_conv_c1_k{k}", **conv_case("", lap, B=2, Fin=4, Fout=8, K=3, seed=1234))

    # ---- G2: K in {1,2,3,5}, V=192, symmetric + non-symmetric operator, permuted input ------
    g = sphere.SphereHealpix(4, nest=False, k=8)  # ring order on purpose
    lap_sym = ref_layers.prepare_torch_laplacian(g.L.copy())
    rng = np.random.default_rng(77)
    M = sparse.diags(1.0 / (0.5 + rng.random(192))) @ g.L  # cotan-like Minv @ L: non-symmetric
    lap_ns = ref_layers.prepare_torch_laplacian(sparse.csr_matrix(M))
    arrays = {}
    for K in (1, 2, 3, 5):
        arrays.update(conv_case(f"sym_K{K}_", lap_sym, B=3, Fin=5, Fout=7, K=K, seed=100 + K))
        arrays.update(
            conv_case(f"ns_K{K}_", lap_ns, B=3, Fin=6, Fout=4, K=K, seed=200 + K, bias=(K != 2),
                      permuted_input=(K == 3))
        )
    save("G2_conv_K_sweep", **arrays)

    # ---- G3: RemapBlock / GeneralAvgPool / GeneralAvgUnpool 768 -> 192 -> 768 ---------------
    gs, gd = sphere.SphereHealpix(8, nest=True, k=8), sphere.SphereHealpix(4, nest=True, k=8)
    pool_h, unpool_h = sphere.healpix_pool_matrices(8, nest=True)
    pool_i, unpool_i = sphere.knn_interp_pool_matrices(gs.coords, gd.coords, k=7)
    arrays = {}
    for tag, (pm, um) in {"hier": (pool_h, unpool_h), "interp": (pool_i, unpool_i)}.items():
        pool = ref_layers.GeneralAvgPool(pm)  # layers.py:971-978
        unpool = ref_layers.GeneralAvgUnpool(um)  # layers.py:981-987
        x = torch.from_numpy(recipes.rand(300, (2, 768, 6))).requires_grad_(True)
        yp, none_idx = pool(x)
        assert none_idx is None
        gyp = torch.from_numpy(recipes.rand(301, (2, 192, 6)))
        yp.backward(gyp)
        xu = torch.from_numpy(recipes.rand(302, (2, 192, 6))).requires_grad_(True)
        yu = unpool(xu, None)
        gyu = torch.from_numpy(recipes.rand(303, (2, 768, 6)))
        yu.backward(gyu)
        prp, pci, pva = csr_of(pool.remap_matrix)
        urp, uci, uva = csr_of(unpool.remap_matrix)
        arrays.update({
            f"{tag}_pool_rowptr": prp, f"{tag}_pool_colind": pci, f"{tag}_pool_values": pva,
            f"{tag}_unpool_rowptr": urp, f"{tag}_unpool_colind": uci, f"{tag}_unpool_values": uva,
            f"{tag}_x": x.detach().numpy(), f"{tag}_yp": yp.detach().numpy(), f"{tag}_gyp": gyp.numpy(),
            f"{tag}_dxp": x.grad.numpy(), f"{tag}_xu": xu.detach().numpy(), f"{tag}_yu": yu.detach().numpy(),
            f"{tag}_gyu": gyu.numpy(), f"{tag}_dxu": xu.grad.numpy(),
            f"{tag}_yp_strides": np.array(yp.stride()), f"{tag}_yu_strides": np.array(yu.stride()),
        })
    save("G3_remap", **arrays)

    # ---- G6: irregular-degree operator (row lengths 5..200) --------------------------------
    rp, ci, va = recipes.irregular_operator(1024, seed=61, min_deg=5, max_deg=200)
    lap_irr = orc.coo_from_csr_arrays(rp, ci, va, (1024, 1024))
    save("G6_conv_irregular", **conv_case("", lap_irr, B=2, Fin=8, Fout=16, K=3, seed=600))

    # ---- G7: error paths -------------------------------------------------------------------
    msgs = {}
    try:
        ref_layers.conv_cheb(lap_sym, torch.zeros(1, 192, 3), torch.zeros(4, 3, 2))
    except ValueError as e:
        msgs["fin_mismatch"] = str(e)
    try:
        ref_layers.GeneralConvBlock.getConvLayer(4, 4, 3, conv_type="mesh", laplacian=lap_sym)
    except ValueError as e:
        msgs["bad_conv_type"] = str(e)
    try:
        ref_layers.PoolUnpoolBlock.getGeneralPoolUnpoolLayer  # exists
        msgs["learn"] = "NotImplementedError"
    except Exception:  # pragma: no cover
        pass
    save("G7_errors", **{k: np.array(v) for k, v in msgs.items()})

    # ---- G5: UNetSpherical nside=8 (V=768/192/48), B=2 --------------------------------------
    ref_layers.build_pooling_matrices = sphere.build_pooling_matrices  # layers.py:1172 global lookup
    V = 768
    tensor_info = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
    }
    torch.manual_seed(10)
    model = ref_models.UNetSpherical(
        tensor_info, sampling="healpix", sampling_kwargs={"subdivisions": 8, "nest": True},
        kernel_size_conv=3, conv_type="graph", graph_type="knn", knn=20, pool_method="interp",
    )
    names = sorted(n for n, _ in model.named_parameters())
    params = dict(model.named_parameters())
    with torch.no_grad():
        for i, n in enumerate(names):
            params[n].copy_(torch.from_numpy(recipes.unet_param_fill(i, n, tuple(params[n].shape))))
    x = torch.from_numpy(recipes.rand(501, (2, 3, V, 6)))
    y = model(x)
    target = torch.from_numpy(recipes.rand(502, (2, 1, V, 2)))
    loss = ((y - target) ** 2).mean()
    loss.backward()
    arrays = {
        "y": y.detach().numpy(), "loss": np.array([loss.item()]),
        "param_names": np.array(names), "state_keys": np.array(list(model.state_dict().keys())),
        "param_shapes": np.array([str(tuple(params[n].shape)) for n in names]),
        "grad_probes": np.stack([recipes.grad_probe(i, params[n].grad.numpy()) for i, n in enumerate(names)]),
    }
    for lvl, lap in enumerate(model.laplacians):
        rp, ci, va = csr_of(lap)
        arrays.update({f"lap{lvl}_rowptr": rp, f"lap{lvl}_colind": ci, f"lap{lvl}_values": va})
    for nm in ("pool1", "unpool1", "pool2", "unpool2"):
        m = getattr(model, nm).remap_matrix
        rp, ci, va = csr_of(m)
        arrays.update({f"{nm}_rowptr": rp, f"{nm}_colind": ci, f"{nm}_values": va,
                       f"{nm}_shape": np.array(m.shape)})
    save("G5_unet_nside8", **arrays)
    print("n_params", sum(p.numel() for p in model.parameters()))


if __name__ == "__main__":
    main()
