#!/usr/bin/env python3
"""Container-only check: the reference's own UNetSpherical (its my_models_graph.py / models.py / utils_models.py, loaded
from /root/reference) built on THIS repository's ``modules.layers``, forward + backward on CPU with the oracle behind
the layers, compared with fixture G5 (which the reference produced on its own layers).

    python tools/reference_model_on_build_layers.py [/path/to/reference]
"""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "deepsphere-weather_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dsw_amd import functional, integrate, sphere  # noqa: E402

ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

# the reference's utils_models / models import pygsp (absent): stand-in with this repo's graph builders
pg, pgg = types.ModuleType("pygsp"), types.ModuleType("pygsp.graphs")
pgg.SphereHealpix, pgg.SphereEquiangular = sphere.SphereHealpix, sphere.SphereEquiangular
pgg.SphereIcosahedral, pgg.SphereCubed, pgg.SphereGaussLegendre = sphere.SphereIcosahedral, sphere.SphereCubed, sphere.SphereGaussLegendre
pg.graphs = pgg
sys.modules["pygsp"], sys.modules["pygsp.graphs"] = pg, pgg

utils_models, models, arch = integrate.load_reference_modules(ref_root)
import modules.layers as layers  # noqa: E402

assert arch.__file__.startswith(ref_root) and layers.__file__.startswith(REPO), (arch.__file__, layers.__file__)

from _oracle_backend import OracleBackend  # noqa: E402
from conftest import load_golden  # noqa: E402
from oracle import cheb_oracle as orc  # noqa: E402
import recipes  # noqa: E402

functional.set_test_backend(OracleBackend())
g = load_golden("G5_unet_nside8")
V = 768
tensor_info = {
    "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
    "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
    "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
}
model = arch.UNetSpherical(tensor_info, sampling="healpix", sampling_kwargs={"subdivisions": 8, "nest": True},
                           kernel_size_conv=3, conv_type="graph", graph_type="knn", knn=20, pool_method="interp")
assert type(model.conv1.convblock1.conv).__module__ == "modules.layers" and isinstance(model.conv1.convblock1.conv, layers.ConvCheb)
assert [str(k) for k in g["state_keys"]] == list(model.state_dict().keys())
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
names = sorted(n for n, _ in model.named_parameters())
for i, n in enumerate(names):
    sd[n] = torch.from_numpy(recipes.unet_param_fill(i, n, tuple(sd[n].shape)))
model.load_state_dict(sd, strict=True)
x = torch.from_numpy(recipes.rand(501, (2, 3, V, 6)))
target = torch.from_numpy(recipes.rand(502, (2, 1, V, 2)))
y = model(x)                      # the reference's ResBlock multiplies / adds IN PLACE on the layer outputs
loss = ((y - target) ** 2).mean()
loss.backward()
err_y = orc.max_rel_err(y.detach(), g["y"])
params = dict(model.named_parameters())
probes = np.stack([recipes.grad_probe(i, params[n].grad.numpy()) for i, n in enumerate(names)])
ref = g["grad_probes"]
err_g = float(np.max(np.abs(probes[:, 0] - ref[:, 0]) / (np.abs(ref[:, 0]) + 1e-12)))
print("max-rel error vs G5: y %.2e, loss %.2e, gradient norms %.2e" % (err_y, abs(loss.item() - float(g["loss"][0])), err_g))
assert err_y <= 1e-5 and abs(loss.item() - float(g["loss"][0])) <= 1e-5 and err_g <= 1e-4
print("REFERENCE MODEL ON BUILD LAYERS: PASS")
