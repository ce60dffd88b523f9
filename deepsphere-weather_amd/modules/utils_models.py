"""Sampling-name helpers (counterpart of ``/root/reference/modules/utils_models.py``).

pygsp's ``sphere-graphs`` branch is used when importable; otherwise the self-contained builders of
``dsw_amd.sphere`` stand in for the samplings they cover (healpix, equiangular).
"""
try:  # pragma: no cover - pygsp is not installable in the build image
    import pygsp as _pygsp

    _GRAPHS = {
        "healpix": _pygsp.graphs.SphereHealpix,
        "equiangular": _pygsp.graphs.SphereEquiangular,
        "icosahedral": _pygsp.graphs.SphereIcosahedral,
        "cubed": _pygsp.graphs.SphereCubed,
        "gauss": _pygsp.graphs.SphereGaussLegendre,
    }
except Exception:
    from dsw_amd import sphere as _sphere

    def _missing(name):
        def ctor(*args, **kwargs):
            raise ImportError(f"sampling '{name}' needs pygsp (sphere-graphs branch), which is not installed")

        return ctor

    _GRAPHS = {
        "healpix": _sphere.SphereHealpix,
        "equiangular": _sphere.SphereEquiangular,
        "icosahedral": _missing("icosahedral"),
        "cubed": _missing("cubed"),
        "gauss": _missing("gauss"),
    }

_SKIP_CONNECTIONS = ("none", "stack", "sum", "avg")


def get_pygsp_graph_dict():
    return dict(_GRAPHS)


def get_valid_pygsp_graph():
    return list(_GRAPHS)


def check_sampling(sampling):
    if not isinstance(sampling, str):
        raise TypeError("'sampling' must be a string.")
    sampling = sampling.lower()
    if sampling not in _GRAPHS:
        raise ValueError("'sampling' must be one of {}.".format(get_valid_pygsp_graph()))
    return sampling


def check_conv_type(conv_type, sampling):
    if not isinstance(conv_type, str):
        raise TypeError("'conv_type' must be a string.")
    if not isinstance(sampling, str):
        raise TypeError("'sampling' must be a string.")
    conv_type = conv_type.lower()
    if conv_type not in ("graph", "image"):
        raise ValueError("'conv_type' must be either 'graph' or 'image'.")
    if conv_type == "image" and sampling.lower() != "equiangular":
        raise ValueError("conv_type='image' is available only if sampling='equiangular'.")
    return conv_type


def check_pool_method(pool_method):
    return pool_method.lower()


def check_skip_connection(skip_connection):
    if skip_connection is None:
        return "none"
    if not isinstance(skip_connection, str):
        raise TypeError("'skip_connection' must be a string.")
    if skip_connection not in _SKIP_CONNECTIONS:
        raise ValueError("'skip_connection' must be one of {}".format(_SKIP_CONNECTIONS))
    return skip_connection


def get_pygsp_graph_fun(sampling):
    return _GRAPHS[check_sampling(sampling)]


def get_pygsp_graph(sampling, sampling_kwargs, knn=20):
    sampling_kwargs["k"] = knn
    return get_pygsp_graph_fun(sampling)(**sampling_kwargs)


def pygsp_graph_coarsening(sampling, sampling_kwargs, coarsening):
    """Sampling kwargs of the next-coarser U-Net level."""
    coarse = dict(sampling_kwargs)
    if sampling == "equiangular":
        coarse["nlat"] //= coarsening
        coarse["nlon"] //= coarsening
    elif sampling in ("icosahedral", "cubed", "healpix"):
        coarse["subdivisions"] //= coarsening
    elif sampling == "gauss":
        coarse["nlat"] //= coarsening
    return coarse
