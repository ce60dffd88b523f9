"""Sampling-name helpers of the spherical models.

API-compatible counterpart of the reference's ``modules/utils_models.py`` (same function names,
arguments, return values and exception types); the implementation is table driven.  Graph classes
come from pygsp's ``sphere-graphs`` branch when it is importable, otherwise from the self-contained
builders in ``dsw_amd.sphere`` (all five samplings of the reference's table, ``utils_models.py:11-20``).
"""
from dsw_amd import sphere as _sphere

_SAMPLINGS = ("healpix", "equiangular", "icosahedral", "cubed", "gauss")
_PYGSP_CLASS = dict(zip(_SAMPLINGS, ("SphereHealpix", "SphereEquiangular", "SphereIcosahedral",
                                     "SphereCubed", "SphereGaussLegendre")))
_BUILTIN = {"healpix": _sphere.SphereHealpix, "equiangular": _sphere.SphereEquiangular,
            "icosahedral": _sphere.SphereIcosahedral, "cubed": _sphere.SphereCubed, "gauss": _sphere.SphereGaussLegendre}
# which sampling kwargs shrink (integer division) when going one U-Net level down
_COARSEN_KEYS = {"equiangular": ("nlat", "nlon"), "healpix": ("subdivisions",), "icosahedral": ("subdivisions",),
                 "cubed": ("subdivisions",), "gauss": ("nlat",)}
_SKIP_MODES = ("none", "stack", "sum", "avg")


def _unavailable(name):
    def ctor(*_args, **_kwargs):
        raise ImportError("sampling '%s' needs pygsp (sphere-graphs branch), which is not installed" % name)

    return ctor


def _graph_table():
    try:  # pragma: no cover - pygsp cannot be installed in the build image
        import pygsp

        return {name: getattr(pygsp.graphs, cls) for name, cls in _PYGSP_CLASS.items()}
    except Exception:
        return {name: _BUILTIN.get(name, _unavailable(name)) for name in _SAMPLINGS}


def _require_str(value, label):
    if not isinstance(value, str):
        raise TypeError("'%s' must be a string." % label)
    return value.lower()


def get_pygsp_graph_dict():
    """sampling name -> graph class."""
    return _graph_table()


def get_valid_pygsp_graph():
    return list(_SAMPLINGS)


def check_sampling(sampling):
    name = _require_str(sampling, "sampling")
    if name not in _SAMPLINGS:
        raise ValueError("'sampling' must be one of {}.".format(list(_SAMPLINGS)))
    return name


def check_conv_type(conv_type, sampling):
    kind = _require_str(conv_type, "conv_type")
    grid = _require_str(sampling, "sampling")
    if kind not in ("graph", "image"):
        raise ValueError("'conv_type' must be either 'graph' or 'image'.")
    if kind == "image" and grid != "equiangular":
        raise ValueError("conv_type='image' is available only if sampling='equiangular'.")
    return kind


def check_pool_method(pool_method):
    return pool_method.lower()


def check_skip_connection(skip_connection):
    mode = "none" if skip_connection is None else skip_connection
    if not isinstance(mode, str):
        raise TypeError("'skip_connection' must be a string.")
    if mode not in _SKIP_MODES:
        raise ValueError("'skip_connection' must be one of {}".format(_SKIP_MODES))
    return mode


def get_pygsp_graph_fun(sampling):
    return _graph_table()[check_sampling(sampling)]


def get_pygsp_graph(sampling, sampling_kwargs, knn=20):
    sampling_kwargs["k"] = knn
    return get_pygsp_graph_fun(sampling)(**sampling_kwargs)


def pygsp_graph_coarsening(sampling, sampling_kwargs, coarsening):
    """kwargs of the sampling one U-Net level down (``coarsening`` = linear factor)."""
    out = dict(sampling_kwargs)
    for key in _COARSEN_KEYS.get(sampling, ()):
        out[key] = out[key] // coarsening
    return out
