"""Drop-in replacement for the hot-path part of the reference's ``modules`` package.

Put ``deepsphere-weather_amd/`` ahead of the reference checkout on ``sys.path`` and
``import modules.my_models_graph as my_architectures`` (``scripts_training/train_predict_state.py:38``)
resolves here; see INTEGRATION.md.
"""
