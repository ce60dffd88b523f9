"""Call-compatibility with the reference's own model code (SURVEY.md 8 a8 / b): the REFERENCE's
``modules/my_models_graph.py`` + ``models.py`` + ``utils_models.py`` run unchanged on top of THIS package's
``modules.layers``.  Container-only (needs /root/reference; skipped elsewhere, e.g. on the GPU box); the oracle stands
behind the layers on CPU, so what is tested is the API surface the reference's model code touches: constructor
signatures, factories, buffer / parameter names, return conventions, in-place use of the outputs."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "modules")), reason="reference checkout not present")
def test_reference_unet_runs_on_build_layers_and_matches_g5():
    # a fresh interpreter: the module table of this test session already holds the build's model files
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "reference_model_on_build_layers.py")],
                         capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "REFERENCE MODEL ON BUILD LAYERS: PASS" in out.stdout
