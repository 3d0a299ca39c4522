import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


_OPTION_DEFAULTS = {"kmeans_window_rows": 0, "kmeans_lane_form": 0, "kmeans_no_graph": 0, "opq_scratch_rows": 0,
                    "opq_fused": 1, "opq_gather_rotation": 1, "adc_single_query": 0, "cross_product_exact": 1,
                    "cross_product_group_bytes": 0, "lookup_two_pass": 2}


@pytest.fixture
def ctx_options():
    """Set per-context options of the default context (include/pqhip.h: pqhip_ctx_set_option) for one test; every
    option is back at its default afterwards."""
    import reductive_amd
    touched = []

    def set_option(name, value):
        assert name in _OPTION_DEFAULTS, name
        reductive_amd.set_option(name, value)
        touched.append(name)
    yield set_option
    for name in touched:
        reductive_amd.set_option(name, _OPTION_DEFAULTS[name])
