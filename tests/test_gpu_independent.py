"""The product on the GPU against the numpy models of tests/ that share no code with it or with the checker (tools/product_vs_independent.py; the CPU-side
twin of each comparison, checker against the same model, is in tests/test_independent_checks.py).  No fixture of the checker is used here."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pvi(hip_lib):
    spec = importlib.util.spec_from_file_location("product_vs_independent", os.path.join(ROOT, "tools", "product_vs_independent.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_product_tsdf_update_rule_against_the_float64_model(pvi):
    res = pvi.tsdf_rule()
    assert len(res) == 12 and all(r["ok"] for r in res.values()), res


def test_product_mesh_against_the_table_free_model(pvi):
    res = pvi.mesh_rules()
    assert res["ok"] and res["same_vertex_order_as_the_model"], res


def test_product_freespace_state_machine_against_the_numpy_model(pvi):
    res = pvi.freespace()
    assert res["ok"], res
