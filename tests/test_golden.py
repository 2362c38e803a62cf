"""Regression pin: committed fixture tests/golden/tiny_sequence.npz (made by tests/golden/make_golden.py from our oracle;
the reference holds no golden vectors for this path) must be reproduced by the oracle (CPU) and by the HIP path (GPU)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
MG = importlib.util.module_from_spec(spec); spec.loader.exec_module(MG)
G = np.load(os.path.join(HERE, "golden", "tiny_sequence.npz"))


def check(summary, tol, lsb):
    assert np.array_equal(summary["tsdf_indices"], G["tsdf_indices"])
    assert np.abs(summary["tsdf_distance"] - G["tsdf_distance"]).max() <= tol
    assert np.abs(summary["tsdf_weight"] - G["tsdf_weight"]).max() <= tol
    assert np.array_equal(summary["color_indices"], G["color_indices"])
    assert np.abs(summary["color_rgb"].astype(int) - G["color_rgb"].astype(int)).max() <= lsb
    assert np.abs(summary["color_weight"] - G["color_weight"]).max() <= tol
    assert summary["esdf_slice"].shape == G["esdf_slice"].shape and np.abs(summary["esdf_slice"] - G["esdf_slice"]).max() <= tol
    assert np.array_equal(summary["esdf_aabb"], G["esdf_aabb"])
    assert np.array_equal(summary["mesh_ntri"], G["mesh_ntri"]) and np.array_equal(summary["mesh_nvert"], G["mesh_nvert"])


def test_oracle_reproduces_golden(oracle_mod):
    o = oracle_mod.OracleMap(oracle_mod.default_params())
    MG.run(o, G["depth_mm"], G["rgb"], G["poses"])
    check(MG.summarize(o, oracle_mod), 0.0, 0)


class _GpuAsOracle:
    """Adapter so make_golden.summarize() can read the HIP mapper."""
    L_TSDF, L_COLOR = 1, 2

    def __init__(self, g):
        self.g = g; self._mesh = None

    def block_indices(self, layer):
        return self.g.block_indices(layer)

    def get_block(self, layer, idx):
        return self.g.get_block(layer, idx)

    def esdf_slice_image(self, unknown):
        return self.g.esdf_slice_image(unknown)

    def mesh_block(self, idx):
        if self._mesh is None:
            self._mesh = self.g.mesh()
        return self._mesh[tuple(int(q) for q in idx)]


@pytest.mark.gpu
def test_hip_path_reproduces_golden(hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=1 << 12)
    cam = tuple(G["cam"].tolist())
    for d, c, T in zip(G["depth_mm"], G["rgb"], G["poses"]):
        g.integrate_depth(d, T, cam)           # uint16 mm path: conversion fused in the kernel
        g.integrate_color(c, T, cam)
    g.update_esdf(); g.update_color_mesh()
    a = _GpuAsOracle(g)
    check(MG.summarize(a, a), 1e-4, 1)


# ------------------------------------------------------------------------------------------------ mapping modes fixture
spec2 = importlib.util.spec_from_file_location("make_golden_modes", os.path.join(HERE, "golden", "make_golden_modes.py"))
MM = importlib.util.module_from_spec(spec2); spec2.loader.exec_module(MM)
GM = np.load(os.path.join(HERE, "golden", "tiny_modes.npz"))


def check_modes(s):
    for k in ("split_unmasked", "split_masked", "occ_indices", "occ_log_odds", "occ_slice", "occ_aabb", "esdf3_indices", "esdf3_sq",
              "esdf3_parent", "esdf3_flags"):
        assert s[k].shape == GM[k].shape, k
        assert np.array_equal(s[k], GM[k]), k            # integer / log-odds / squared-distance work: bit-exact


def test_oracle_reproduces_modes_golden(oracle_mod):
    split = lambda d, mk: oracle_mod.split_depth_by_mask(d, mk, MM.t_cm_cd(), MM.CAM, MM.MASK_CAM, 0.25)
    mk_map = lambda **kw: oracle_mod.OracleMap(oracle_mod.default_params(**kw))
    splits, occ, e3 = MM.run(split, mk_map, GM["depth_mm"], GM["masks"], GM["poses"])
    check_modes(MM.summarize(splits, occ, e3, oracle_mod.L_TSDF, oracle_mod.L_ESDF, "distance"))


@pytest.mark.gpu
def test_hip_path_reproduces_modes_golden(hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    helper = M.Mapper(M.default_params(), block_capacity=256)

    def split(d, mk):
        un, ma = helper.split_depth_by_mask(d, mk, MM.t_cm_cd(), MM.CAM, MM.MASK_CAM, 0.25)
        return un.cpu().numpy(), ma.cpu().numpy()
    mk_map = lambda **kw: M.Mapper(M.default_params(**kw), block_capacity=1 << 12)
    splits, occ, e3 = MM.run(split, mk_map, GM["depth_mm"], GM["masks"], GM["poses"])
    check_modes(MM.summarize(splits, occ, e3, M.LAYER_OCCUPANCY, M.LAYER_ESDF, "log_odds"))
