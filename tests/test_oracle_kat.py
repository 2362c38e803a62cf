"""Known-answer tests that pin the oracle to what /root/reference itself holds for this path.

1. nvblox_ros/test/unit_tests/test_esdf_and_gradient_conversions.cpp:36-83 (FloatGrid linearisation) and :110-157
   (EsdfValues: squared_distance_vox = Index3DHash(voxel) % 1000, observed, output = voxel_size*sqrt(v) inside the
   block, default -1000 outside, tolerance 1e-6).
2. nvblox_rviz_plugin/include/nvblox_rviz_plugin/nvblox_hash_utils.h:40-50 (Index3DHash).
Everything else about TSDF / colour / ESDF propagation / mesh is parity-unpinned in the reference.
"""
import numpy as np


def ref_hash(x, y, z):
    sl = 17191
    return (x + y * sl + z * sl * sl) % (1 << 64) % (1 << 32)      # static_cast<unsigned int>(size_t sum)


def test_index3d_hash_matches_reference_formula(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.default_rng(0)
    for x, y, z in rng.integers(-5000, 5000, size=(200, 3)).tolist() + [[0, 0, 0], [7, 7, 7], [-1, -1, -1]]:
        assert L.orc_index_hash(x, y, z) == ref_hash(x, y, z)


def test_kat_esdf_values(oracle_mod):
    o = oracle_mod.OracleMap(oracle_mod.default_params(voxel_size=0.05))
    vox = np.zeros(512, oracle_mod.ESDF_DT)
    for x in range(8):
        for y in range(8):
            for z in range(8):
                vox[z + 8 * y + 64 * x]["squared_distance_vox"] = ref_hash(x, y, z) % 1000
                vox[z + 8 * y + 64 * x]["observed"] = 1
    o.set_block(oracle_mod.L_ESDF, (0, 0, 0), vox)
    grid = o.esdf_dense_grid((0, 0, 0), (9, 9, 9), -1000.0)     # aabb min..max voxel inclusive, like the reference loop
    for x in range(9):
        for y in range(9):
            for z in range(9):
                if x < 8 and y < 8 and z < 8:
                    want = np.float32(0.05) * np.sqrt(np.float32(ref_hash(x, y, z) % 1000))
                    assert abs(grid[x, y, z] - want) <= 1e-6
                else:
                    assert abs(grid[x, y, z] + 1000.0) <= 1e-6


def test_kat_float_grid_linearisation(oracle_mod):
    """test_esdf_and_gradient_conversions.cpp:36-83: message index = x*stride_y + y*stride_z + z."""
    o = oracle_mod.OracleMap(oracle_mod.default_params())
    vox = np.zeros(512, oracle_mod.ESDF_DT)
    vox["observed"] = 1
    for x in range(8):
        for y in range(8):
            for z in range(8):
                vox[z + 8 * y + 64 * x]["squared_distance_vox"] = (ref_hash(x, y, z) % 1000) ** 2   # sqrt -> exact integer
    o.set_block(oracle_mod.L_ESDF, (0, 0, 0), vox)
    g = o.esdf_dense_grid((0, 0, 0), (2, 2, 2), 0.0)
    flat = g.reshape(-1)
    for x in range(2):
        for y in range(2):
            for z in range(2):
                assert flat[x * 4 + y * 2 + z] == np.float32(0.05) * np.float32(ref_hash(x, y, z) % 1000)


def test_inside_voxels_are_negative(oracle_mod):
    """SignedDistanceFunctor: is_inside flips the sign (esdf_and_gradients_conversions.cu:37-41)."""
    o = oracle_mod.OracleMap(oracle_mod.default_params())
    vox = np.zeros(512, oracle_mod.ESDF_DT)
    vox["observed"] = 1; vox["squared_distance_vox"] = 4.0; vox["is_inside"][:256] = 1
    o.set_block(oracle_mod.L_ESDF, (1, 2, 3), vox)
    g = o.esdf_dense_grid((8, 16, 24), (8, 8, 8), 7.0)
    assert np.allclose(g[:4], -0.1) and np.allclose(g[4:], 0.1)
    assert np.allclose(o.esdf_dense_grid((0, 0, 0), (8, 8, 8), 7.0), 7.0)
