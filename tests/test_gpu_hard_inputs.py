"""GPU parity on inputs that are NOT the noise-free axis-aligned room: sensor noise, speckle invalids, a scene oblique to the voxel
grid, non-square intrinsics, odd image sizes, other ray subsampling factors.  Each case runs the whole path -- TSDF, colour, ESDF
(incremental), mesh -- through the C-ABI against the oracle: block-index sets bit-exact, voxel values 1e-4 (SURVEY.md 8d [D] names
the sigma = 5 mm variant; BASELINE.json north_star states the tolerances)."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S
from test_gpu_parity import TOL, compare_layer, make_pair

pytestmark = pytest.mark.gpu

ESDF_FIELDS = ("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site")


def world_rotation(yaw_deg=17.0, roll_deg=9.0, shift=(0.013, -0.027, 0.041)):
    """Rigid transform W applied to every pose: the room's walls end up oblique to all three voxel axes."""
    a, b = np.deg2rad(yaw_deg), np.deg2rad(roll_deg)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    Rx = np.array([[1.0, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    W = np.eye(4); W[:3, :3] = Rz @ Rx; W[:3, 3] = shift
    return W


CASES = {
    # name: (camera, n_frames, stride, noise sigma, invalid fraction, world transform, extra params)
    "noise_5mm": (H.SMALL_CAM, 5, 7, 0.005, 0.0, None, {}),
    "speckle_2pct_invalid": (H.SMALL_CAM, 5, 7, 0.0, 0.02, None, {}),
    "noise_and_speckle_inverse_square": (H.SMALL_CAM, 4, 9, 0.005, 0.02, None, dict(weighting_mode=4)),
    "oblique_17_9_deg": (H.SMALL_CAM, 5, 7, 0.0, 0.0, world_rotation(), {}),
    "oblique_noisy": (H.SMALL_CAM, 4, 9, 0.005, 0.01, world_rotation(33.0, -14.0), dict(esdf_slice_height=0.3, esdf_slice_min_height=0.1, esdf_slice_max_height=0.9)),
    "fu_ne_fv_odd_size": ((85.0, 78.0, 81.3, 58.1, 161, 119), 5, 7, 0.002, 0.0, None, {}),
    "subsample_3": (H.SMALL_CAM, 4, 7, 0.0, 0.0, None, dict(raycast_subsampling_factor=3, sphere_tracing_subsampling=2)),
    "subsample_1_nearest": ((85.0, 78.0, 81.3, 58.1, 161, 119), 3, 11, 0.003, 0.01, None, dict(raycast_subsampling_factor=1, depth_interp_nearest=1)),
    "full_res_641x479": ((321.0, 317.0, 320.2, 238.7, 641, 479), 2, 15, 0.005, 0.02, world_rotation(), {}),
}


def make_frames(cam, n, stride, sigma, invalid, W):
    sc = S.Scene()
    rng = np.random.default_rng(7)
    out = []
    for i in range(n):
        T = S.trajectory_pose(i * stride)
        d, rgb = S.render(sc, T, cam, color=True, noise_sigma=sigma, rng=rng)
        if invalid > 0:
            d = np.where(rng.random(d.shape) < invalid, np.float32(0.0), d).astype(np.float32)
        if W is not None:
            T = (W @ T.astype(np.float64)).astype(np.float32)
        out.append((d, rgb, T))
    return out


@pytest.mark.parametrize("case", list(CASES))
def test_whole_path_parity_on_hard_inputs(oracle_mod, hip_lib, case):
    cam, n, stride, sigma, invalid, W, kw = CASES[case]
    M, g, o = make_pair(oracle_mod, **kw)
    for k, (d, rgb, T) in enumerate(make_frames(cam, n, stride, sigma, invalid, W)):
        g.integrate_depth(d, T, cam); o.integrate_depth(d, T, cam)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view()), (case, k)
        g.integrate_color(rgb, T, cam); o.integrate_color(rgb, T, cam)
        sg, so = g.synthetic_depth(), o.synthetic_depth()
        assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL
        assert H.idx_set(g.last_color_view()) == H.idx_set(o.last_color_view())
        if k % 2 == 1 or k == n - 1:
            g.update_esdf(); o.update_esdf()
            ig, ag = g.esdf_slice_image(1000.0); io, ao = o.esdf_slice_image(1000.0)
            assert ig.shape == io.shape and np.array_equal(ag, ao) and np.abs(ig - io).max() <= TOL
    nt, _ = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    nc, _ = compare_layer(M, g, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    ne, _ = compare_layer(M, g, o, M.LAYER_ESDF, oracle_mod.L_ESDF, fields_exact=ESDF_FIELDS)
    assert nt > 100 and nc > 20 and ne > 5
    g.update_color_mesh(); o.update_mesh()
    mg = g.mesh()
    nonempty = 0
    for idx in o.block_indices(oracle_mod.L_TSDF):
        mo = o.mesh_block(idx)
        a = mg[tuple(idx)]
        assert a["triangles"].shape == mo["triangles"].shape and np.array_equal(a["triangles"], mo["triangles"]), (case, idx)
        if len(mo["vertices"]):
            nonempty += 1
            assert np.abs(a["vertices"] - mo["vertices"]).max() <= TOL
            assert np.abs(a["normals"] - mo["normals"]).max() <= 1e-3
            assert np.abs(a["colors"].astype(int) - mo["colors"].astype(int)).max() <= 1
    assert nonempty > 20
    assert g.counters()["capacity_overflow"] == 0


def test_non_finite_and_negative_depth_pixels(oracle_mod, hip_lib):
    """A depth image with NaN, +-inf, negative and absurdly large pixels (a driver hiccup): the view and the TSDF of the HIP path equal the
    oracle's (NaN and non-positive pixels are invalid depth; an infinite range is cut at the integration distance), nothing non-finite is
    written into the map, and the next clean frame integrates normally."""
    M, g, o = make_pair(oracle_mod)
    rng = np.random.default_rng(5)
    fr = H.frames(3, H.SMALL_CAM, color=False, stride=9)
    for k, (d, _, T) in enumerate(fr):
        d = d.copy()
        if k < 2:
            idx = rng.integers(0, d.size, 400)
            d.reshape(-1)[idx[:100]] = np.nan; d.reshape(-1)[idx[100:200]] = np.inf; d.reshape(-1)[idx[200:250]] = -np.inf
            d.reshape(-1)[idx[250:330]] = -1.5; d.reshape(-1)[idx[330:]] = 1.0e30
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    n, _ = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 100
    b, _ = g.get_blocks(M.LAYER_TSDF, g.block_indices(M.LAYER_TSDF))
    assert np.isfinite(b["distance"]).all() and np.isfinite(b["weight"]).all()
    g.update_esdf(); o.update_esdf()
    compare_layer(M, g, o, M.LAYER_ESDF, oracle_mod.L_ESDF, fields_exact=ESDF_FIELDS)
    assert g.counters()["capacity_overflow"] == 0
