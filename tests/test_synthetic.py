import numpy as np

from isaac_ros_nvblox_amd import synthetic as S
import helpers as H


def test_render_is_deterministic_and_metric():
    sc = S.Scene()
    T = S.trajectory_pose(3)
    d1, c1 = S.render(sc, T, H.SMALL_CAM); d2, c2 = S.render(sc, T, H.SMALL_CAM)
    assert np.array_equal(d1, d2) and np.array_equal(c1, c2)
    assert d1.dtype == np.float32 and c1.dtype == np.uint8 and c1.shape == (120, 160, 3)
    assert (d1 > 0).all() and d1.max() < 8.0            # closed room: every ray hits something
    assert set(np.unique(c1).tolist()) <= {64, 192}


def test_pose_is_rigid_and_looks_outward():
    for i in (0, 17, 150):
        T = S.trajectory_pose(i).astype(np.float64)
        R = T[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1.0) < 1e-6
        pos = T[:3, 3]; fwd = R[:, 2]
        assert np.dot(fwd[:2], pos[:2]) > 0 and abs(pos[2] - 1.5) < 1e-6


def test_depth_is_z_depth_of_the_analytic_hit():
    sc = S.Scene()
    T = S.look_pose(np.array([0.0, 0.0, 1.5]), 0.0, 0.0)    # looking along +x at the wall x = 3
    d, _ = S.render(sc, T, H.SMALL_CAM, color=False)
    assert abs(d[60, 80] - 3.0) < 1e-3
