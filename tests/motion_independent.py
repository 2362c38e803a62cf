"""An INDEPENDENT restatement of the LiDAR motion compensation ([U] MultiMapper::integrateDepth(pointcloud, ..., use_lidar_motion_compensation,
T_L_S_scanEnd, scan_duration_ms), nvblox_node.cpp:1339-1384) in float64 with numpy's linear algebra and scipy's Rotation -- nothing here is shared
with the product or with the checker's C code (both take the per-point arithmetic from csrc/nvbx_motion_math.h for bit parity).  Written from the
model's description only: the sensor moves from T_L_S(start) to T_L_S(end) during the scan; a point measured at fraction a of the scan duration
(clamped to [0, 1]) was seen from the pose interpolated at a -- translation linearly, rotation by NORMALISED LINEAR interpolation of the unit
quaternions of identity and of the relative rotation (shortest arc) -- and is re-expressed in the sensor frame at scan start."""
import numpy as np
from scipy.spatial.transform import Rotation


def compensate(points, rel_time_ms, T_L_S_start, T_L_S_end, scan_duration_ms):
    p = np.asarray(points, np.float64)
    a = np.clip(np.asarray(rel_time_ms, np.float64) / float(scan_duration_ms), 0.0, 1.0)
    rel = np.linalg.inv(np.asarray(T_L_S_start, np.float64)) @ np.asarray(T_L_S_end, np.float64)      # sensor(end) in sensor(start)
    q = Rotation.from_matrix(rel[:3, :3]).as_quat()                      # scipy order: x, y, z, w
    if q[3] < 0:
        q = -q
    ident = np.array([0.0, 0.0, 0.0, 1.0])
    qa = (1.0 - a)[:, None] * ident[None, :] + a[:, None] * q[None, :]
    qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    return Rotation.from_quat(qa).apply(p) + a[:, None] * rel[:3, 3][None, :]
