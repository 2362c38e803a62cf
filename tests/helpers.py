"""Shared helpers for the parity tests: run the same seeded sequence through the CPU oracle and the HIP mapper."""
import numpy as np

from isaac_ros_nvblox_amd import synthetic as S

SMALL_CAM = (80.0, 80.0, 79.5, 59.5, 160, 120)      # 160x120, same 90 deg HFOV as the 640x480 camera


def frames(n, cam=S.REPLICA_LIKE_CAM, start=0, color=True, stride=1, **kw):
    sc = S.Scene()
    out = []
    for i in range(n):
        T = S.trajectory_pose(start + i * stride, 200, **kw)
        d, rgb = S.render(sc, T, cam, color=color)
        out.append((d, rgb, T))
    return out


def idx_set(a):
    return set(map(tuple, np.asarray(a).reshape(-1, 3).tolist()))


def copy_params(src, dst_cls):
    dst = dst_cls()
    for name, _ in dst_cls._fields_:
        setattr(dst, name, getattr(src, name))
    return dst
