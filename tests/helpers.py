"""Shared helpers for the parity tests: run the same seeded sequence through the CPU oracle and the HIP mapper."""
import numpy as np

from isaac_ros_nvblox_amd import synthetic as S

SMALL_CAM = (80.0, 80.0, 79.5, 59.5, 160, 120)      # 160x120, same 90 deg HFOV as the 640x480 camera


def frames(n, cam=S.REPLICA_LIKE_CAM, start=0, color=True, stride=1, **kw):
    sc = S.Scene()

    def one(i):
        T = S.trajectory_pose(start + i * stride, 200, **kw)
        d, rgb = S.render(sc, T, cam, color=color)
        return (d, rgb, T)
    if n >= 16:          # (numpy releases the GIL in the ray casts: long sequences render on a few threads)
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
            return list(pool.map(one, range(n)))
    return [one(i) for i in range(n)]


def idx_set(a):
    return set(map(tuple, np.asarray(a).reshape(-1, 3).tolist()))


def copy_params(src, dst_cls):
    dst = dst_cls()
    for name, _ in dst_cls._fields_:
        setattr(dst, name, getattr(src, name))
    return dst
