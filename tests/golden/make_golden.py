#!/usr/bin/env python3
"""Generates tests/golden/tiny_sequence.npz.

The reference holds no golden vectors for TSDF / colour / ESDF propagation / mesh (its rosbag is a
Git-LFS pointer and its integration tests only check that messages arrive), and its core is not buildable here, so these
vectors are produced by OUR oracle (oracle/nvblox_oracle.c) on a small deterministic input and serve as a regression pin
for both the oracle and the HIP path.  They are NOT outputs of the reference -- tests/test_oracle_kat.py holds the
only reference-derived known answers.  Inputs (uint16 mm depth, rgb, poses) are stored too, so the fixture does not
depend on the synthetic generator staying unchanged.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from isaac_ros_nvblox_amd import synthetic as S  # noqa: E402

CAM = (40.0, 40.0, 39.5, 29.5, 80, 60)


def run(o, depth_mm, rgb, poses):
    for d, c, T in zip(depth_mm, rgb, poses):
        depth = d.astype(np.float32) * np.float32(1.0 / 1000.0)
        o.integrate_depth(depth, T, CAM)
        o.integrate_color(c, T, CAM)
    o.update_esdf()
    o.update_mesh()


def summarize(o, L):
    idx = o.block_indices(L.L_TSDF)
    tsdf = np.stack([o.get_block(L.L_TSDF, i) for i in idx])
    cidx = o.block_indices(L.L_COLOR)
    col = np.stack([o.get_block(L.L_COLOR, i) for i in cidx])
    img, aabb = o.esdf_slice_image(1000.0)
    ntri = np.array([len(o.mesh_block(i)["triangles"]) for i in idx], np.int32)
    nvert = np.array([len(o.mesh_block(i)["vertices"]) for i in idx], np.int32)
    return dict(tsdf_indices=idx, tsdf_distance=tsdf["distance"], tsdf_weight=tsdf["weight"],
                color_indices=cidx, color_rgb=np.stack([col["r"], col["g"], col["b"]], -1), color_weight=col["weight"],
                esdf_slice=img, esdf_aabb=aabb, mesh_ntri=ntri, mesh_nvert=nvert)


def main():
    sc = S.Scene()
    depth_mm, rgb, poses = [], [], []
    for i in (0, 30, 60):
        T = S.trajectory_pose(i)
        d, c = S.render(sc, T, CAM)
        depth_mm.append(np.round(d * 1000.0).astype(np.uint16)); rgb.append(c); poses.append(T)
    o = oracle.OracleMap(oracle.default_params())
    run(o, depth_mm, rgb, poses)
    out = summarize(o, oracle)
    out.update(depth_mm=np.stack(depth_mm), rgb=np.stack(rgb), poses=np.stack(poses), cam=np.array(CAM, np.float32))
    path = os.path.join(ROOT, "tests", "golden", "tiny_sequence.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["tsdf_indices"]), "blocks")


if __name__ == "__main__":
    main()
