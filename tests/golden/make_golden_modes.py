#!/usr/bin/env python3
"""Generates tests/golden/tiny_modes.npz: regression pins for the mapping modes beside the benchmark path -- the mask split of
the human mapping types, an occupancy mapper (log-odds layer, its ESDF slice, after one decayOccupancyAllVoxels), and a 3-D ESDF.

Like tiny_sequence.npz these are outputs of OUR oracle (the reference holds no golden vectors for any of this and its core is
not buildable here): they pin the oracle and the HIP path against silent change, they are not reference outputs.  The inputs
(uint16 mm depth, mask, poses) are stored in the fixture.

Run from the repo root:  python tests/golden/make_golden_modes.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CAM = (40.0, 40.0, 39.5, 29.5, 80, 60)
MASK_CAM = (42.0, 42.0, 39.5, 29.5, 80, 60)
OCC = dict(projective_layer_type=1, free_region_occupancy_probability=0.3, occupied_region_occupancy_probability=0.9,
           unobserved_region_occupancy_probability=0.35, occupied_region_half_width_m=0.2, max_integration_distance_m=4.0)
ESDF3 = dict(esdf_mode=1, esdf_max_distance_m=0.6, max_integration_distance_m=3.0)


def t_cm_cd():
    T = np.eye(4, dtype=np.float32); T[0, 3] = 0.05
    return T


def run(split_fn, make_map, depth_mm, masks, poses):
    """split_fn(depth, mask) -> (unmasked, masked); make_map(**params) -> a mapper with the oracle's method names."""
    occ = make_map(**OCC); e3 = make_map(**ESDF3)
    splits = []
    for d, mk, T in zip(depth_mm, masks, poses):
        depth = d.astype(np.float32) * np.float32(1.0 / 1000.0)
        un, ma = split_fn(depth, mk)
        splits.append((np.asarray(un), np.asarray(ma)))
        occ.integrate_depth(np.asarray(ma), T, CAM)          # the "person" feeds the occupancy mapper
        e3.integrate_depth(np.asarray(un), T, CAM)           # the rest feeds a TSDF mapper with a 3-D ESDF
    occ.update_esdf(); occ.decay_occupancy(); occ.update_esdf()
    e3.update_esdf()
    return splits, occ, e3


def summarize(splits, occ, e3, layer_occ, layer_esdf, occ_field):
    oi = occ.block_indices(layer_occ)
    lo = np.stack([np.asarray(occ.get_block(layer_occ, i)[occ_field]) for i in oi])
    oslice, oaabb = occ.esdf_slice_image(1000.0)
    ei = e3.block_indices(layer_esdf)
    eb = [e3.get_block(layer_esdf, i) for i in ei]
    return dict(split_unmasked=np.stack([s[0] for s in splits]), split_masked=np.stack([s[1] for s in splits]),
                occ_indices=oi, occ_log_odds=lo, occ_slice=oslice, occ_aabb=oaabb,
                esdf3_indices=ei, esdf3_sq=np.stack([b["squared_distance_vox"] for b in eb]).astype(np.float32),
                esdf3_parent=np.stack([b["parent_direction"] for b in eb]).astype(np.int8),
                esdf3_flags=np.stack([b["observed"] + 2 * b["is_inside"] + 4 * b["is_site"] for b in eb]).astype(np.uint8))


def main():
    import oracle
    from isaac_ros_nvblox_amd import synthetic as S
    sc = S.Scene()
    depth_mm, masks, poses = [], [], []
    for k in range(3):
        T = S.trajectory_pose(10 * k, 200)
        d, _ = S.render(sc, T, CAM, color=False)
        depth_mm.append(np.round(d * 1000.0).astype(np.uint16))
        mk = np.zeros((60, 80), np.uint8); mk[15:50, 25 + 3 * k:45 + 3 * k] = 255
        masks.append(mk); poses.append(T)
    depth_mm = np.stack(depth_mm); masks = np.stack(masks); poses = np.stack(poses)
    split = lambda d, mk: oracle.split_depth_by_mask(d, mk, t_cm_cd(), CAM, MASK_CAM, 0.25)
    mk_map = lambda **kw: oracle.OracleMap(oracle.default_params(**kw))
    splits, occ, e3 = run(split, mk_map, depth_mm, masks, poses)
    out = summarize(splits, occ, e3, oracle.L_TSDF, oracle.L_ESDF, "distance")
    out.update(depth_mm=depth_mm, masks=masks, poses=poses)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_modes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["occ_indices"]), "occupancy blocks,", len(out["esdf3_indices"]), "3-D ESDF blocks")


if __name__ == "__main__":
    main()
