"""Independent float64 model of the projective TSDF update (UpdateTsdfVoxelFunctor as DESIGN.md 3 / SEMANTICS.md restate it): numpy only, whole arrays,
no code shared with the product or with oracle/nvblox_oracle.c.  Per voxel centre of a block: layer -> camera frame, pinhole projection, bilinear
depth with validity (all four taps > 0), sdf = measured depth - voxel depth, the weighting function (six modes, two formula sets), weighted blend,
clamps.  Returns the expected {distance, weight} and a per-voxel "robust" mask: voxels whose DECISIONS (in the image, taps valid, sdf >= -trunc,
clamps) do not hang on the last bits of a float32 evaluation."""
import numpy as np


def weight_fn(mode, variant, ds, vd, trunc, vs, max_dist):
    w = np.ones_like(ds)
    if mode in (2, 3, 4):
        w = 1.0 / (ds * ds)
    elif mode == 5:
        w = np.minimum(1.0, 1.0 / ds) if variant == 0 else np.maximum(0.01, 1.0 - ds / max_dist)
    sdf = ds - vd
    if mode in (1, 3):
        if variant == 0:
            g = np.clip((trunc + sdf) / trunc, 0.0, None)
            w = np.where(sdf < 0.0, w * g, w)
        else:
            g = np.clip((trunc + sdf) / (trunc - vs), 0.0, None) if trunc > vs else np.zeros_like(sdf)
            w = np.where(sdf < -vs, w * g, w)
    elif mode == 4:
        if variant == 0:
            w = np.where(sdf > trunc, w * (trunc / np.where(sdf > trunc, sdf, 1.0)), w)
        else:
            g = np.clip((trunc + sdf) / trunc, 0.0, None)
            w = np.where(sdf < 0.0, w * g * g, w)
    return w


def update_block(prev_d, prev_w, block_index, depth, T_L_C, cam, p, margin=2e-4):
    """prev_d / prev_w: [512] float64 in voxel order z + 8 y + 64 x; depth [rows, cols]; T_L_C 4x4; cam (fu, fv, cu, cv, w, h); p: the parameter struct."""
    fu, fv, cu, cv, w, h = cam
    vs = float(p.voxel_size); bs = 8.0 * vs; trunc = float(p.truncation_distance_vox) * vs
    max_dist = float(p.max_integration_distance_m); max_w = float(p.max_weight)
    rows, cols = depth.shape
    lin = np.arange(512); vx, vy, vz = lin // 64, (lin // 8) % 8, lin % 8
    pl = np.stack([block_index[0] * bs + vx * vs + vs / 2, block_index[1] * bs + vy * vs + vs / 2, block_index[2] * bs + vz * vs + vs / 2], 1)
    T = np.asarray(T_L_C, np.float64); R = T[:3, :3]; t = T[:3, 3]
    pc = (pl - t) @ R                                   # R^T (p - t)
    z = pc[:, 2]
    zs = np.where(z > 0, z, 1.0)
    u = fu * pc[:, 0] / zs + cu; v = fv * pc[:, 1] / zs + cv
    in_img = (z > 0) & (u >= 0) & (v >= 0) & (u <= w) & (v <= h)
    in_rng = ~(max_dist > 0) | (z <= max_dist)
    uc, vc = u - 0.5, v - 0.5
    x0 = np.floor(uc).astype(np.int64); y0 = np.floor(vc).astype(np.int64)
    in_taps = (x0 >= 0) & (y0 >= 0) & (x0 + 1 <= cols - 1) & (y0 + 1 <= rows - 1)
    xs = np.clip(x0, 0, cols - 2); ys = np.clip(y0, 0, rows - 2)
    d = depth.astype(np.float64)
    f00, f10, f01, f11 = d[ys, xs], d[ys, xs + 1], d[ys + 1, xs], d[ys + 1, xs + 1]
    taps_ok = (f00 > 0) & (f10 > 0) & (f01 > 0) & (f11 > 0)
    ax, ay = uc - np.floor(uc), vc - np.floor(vc)
    ds = (1 - ay) * ((1 - ax) * f00 + ax * f10) + ay * ((1 - ax) * f01 + ax * f11)
    sdf = ds - z
    upd = in_img & in_rng & in_taps & taps_ok & (sdf >= -trunc)
    ds_s = np.where(upd, ds, 1.0)
    wm = weight_fn(int(p.weighting_mode), int(p.tsdf_weighting_variant), ds_s, np.where(upd, z, 0.0), trunc, vs, max_dist)
    wsum = wm + prev_w
    upd = upd & (wsum > 0)
    fused = np.clip((np.where(upd, sdf, 0.0) * wm + prev_d * prev_w) / np.where(wsum > 0, wsum, 1.0), -trunc, trunc)
    out_d = np.where(upd, fused, prev_d); out_w = np.where(upd, np.minimum(wsum, max_w), prev_w)
    # decisions that hang on the last bits: close to the image border / a tap boundary / the range limit / sdf == -trunc / depth edges (taps that differ much make ds sensitive)
    spread = np.maximum.reduce([f00, f10, f01, f11]) - np.minimum.reduce([f00, f10, f01, f11])
    robust = ((np.abs(z) > 1e-3) & (np.minimum.reduce([np.abs(u), np.abs(v), np.abs(u - w), np.abs(v - h)]) > margin * 50) &
              (np.minimum(np.abs(uc - np.round(uc)), np.abs(vc - np.round(vc))) > margin * 50) & (np.abs(z - max_dist) > margin) &
              (np.abs(sdf + trunc) > margin) & (spread < 0.5))
    return out_d, out_w, upd, robust


def project_and_sample(block_index, depth, T_L_C, cam, vs, max_dist):
    """shared geometry of the update rules below: voxel depth z, measured depth ds, masks (in view & range, four valid taps, some tap invalid), robustness inputs"""
    fu, fv, cu, cv, w, h = cam
    bs = 8.0 * vs
    rows, cols = depth.shape
    lin = np.arange(512); vx, vy, vz = lin // 64, (lin // 8) % 8, lin % 8
    pl = np.stack([block_index[0] * bs + vx * vs + vs / 2, block_index[1] * bs + vy * vs + vs / 2, block_index[2] * bs + vz * vs + vs / 2], 1)
    T = np.asarray(T_L_C, np.float64); pc = (pl - T[:3, 3]) @ T[:3, :3]
    z = pc[:, 2]; zs = np.where(z > 0, z, 1.0)
    u = fu * pc[:, 0] / zs + cu; v = fv * pc[:, 1] / zs + cv
    seen = (z > 0) & (u >= 0) & (v >= 0) & (u <= w) & (v <= h) & (~(max_dist > 0) | (z <= max_dist))
    uc, vc = u - 0.5, v - 0.5
    x0 = np.floor(uc).astype(np.int64); y0 = np.floor(vc).astype(np.int64)
    in_taps = (x0 >= 0) & (y0 >= 0) & (x0 + 1 <= cols - 1) & (y0 + 1 <= rows - 1)
    xs = np.clip(x0, 0, cols - 2); ys = np.clip(y0, 0, rows - 2)
    d = depth.astype(np.float64)
    f00, f10, f01, f11 = d[ys, xs], d[ys, xs + 1], d[ys + 1, xs], d[ys + 1, xs + 1]
    valid = (f00 > 0) & (f10 > 0) & (f01 > 0) & (f11 > 0)
    ax, ay = uc - np.floor(uc), vc - np.floor(vc)
    ds = (1 - ay) * ((1 - ax) * f00 + ax * f10) + ay * ((1 - ax) * f01 + ax * f11)
    spread = np.maximum.reduce([f00, f10, f01, f11]) - np.minimum.reduce([f00, f10, f01, f11])
    near_edge = ((np.abs(z) < 1e-3) | (np.minimum.reduce([np.abs(u), np.abs(v), np.abs(u - w), np.abs(v - h)]) < 0.01) |
                 (np.minimum(np.abs(uc - np.round(uc)), np.abs(vc - np.round(vc))) < 0.01) | (np.abs(z - max_dist) < 2e-4))
    return z, ds, seen & in_taps & valid, seen & in_taps & ~valid, spread, near_edge


def update_block_occupancy(prev_lo, block_index, depth, T_L_C, cam, p):
    """ProjectiveOccupancyIntegrator as restated: a voxel in front of the measured surface by more than the half width adds logit(free), within +- the half
    width logit(occupied), behind it logit(unobserved); clamp +-10.  -> (log-odds [512], robust mask)"""
    lo = lambda q: np.log(np.float32(q) / (np.float32(1.0) - np.float32(q)))
    hw = float(p.occupied_region_half_width_m)
    z, ds, got, _, spread, near_edge = project_and_sample(block_index, depth, T_L_C, cam, float(p.voxel_size), float(p.max_integration_distance_m))
    upd = np.where(z < ds - hw, lo(p.free_region_occupancy_probability), np.where(z <= ds + hw, lo(p.occupied_region_occupancy_probability), lo(p.unobserved_region_occupancy_probability)))
    out = np.where(got, np.clip(prev_lo + upd, -10.0, 10.0), prev_lo)
    robust = ~near_edge & (np.abs(z - (ds - hw)) > 2e-4) & (np.abs(z - (ds + hw)) > 2e-4) & (spread < 0.5)
    return out, robust


def update_block_invalid_decay(prev_d, prev_w, block_index, depth, T_L_C, cam, p):
    """[U] invalid-depth decay: a voxel that projects into the image onto an invalid (<= 0) depth tap has its weight multiplied by the factor (and is not
    integrated); everything else as update_block"""
    nd, nw, upd, rob = update_block(prev_d, prev_w, block_index, depth, T_L_C, cam, p)
    z, ds, got, bad, spread, near_edge = project_and_sample(block_index, depth, T_L_C, cam, float(p.voxel_size), float(p.max_integration_distance_m))
    nw = np.where(bad, prev_w * float(p.invalid_depth_decay_factor), nw); nd = np.where(bad, prev_d, nd)
    return nd, nw, rob & ~near_edge
