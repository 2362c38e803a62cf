"""Debug helper: replay the randomised API sequence of tests/test_gpu_sequences.py for one seed and report the first operation after which
the HIP path and the oracle diverge (block sets after every op, ESDF slice after every updateEsdf).  Usage: python tools/seq_diverge.py SEED"""
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, helpers as H, oracle as O
from isaac_ros_nvblox_amd import synthetic as S, mapper as M
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(100 + seed)
kw = dict(tsdf_decay_factor=0.7, tsdf_decayed_weight_threshold=0.2, invalid_depth_decay_factor=(0.8 if seed % 3 == 2 else -1.0), weighting_mode=(4 if seed % 3 == 1 else 0))
pg = M.default_params(**kw); po = H.copy_params(pg, O.OrcParams)
g = M.Mapper(pg, block_capacity=1 << 14); o = O.OracleMap(po)
sc = S.Scene(); CAM = H.SMALL_CAM; last_T = None; hist = []
for step in range(36):
    op = rng.choice(["depth", "depth", "depth", "color", "color", "esdf", "esdf", "mesh", "decay", "radius", "shapes"])
    if step < 3: op = "depth"
    desc = op
    if op in ("depth", "color"):
        i = int(rng.integers(0, 200))
        T = S.trajectory_pose(i, 200, radius=float(rng.uniform(0.3, 1.6)), height=float(rng.uniform(0.8, 2.0)), pitch_deg=float(rng.uniform(-35.0, 15.0)), yaw_offset_deg=float(rng.uniform(-60.0, 60.0)))
        d, rgb = S.render(sc, T, CAM, max_range=(6.0 if rng.random() < 0.3 else None))
        if op == "depth": g.integrate_depth(d, T, CAM); o.integrate_depth(d, T, CAM); last_T = T
        else: g.integrate_color(rgb, T, CAM); o.integrate_color(rgb, T, CAM)
    elif op == "esdf": g.update_esdf(); o.update_esdf()
    elif op == "mesh":
        full = bool(rng.random() < 0.3); g.update_color_mesh(full=full); o.update_mesh(full=full)
    elif op == "decay":
        ex = bool(rng.random() < 0.5); g.decay_tsdf(ex); o.decay_tsdf(ex); desc += str(ex)
    elif op == "radius":
        c = (float(last_T[0, 3]), float(last_T[1, 3]), float(last_T[2, 3])); r = float(rng.uniform(2.0, 4.0))
        g.clear_outside_radius(c, r); o.clear_outside_radius(c, r); desc += " %.2f" % r
    elif op == "shapes":
        ctr = tuple(float(v) for v in rng.uniform([-2.5, -2.0, 0.2], [2.5, 2.0, 2.0])); lo = tuple(float(v) for v in rng.uniform([-3.0, -2.5, 0.0], [2.0, 1.5, 1.0]))
        shapes = [("sphere", ctr, float(rng.uniform(0.3, 0.9))), ("aabb", lo, tuple(v + float(rng.uniform(0.3, 1.2)) for v in lo))]
        g.clear_tsdf_inside_shapes(shapes); o.clear_tsdf_inside_shapes(shapes)
    hist.append(desc)
    tg, to = H.idx_set(g.block_indices(M.LAYER_TSDF)), H.idx_set(o.block_indices(O.L_TSDF))
    eg, eo = H.idx_set(g.block_indices(M.LAYER_ESDF)), H.idx_set(o.block_indices(O.L_ESDF))
    print(step, desc, "tsdf", len(tg), len(to), "diff", len(tg ^ to), "esdf", len(eg), len(eo), "only_g", sorted(eg - eo)[:4], "only_o", sorted(eo - eg)[:4])
    if not (eg ^ eo) and len(eo) and op in ("esdf",):
        sg, ag = g.esdf_slice_image(); so, ao = o.esdf_slice_image()
        if sg.shape != so.shape or np.abs(sg - so).max() > 1e-4:
            bad = np.argwhere(np.abs(sg - so) > 1e-4)
            print("   SLICE differs at", len(bad), "pixels; first", bad[:5].tolist(), "gpu", [float(sg[tuple(b)]) for b in bad[:5]], "oracle", [float(so[tuple(b)]) for b in bad[:5]], "aabb", ag)
            by0 = int(np.floor(ag[1] / 0.4 + 0.5)); bx0 = int(np.floor(ag[0] / 0.4 + 0.5))
            blks = sorted(set((int(b[1]) // 8 + bx0, int(b[0]) // 8 + by0) for b in bad))
            print("   blocks (x,y):", blks[:10])
            for (x, y) in blks[:3]:
                print("    col", (x, y), "oracle tsdf", sorted(t for t in to if t[0] == x and t[1] == y), "gpu tsdf", sorted(t for t in tg if t[0] == x and t[1] == y))
                eb_g = g.get_block(M.LAYER_ESDF, (x, y, 0)); eb_o = o.get_block(O.L_ESDF, np.array((x, y, 0), np.int32))
                for f in ("observed", "is_site", "is_inside"):
                    print("     ", f, "gpu", int(eb_g[f].reshape(8, 8, 8)[:, :, 1].sum()), "oracle", int(eb_o[f].reshape(8, 8, 8)[:, :, 1].sum()))
            print("   history:", hist)
            break
    if (eg ^ eo):
        miss = sorted(eo - eg)[:3] + sorted(eg - eo)[:3]
        for (x, y, z) in miss:
            col = [t for t in to if t[0] == x and t[1] == y]
            print("   column", (x, y), "oracle tsdf blocks", sorted(col), "gpu tsdf", sorted(t for t in tg if t[0] == x and t[1] == y))
        break
