#!/usr/bin/env python3
"""The PRODUCT (libnvblox_hip.so on the GPU) against the numpy models under tests/ that share no code with it -- with the CPU checker (oracle/) not
involved at all.  Run on a GPU box; prints one JSON object (committed as profiles/r05_product_vs_independent.json).  Each model is the one
tests/test_independent_checks.py holds the checker against, so this closes the triangle product <-> checker <-> model by its third side.

  python tools/product_vs_independent.py > gpurun_out/product_vs_independent.json
"""
import json, os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import tsdf_independent as TI
import mesh_independent as MI
from isaac_ros_nvblox_amd import mapper as M

cam = H.SMALL_CAM
L = types.SimpleNamespace(L_TSDF=M.LAYER_TSDF, L_COLOR=M.LAYER_COLOR)        # (the models ask `oracle_mod` only for the layer constants)


def tsdf_rule():
    fr = H.frames(2, cam, stride=9, color=False)
    res = {}
    for mode in range(6):
        for variant in (0, 1):
            p = M.default_params(weighting_mode=mode, tsdf_weighting_variant=variant, max_weight=1.7)
            g = M.Mapper(p, device=0); model = {}
            for d, _, T in fr:
                g.integrate_depth(d, T, cam); g.synchronize()
                for idx in g.block_indices(M.LAYER_TSDF):
                    key = tuple(int(v) for v in idx)
                    pd_, pw_ = model.get(key, (np.zeros(512), np.zeros(512)))
                    nd, nw, upd, rob = TI.update_block(pd_, pw_, key, d, T, cam, p)
                    model[key] = (nd, nw); model[("robust", key)] = model.get(("robust", key), np.ones(512, bool)) & rob
            idx = g.block_indices(M.LAYER_TSDF); blocks, found = g.get_blocks(M.LAYER_TSDF, idx)
            assert found.all()
            n = 0; ed_max = ew_max = 0.0
            for i, b in zip(idx, blocks):
                key = tuple(int(v) for v in i); ed, ew = model[key]; rob = model[("robust", key)]
                n += int(rob.sum())
                ed_max = max(ed_max, float(np.abs(b["distance"][rob] - ed[rob]).max(initial=0.0)))
                ew_max = max(ew_max, float((np.abs(b["weight"][rob] - ew[rob]) / max(1.0, float(ew.max()))).max(initial=0.0)))
            res["mode%d_variant%d" % (mode, variant)] = {"voxels": n, "max_abs_distance_error": ed_max, "max_rel_weight_error": ew_max, "ok": bool(n > 100000 and ed_max <= 2e-5 and ew_max <= 2e-5)}
            g.close()
    return res


def mesh_rules():
    p = M.default_params()
    g = M.Mapper(p, device=0); g.set_color_deferral(False)
    for d, rgb, T in H.frames(5, cam, stride=9, color=True):
        g.integrate_depth(d, T, cam); g.integrate_color(rgb, T, cam)
    g.update_color_mesh(full=True); g.synchronize()
    mesh = g.mesh()
    vs = float(p.voxel_size)
    ti, lo, d, w, has, col, cw = MI.dense_layers(g, L)
    nv = nt = 0; bad = []; n_flip = 0; order_equal = True
    for i in ti:
        key = tuple(int(v) for v in i)
        eids, pos, cols, active = MI.expected_block(i, lo, d, w, has, col, cw, vs, float(p.mesh_min_weight))
        mb = mesh.get(key)
        if mb is None or len(mb["vertices"]) == 0:
            if len(eids): bad.append((key, "missing"))
            continue
        v, t, c = mb["vertices"], mb["triangles"], mb["colors"]
        if len(v) != len(eids): bad.append((key, "vertex count %d != %d" % (len(v), len(eids)))); continue
        if np.abs(v - pos).max() > 1e-6:
            order_equal = False                                              # (the product may order differently: compare as sets)
            a = v[np.lexsort(v.T[::-1])]; b = pos[np.lexsort(pos.T[::-1])]
            if np.abs(a - b).max() > 1e-6: bad.append((key, "vertex positions")); continue
        else:
            if not np.array_equal(c[:, :3], cols): bad.append((key, "colours"))
        cube, inside = MI.triangle_cubes(i, v, t, vs)
        if not (inside.all() and cube.min() >= 0 and cube.max() <= 7): bad.append((key, "triangle outside its cube")); continue
        seen = np.zeros((8, 8, 8), bool); seen[tuple(cube.T)] = True
        if not np.array_equal(seen, active): bad.append((key, "meshed cubes"))
        nv += len(v); nt += len(t)
    g.close()
    return {"vertices": nv, "triangles": nt, "same_vertex_order_as_the_model": order_equal, "blocks_differing": len(bad), "first_differences": [str(b) for b in bad[:5]],
            "ok": bool(nv > 5000 and not bad)}


def freespace():
    kw = dict(projective_layer_type=2, min_duration_since_occupied_for_freespace_ms=250, max_unobserved_to_keep_consecutive_occupancy_ms=150,
              min_consecutive_occupancy_duration_for_reset_ms=300, check_neighborhood=1)
    p = M.default_params(**kw)
    g = M.Mapper(p, device=0)
    fr = H.frames(8, cam, stride=3, color=False)
    times = [0, 90, 210, 260, 400, 520, 640, 760]
    NB_ = 40; G = NB_ * 8; off = np.array([NB_ // 2] * 3) * 8
    init = np.zeros((G, G, G), bool); last = np.zeros((G, G, G), np.int64); dur = np.zeros((G, G, G), np.int64); hc = np.zeros((G, G, G), bool)
    thr = np.float32(p.max_tsdf_distance_for_occupancy_m)
    sl = lambda i: tuple(slice(int(a) * 8 + int(c), int(a) * 8 + int(c) + 8) for a, c in zip(i, off))
    n_checked = 0; n_bad = 0; n_reset = 0
    for k, ((d_, _, T), now) in enumerate(zip(fr, times)):
        d_ = d_.copy()
        if k >= 4:
            d_[40:90, 60:110] = np.minimum(d_[40:90, 60:110], 0.9)
        g.set_time_ms(now); g.integrate_depth(d_, T, cam); g.synchronize()
        dist = np.zeros((G, G, G), np.float32); wgt = np.zeros((G, G, G), np.float32)
        ti = g.block_indices(M.LAYER_TSDF); blocks, _ = g.get_blocks(M.LAYER_TSDF, ti)
        for i, b in zip(ti, blocks):
            b = b.reshape(8, 8, 8); dist[sl(i)] = b["distance"]; wgt[sl(i)] = b["weight"]
        view = np.zeros((G, G, G), bool)
        for i in g.last_view():
            view[sl(i)] = True
        occ_self = (wgt > 0) & (dist < thr); occ = occ_self.copy()
        for ax in range(3):
            occ |= np.roll(occ_self, 1, ax) | np.roll(occ_self, -1, ax)
        first = view & ~init
        init |= first; last[first] = now; dur[first] = 0; hc[first] = bool(p.initialize_to_high_confidence_freespace)
        obs = view & (wgt > 0); is_occ = obs & occ; gap = now - last
        dur = np.where(is_occ, np.where(gap <= int(p.max_unobserved_to_keep_consecutive_occupancy_ms), dur + gap, 0), dur)
        last = np.where(is_occ, now, last)
        reset = is_occ & (dur >= int(p.min_consecutive_occupancy_duration_for_reset_ms)); n_reset += int((reset & hc).sum())
        hc = np.where(reset, False, hc)
        hc = np.where(obs & ~is_occ & (now - last >= int(p.min_duration_since_occupied_for_freespace_ms)), True, hc)
        fi = g.block_indices(M.LAYER_FREESPACE); fb, _ = g.get_blocks(M.LAYER_FREESPACE, fi)
        for i, f in zip(fi, fb):
            f = f.reshape(8, 8, 8); s = sl(i)
            same = (np.array_equal(f["is_high_confidence_freespace"].astype(bool), hc[s]) and np.array_equal(f["last_occupied_timestamp_ms"], last[s])
                    and np.array_equal(f["consecutive_occupancy_duration_ms"], dur[s]))
            n_checked += 512; n_bad += 0 if same else 1
    g.close()
    return {"voxel_checks": n_checked, "blocks_differing": n_bad, "high_confidence_voxels_reset": n_reset, "ok": bool(n_checked > 10 ** 6 and n_bad == 0 and n_reset > 100)}


def main():
    assert "oracle" not in sys.modules
    out = {}
    for name, fn in (("tsdf_update_rule", tsdf_rule), ("mesh_table_free_rules", mesh_rules), ("freespace_state_machine", freespace)):
        try:
            out[name] = fn()
        except Exception as e:      # (evidence script: report, do not hide)
            import traceback
            out[name] = {"ok": False, "error": repr(e), "trace": traceback.format_exc()[-1500:]}
    out["oracle_imported"] = "oracle" in sys.modules
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
