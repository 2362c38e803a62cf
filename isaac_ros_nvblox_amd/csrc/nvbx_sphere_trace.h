// nvbx_sphere_trace.h -- the sphere-tracing worker of MultiMapper::integrateColor (synthetic depth image for the occlusion test), shared by
// k_sphere_trace (color.hip) and by k_mark_view (tsdf.hip), which runs it in extra workgroups of the NEXT depth frame's view-marking launch
// when the colour frame was held back (nvbx_mapper_set_color_deferral): the two are independent -- sphere tracing reads the
// TSDF and the insert-only hash, view marking inserts hash entries whose pool slots are all-zero (= unobserved, exactly like a missing block).
#pragma once
#include "nvbx_mapper.h"

namespace nvbx {

// TSDF reads below skip the layer-flag load: the TSDF pool of a slot that does not carry F_TSDF is all-zero (freed /
// ESDF-only slots are zeroed, maintenance.hip), and weight 0 reads as "unobserved" exactly like a missing block.
// RAY_LANES (template parameter RL below) = lanes cooperating on one ray = samples fetched per round trip.  One camera: 8 (4: 13.7, 8: 13.4,
// 16: 14.5, 32: 21.4 us -- 19 200 rays x 8 lanes = 2 400 wavefronts, about one resident round of the chip).  A BATCH of n cameras has n
// times the rays: the latency trick that fills an idle chip for one camera turns into n occupancy rounds of mostly speculative samples, so
// the batch launches with fewer lanes per ray (sphere_trace_lanes() below; tools: NVBX_ST_LANES).

// [U] SphereTracer::cast restated, sample-parallel.  The serial march t <- t + tsdf(t) (nearest voxel) is a chain of
// dependent HBM round trips (hash entry, then voxel) plus ~150 ALU ops per step, and a ray takes 10-20 steps.  But the
// step is PREDICTABLE: exactly `trunc` through free (clamped) or unobserved space, and the same small value while the
// ray stays inside one voxel near the surface.  So RAY_LANES lanes serve one ray: lane j fetches the sample at
// t + j*ps (ps = predicted step, accumulated with the same float additions the serial march performs), all their hash
// probes and voxel loads are in flight together, and the group consumes the samples in order with ballots while each
// sample's step equals the prediction.  The first sample that breaks it supplies the next t and the next prediction.
// The sequence of t values -- and so the result -- is bit-identical to the one-sample-at-a-time march.
// The colour frames of one launch set: one frame, or a batch of up to MAX_BATCH (nvbx_integrate_color_batch); kernel arguments.
template <int NB> struct PoseSet { FrameCore f[NB]; int32_t n; };

// worker `wgi` (a 256-thread workgroup) of NSH * ceil(patches / NSH) * n workers; every thread of the workgroup calls
template <int NB, int RAY_LANES>
__device__ inline void sphere_trace_worker(const DMap& m, const PoseSet<NB>& poses, float* synth_all, int32_t srows, int32_t scols, int32_t max_steps,
                                           float max_len, float eps_m, int wgi) {
  const int tid = threadIdx.x;
  NVBX_INV_TSDF_READER(m);
  if (wgi == 0 && tid == 0) list_reset(m, S_LIST_COLOR);
  const int lane = tid & 63;
  const int sub = lane & (RAY_LANES - 1);              // sample index within the ray's group
  const int gsh = lane & ~(RAY_LANES - 1);             // first lane of the group (= shift of its bits in a ballot)
  // XCD-aware ray -> workgroup mapping.  Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2: with rays
  // dealt out in row-major order every XCD marches through EVERY part of the frustum and fetches its own copy of every TSDF block
  // and hash line (PMC: 6.6 MB of HBM traffic for 1.0 MB of blocks).  Instead a workgroup takes an 8 x 4 patch of rays, and the
  // patches are numbered so that the workgroups of one XCD (blockIdx.x & 7) own a contiguous band of patch rows.
  constexpr int PW = 8, PH = (256 / RAY_LANES) / PW;       // 256 / RAY_LANES rays per 256-thread workgroup (32 at 8 lanes per ray)
  const int patches_x = (scols + PW - 1) / PW, patches_y = (srows + PH - 1) / PH;
  const int n_patch = patches_x * patches_y, per_xcd = (n_patch + NSH - 1) / NSH;
  const int cam = NB > 1 ? wgi / (NSH * per_xcd) : 0;          // batch: NSH * per_xcd workgroups per camera, camera after camera
  const int wg = wgi - cam * (NSH * per_xcd);
  const FrameCore& f = poses.f[cam];
  float* synth = synth_all + (size_t)cam * srows * scols;
  const int patch = (wg & (NSH - 1)) * per_xcd + (wg >> 3);
  const int pr = tid / RAY_LANES;                            // ray within the patch
  const int py = patch / patches_x, px = patch - py * patches_x;
  const int r = py * PH + pr / PW, c = px * PW + pr % PW;
  const bool valid = cam < poses.n && patch < n_patch && (wg >> 3) < per_xcd && r < srows && c < scols;
  const float rx = (((float)((valid ? c : 0) * f.subsample) + 0.5f) - f.cu) / f.fu;
  const float ry = (((float)((valid ? r : 0) * f.subsample) + 0.5f) - f.cv) / f.fv;
  const float n = NVBX_SQRT((rx * rx + ry * ry) + 1.0f);          // (NVBX_SQRT / NVBX_DIV: the IEEE results, shorter sequences -- nvbx_arith.h)
  const float dcx = NVBX_DIV(rx, n), dcy = NVBX_DIV(ry, n), dcz = NVBX_DIV(1.0f, n);
  float dl[3];
  rotate(f.R_LC, dcx, dcy, dcz, dl);
  // group-uniform march state (replicated in the group's lanes)
  bool last_positive = false, hit = false, done = !valid;
  float t = 0.0f, ps = f.trunc;
  int i = 0;
  int n_rounds = 0;
  while (__ballot(!done)) {                              // wave-uniform loop: ballots / shuffles below need all lanes
    n_rounds++;
    // this lane's sample: t advanced `sub` times by the predicted step (the serial march's additions, replayed)
    float tc = t;
    for (int j = 0; j < RAY_LANES - 1; j++) if (j < sub) tc = tc + ps;
    const float px = f.t_LC[0] + tc * dl[0], py = f.t_LC[1] + tc * dl[1], pz = f.t_LC[2] + tc * dl[2];
    const int32_t gx = (int32_t)floorf(NVBX_DIV(px, f.voxel_size)), gy = (int32_t)floorf(NVBX_DIV(py, f.voxel_size)), gz = (int32_t)floorf(NVBX_DIV(pz, f.voxel_size));
    const int32_t bx = gx >> 3, by = gy >> 3, bz = gz >> 3;
    const uint32_t h = done ? 0u : table_pos(m, bx, by, bz);
    const uint4 e = *reinterpret_cast<const uint4*>(&m.table[h]);
    const uint32_t slot = done ? SLOT_NONE : resolve_any(m, pack_key(bx, by, bz), h, e);
    const float2 v = m.tsdf[slot_ok(slot) ? (size_t)slot * 512 + (gz & 7) + 8 * (gy & 7) + 64 * (gx & 7) : 0];
    // classify the sample as the serial loop body would, assuming every earlier sample of the round kept the prediction
    const bool in_bounds = (i + sub < max_steps) && (tc < max_len);
    const bool observed = slot_ok(slot) && (v.y > 1e-4f);
    const bool surf = observed && (v.x < eps_m);                     // hit test
    const bool keep = observed && !surf && (v.x == ps);              // observed, step == prediction
    const u64 obs_mask = __ballot(observed && !surf);                 // samples that set last_positive
    const uint32_t before = (uint32_t)((obs_mask >> gsh) & ((1u << sub) - 1u));
    const bool pos_before = last_positive || before != 0;            // last_positive when the serial loop reaches this sample
    const bool unobs_keep = !observed && !pos_before && (ps == f.trunc);   // unobserved: step = trunc, if that is the prediction
    const bool event = !done && !(in_bounds && (keep || unobs_keep));
    const uint32_t ev = (uint32_t)((__ballot(event) >> gsh) & (uint32_t)((1ull << RAY_LANES) - 1ull));
    const int e_sub = ev ? (__ffs((int)ev) - 1) : RAY_LANES;         // first sample that breaks the prediction
    const int src = gsh + (e_sub < RAY_LANES ? e_sub : RAY_LANES - 1);
    // values at the event sample (or at the last sample if the whole round kept the prediction)
    const float e_tc = __shfl(tc, src);
    const float e_vx = __shfl(v.x, src);
    const int e_inb = __shfl((int)in_bounds, src), e_obs = __shfl((int)observed, src), e_surf = __shfl((int)surf, src);
    const int e_posb = __shfl((int)pos_before, src);
    const int pos_last = __shfl((int)(pos_before || (observed && !surf)), gsh + RAY_LANES - 1);   // last_positive after a fully kept round
    if (!done) {
      if (e_sub == RAY_LANES) {                      // all samples consumed with the predicted step
        t = e_tc + ps; i += RAY_LANES; last_positive = pos_last != 0;
      } else {
        i += e_sub;                                  // samples before the event were regular steps
        last_positive = e_posb != 0;
        if (!e_inb) { done = true; }
        else if (!e_obs) {                           // unobserved / missing
          if (!last_positive) { t = e_tc + f.trunc; i += 1; ps = f.trunc; }   // (prediction was not trunc)
          else done = true;
        } else if (e_surf) {
          if (last_positive) { t = e_tc + e_vx; hit = true; }
          done = true;
        } else {                                     // observed, step differs from the prediction
          t = e_tc + e_vx; i += 1; last_positive = true; ps = e_vx;
        }
      }
    }
  }
  if (valid && sub == 0) synth[(int64_t)r * scols + c] = hit ? t * dcz : 0.0f;
  NVBX_TV(0, 6, n_rounds); (void)n_rounds;
}


// The sphere tracing of a held-back colour frame riding in the next depth frame's k_mark_view launch (kernel argument; n_wg = 0: none).
// n_tile_wg > 0: the launch's first n_tile_wg workgroups are the view-marking tiles and the riders (EDT, then sphere tracing) follow; 0: riders first.
// n_scan_wg > 0: behind the sphere-tracing workers, n_scan_wg workgroups discover the colour frame's candidate blocks (color_scan_worker,
// nvbx_color_worker.h) into cand[], count in counters[cand_cnt_idx]; they also zero counters[cand_reset_idx] (the other parity's count).
template <int NB>
struct TraceRiderT { PoseSet<NB> ps; float* synth; int32_t srows, scols, max_steps; float max_len, eps_m; int32_t n_wg; int32_t n_tile_wg; int32_t lanes;
                     int32_t n_scan_wg; int4* cand; int32_t cand_cnt_idx, cand_reset_idx;
                     int32_t fence_report;     // written to host_mirror[4] as the launch's first action: colour-reading launches enqueued before it (frames.hip)
                     int32_t n_mark_wg; };     // n_mark_wg > 0: behind those, the ESDF site marking of the held-back update (k_mark_view's EsdfArgs)
using TraceRider = TraceRiderT<1>;

}  // namespace nvbx
