// nvbx_esdf_edt.h -- the 2-D Euclidean distance transform worker (second half of MultiMapper::updateEsdf), shared by k_esdf_edt
// (esdf.hip) and by k_mark_view (tsdf.hip), which runs a pending EDT in extra workgroups of the next depth frame's first
// launch: the EDT touches only the ESDF layer and the (insert-only) hash, so it overlaps the view marking of the next frame.
#pragma once
#include "nvbx_mapper.h"

namespace nvbx {

constexpr int8_t DX_NONE = 127;
constexpr int EDT_MAX_RB = 8;                       // ri <= 63 voxels -> <= 8 blocks each side
constexpr int EDT_MAX_NN = 2 * EDT_MAX_RB + 1;      // 17 x 17 neighbourhood
constexpr int EDT_MAX_ROWS = 8 + 2 * 63;            // 134 rows of the local strip
constexpr int EDT_ROW_WORDS = 5;                    // zero pad word + 3 data words (<= 136 bits) + zero pad word

// Four wavefronts per ESDF block.  Dependent-access chain: {window record} -> {hash entries of the (2rb+1)^2
// neighbourhood, own block included} -> {site masks, own layer flag, own voxel flags} -> LDS phases -> store.
// LDS carve-up of one EDT workgroup (256 or 512 threads): 10800 bytes
struct EdtShared {
  u64 s_bits[EDT_MAX_NN * EDT_MAX_NN];
  u64 s_rows[EDT_MAX_ROWS * EDT_ROW_WORDS];
  int32_t s_part[8 * 64];          // (one row per wavefront: 4 in a 256-thread workgroup, 8 in a 512-thread one)
  uint32_t s_own[2];               // own slot, own layer flags
  u64 s_masks[2];                  // own observed / inside masks
  int8_t s_dx[EDT_MAX_ROWS * 8];
};
// worker `wg` of `nwg` NT-thread workgroups (NT = 256: k_esdf_edt and the EDT workgroups riding in k_mark_view; NT = 512: those riding in the
// fused colour + TSDF launch, tsdf.hip): NT / 64 wavefronts per ESDF block
template <int NT = 256>
__device__ inline void esdf_edt_worker(const DMap& m, const EsdfArgs& a, int wg, int nwg, EdtShared* sh_) {
  u64* s_bits = sh_->s_bits; u64* s_rows = sh_->s_rows; int8_t* s_dx = sh_->s_dx; int32_t* s_part = sh_->s_part; uint32_t* s_own = sh_->s_own; u64* s_masks = sh_->s_masks;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 7, vy = lane >> 3;
  // sweep window = dirty AABB of this update (min / max over the shard copies) + R (in blocks), decided on the device
  const int srec = S_ESDF_REC + (int)(a.epoch & 1), srec_next = S_ESDF_REC + (int)((a.epoch + 1) & 1);
  int32_t x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
#pragma unroll
  for (int s = 0; s < NSH; s++) {
    x0 = min(x0, *shc_at(m, srec, s, 0)); y0 = min(y0, *shc_at(m, srec, s, 1));
    x1 = max(x1, *shc_at(m, srec, s, 2)); y1 = max(y1, *shc_at(m, srec, s, 3));
  }
  const bool ok = x0 <= x1 && y0 <= y1;
  const int32_t wx0 = x0 - a.rb, wy0 = y0 - a.rb, ww = x1 - x0 + 1 + 2 * a.rb, wh = y1 - y0 + 1 + 2 * a.rb;
  if (wg == 0 && tid < NSH) {                                     // next update's record, one shard per thread
    *shc_at(m, srec_next, tid, 0) = INT32_MAX; *shc_at(m, srec_next, tid, 1) = INT32_MAX;
    *shc_at(m, srec_next, tid, 2) = INT32_MIN; *shc_at(m, srec_next, tid, 3) = INT32_MIN;
    *shc_at(m, srec_next, tid, 4) = 0; *shc_at(m, srec_next, tid, 5) = 0;
    if (!a.keep_list) *shc_at(m, S_LIST_ESDF_DIRTY, tid, 0) = 0;   // dirty list consumed by k_esdf_mark (pipelined order: emptied by the marking pass itself)
    if (tid == 0) { m.counters[a.rec_next + 6] = 0; if (ok) m.counters[a.rec + 6] = ww * 8 * wh * 8; }
  }
  if (!ok) return;
  const int nn = 2 * a.rb + 1;                // neighbourhood side in blocks
  const int ctr = a.rb * nn + a.rb;           // own block's position in the neighbourhood
  const int rows = 8 + 2 * a.ri;              // local strip: rows Y0 - ri .. Y0 + 7 + ri
  const int yoff = 8 * a.rb - a.ri;           // strip row 0 in neighbourhood voxel rows
  const int32_t ncell = ww * wh;
  for (int32_t c = wg; c < ncell; c += nwg) {
    const int32_t cy = c / ww, cx = c - cy * ww;
    const int32_t bx = wx0 + cx, by = wy0 + cy;
    __syncthreads();                     // previous block's LDS reads are done
    // 1. site masks of the nn x nn surrounding blocks (zero where there is no block: site_bits of non-ESDF slots is 0)
    for (int q = tid; q < nn * nn; q += NT) {
      const int qy = q / nn, qx = q - qy * nn;
      const uint32_t s = any_slot(m, bx + qx - a.rb, by + qy - a.rb, a.bz_out);
      s_bits[q] = slot_ok(s) ? m.site_bits[s] : 0ull;
      if (q == ctr) {
        s_own[0] = s; s_own[1] = slot_ok(s) ? m.slot_flags[s] : 0u;
        s_masks[0] = slot_ok(s) ? m.obs_bits[s] : 0ull; s_masks[1] = slot_ok(s) ? m.inside_bits[s] : 0ull;
      }
    }
    __syncthreads();
    const uint32_t es = s_own[0];
    if (!slot_ok(es) || !(s_own[1] & (F_ESDF | F_ESDF_PENDING))) continue;   // uniform: no ESDF block in this window cell
    if ((s_own[1] & F_ESDF_REMARK) && tid == 0) atomicAnd(&m.slot_flags[es], ~F_ESDF_REMARK);     // resolved by this update
    if ((s_own[1] & F_ESDF_PENDING) && tid == 0) {                // the block joins the ESDF layer with this update
      atomicOr(&m.slot_flags[es], F_ESDF); atomicAnd(&m.slot_flags[es], ~F_ESDF_PENDING);
      atomicMin(&m.counters[C_ESDF_AABB + 0], bx); atomicMin(&m.counters[C_ESDF_AABB + 1], by);
      atomicMax(&m.counters[C_ESDF_AABB + 2], bx); atomicMax(&m.counters[C_ESDF_AABB + 3], by);
    }
    uint2* vp = &m.esdf[(size_t)es * 512 + a.vz_out * 64 + lane];
    // voxel flags = the masks of the last marking pass (own site mask = centre of the neighbourhood)
    const uint32_t vflags = (((s_masks[0] >> lane) & 1ull) ? ESDF_OBSERVED : 0u) | (((s_masks[1] >> lane) & 1ull) ? ESDF_INSIDE : 0u) |
                            (((s_bits[ctr] >> lane) & 1ull) ? ESDF_SITE : 0u);
    // 2. row bitmap: word w of row r holds neighbourhood voxel columns 64(w-1) .. 64(w-1)+63 (bit = column & 63)
    for (int q = tid; q < rows * EDT_ROW_WORDS; q += NT) {
      const int r = q / EDT_ROW_WORDS, w = q - r * EDT_ROW_WORDS;
      u64 word = 0ull;
      if (w >= 1 && w <= 3) {
        const int yy = r + yoff, qy = yy >> 3, sh = 8 * (yy & 7);
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const int qx = (w - 1) * 8 + b;
          if (qx < nn) word |= ((s_bits[qy * nn + qx] >> sh) & 0xFFull) << (8 * b);
        }
      }
      s_rows[q] = word;
    }
    __syncthreads();
    // 3. row pass: nearest site along x within ri (ties -> -x), for the block's own 8 columns on every strip row
    for (int q = tid; q < rows * 8; q += NT) {
      const int r = q >> 3, X = 8 * a.rb + (q & 7);
      const u64* row = &s_rows[r * EDT_ROW_WORDS];
      const int wi = 1 + (X >> 6), b = X & 63;
      const u64 cur = row[wi], prev = row[wi - 1], next = row[wi + 1];
      const u64 left = (b == 63) ? cur : ((cur << (63 - b)) | (prev >> (b + 1)));   // bit 63 <-> x, bit 62 <-> x-1, ...
      const u64 right = (b == 0) ? cur : ((cur >> b) | (next << (64 - b)));         // bit 0 <-> x, bit 1 <-> x+1, ...
      const int dl = left ? __clzll((long long)left) : 64;
      const int dr = right ? (__ffsll((long long)right) - 1) : 64;
      int8_t v = DX_NONE;
      if (dl <= dr) { if (dl <= a.ri) v = (int8_t)(-dl); }
      else { if (dr <= a.ri) v = (int8_t)dr; }
      s_dx[q] = v;
    }
    __syncthreads();
    // 4. column pass: argmin over dy of (dy^2 + dx^2, dy) -- the oracle scans dy ascending with strict improvement,
    //    i.e. the smallest dy among equal distances.  Wave w takes |dy| = w, w + NT/64, ... (increasing, stop at dy^2 > best);
    //    the partial minima are merged with the same lexicographic rule.
    int32_t best = INT32_MAX, bdx = 0, bdy = 0;
    for (int ady = wave; ady <= a.ri; ady += NT / 64) {
      if (ady * ady > best) break;
#pragma unroll
      for (int sgn = 0; sgn < 2; sgn++) {
        if (sgn == 1 && ady == 0) continue;
        const int dy = sgn == 0 ? -ady : ady;
        const int8_t dx = s_dx[(vy + a.ri + dy) * 8 + vx];
        if (dx == DX_NONE) continue;
        const int32_t sq = dy * dy + (int32_t)dx * dx;
        if (sq < best || (sq == best && dy < bdy)) { best = sq; bdx = dx; bdy = dy; }
      }
    }
    // pack (sq, dy, dx) so that integer min == lexicographic (sq, dy) min: sq < 2^13, dy + 64 < 2^7, dx + 64 < 2^7
    s_part[wave * 64 + lane] = best == INT32_MAX ? INT32_MAX : ((best << 14) | ((bdy + 64) << 7) | (bdx + 64));
    __syncthreads();
    if (wave == 0) {
      int32_t p = s_part[lane];
#pragma unroll
      for (int w = 1; w < NT / 64; w++) { const int32_t o = s_part[w * 64 + lane]; if (o < p) p = o; }
      if (p != INT32_MAX && (float)(p >> 14) <= a.max_sq) {
        const int32_t fdx = (p & 127) - 64, fdy = ((p >> 7) & 127) - 64;
        *vp = make_uint2(__float_as_uint((float)(p >> 14)), vflags | ((uint32_t)(uint8_t)(int8_t)fdx) | (((uint32_t)(uint8_t)(int8_t)fdy) << 8));
      } else {
        *vp = make_uint2(__float_as_uint(a.max_sq), vflags);
      }
      if (lane == 0) atomicAdd(shc_at(m, srec, my_shard(), 5), 1);
    }
  }
}


}  // namespace nvbx
