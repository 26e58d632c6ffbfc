// cull.h -- the brick cull of the fusion pass (integrate.hip) as device code that more than one launch can carry: k_integrate_cull itself, and the
// persistent ICP loop, whose workgroups run it as their tail once the pose is committed (track.hip: a steady-state frame then has three launches).
// Reference: the update predicate of integrateKernel, src/cuda/integrateVolume.cu:39-67 -- a brick is dropped only when no voxel of it can pass it.
#pragma once
#include "kf_internal.h"
#ifndef KF_CULL_PX_SLACK
#define KF_CULL_PX_SLACK 0.0625f          // pixels: see cull_test_cell
#endif

struct IntegrateArgs {
  KfVolume vol;
  KfCam dcam, rcam;
  const float* depth;            // trunced_depth
  const float4* normals;         // new_normals_pyramid[0]
  const uchar4* rgb;             // raw_rgb
  const float* tinv;             // device: world -> camera = Mat44::getInverse of the pose (integrateVolume.cu:84), kept next to the
                                 // device-resident pose by whoever commits it (KfTrackState::pose_inv); null: tinv_val
  KfMat tinv_val;                // the inverse of a host-supplied transform, computed on the host with the same arithmetic
  float* tile_max;               // device: max of the depth gated by max_dist over 8-pixel and 16-pixel tiles (two tables, see tile_off)
  int tile_off[2], tile_w[2], tile_h[2];   // offset / width / height of each level's table inside tile_max
  int fine_tiles;                // 1: bricks are small on screen, the cull reads the 8-pixel table where the footprint allows
  unsigned* queue;               // active brick slots
  const unsigned* queue_pad;     // 16 aligned bytes nobody writes during the fusion pass (what non-updating lanes load)
  unsigned queue_cap;            // entries the queue can hold (= stored bricks)
  KfCounters* cnt;
  const KfTrackState* track;     // non-null: integrate only when track->tracked
  float sdf_trunc, max_dist;
  int has_color, color_angled;
  float fr_slope[4], fr_norm[4]; // frustum side planes through the eye (left, right, top, bottom), widened by one pixel: slope and sqrt(1+slope^2)
  int exp_mode;                  // timing experiments only (KF_INTEGRATE_EXP): 1 = no store, 2 = no load/store
  int parity;                    // which of the double-buffered counter sets (KfCounters) this call uses
  int clear_tiles, n_tile_floats;   // the fusion pass clears the tile tables (maxima to 0, minima to +inf) once the cull has read them
  int defer_cull;                // deferred-weight words are in use (k_integrate_pairs<.., DEFER>) and the tile minima describe this depth map: the cull may
                                 // retire whole free-space bricks (counted, their pending counts bumped, never queued)
  int macro_depth;               // 1: the depth-range test once per macro cell in front of the 64 brick tests (cull_test_cell; KF_CULL_MACRO_DEPTH=0: off)
  int free_ok;                   // sdf_trunc > 0 (and the shortcut not disabled): free-space waves skip the quotients (k_integrate_pairs)
  unsigned long long* layer_work;   // non-null on sampled frames (kf_count_layer_work): voxels updated per BRICK LAYER of the whole volume, queued bricks only --
                                    // what z-slab ranks balance their boundaries on (pipeline.SlabPipeline.rebalance)
};

// pass 1: one WAVE per 32^3-voxel macro cell (4x4x4 bricks = 64 lanes).  The wave first tests the macro cell's bounding
// sphere against the depth range and the frustum (uniform: a culled macro cell costs a few scalar-ish instructions for 64
// bricks), then each lane tests its own brick, and the survivors are appended with ONE atomic per wave.  All tests are
// conservative: a brick is dropped only when no voxel of it can pass the reference's predicate.
__device__ __forceinline__ bool cull_sphere_visible(const IntegrateArgs& a, const float* m, float cx, float cy, float cz, float r,
                                                    float& px, float& py, float& pz) {
  px = m[0] * cx + m[1] * cy + m[2] * cz + m[3];
  py = m[4] * cx + m[5] * cy + m[6] * cz + m[7];
  pz = m[8] * cx + m[9] * cy + m[10] * cz + m[11];
  if (pz + r <= 0.f) return false;                                     // every voxel has pf.z <= 0
  if (pz - r >= a.max_dist + a.sdf_trunc) return false;                // needs pf.z < depth + trunc < max_dist + trunc
  if ((px - a.fr_slope[0] * pz) < -r * a.fr_norm[0]) return false;
  if ((a.fr_slope[1] * pz - px) < -r * a.fr_norm[1]) return false;
  if ((py - a.fr_slope[2] * pz) < -r * a.fr_norm[2]) return false;
  if ((a.fr_slope[3] * pz - py) < -r * a.fr_norm[3]) return false;
  return true;
}

// one more whole-quarter free-space observation of a quarter in a deferred state p >= 1 (k = p - 1 pending): p + 1, or KF_PEND_SAT once
// k + 1 >= max_weight - 1 -- every stored weight is >= 1, so every true weight fminf(w + k + 1, max) has then reached max_weight
__device__ __forceinline__ unsigned kf_pend_step(unsigned p, float max_weight) {
  if (p >= KF_PEND_SAT) return KF_PEND_SAT;
  const unsigned n = p + 1u;
  return ((float)n >= max_weight || n >= KF_PEND_SAT) ? KF_PEND_SAT : n;
}

// The cull's test of ONE macro cell by ONE wave (lane = brick of the cell): `keep` = the lane's brick must be visited by the fusion pass; DEFER: `noop` =
// the brick was retired here (counted by the caller, its pending counts bumped here).  m = the world -> camera transform (a uniform pointer: kernel
// arguments, global memory or LDS).  Shared by k_integrate_cull (integrate.hip) and the cull that runs as the tail of the tracking launch (track.hip).
template <bool DEFER>
__device__ __forceinline__ bool cull_test_cell(const IntegrateArgs& a, const float* __restrict__ m, int mx, int my, int mz, bool cell_exists, int lane,
                                               int& bx_out, int& by_out, int& bz_out, bool& noop_out) {
  const KfVolume& v = a.vol;
  const float cell = v.cell;
  float px, py, pz;
  const int bx = mx * 4 + (lane & 3), by = my * 4 + ((lane >> 2) & 3), bz = mz * 4 + (lane >> 4);
  // macro cell: voxel centres span [(32m+0.5), (32m+31.5)] * cell -> centre (32m+16)*cell, half-diagonal 15.5*sqrt(3)*cell
  bool keep = cell_exists &&
              cull_sphere_visible(a, m, (float)(mx * 32 + 16) * cell, (float)(my * 32 + 16) * cell, (float)(mz * 32 + 16) * cell,
                                  27.0f * cell + 1e-4f * v.size, px, py, pz);
  // The brick test's depth-range argument once for the whole macro cell (wave-uniform): its 32^3 voxel centres span +-15.5 cells around the centre; when the
  // cell's pixel footprint spans at most 4 x 4 of the 16-pixel tiles, sixteen lanes read one tile maximum each, and unless SOME tile holds a depth whose
  // surface the cell's near side can reach (z_near < tile maximum + truncation), no voxel of the cell passes the reference's predicate (:50, :64-67) -- the
  // 64 brick tests (their loads, their quotients) are not made.  In a room seen from inside, most of the frustum's cells lie behind the walls.
  if (keep && a.macro_depth) {
    const float hM = 15.5f * cell, epsM = 1e-4f * v.size;
    const float exM = hM * (fabsf(m[0]) + fabsf(m[1]) + fabsf(m[2])) + epsM, eyM = hM * (fabsf(m[4]) + fabsf(m[5]) + fabsf(m[6])) + epsM;
    const float ezM = hM * (fabsf(m[8]) + fabsf(m[9]) + fabsf(m[10])) + epsM;
    const float znM = pz - ezM, zfM = pz + ezM;
    if (znM > 4.f * cell) {
      const float xl = px - exM, xr = px + exM, yl = py - eyM, yr = py + eyM;
      const float u0 = (xl < 0.f ? xl / znM : xl / zfM) * a.dcam.fx + a.dcam.cx, u1 = (xr > 0.f ? xr / znM : xr / zfM) * a.dcam.fx + a.dcam.cx;
      const float w0 = (yl < 0.f ? yl / znM : yl / zfM) * a.dcam.fy + a.dcam.cy, w1 = (yr > 0.f ? yr / znM : yr / zfM) * a.dcam.fy + a.dcam.cy;
      int ix0 = (int)floorf(u0 + (0.5f - KF_CULL_PX_SLACK)), ix1 = (int)floorf(u1 + (0.5f + KF_CULL_PX_SLACK));
      int iy0 = (int)floorf(w0 + (0.5f - KF_CULL_PX_SLACK)), iy1 = (int)floorf(w1 + (0.5f + KF_CULL_PX_SLACK));
      ix0 = max(ix0, 0); iy0 = max(iy0, 0); ix1 = min(ix1, a.dcam.cols - 1); iy1 = min(iy1, a.dcam.rows - 1);
      if (ix0 > ix1 || iy0 > iy1) keep = false;                              // no voxel's pixel lies in the image
      else {
        const int tx0 = ix0 >> 4, tx1 = ix1 >> 4, ty0 = iy0 >> 4, ty1 = iy1 >> 4;
        if (tx1 - tx0 < 4 && ty1 - ty0 < 4) {
          const float t = a.tile_max[a.tile_off[1] + min(ty0 + ((lane >> 2) & 3), ty1) * a.tile_w[1] + min(tx0 + (lane & 3), tx1)];
          if (__ballot(t != 0.f && znM < t + a.sdf_trunc) == 0ull) keep = false;
        }
      }
    }
  }
  keep = keep && bx < v.nb && by < v.nb && bz >= v.bz0 && bz < v.bz1;
  // brick: voxel centres span [(8b+0.5), (8b+7.5)] * cell per axis: an axis-aligned box of half-extent 3.5 cells around
  // (8b+4)*cell.  Every test below is linear in the voxel position, so its extreme over the box is the value at the centre plus
  // the box's support h * sum |coefficients| -- up to 42 % tighter than the bounding sphere for planes along the axes, which is
  // what decides bricks on the frustum's sides and just behind a surface.
  const float h = 3.5f * cell, eps = 1e-4f * v.size;
  const float ex = h * (fabsf(m[0]) + fabsf(m[1]) + fabsf(m[2])) + eps, ey = h * (fabsf(m[4]) + fabsf(m[5]) + fabsf(m[6])) + eps;
  const float ez = h * (fabsf(m[8]) + fabsf(m[9]) + fabsf(m[10])) + eps;
  if (keep) {
    const float cx = (float)(bx * 8 + 4) * cell, cy = (float)(by * 8 + 4) * cell, cz = (float)(bz * 8 + 4) * cell;
    px = m[0] * cx + m[1] * cy + m[2] * cz + m[3];
    py = m[4] * cx + m[5] * cy + m[6] * cz + m[7];
    pz = m[8] * cx + m[9] * cy + m[10] * cz + m[11];
    keep = pz + ez > 0.f && pz - ez < a.max_dist + a.sdf_trunc;         // some voxel with 0 < pf.z < max_dist + trunc
    // frustum sides (through the eye, one pixel wider): inside means x - tl*z >= 0, tr*z - x >= 0, y - tt*z >= 0, tb*z - y >= 0
    const float tl = a.fr_slope[0], tr = a.fr_slope[1], tt = a.fr_slope[2], tb = a.fr_slope[3];
    const float sl = h * (fabsf(m[0] - tl * m[8]) + fabsf(m[1] - tl * m[9]) + fabsf(m[2] - tl * m[10])) + eps * a.fr_norm[0];
    const float sr = h * (fabsf(tr * m[8] - m[0]) + fabsf(tr * m[9] - m[1]) + fabsf(tr * m[10] - m[2])) + eps * a.fr_norm[1];
    const float st = h * (fabsf(m[4] - tt * m[8]) + fabsf(m[5] - tt * m[9]) + fabsf(m[6] - tt * m[10])) + eps * a.fr_norm[2];
    const float sb = h * (fabsf(tb * m[8] - m[4]) + fabsf(tb * m[9] - m[5]) + fabsf(tb * m[10] - m[6])) + eps * a.fr_norm[3];
    keep = keep && (px - tl * pz) + sl >= 0.f && (tr * pz - px) + sr >= 0.f && (py - tt * pz) + st >= 0.f && (tb * pz - py) + sb >= 0.f;
  }
  // depth test against the tile max over the brick's pixel footprint (only when the brick is clear of the eye plane)
  const float zn = pz - ez, zf = pz + ez;
  bool noop = false;                                                       // a brick retired here: its 512 voxels are counted, not visited
  // DEFER: the brick's four deferred-weight words, requested before the depth tests so that they travel together with the tile maxima
  // (behind them they were one more dependent round trip at the end of every surviving lane's chain)
  const unsigned cull_slot = keep ? kf_brick_slot(v, bx, by, bz) : 0u;
  const unsigned long long pp = DEFER ? v.pend[kf_opaque(cull_slot)] : 0ull;
  if (keep && zn > 4.f * cell) {
    const float xl = px - ex, xr = px + ex, yl = py - ey, yr = py + ey;
    float u0 = (xl < 0.f ? xl / zn : xl / zf) * a.dcam.fx + a.dcam.cx, u1 = (xr > 0.f ? xr / zn : xr / zf) * a.dcam.fx + a.dcam.cx;
    float w0 = (yl < 0.f ? yl / zn : yl / zf) * a.dcam.fy + a.dcam.cy, w1 = (yr > 0.f ? yr / zn : yr / zf) * a.dcam.fy + a.dcam.cy;
#ifndef KF_CULL_WIDE_MARGIN
    // A voxel's pixel is (int)(u + 0.5) of ITS projection u (DepthCamera.h:30-43), and u0 .. u1 bound the projections of all the brick's voxels (the box's
    // support, widened by eps = 1e-4 of the volume, ~0.1 pixel): the pixels lie in [floor(u0 + 0.5), floor(u1 + 0.5)].  KF_CULL_PX_SLACK more on either side
    // covers the last-bit differences between these expressions and the fusion kernel's own (a few 1e-4 of a pixel).  (Rounds 1-4 used floor(u0) - 1 ..
    // ceil(u1) + 2: three pixels wider than needed, on footprints of 4-8 pixels -- a fifth of the queued bricks of a 1024^3 volume were free space that the
    // tile minima of those extra pixels did not vouch for: profiles/r04_cull_margin.txt.)
    int ix0 = (int)floorf(u0 + (0.5f - KF_CULL_PX_SLACK)), ix1 = (int)floorf(u1 + (0.5f + KF_CULL_PX_SLACK));
    int iy0 = (int)floorf(w0 + (0.5f - KF_CULL_PX_SLACK)), iy1 = (int)floorf(w1 + (0.5f + KF_CULL_PX_SLACK));
#else
    int ix0 = (int)floorf(u0) - 1, ix1 = (int)ceilf(u1) + 2, iy0 = (int)floorf(w0) - 1, iy1 = (int)ceilf(w1) + 2;
#endif
    // every voxel's pixel lies in [ix0, ix1] x [iy0, iy1]: inside the reference's 1 .. cols-2 / rows-2 window (integrateVolume.cu:43)?
    const bool all_inside = ix0 >= 1 && iy0 >= 1 && ix1 <= a.dcam.cols - 2 && iy1 <= a.dcam.rows - 2;
    ix0 = max(ix0, 0); iy0 = max(iy0, 0); ix1 = min(ix1, a.dcam.cols - 1); iy1 = min(iy1, a.dcam.rows - 1);
    if (ix0 > ix1 || iy0 > iy1) keep = false;
    else {
      // 8-pixel tiles when the footprint spans at most 4 x 4 of them (distant bricks: a far tighter maximum), else 16-pixel tiles
      const int lvl = (!a.fine_tiles || ((ix1 >> 3) - (ix0 >> 3)) >= 4 || ((iy1 >> 3) - (iy0 >> 3)) >= 4) ? 1 : 0;
      const int sh = 3 + lvl;
      const int tx0 = ix0 >> sh, tx1 = ix1 >> sh, ty0 = iy0 >> sh, ty1 = iy1 >> sh;
      if (tx1 - tx0 < 4 && ty1 - ty0 < 4) {
        // sixteen independent clamped loads instead of a dependent loop
        const float* tbl = a.tile_max + a.tile_off[lvl];
        const int tw = a.tile_w[lvl];
        float dmax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) dmax = fmaxf(dmax, tbl[min(ty0 + j, ty1) * tw + min(tx0 + i, tx1)]);
        if (dmax == 0.f) keep = false;                                     // no pixel under the brick can integrate
        else if (zn >= dmax + a.sdf_trunc) keep = false;                   // every voxel lies behind every surface it can see
        else if (DEFER && all_inside) {
          // Free space in a deferred state seen as free space again: all four quarters of the brick carry a deferred-weight word (its 512
          // voxels hold tsdf 1 and a weight >= 1), every voxel projects inside the image window, every pixel it can land on holds a depth that
          // integrates (tile minimum > 0 means: all valid, all < max_dist) and that depth is at least one truncation distance behind the
          // brick's far side -> every voxel passes the reference's predicate (:39-67), observes tsdf min(1, sdf / trunc) = 1, and
          // (1 * w + 1) / (w + 1) = 1 leaves the tsdf as it is; the weight's min(w + 1, max) is one more pending step of each quarter
          // (nothing at all once a quarter is saturated).  The brick is counted, not queued: no other wave touches it in this frame.
          const unsigned slot = cull_slot;
          const unsigned q0 = (unsigned)(pp & 0xFFFFull), q1 = (unsigned)((pp >> 16) & 0xFFFFull), q2 = (unsigned)((pp >> 32) & 0xFFFFull), q3 = (unsigned)(pp >> 48);
          if (q0 && q1 && q2 && q3) {
            const float* tmn = tbl + a.n_tile_floats;
            float dmin = __builtin_huge_valf();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < 4; ++i) dmin = fminf(dmin, tmn[min(ty0 + j, ty1) * tw + min(tx0 + i, tx1)]);
            if (dmin > 0.f && dmin - zf >= a.sdf_trunc + eps) {
              keep = false; noop = true;
              if (pp != ~0ull) v.pend[slot] = (unsigned long long)kf_pend_step(q0, v.max_weight) | ((unsigned long long)kf_pend_step(q1, v.max_weight) << 16) |
                                              ((unsigned long long)kf_pend_step(q2, v.max_weight) << 32) | ((unsigned long long)kf_pend_step(q3, v.max_weight) << 48);
            }
          }
        }
      }                                                                    // larger footprints (bricks close to the eye) are kept
    }
  }
#ifdef KF_EXPERIMENTS
  // KF_INTEGRATE_EXP=17 (tools/exp_cull_deferred.py): what would an EXACT maximum / minimum of the depth image over the brick's pixel footprint drop or
  // retire (the upper bound of what finer tile tables can give), and what a 4-pixel table level?  rc_steps lo: kept bricks; hi: dropped by the exact
  // maximum; rc_hits lo: of the rest, retired by the exact minimum (all four quarters deferred); hi: dropped with 4-pixel tiles; mc_blocks lo: retired with them
  if (KF_EXP_MODE(a) == 17 && keep) {
    const unsigned sh = (unsigned)(lane & 63) * 16u;
    atomicAdd(&a.cnt->rc_steps[sh], 1ull);
    if (zn > 4.f * cell) {
      const float xl = px - ex, xr = px + ex, yl = py - ey, yr = py + ey;
      float u0 = (xl < 0.f ? xl / zn : xl / zf) * a.dcam.fx + a.dcam.cx, u1 = (xr > 0.f ? xr / zn : xr / zf) * a.dcam.fx + a.dcam.cx;
      float w0 = (yl < 0.f ? yl / zn : yl / zf) * a.dcam.fy + a.dcam.cy, w1 = (yr > 0.f ? yr / zn : yr / zf) * a.dcam.fy + a.dcam.cy;
      int ix0 = (int)floorf(u0 + (0.5f - KF_CULL_PX_SLACK)), ix1 = (int)floorf(u1 + (0.5f + KF_CULL_PX_SLACK));
      int iy0 = (int)floorf(w0 + (0.5f - KF_CULL_PX_SLACK)), iy1 = (int)floorf(w1 + (0.5f + KF_CULL_PX_SLACK));
      const bool all_inside = ix0 >= 1 && iy0 >= 1 && ix1 <= a.dcam.cols - 2 && iy1 <= a.dcam.rows - 2;
      ix0 = max(ix0, 0); iy0 = max(iy0, 0); ix1 = min(ix1, a.dcam.cols - 1); iy1 = min(iy1, a.dcam.rows - 1);
      if (ix0 <= ix1 && iy0 <= iy1 && (ix1 - ix0) < 64 && (iy1 - iy0) < 64) {
        float dmax = 0.f, dmin = __builtin_huge_valf(), dmax4 = 0.f, dmin4 = __builtin_huge_valf();
        for (int y = iy0; y <= iy1; ++y) for (int x = ix0; x <= ix1; ++x) {
          const float dv = a.depth[y * a.dcam.cols + x]; const float g = dv < a.max_dist ? dv : 0.f;
          dmax = fmaxf(dmax, g); dmin = fminf(dmin, g);
        }
        for (int y = iy0 & ~3; y <= min(iy1 | 3, a.dcam.rows - 1); ++y) for (int x = ix0 & ~3; x <= min(ix1 | 3, a.dcam.cols - 1); ++x) {
          const float dv = a.depth[y * a.dcam.cols + x]; const float g = dv < a.max_dist ? dv : 0.f;
          dmax4 = fmaxf(dmax4, g); dmin4 = fminf(dmin4, g);
        }
        const unsigned q0 = (unsigned)(pp & 0xFFFFull), q1 = (unsigned)((pp >> 16) & 0xFFFFull), q2 = (unsigned)((pp >> 32) & 0xFFFFull), q3 = (unsigned)(pp >> 48);
        const bool alldef = DEFER && q0 && q1 && q2 && q3 && all_inside;
        if (dmax == 0.f || zn >= dmax + a.sdf_trunc) atomicAdd(&a.cnt->rc_steps[sh], 1ull << 32);
        else if (alldef && dmin > 0.f && dmin - zf >= a.sdf_trunc + eps) atomicAdd(&a.cnt->rc_hits[sh], 1ull);
        if (dmax4 == 0.f || zn >= dmax4 + a.sdf_trunc) atomicAdd(&a.cnt->rc_hits[sh], 1ull << 32);
        else if (alldef && dmin4 > 0.f && dmin4 - zf >= a.sdf_trunc + eps) atomicAdd(&a.cnt->mc_blocks[sh], 1ull);
      }
    }
  }
  // KF_INTEGRATE_EXP=16 (tools/exp_cull_deferred.py): why are bricks whose quarters are in a deferred state still queued?  Counted per kept brick.
  if (DEFER && KF_EXP_MODE(a) == 16 && keep) {
    const unsigned q0 = (unsigned)(pp & 0xFFFFull), q1 = (unsigned)((pp >> 16) & 0xFFFFull), q2 = (unsigned)((pp >> 32) & 0xFFFFull), q3 = (unsigned)(pp >> 48);
    const int nq = (q0 != 0) + (q1 != 0) + (q2 != 0) + (q3 != 0);
    const unsigned sh = (unsigned)(lane & 63) * 16u;
    if (nq == 4) {
      atomicAdd(&a.cnt->rc_steps[sh], 1ull);                                              // all four quarters deferred, queued all the same
      bool near_eye = !(zn > 4.f * cell), outside = false, large = false, invalid = false, close = false;
      if (!near_eye) {
        const float xl = px - ex, xr = px + ex, yl = py - ey, yr = py + ey;
        float u0 = (xl < 0.f ? xl / zn : xl / zf) * a.dcam.fx + a.dcam.cx, u1 = (xr > 0.f ? xr / zn : xr / zf) * a.dcam.fx + a.dcam.cx;
        float w0 = (yl < 0.f ? yl / zn : yl / zf) * a.dcam.fy + a.dcam.cy, w1 = (yr > 0.f ? yr / zn : yr / zf) * a.dcam.fy + a.dcam.cy;
        int ix0 = (int)floorf(u0) - 1, ix1 = (int)ceilf(u1) + 2, iy0 = (int)floorf(w0) - 1, iy1 = (int)ceilf(w1) + 2;
        outside = !(ix0 >= 1 && iy0 >= 1 && ix1 <= a.dcam.cols - 2 && iy1 <= a.dcam.rows - 2);
        ix0 = max(ix0, 0); iy0 = max(iy0, 0); ix1 = min(ix1, a.dcam.cols - 1); iy1 = min(iy1, a.dcam.rows - 1);
        const int lvl = (!a.fine_tiles || ((ix1 >> 3) - (ix0 >> 3)) >= 4 || ((iy1 >> 3) - (iy0 >> 3)) >= 4) ? 1 : 0;
        const int sh2 = 3 + lvl, tx0 = ix0 >> sh2, tx1 = ix1 >> sh2, ty0 = iy0 >> sh2, ty1 = iy1 >> sh2;
        large = !(tx1 - tx0 < 4 && ty1 - ty0 < 4);
        if (!outside && !large) {
          const float* tmn = a.tile_max + a.tile_off[lvl] + a.n_tile_floats;
          float dmin = __builtin_huge_valf();
          for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) dmin = fminf(dmin, tmn[min(ty0 + j, ty1) * a.tile_w[lvl] + min(tx0 + i, tx1)]);
          invalid = !(dmin > 0.f);
          close = !invalid;                                                                // (what is left: the minimum depth is within a truncation distance of the brick)
        }
      }
      if (near_eye || large) atomicAdd(&a.cnt->rc_steps[sh], 1ull << 32);               // no tile test possible (next to the eye / a footprint beyond 4 x 4 tiles)
      else if (outside) atomicAdd(&a.cnt->rc_hits[sh], 1ull);                            // some voxel may project outside the image window
      else if (invalid) atomicAdd(&a.cnt->rc_hits[sh], 1ull << 32);                      // a pixel under the footprint's tiles holds no depth that integrates
      else if (close) atomicAdd(&a.cnt->mc_blocks[sh], 1ull);                            // the tiles' minimum depth is not a truncation distance behind the brick
    } else if (nq > 0) atomicAdd(&a.cnt->mc_blocks[sh], 1ull << 32);                     // one to three quarters deferred
  }
#endif
  bx_out = bx; by_out = by; bz_out = bz; noop_out = noop;
  return keep;
}

// the same for macro cell number `wave` of the stored volume, counted x-fastest from macro layer mz0 (n_macro cells in all)
template <bool DEFER>
__device__ __forceinline__ bool cull_test(const IntegrateArgs& a, const float* __restrict__ m, int wave, int lane, int n_macro, int nmxy, int mz0,
                                          int& bx_out, int& by_out, int& bz_out, bool& noop_out) {
  return cull_test_cell<DEFER>(a, m, wave % nmxy, (wave / nmxy) % nmxy, wave / (nmxy * nmxy) + mz0, wave < n_macro, lane, bx_out, by_out, bz_out, noop_out);
}

// ---- the cull as the TAIL of another launch (the persistent ICP loop, track.hip) -------------------------------------------------------------------
// n_wg workgroups share the macro cells: workgroup wg's wave wid takes the cells (it * n_wg + wg) * waves + wid, it = 0 .. CULL_TAIL_ROUNDS - 1 (the host
// arms the tail only when that covers the volume: kf_cull_tail_fits).  All of a wave's tests run first (their tile-table loads travel together), then ONE
// queue atomic per workgroup.  Plain cull only (no deferred-weight words: those volumes have too many macro cells for a tail anyway), no side effect but
// the queue and its counter -- a tail nobody consumes is undone by zeroing that counter (kf_tail_cull_discard).  m: world -> camera, in LDS;
// s_cnt: 17 words of LDS.  Every lane of the workgroup must call.
#define CULL_TAIL_ROUNDS 4
__device__ __forceinline__ void cull_tail(const IntegrateArgs& a, const float* m, int wg, int n_wg, unsigned* s_cnt) {
  const KfVolume& v = a.vol;
  const int nmxy = (v.nb + 3) >> 2, mz0 = v.bz0 >> 2, mz1 = (v.bz1 + 3) >> 2;
  const int n_macro = nmxy * nmxy * (mz1 - mz0);
  const int waves = (int)(blockDim.x >> 6), wid = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int per_round = n_wg * waves;
  unsigned packed[CULL_TAIL_ROUNDS];
  unsigned long long mask[CULL_TAIL_ROUNDS];
  unsigned mine = 0;
#pragma unroll
  for (int it = 0; it < CULL_TAIL_ROUNDS; ++it) {
    packed[it] = 0u; mask[it] = 0ull;
    if (it * per_round < n_macro) {                                         // (uniform over the launch)
      int bx, by, bz; bool noop;
      const bool keep = cull_test<false>(a, m, (it * n_wg + wg) * waves + wid, lane, n_macro, nmxy, mz0, bx, by, bz, noop);
      packed[it] = (unsigned)bx | ((unsigned)by << 10) | ((unsigned)(bz - v.bz0) << 20);
      mask[it] = __ballot(keep);
      mine += (unsigned)__popcll(mask[it]);
    }
  }
  __syncthreads();                                                          // (s_cnt may still be read by a previous call: the solo finish calls in a loop)
  if (lane == 0) s_cnt[wid] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0;
    for (int w = 0; w < waves; ++w) { const unsigned n = s_cnt[w]; s_cnt[w] = total; total += n; }
    s_cnt[16] = total ? atomicAdd(&a.cnt->n_active[a.parity], total) : 0u;
  }
  __syncthreads();
  unsigned pos = s_cnt[16] + s_cnt[wid];
#pragma unroll
  for (int it = 0; it < CULL_TAIL_ROUNDS; ++it) {
    if ((mask[it] >> lane) & 1ull) a.queue[pos + (unsigned)__popcll(mask[it] & ((1ull << lane) - 1ull))] = packed[it];
    pos += (unsigned)__popcll(mask[it]);
  }
}

// host side (integrate.hip)
struct kf_ctx;
void kf_fill_cull_args(kf_ctx* c, IntegrateArgs& a, const kf_camera_params* dcam, float sdf_trunc, float max_dist);   // everything the cull reads; pose = the device-resident one
bool kf_cull_tail_fits(const kf_ctx* c, int n_wg, int waves);
