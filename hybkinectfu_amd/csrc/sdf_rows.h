// sdf_rows.h -- one pixel's row of the direct SDF tracker's system: buildSDFSolverRows (src/cuda/CalSDFErrSolverParams.cu:7-66) over
// tsdfvolume::interpolateSDF (src/cuda/tsdfVolume.h:98-122, :151-172).
//
// The reference looks the volume up 13 times per pixel, 8 voxels each: at the pixel's world point pw0, at six points rotated about the
// world axes by -+0.001 rad, and at pw0 -+ one cell along each axis.  Round 4 issued them as 13 independent trilinear lookups, two at a
// time: ~170 vector instructions each (three exact quotients for the cell, eight brick-slot computations, 24 multiplies), i.e. the
// step was bound by VALU issue, not by the gathers (profiles/r05_c3_before_*).  Here:
//   * a voxel's element index is separable in the bricked layout: idx(x, y, z) = ox(x) + oy(y) + oz(z)
//     (ox = (x >> 3) * 512 + (x & 7), oy = (y >> 3) * nb * 512 + (y & 7) * 8, oz = ((z >> 3) - bz0) * nb^2 * 512 + (z & 7) * 64):
//     three offsets per axis position, then one or two adds per voxel;
//   * pw0 and its six axis neighbours share ONE gather of 32 voxels: the 2x2x2 cell of pw0 extended by one voxel on either side along
//     each axis (the reference reads 7 x 8 = 56).  A lookup at pw0 -+ v_h e_x has the SAME y / z cell and fractions as pw0 (same
//     operands, same operations) and an x cell one to the side; its x cell and fraction are computed the reference's way and select two
//     of the four bilinear (y, z) interpolants along the line.  Should rounding ever put its cell outside the line (it cannot for
//     v_h = cell, but nothing here relies on that), the lookup falls back to the generic path under a wave-uniform branch;
//   * the six rotated lookups stay generic (their cells differ from pw0's in two axes at once), with the separable addressing;
//   * the trilinear value is formed by nested lerps (14 operations instead of 31).  The cell and the validity test (all 8 weights
//     non-zero, the cell inside the volume and inside the stored layers) are the reference's own, bit for bit -- they decide WHICH pixels
//     count; the fractions and the interpolation are tolerance-side (the 27 sums are checked at 1e-5 of the largest entry against fp64).
#pragma once
#include "kf_internal.h"
#ifndef SDF_TAP_BATCH
#define SDF_TAP_BATCH 6
#endif

// adjusted base cell along one axis, the fraction inside it, and the range test of the unadjusted cell (tsdfVolume.h:50-56, :155-169)
struct SdfCell { int g; float f; bool ok; };
__device__ __forceinline__ SdfCell sdf_cell(float pos, float r, const KfRecip& rS, float cell, float rcell, int R) {
  SdfCell c;
  const int g = kf_f2i(kf_div(pos * r, rS));                         // the exact quotient: the cell decides which voxels vouch for the lookup
  c.ok = g > 0 && g < R - 1;
  const int ga = (pos < ((float)g + 0.5f) * cell) ? g - 1 : g;
  c.g = ga;
  c.f = (pos - ((float)ga + 0.5f) * cell) * rcell;                    // (the reference divides by the cell size: tolerance side)
  return c;
}

// Two ways to address the bricked volume, same interface.  offsets are separable: off(x, y, z) = ox(x) + oy(y) + oz(z).
//   SdfBufAddr  volumes below 4 GB (up to 768^3): raw buffer loads (MI355X guide: __builtin_amdgcn_make_buffer_rsrc + raw_buffer_load), byte offsets in 32
//               bits, and the descriptor's range check in place of clamps: an offset beyond the stored bytes reads as (0, 0), one that wraps into them
//               reads some other voxel -- either way on behalf of a lookup whose own range test has already failed (see sdf_pixel_row)
//   SdfFlatAddr any size: 64-bit element offsets, coordinates clamped into the stored volume
typedef unsigned sdf_v2u __attribute__((ext_vector_type(2)));
struct SdfBufAddr {
  typedef unsigned IDX;
  __amdgpu_buffer_rsrc_t rsrc; unsigned nb2; unsigned nb; int zlo, zhi, R, bz0;
  __device__ __forceinline__ explicit SdfBufAddr(const KfVolume& v) {
    const unsigned long long bytes = (unsigned long long)(v.bz1 - v.bz0) * v.nb * v.nb * KF_BRICK_VOX * sizeof(float2);
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)v.tw, 0, (unsigned)bytes, 0x00020000);
    nb = (unsigned)v.nb; nb2 = nb * nb; zlo = v.bz0 * KF_BRICK; zhi = v.bz1 * KF_BRICK - 1; R = v.res; bz0 = v.bz0;
  }
  __device__ __forceinline__ IDX ox(int x) const { return ((unsigned)(x >> 3) << 12) + ((unsigned)(x & 7) << 3); }
  __device__ __forceinline__ IDX oy(int y) const { return (kf_opaque(__umul24((unsigned)(y >> 3), nb)) << 12) + ((unsigned)(y & 7) << 6); }     // (opaque: else the shift is folded into a full 32-bit multiply)
  __device__ __forceinline__ IDX oz(int z) const { return (kf_opaque(__umul24((unsigned)((z >> 3) - bz0), nb2)) << 12) + ((unsigned)(z & 7) << 9); }
  __device__ __forceinline__ int cxy(int x) const { return x; }
  __device__ __forceinline__ int cz(int z) const { return z; }
  __device__ __forceinline__ float2 load(IDX i) const {
    const sdf_v2u a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, i, 0, 0);
    return make_float2(__uint_as_float(a.x), __uint_as_float(a.y));
  }
};
struct SdfFlatAddr {
  typedef size_t IDX;
  const float2* __restrict__ tw; size_t sy, sz; int zlo, zhi, R, bz0;
  __device__ __forceinline__ explicit SdfFlatAddr(const KfVolume& v) {
    tw = v.tw; sy = (size_t)v.nb * KF_BRICK_VOX; sz = sy * (size_t)v.nb; zlo = v.bz0 * KF_BRICK; zhi = v.bz1 * KF_BRICK - 1; R = v.res; bz0 = v.bz0;
  }
  __device__ __forceinline__ IDX ox(int x) const { return (size_t)(unsigned)(x >> 3) * KF_BRICK_VOX + (size_t)(unsigned)(x & 7); }
  __device__ __forceinline__ IDX oy(int y) const { return (size_t)(unsigned)(y >> 3) * sy + (size_t)(unsigned)((y & 7) << 3); }
  __device__ __forceinline__ IDX oz(int z) const { return (size_t)(unsigned)((z >> 3) - bz0) * sz + (size_t)(unsigned)((z & 7) << 6); }
  __device__ __forceinline__ int cxy(int x) const { return min(max(x, 0), R - 1); }
  __device__ __forceinline__ int cz(int z) const { return min(max(z, zlo), zhi); }
  __device__ __forceinline__ float2 load(IDX i) const { return tw[i]; }
};
__device__ __forceinline__ float sdf_lerp(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }
// the smallest of four WEIGHTS, as bits: weights are >= +0 and never NaN, so they order like their bit patterns -- two v_min3_u32 / v_min_u32
// instead of three fminf with their operand canonicalisation; "some weight is zero" <=> the minimum's bits are zero
__device__ __forceinline__ unsigned sdf_min4(float a, float b, float c, float d) {
  return min(min(min(__float_as_uint(a), __float_as_uint(b)), __float_as_uint(c)), __float_as_uint(d));
}

// a generic lookup split into prepare / load / finish, so that several can have their gathers in flight together
template <typename AD> struct SdfTap { typename AD::IDX i[8]; float a, b, c; bool ok; };
template <typename AD>
__device__ __forceinline__ SdfTap<AD> sdf_tap_prepare(float3 pos, float r, const KfRecip& rS, float cell, float rcell, const AD& s) {
  typedef typename AD::IDX IDX;
  SdfTap<AD> t;
  const SdfCell cx = sdf_cell(pos.x, r, rS, cell, rcell, s.R), cy = sdf_cell(pos.y, r, rS, cell, rcell, s.R), cz = sdf_cell(pos.z, r, rS, cell, rcell, s.R);
  t.a = cx.f; t.b = cy.f; t.c = cz.f;
  // (kf_interpolate_sdf: the range tests, then both z layers stored; a position outside is clamped for the address only -- its lookup has failed already)
  t.ok = cx.ok && cy.ok && cz.ok && cz.g >= s.zlo && cz.g + 1 <= s.zhi;
  const IDX x0 = s.ox(s.cxy(cx.g)), x1 = s.ox(s.cxy(cx.g + 1));
  const IDX y0 = s.oy(s.cxy(cy.g)), y1 = s.oy(s.cxy(cy.g + 1));
  const IDX z0 = s.oz(s.cz(cz.g)), z1 = s.oz(s.cz(cz.g + 1));
  const IDX yz00 = y0 + z0, yz01 = y0 + z1, yz10 = y1 + z0, yz11 = y1 + z1;
  t.i[0] = x0 + yz00; t.i[1] = x0 + yz01; t.i[2] = x0 + yz10; t.i[3] = x0 + yz11;       // k = (dx << 2) | (dy << 1) | dz, as kf_interpolate_sdf numbers them
  t.i[4] = x1 + yz00; t.i[5] = x1 + yz01; t.i[6] = x1 + yz10; t.i[7] = x1 + yz11;
  return t;
}
template <typename AD> __device__ __forceinline__ void sdf_tap_load(const AD& s, const SdfTap<AD>& t, float2 q[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = s.load(t.i[k]);
}
template <typename AD> __device__ __forceinline__ bool sdf_tap_finish(const SdfTap<AD>& t, const float2 q[8], float& dist) {
  const unsigned wmin = min(sdf_min4(q[0].y, q[1].y, q[2].y, q[3].y), sdf_min4(q[4].y, q[5].y, q[6].y, q[7].y));
  const float z00 = sdf_lerp(q[0].x, q[1].x, t.c), z01 = sdf_lerp(q[2].x, q[3].x, t.c), z10 = sdf_lerp(q[4].x, q[5].x, t.c), z11 = sdf_lerp(q[6].x, q[7].x, t.c);
  dist = sdf_lerp(sdf_lerp(z00, z01, t.b), sdf_lerp(z10, z11, t.b), t.a);
  return t.ok && wmin != 0u;
}

// the generic lookup as a real call: the fallback of sdf_pixel_row's neighbour lookups (a few dozen pixels per iteration at VGA) stays out of the hot
// path's register allocation
template <typename AD>
__device__ __attribute__((noinline)) bool sdf_lookup_cold(const AD& s, float3 pos, float r, const KfRecip& rS, float cell, float rcell, float& dist) {
  const SdfTap<AD> t = sdf_tap_prepare<AD>(pos, r, rS, cell, rcell, s);
  float2 q[8];
  sdf_tap_load<AD>(s, t, q);
  return sdf_tap_finish<AD>(t, q, dist);
}

// One pixel's row (7 floats).  s_m: cur, then delta * cur for +w1, -w1, +w2, -w2, +w3, -w3 (CalSDFErrSolverParams.cu:118-133); p: the pixel's camera-space
// point; w_h, v_h: the reference's two steps (:119-120).  Returns false where the reference's buildSDFSolverRows does (any of the 13 lookups fails).
// `slab`: count the pixel only when pw0's own voxel layer lies in [own_z0, own_z1) (z-slab partition).
template <typename AD>
__device__ __forceinline__ bool sdf_pixel_row(const KfVolume& v, const AD& S, const float (*s_m)[16], float3 p, float w_h, float v_h, const KfRecip& rS, float rcell,
                                              bool slab, float row[7]) {
  typedef typename AD::IDX IDX;
  const float r = (float)v.res, cell = v.cell;
  const float4 p4 = make_float4(p.x, p.y, p.z, 1.0f);
  const float4 pw0 = kf_mat_vec(s_m[0], p4);
  bool ok = true;
  if (slab) {
    const int gz = kf_world_to_voxel(v, kf3(pw0.x, pw0.y, pw0.z)).z;
    ok = gz >= v.own_z0 && gz < v.own_z1;
  }
  // ---- pw0 and its six axis neighbours: one 32-voxel gather ---------------------------------------------------------------------------
  const SdfCell cx = sdf_cell(pw0.x, r, rS, cell, rcell, S.R), cy = sdf_cell(pw0.y, r, rS, cell, rcell, S.R), cz = sdf_cell(pw0.z, r, rS, cell, rcell, S.R);
  const SdfCell cxp = sdf_cell(pw0.x + v_h, r, rS, cell, rcell, S.R), cxm = sdf_cell(pw0.x - v_h, r, rS, cell, rcell, S.R);
  const SdfCell cyp = sdf_cell(pw0.y + v_h, r, rS, cell, rcell, S.R), cym = sdf_cell(pw0.y - v_h, r, rS, cell, rcell, S.R);
  const SdfCell czp = sdf_cell(pw0.z + v_h, r, rS, cell, rcell, S.R), czm = sdf_cell(pw0.z - v_h, r, rS, cell, rcell, S.R);
  IDX ox[4], oy[4], oz[4]; bool zs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ox[i] = S.ox(S.cxy(cx.g - 1 + i));
    oy[i] = S.oy(S.cxy(cy.g - 1 + i));
    const int z = cz.g - 1 + i;
    zs[i] = z >= S.zlo && z <= S.zhi;
    oz[i] = S.oz(S.cz(z));
  }
  // core[i][j][k], i, j, k in {0, 1} = line positions 1, 2; ex[e][j][k]: x position 0 / 3; ey[i][e][k]; ez[i][j][e]
  float2 core[2][2][2], ex[2][2][2], ey[2][2][2], ez[2][2][2];
  {
    IDX yz[2][2], xz[2][2], xy[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) { yz[a][b] = oy[1 + a] + oz[1 + b]; xz[a][b] = ox[1 + a] + oz[1 + b]; xy[a][b] = ox[1 + a] + oy[1 + b]; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          core[i][j][k] = S.load(ox[1 + i] + yz[j][k]);
          ex[i][j][k] = S.load(ox[3 * i] + yz[j][k]);
          ey[i][j][k] = S.load(oy[3 * j] + xz[i][k]);
          ez[i][j][k] = S.load(oz[3 * k] + xy[i][j]);
        }
  }
  // (the 32 gathers above are in flight; the line interpolants below wait for them)
  // ---- the line interpolants --------------------------------------------------------------------------------------------------------------
  const float a = cx.f, b = cy.f, c = cz.f;
  float Lx[4], Ly[4], Lz[4]; unsigned Wx[4], Wy[4], Wz[4];
  {
    float mz[2][2], mx[2][2];                    // core lerped along z (shared by the x and y lines) and along x (the z line)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) { mz[i][j] = sdf_lerp(core[i][j][0].x, core[i][j][1].x, c); mx[i][j] = sdf_lerp(core[0][i][j].x, core[1][i][j].x, a); }   // mx[j][k]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      Lx[1 + i] = sdf_lerp(mz[i][0], mz[i][1], b);
      Ly[1 + i] = sdf_lerp(mz[0][i], mz[1][i], a);
      Lz[1 + i] = sdf_lerp(mx[0][i], mx[1][i], b);
      Wx[1 + i] = sdf_min4(core[i][0][0].y, core[i][0][1].y, core[i][1][0].y, core[i][1][1].y);
      Wy[1 + i] = sdf_min4(core[0][i][0].y, core[0][i][1].y, core[1][i][0].y, core[1][i][1].y);
      Wz[1 + i] = sdf_min4(core[0][0][i].y, core[0][1][i].y, core[1][0][i].y, core[1][1][i].y);
      Lx[3 * i] = sdf_lerp(sdf_lerp(ex[i][0][0].x, ex[i][0][1].x, c), sdf_lerp(ex[i][1][0].x, ex[i][1][1].x, c), b);
      Ly[3 * i] = sdf_lerp(sdf_lerp(ey[0][i][0].x, ey[0][i][1].x, c), sdf_lerp(ey[1][i][0].x, ey[1][i][1].x, c), a);
      Lz[3 * i] = sdf_lerp(sdf_lerp(ez[0][0][i].x, ez[1][0][i].x, a), sdf_lerp(ez[0][1][i].x, ez[1][1][i].x, a), b);
      Wx[3 * i] = sdf_min4(ex[i][0][0].y, ex[i][0][1].y, ex[i][1][0].y, ex[i][1][1].y);
      Wy[3 * i] = sdf_min4(ey[0][i][0].y, ey[0][i][1].y, ey[1][i][0].y, ey[1][i][1].y);
      Wz[3 * i] = sdf_min4(ez[0][0][i].y, ez[0][1][i].y, ez[1][0][i].y, ez[1][1][i].y);
    }
  }
  const bool in_xyz = cx.ok && cy.ok && cz.ok;
  const bool z_core = zs[1] && zs[2];
  const float sdf0 = sdf_lerp(Lx[1], Lx[2], a);
  ok = ok && in_xyz && z_core && min(Wx[1], Wx[2]) != 0u;
  // a neighbour along one axis: its own cell (s = position of its base on the line, 0..2) picks two of the four interpolants; its range test is its own,
  // the other two axes' tests and fractions are pw0's.  off_line: the cell left the line (generic fallback below).
  float sv[6]; bool off_line = false;
#define SDF_NEIGHBOUR(out, nc, base, L, W, other_ok, zlo_ok, zmid_ok, zhi_ok)                                                      \
  { const int s_ = (nc).g - ((base).g - 1);                                                                                          \
    const bool on_ = s_ >= 0 && s_ <= 2;                                                                                             \
    const float l0 = s_ <= 0 ? L[0] : (s_ == 1 ? L[1] : L[2]), l1 = s_ <= 0 ? L[1] : (s_ == 1 ? L[2] : L[3]);                       \
    const unsigned w0 = s_ <= 0 ? W[0] : (s_ == 1 ? W[1] : W[2]), w1 = s_ <= 0 ? W[1] : (s_ == 1 ? W[2] : W[3]);                    \
    const bool st_ = s_ <= 0 ? (zlo_ok) : (s_ == 1 ? (zmid_ok) : (zhi_ok));                                                          \
    out = sdf_lerp(l0, l1, (nc).f);                                                                                                  \
    off_line = off_line || (!on_ && (nc).ok && (other_ok));                                                                          \
    ok = ok && (nc).ok && (other_ok) && st_ && (!on_ || min(w0, w1) != 0u); }
  SDF_NEIGHBOUR(sv[0], cxp, cx, Lx, Wx, cy.ok && cz.ok, z_core, z_core, z_core)
  SDF_NEIGHBOUR(sv[1], cxm, cx, Lx, Wx, cy.ok && cz.ok, z_core, z_core, z_core)
  SDF_NEIGHBOUR(sv[2], cyp, cy, Ly, Wy, cx.ok && cz.ok, z_core, z_core, z_core)
  SDF_NEIGHBOUR(sv[3], cym, cy, Ly, Wy, cx.ok && cz.ok, z_core, z_core, z_core)
  SDF_NEIGHBOUR(sv[4], czp, cz, Lz, Wz, cx.ok && cy.ok, zs[0] && zs[1], zs[1] && zs[2], zs[2] && zs[3])
  SDF_NEIGHBOUR(sv[5], czm, cz, Lz, Wz, cx.ok && cy.ok, zs[0] && zs[1], zs[1] && zs[2], zs[2] && zs[3])
#undef SDF_NEIGHBOUR
#if SDF_TAP_BATCH < 6
  __builtin_amdgcn_sched_barrier(0);
#endif
  // ---- the six rotated lookups, SDF_TAP_BATCH at a time (6: ONE batch of 48 gathers -- the pixel then costs two memory round trips, this one and the
  // 32-voxel gather above; 2: three batches of 16, for launch shapes with fewer registers per lane and more waves to cover the round trips) --------
  float sw[6];
#pragma unroll
  for (int k0 = 0; k0 < 6; k0 += SDF_TAP_BATCH) {
    SdfTap<AD> tap[SDF_TAP_BATCH];
#pragma unroll
    for (int k = 0; k < SDF_TAP_BATCH; ++k) {
      const float4 pr = kf_mat_vec(s_m[1 + k0 + k], p4);
      tap[k] = sdf_tap_prepare<AD>(kf3(pr.x, pr.y, pr.z), r, rS, cell, rcell, S);
    }
    float2 qt[SDF_TAP_BATCH][8];
#pragma unroll
    for (int k = 0; k < SDF_TAP_BATCH; ++k) sdf_tap_load<AD>(S, tap[k], qt[k]);
#pragma unroll
    for (int k = 0; k < SDF_TAP_BATCH; ++k) { const bool o = sdf_tap_finish<AD>(tap[k], qt[k], sw[k0 + k]); ok = ok && o; }
#if SDF_TAP_BATCH < 6
    __builtin_amdgcn_sched_barrier(0);           // (else the scheduler hoists every batch's gathers to the top: 80 loads in flight and their registers spilled)
#endif
  }
  if (__builtin_expect(__any(off_line && ok), 0)) {
    // never taken for v_h = cell (a neighbour's cell is one to the side, give or take a rounding that keeps it on the line); kept so that the
    // result does not depend on that argument: the generic lookup, the reference's own order of tests
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float3 q = kf3(pw0.x, pw0.y, pw0.z);
      const float d = (k & 1) ? -v_h : v_h;
      if ((k >> 1) == 0) q.x += d; else if ((k >> 1) == 1) q.y += d; else q.z += d;
      const SdfCell nc = (k == 0) ? cxp : (k == 1) ? cxm : (k == 2) ? cyp : (k == 3) ? cym : (k == 4) ? czp : czm;
      const SdfCell bc = (k >> 1) == 0 ? cx : ((k >> 1) == 1 ? cy : cz);
      const int s_ = nc.g - (bc.g - 1);
      if (s_ < 0 || s_ > 2) {
        float dv = 0.f;
        const bool o = sdf_lookup_cold<AD>(S, q, r, rS, cell, rcell, dv);
        sv[k] = dv; ok = ok && o;
      }
    }
  }
  // :57-62 divide the differences by 2 w_h / 2 v_h; here by the reciprocals (tolerance side: one more rounding per entry)
  const float rw = 1.0f / (2 * w_h), rv = 1.0f / (2 * v_h);
  row[0] = (sw[0] - sw[1]) * rw; row[1] = (sw[2] - sw[3]) * rw; row[2] = (sw[4] - sw[5]) * rw;
  row[3] = (sv[0] - sv[1]) * rv; row[4] = (sv[2] - sv[3]) * rv; row[5] = (sv[4] - sv[5]) * rv;
  row[6] = sdf0;                                                                                                         // :63
  return ok;
}
