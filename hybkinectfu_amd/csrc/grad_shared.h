// grad_shared.h -- the six taps of gradientForPoint (src/cuda/raycastingVolume.cu:16-42) from ONE 32-voxel neighbourhood.
//
// The reference looks the volume up at vtx -+ one cell along each axis: six trilinear interpolations (tsdfVolume.h:98-122, :151-172) of
// 8 voxels each.  A tap along x has vtx's own y / z operands (vtx.y + 0.f), hence vtx's own y / z cells and fractions, and an x cell that
// lies one to the side of vtx's: the six taps' 48 voxels are 32 distinct ones -- the 2x2x2 cell of vtx extended by one voxel on either side
// along each axis -- and their 18 per-axis cell computations (two exact quotients each) are 9 distinct ones.
//
// What is shared is the ADDRESSING and the LOADS; each tap's value is the reference's own 8-term sum over its own 8 voxels and its own
// fractions, operation for operation (rc_trilinear), so the maps stay bit-identical.  "One to the side" is the typical outcome of the
// tap's own cell computation, not a premise: every tap's cell is computed the reference's way, and a wave in which some lane's tap lands
// elsewhere (a rounding at a cell boundary; not met once in 8 M random points at four volume geometries, so tests force it) takes the generic path -- see gradient_for_point in raycast.hip.
//
// The separable brick addressing follows sdf_rows.h (offset(x, y, z) = bx(x) + by(y) + bz(z)); sdf_rows.h is included for its 8-byte load type.
#pragma once
#include "kf_internal.h"
#include "sdf_rows.h"

// one axis of tsdfVolume.h:151-172: the range test on the unadjusted cell, the adjusted base cell, the fraction inside it (exact quotients)
struct RcCell { int g; float f; bool in; };
__device__ __forceinline__ RcCell rc_cell(float pos, float r, const KfRecip& rS, float cell, const KfRecip& rcell, int R) {
  RcCell c;
  const int g = kf_f2i(kf_div(pos * r, rS));
  c.in = !(g <= 0 || g >= R - 1);
  const int ga = (pos < ((float)g + 0.5f) * cell) ? g - 1 : g;
  c.g = ga;
  c.f = kf_div(pos - ((float)ga + 0.5f) * cell, rcell);
  return c;
}
// pins a value to this point of the program: the compiler may neither sink its computation into a later branch nor (with the sched_barrier that
// follows every use here) start the next round's gathers before it -- without it all 32 gathers go out together and the kernel needs 106 registers
__device__ __forceinline__ float rc_pin(float v) { asm volatile("" : "+v"(v)); return v; }
// tsdfVolume.h:113-120, the sum as written there (k = (dx << 2) | (dy << 1) | dz)
__device__ __forceinline__ float rc_trilinear(float q0, float q1, float q2, float q3, float q4, float q5, float q6, float q7, float a, float b, float c) {
  const float ia = 1 - a, ib = 1 - b, ic = 1 - c;
  return q0 * ia * ib * ic + q1 * ia * ib * c + q2 * ia * b * ic + q3 * ia * b * c + q4 * a * ib * ic + q5 * a * ib * c + q6 * a * b * ic + q7 * a * b * c;
}

struct RcGradCells { RcCell c0[3], cp[3], cm[3]; bool on_line; };
// the nine cells of the six taps; on_line: every tap's cell is vtx's cell moved by exactly one along the tap's axis
__device__ __forceinline__ RcGradCells rc_grad_cells(const KfVolume& v, float3 vtx, const KfRecip& rS, const KfRecip& rcell) {
  RcGradCells G;
  const float r = (float)v.res, cell = v.cell;
  const int R = v.res;
  const float p[3] = {vtx.x, vtx.y, vtx.z};
  G.on_line = true;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    G.c0[ax] = rc_cell(p[ax] + 0.f, r, rS, cell, rcell, R);
    G.cp[ax] = rc_cell(p[ax] + cell, r, rS, cell, rcell, R);
    G.cm[ax] = rc_cell(p[ax] + -cell, r, rS, cell, rcell, R);
    G.on_line = G.on_line && G.cp[ax].g == G.c0[ax].g + 1 && G.cm[ax].g == G.c0[ax].g - 1;
  }
  return G;
}

// Addressing: BYTE offsets modulo 2^32, separable -- off(x, y, z) = bx(x) + by(y) + bz(z) -- relative to a brick layer of the stored volume (the view's first:
// below), coordinates clamped into the stored volume.  Every component may wrap; the sum is right whenever the true offset is below 2^32, which the view test
// guarantees for every voxel a lane loads.  Any volume size (2048^3: 256 MB per brick layer, a view of 16 layers).
struct RcViewAddr {
  unsigned nb, nb2; int zlo, zhi, R, bz0;
  __device__ __forceinline__ explicit RcViewAddr(const KfVolume& v) {
    nb = (unsigned)v.nb; nb2 = nb * nb; zlo = v.bz0 * KF_BRICK; zhi = v.bz1 * KF_BRICK - 1; R = v.res; bz0 = v.bz0;
  }
  __device__ __forceinline__ unsigned bx(int x) const { x = min(max(x, 0), R - 1); return ((unsigned)(x >> 3) << 12) + ((unsigned)(x & 7) << 3); }
  __device__ __forceinline__ unsigned by(int y) const { y = min(max(y, 0), R - 1); return (kf_opaque(__umul24((unsigned)(y >> 3), nb)) << 12) + ((unsigned)(y & 7) << 6); }
  // z relative to brick layer `zb0` of the stored volume (the view's first layer)
  __device__ __forceinline__ unsigned bz(int z, int zb0) const { z = min(max(z, zlo), zhi); return (kf_opaque(__umul24((unsigned)((z >> 3) - bz0 - zb0), nb2)) << 12) + ((unsigned)(z & 7) << 9); }
  __device__ __forceinline__ int zbrick(int z) const { z = min(max(z, zlo), zhi); return (z >> 3) - bz0; }
};
// The loads go through a raw buffer descriptor (MI355X guide: __builtin_amdgcn_make_buffer_rsrc + raw_buffer_load: one 32-bit offset register per gather,
// no 64-bit address pairs) whose base is WAVE-UNIFORM: the stored volume may hold 8 GB, a 32-bit byte offset reaches 4 GB, but the 64 neighbourhoods of one
// 8x8-pixel wave lie within a few bricks of each other.  The view starts `half` brick layers (2 GB worth, at least one) below the first lane's
// neighbourhood and is 2 * half layers deep; a wave with a lane outside it takes the generic path (1024^3: a silhouette spanning more than a quarter of
// the volume in z inside one 8x8 patch, rare; 2048^3: 16 brick layers = 0.5 m at 8 m, i.e. the waves on depth discontinuities).
struct RcWaveView { __amdgpu_buffer_rsrc_t rsrc; int zb0; };
__device__ __forceinline__ int rc_view_half_layers(const KfVolume& v, int forced) {
  const unsigned long long layer = (unsigned long long)v.nb * v.nb * KF_BRICK_VOX * sizeof(float2);       // bytes per brick layer
  const unsigned long long h = (1ull << 31) / layer;                     // (>= 1 up to 5792 bricks per axis; kf_create stops at 1024)
  const int most = h < 1 ? 1 : (h > 4096 ? 4096 : (int)h);
  return forced > 0 && forced < most ? forced : most;                    // forced (KF_RAYCAST_VIEW_HALF, tests): a shallower view, so that small volumes meet its edges too
}
__device__ __forceinline__ RcWaveView rc_wave_view(const KfVolume& v, int zb0) {
  RcWaveView w; w.zb0 = zb0;
  const unsigned long long layer = (unsigned long long)v.nb * v.nb * KF_BRICK_VOX * sizeof(float2);
  const unsigned long long rem = (unsigned long long)(v.bz1 - v.bz0 - zb0) * layer;
  w.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)v.tw + (unsigned long long)zb0 * layer), 0, rem > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)rem, 0x00020000);
  return w;
}
__device__ __forceinline__ float2 rc_view_load(const RcWaveView& w, unsigned byte_off) {
  const sdf_v2u q = __builtin_amdgcn_raw_buffer_load_b64(w.rsrc, byte_off, 0, 0);
  return make_float2(__uint_as_float(q.x), __uint_as_float(q.y));
}

// The taps' values f[0..5] (+x, -x, +y, -y, +z, -z) for a lane whose cells are on the line and whose neighbourhood lies in the view; returns what the
// reference's six lookups return together (all in range, all layers stored, all 48 = 32 distinct weights non-zero).
// ROUNDS: 1 = the 32 gathers in one batch (kernels with registers to spare); 2 = the cell and the x line, then the y and z lines (16 + 16);
// 4 = cell, x line, y line, z line (8 each: the cell's eight values stay, every line's values go as soon as its two taps are formed -- the raycast's
// own launch runs under an 80-register cap).  Offsets are formed where they are used.
template <int ROUNDS>
__device__ __forceinline__ bool rc_grad_taps(const KfVolume& v, const RcGradCells& G, const RcWaveView& w, float f[6]) {
  static_assert(ROUNDS == 1 || ROUNDS == 2 || ROUNDS == 4, "");
  const RcViewAddr E(v);
  const int gx = G.c0[0].g, gy = G.c0[1].g, gz = G.c0[2].g;
  bool ok = gz - 1 >= E.zlo && gz + 2 <= E.zhi;                               // the four layers of the z line stored
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) ok = ok && G.c0[ax].in && G.cp[ax].in && G.cm[ax].in;
  const float a = G.c0[0].f, b = G.c0[1].f, c = G.c0[2].f;
  const unsigned x1[2] = {E.bx(gx), E.bx(gx + 1)}, y1[2] = {E.by(gy), E.by(gy + 1)}, z1[2] = {E.bz(gz, w.zb0), E.bz(gz + 1, w.zb0)};
  // core[i][j][k]: vtx's own cell (line positions 1, 2 of each axis); ex[e][j][k]: x position 0 / 3; ey[i][e][k]; ez[i][j][e]
  float2 core[2][2][2], ex[2][2][2], ey[2][2][2], ez[2][2][2];
  bool nz = true;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i) core[i][j][k] = rc_view_load(w, x1[i] + (y1[j] + z1[k]));
  if (ROUNDS == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { nz = nz && !(core[i >> 2][(i >> 1) & 1][i & 1].y == 0.f); core[i >> 2][(i >> 1) & 1][i & 1].x = rc_pin(core[i >> 2][(i >> 1) & 1][i & 1].x); }
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const unsigned xe[2] = {E.bx(gx - 1), E.bx(gx + 2)};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) ex[e][j][k] = rc_view_load(w, xe[e] + (y1[j] + z1[k]));
  }
  if (ROUNDS >= 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      nz = nz && !(ex[i >> 2][(i >> 1) & 1][i & 1].y == 0.f);
      if (ROUNDS == 2) nz = nz && !(core[i >> 2][(i >> 1) & 1][i & 1].y == 0.f);
    }
    f[0] = rc_trilinear(core[1][0][0].x, core[1][0][1].x, core[1][1][0].x, core[1][1][1].x, ex[1][0][0].x, ex[1][0][1].x, ex[1][1][0].x, ex[1][1][1].x, G.cp[0].f, b, c);
    f[1] = rc_trilinear(ex[0][0][0].x, ex[0][0][1].x, ex[0][1][0].x, ex[0][1][1].x, core[0][0][0].x, core[0][0][1].x, core[0][1][0].x, core[0][1][1].x, G.cm[0].f, b, c);
    f[0] = rc_pin(f[0]); f[1] = rc_pin(f[1]);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const unsigned ye[2] = {E.by(gy - 1), E.by(gy + 2)};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) ey[i][e][k] = rc_view_load(w, ye[e] + (x1[i] + z1[k]));
  }
  if (ROUNDS == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) nz = nz && !(ey[i >> 2][(i >> 1) & 1][i & 1].y == 0.f);
    f[2] = rc_trilinear(core[0][1][0].x, core[0][1][1].x, ey[0][1][0].x, ey[0][1][1].x, core[1][1][0].x, core[1][1][1].x, ey[1][1][0].x, ey[1][1][1].x, a, G.cp[1].f, c);
    f[3] = rc_trilinear(ey[0][0][0].x, ey[0][0][1].x, core[0][0][0].x, core[0][0][1].x, ey[1][0][0].x, ey[1][0][1].x, core[1][0][0].x, core[1][0][1].x, a, G.cm[1].f, c);
    f[2] = rc_pin(f[2]); f[3] = rc_pin(f[3]);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const unsigned ze[2] = {E.bz(gz - 1, w.zb0), E.bz(gz + 2, w.zb0)};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) ez[i][j][e] = rc_view_load(w, ze[e] + (x1[i] + y1[j]));
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    nz = nz && !(ez[i >> 2][(i >> 1) & 1][i & 1].y == 0.f);
    if (ROUNDS <= 2) nz = nz && !(ey[i >> 2][(i >> 1) & 1][i & 1].y == 0.f);
    if (ROUNDS == 1) nz = nz && !(core[i >> 2][(i >> 1) & 1][i & 1].y == 0.f) && !(ex[i >> 2][(i >> 1) & 1][i & 1].y == 0.f);
  }
  if (ROUNDS == 1) {
    f[0] = rc_trilinear(core[1][0][0].x, core[1][0][1].x, core[1][1][0].x, core[1][1][1].x, ex[1][0][0].x, ex[1][0][1].x, ex[1][1][0].x, ex[1][1][1].x, G.cp[0].f, b, c);
    f[1] = rc_trilinear(ex[0][0][0].x, ex[0][0][1].x, ex[0][1][0].x, ex[0][1][1].x, core[0][0][0].x, core[0][0][1].x, core[0][1][0].x, core[0][1][1].x, G.cm[0].f, b, c);
  }
  if (ROUNDS <= 2) {
    f[2] = rc_trilinear(core[0][1][0].x, core[0][1][1].x, ey[0][1][0].x, ey[0][1][1].x, core[1][1][0].x, core[1][1][1].x, ey[1][1][0].x, ey[1][1][1].x, a, G.cp[1].f, c);
    f[3] = rc_trilinear(ey[0][0][0].x, ey[0][0][1].x, core[0][0][0].x, core[0][0][1].x, ey[1][0][0].x, ey[1][0][1].x, core[1][0][0].x, core[1][0][1].x, a, G.cm[1].f, c);
  }
  f[4] = rc_trilinear(core[0][0][1].x, ez[0][0][1].x, core[0][1][1].x, ez[0][1][1].x, core[1][0][1].x, ez[1][0][1].x, core[1][1][1].x, ez[1][1][1].x, a, b, G.cp[2].f);
  f[5] = rc_trilinear(ez[0][0][0].x, core[0][0][0].x, ez[0][1][0].x, core[0][1][0].x, ez[1][0][0].x, core[1][0][0].x, ez[1][1][0].x, core[1][1][0].x, a, b, G.cm[2].f);
  return ok && nz;
}

// 0: the reference's verdict is false; 1: `grad` is the reference's gradient; 2: (wave-uniform) this wave must evaluate the generic way
// force_generic (uniform; tests only): answer 2 -- the outcome never seen with real data (no tap of 8 M random points left its line), so the suite forces it;
// view_half (tests only, 0 = as deep as 32-bit offsets allow): see rc_view_half_layers
template <int ROUNDS>
__device__ __forceinline__ int rc_gradient_shared(const KfVolume& v, float3 samplepos, float3 vtx, const KfRecip& rS, const KfRecip& rcell, bool force_generic, int view_half, float3& grad) {
  const float rf = (float)v.res;
  const int3 g = make_int3(kf_f2i(kf_div(samplepos.x * rf, rS)), kf_f2i(kf_div(samplepos.y * rf, rS)), kf_f2i(kf_div(samplepos.z * rf, rS)));
  const int R = v.res;
  const bool in_g = !(g.x <= 1 || g.x >= R - 2 || g.y <= 1 || g.y >= R - 2 || g.z <= 1 || g.z >= R - 2);      // raycastingVolume.cu:17-21, on the LAST sample's voxel
  const RcGradCells G = rc_grad_cells(v, vtx, rS, rcell);
  bool all_in = true;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) all_in = all_in && G.c0[ax].in && G.cp[ax].in && G.cm[ax].in;
  const bool wanted = in_g && all_in;                                 // otherwise the verdict is false whatever the voxels hold
  const bool fast = wanted && G.on_line;
  const RcViewAddr E(v);
  const int zb_lo = E.zbrick(G.c0[2].g - 1), zb_hi = E.zbrick(G.c0[2].g + 2);
  const unsigned long long fm = __ballot(fast);
  const int half = rc_view_half_layers(v, view_half);
  int zb0 = 0;
  if (fm) zb0 = max(__builtin_amdgcn_readlane(zb_lo, (int)(__ffsll((long long)fm) - 1)) - half, 0);
  const bool in_view = zb_lo >= zb0 && zb_hi < zb0 + 2 * half;
  if (__builtin_expect(__any(wanted && !(G.on_line && in_view)) || force_generic, 0)) return 2;
  if (!fast) return 0;
  const RcWaveView w = rc_wave_view(v, zb0);
  float f[6];
  if (!rc_grad_taps<ROUNDS>(v, G, w, f)) return 0;
  float3 n;
  n.x = f[0] - f[1]; n.y = f[2] - f[3]; n.z = f[4] - f[5];
  const float len = kf_norm(n);
  if ((double)len < 1e-8) return 0;
  grad = kf_scale(n, 1 / len);                                         // fp32 reciprocal (:40)
  return 1;
}
