// track.hip -- camera tracking: projective point-to-plane ICP and the direct SDF tracker, with the whole Gauss-Newton
// loop resident on the device.
//
// Reference: computeGbufKernel / findCorrs / buildPointToPlaneSolverRows (src/cuda/CalPointToPlaneErrSolverParams.cu:7-129),
// reduceGbufKernel (src/cuda/device_functions.h:17-55), computeSDFSolverbufKernel / buildSDFSolverRows
// (src/cuda/CalSDFErrSolverParams.cu:7-138), CameraPoseFinderICP (src/CameraPoseFinderICP.cpp:12-145),
// CameraPoseFinderSDF (src/CameraPoseFinderSDF.cpp:25-106), direct_exponential_map (src/utils/eigen_utils.cpp:60-127).
//
// gfx950 design.  The reference builds the 6x6 system with 27 sequential 256-wide shared-memory tree reductions
// (54 barriers) per launch, a second 27-block launch, a device sync and a 108-byte read-back, 19 times per frame; the
// loop is latency-bound.  Here a Gauss-Newton step is ONE launch and the host never sees the 27 floats:
//   * every lane keeps the 27 partial sums of its pixels in registers (no MFMA: a 6x6 outer product is far too small);
//     a wave64 __shfl_down tree folds them, LDS holds the four per-wave partials, 27 lanes write the workgroup partial;
//   * there is no second "reduce" launch, no atomic and no fence: the NEXT step's launch starts by having every workgroup
//     sum the previous step's workgroup partials (<= 512 x 27 floats, L2 resident) in the same fixed order and run the same
//     6x6 determinant test / Cholesky solve / shake test / pose composition on one lane, so all workgroups hold the
//     identical new transform (bitwise) and go straight on to their pixels.  Partials and the running transform are
//     double-buffered by step parity; the kernel boundary is the only synchronisation (1.5-2 us, MI355X_MICROARCH.md
//     'boundary') -- cheaper than a ticket on one contended word plus a release/acquire pair per step;
//   * a last 1-workgroup launch folds the final step and commits CameraPoseFinder::_pose.
// Sums are fp32 in a fixed (reproducible) association that differs from the reference's tree, so parity for this stage is
// by tolerance (pose within 1e-4 m / 1e-4 rad), as SURVEY.md section 8a row a6 states.
#include "kf_internal.h"
#include "cull.h"
#include "bilateral_tile.h"
#include "sdf_rows.h"
#include <string.h>
#include <stdlib.h>

#define TRK_THREADS 256
#define TRK_PX 2                   // 8x8 tiles per wave of the SDF step (grid-stride beyond that): the lookups want many waves, not long lanes

struct TrackArgs {
  const float4* new_v; const float4* new_n; const float4* model_v; const float4* model_n;
  KfCam cam;
  KfMat cur_val, linv_val;                       // host-supplied transforms (per-call wrappers only)
  int use_state;                                 // 1: transforms come from the device-resident KfTrackState
  float dist_thres, sin_thres, dist_shake, angle_shake;
  float cos_shake, dist_shake2;                  // cos(angle_shake), dist_shake^2: the shake test without acos / sqrt (set next to the two above)
  float dist_thres2, sin_thres2;                 // the correspondence gates' squares (gate_square: a negative threshold rejects everything, as `norm > t` does)
  float* partials;                               // 2 x KF_ICP_MAX_WG x 32 floats, indexed by step parity
  KfTrackState* track;
  int step;                                      // index of this Gauss-Newton step within the frame (buffer parity)
  int consume;                                   // 1: first fold + apply the previous step's system
  int n_prev_wg;                                 // workgroups that wrote the previous step's partials
  int fold_group;                                // > 0: fold them in groups g, g + fold_group, ... first (fold_partials: the batched loop's publishers)
  int sdf;                                       // consume with the SDF tracker's update rule (exp map, convergence test)
  // pixel-partitioned ICP (multi-GPU): this context sums only pixels [px_begin, px_end) (0,0 = all); the previous step's
  // system arrives all-reduced in ext_prev (27 floats) instead of workgroup partials; fold_out receives this step's 27 sums
  int px_begin, px_end;
  // replicated ICP, one launch per step: the persistent loop's pixel dealing (icp_level_geometry) -- px_l pixels per lane dealt in
  // 64-pixel chunks round robin over deal_grid workgroups -- so that both launch forms add the same numbers in the same order
  // (px_l == 0: a contiguous pixel range per workgroup, the pixel-partitioned form)
  int px_l, deal_grid;
  const float* ext_prev;
  float* fold_out;
  // SDF tracker only
  KfVolume vol; const float* depth;
  int slab_pixels;                               // 1: sum only pixels whose own voxel (pworld0) lies in the layers this context OWNS (z-slab partition)
#ifdef KF_EXPERIMENTS
  unsigned long long* dbg;                       // KF_ICP_EXP=11: where workgroup 0 accumulates the solve's sub-phase times (10 ns ticks)
  int exp_nodet;                                 // KF_ICP_EXP=12: skip the determinant (timing only)
#endif
};

// ---- 6x6 dense algebra on one lane (stands in for Eigen); every index is static after unrolling -> registers only ----
// src/CameraPoseFinderICP.cpp:119-136
__device__ __forceinline__ void unpack27(const float* in, float A[36], float b[6]) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 7; ++j) {
      float v = in[s++];
      if (j == 6) b[i] = v; else { A[i * 6 + j] = v; A[j * 6 + i] = v; }
    }
}
// determinant by partial-pivot LU (Eigen's path for a 6x6 `determinant()`), fp32; destroys m
__device__ __forceinline__ float det6(float m[36]) {
  float det = 1.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int p = k; float best = fabsf(m[k * 6 + k]);
#pragma unroll
    for (int r = k + 1; r < 6; ++r) { float v = fabsf(m[r * 6 + k]); if (v > best) { best = v; p = r; } }
    if (best == 0.f) return 0.f;
#pragma unroll
    for (int r = k + 1; r < 6; ++r)
      if (p == r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) { float t = m[k * 6 + c]; m[k * 6 + c] = m[r * 6 + c]; m[r * 6 + c] = t; }
        det = -det;
      }
    const float piv = m[k * 6 + k];
    det *= piv;
#ifndef KF_SOLVE_EXACT
    const float rp = __builtin_amdgcn_rcpf(piv);             // 1-ulp reciprocal, as in llt_solve6 below: the result only meets a threshold
#else
    const KfRecip rp = kf_recip(piv);                        // one reciprocal refinement per pivot, exact quotients (kf_div)
#endif
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
#ifndef KF_SOLVE_EXACT
      const float f = m[r * 6 + k] * rp;
#else
      const float f = kf_div(m[r * 6 + k], rp);
#endif
#pragma unroll
      for (int c = k + 1; c < 6; ++c) m[r * 6 + c] -= f * m[k * 6 + c];
    }
  }
  return det;
}
// x = LLT(A) \ b, fp32 (ICP.cpp:143, SDF.cpp:79: Eigen's llt().solve in float).  The solve is one chain of dependent operations on
// one lane (about 1 us of the 1.7 us a Gauss-Newton step spends between two pixel phases), so its square roots, reciprocals and
// quotients are the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 and a product with the reciprocal, not the correctly rounded forms:
// -7.5 us per frame.  What that costs in fidelity is below what is there anyway -- the 27 sums the solve starts from differ from
// the reference's in their last bits (fp32 sums in another association order, asserted <= 1e-5), the tracked pose is asserted to
// 1e-4 against an exactly rounded CPU restatement of Eigen's chain on identical inputs and stays there, and every
// workgroup, rank and context runs the same instructions, so poses stay bitwise reproducible.  -DKF_SOLVE_EXACT restores the
// correctly rounded chain.
__device__ __forceinline__ void llt_solve6(const float A[36], const float b[6], float x[6]) {
#ifndef KF_SOLVE_EXACT
  float L[36], rd[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float s = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
#ifndef KF_SOLVE_SQRT_RCP
    rd[j] = __builtin_amdgcn_rsqf(s);                        // 1 / L_jj in ONE transcendental (L_jj itself is never used: every use divides by it)
#else
    const float d = __builtin_amdgcn_sqrtf(s);
    L[j * 6 + j] = d;
    rd[j] = __builtin_amdgcn_rcpf(d);
#endif
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float t = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = t * rd[j];
    }
  }
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float t = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= L[i * 6 + k] * y[k];
    y[i] = t * rd[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float t = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) t -= L[k * 6 + i] * x[k];
    x[i] = t * rd[i];
  }
#else
  float L[36]; KfRecip rd[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float s = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
    const float d = sqrtf(s);
    L[j * 6 + j] = d;
    rd[j] = kf_recip(d);                                     // every division by L_jj (3 per column) shares this reciprocal
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float t = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = kf_div(t, rd[j]);
    }
  }
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float t = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= L[i * 6 + k] * y[k];
    y[i] = kf_div(t, rd[i]);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float t = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) t -= L[k * 6 + i] * x[k];
    x[i] = kf_div(t, rd[i]);
  }
#endif
}
// The same 6x6 system solved by a WAVE: lane i (< 6) holds row i of the symmetric matrix and b_i; a right-looking Cholesky -- column j is scaled by
// rsq(pivot), then every lane takes l_i * l_k off its trailing entries, l_k by v_readlane -- with the forward substitution carried along as a seventh
// column, then a column-oriented back substitution on wave-uniform values.  About 110 instructions on a dependent chain of six (readlane, rsq, product,
// fused update) links instead of ~430 instructions and two transcendentals per pivot on one lane: the Gauss-Newton step's solve phase was that chain.
// The pivots come for free, and with them the determinant the reference tests (ICP.cpp:138 `determinant() < 1e-10` on the float matrix): for the
// symmetric positive (semi-)definite J^T J it is the product of the Cholesky pivots, so the second 6x6 factorization (det6: partial-pivot LU on a lane of
// another wave, the longer of the two chains) is not run at all; a pivot <= 0 means the product has no business being positive: singular.  Same inputs
// in every workgroup / launch form -> same bits everywhere; against the reference's Eigen chain the increment agrees as before to the asserted 1e-4
// (1-ulp rsq and fused updates instead of sqrt / divide: DESIGN.md section 2.4).  -DKF_SOLVE_ONE_LANE restores the one-lane solve + LU determinant.
__device__ __forceinline__ void llt_solve6_wave(const float* s_tot, float x[6], bool& singular, float& det) {
  const int i = min((int)(threadIdx.x & 63u), 5);
  float a[6], bb;
#pragma unroll
  for (int j = 0; j < 6; ++j) { const int r = i < j ? i : j, c = i < j ? j : i; a[j] = s_tot[7 * r - (r * (r - 1)) / 2 + (c - r)]; }   // the packing of unpack27
  bb = s_tot[7 * i - (i * (i - 1)) / 2 + (6 - i)];
  float Lc[6], y[6], rd[6];
  bool bad = false; float prod = 1.f;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[j]), j));      // the pivot: row j's diagonal after the earlier columns' updates
    bad = bad || (sj <= 0.f);
    prod *= sj;
    const float r = __builtin_amdgcn_rsqf(sj);
    rd[j] = r;
    const float l = a[j] * r;                                                                  // lane i >= j: L[i][j]
    Lc[j] = l;
#pragma unroll
    for (int k = j + 1; k < 6; ++k) a[k] = __builtin_fmaf(-l, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), k)), a[k]);
    const float yj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb), j)) * r;     // forward substitution, as column 7
    y[j] = yj;
    bb = __builtin_fmaf(-l, yj, bb);
  }
#pragma unroll
  for (int k = 5; k >= 0; --k) {
    const float xk = y[k] * rd[k];
    x[k] = xk;
#pragma unroll
    for (int r = 0; r < k; ++r) y[r] = __builtin_fmaf(-__int_as_float(__builtin_amdgcn_readlane(__float_as_int(Lc[r]), k)), xk, y[r]);   // L[k][r] lives in lane k
  }
  singular = bad; det = prod;
}
__device__ __forceinline__ void mat3_mul(const float a[9], const float b[9], float o[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}
// Mat.h:240-262
__device__ __forceinline__ void mat44_mul(const float* a, const float* b, float* out) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r[i * 4 + j] = a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j] + a[i * 4 + 2] * b[8 + j] + a[i * 4 + 3] * b[12 + j];
#pragma unroll
  for (int i = 0; i < 16; ++i) out[i] = r[i];
}
// vector6ToTransformMatrix (ICP.cpp:95-111, SDF.cpp:25-43): R = Rx Ry Rz, shake test on the rotation angle and |t|
__device__ __forceinline__ bool transform_from_sincos(const float x[6], float c0, float s0, float c1, float s1, float c2, float s2,
                                                      float dist_shake2, float cos_shake, float t[16]) {
  // R = Rx Ry Rz with Rx = {1,0,0, 0,c0,-s0, 0,s0,c0}, Ry = {c1,0,s1, 0,1,0, -s1,0,c1}, Rz = {c2,-s2,0, s2,c2,0, 0,0,1} (ICP.cpp:98-100).
  // The general 3x3 products spend most of their 90 operations on the literal zeros and ones; written out, every element is the same
  // sum of the same products in the same order with the x * 0, x * 1 and + 0 terms dropped -- equal values for every finite input
  // (only the sign of an exact zero can differ, and a NaN angle still poisons R0 / R4 / R8, i.e. the shake verdict).
  const float xy3 = s0 * s1, xy5 = -s0 * c1, xy6 = -c0 * s1, xy8 = c0 * c1;            // Rx Ry = {c1, 0, s1,  xy3, c0, xy5,  xy6, s0, xy8}
  const float R[9] = {c1 * c2, c1 * -s2, s1,
                      xy3 * c2 + c0 * s2, xy3 * -s2 + c0 * c2, xy5,
                      xy6 * c2 + s0 * s2, xy6 * -s2 + s0 * c2, xy8};
  // the reference compares AngleAxisf(R).angle() = acos(ca) with angle_shake and |t| with dist_shake (ICP.cpp:101-107); acos is
  // decreasing and the square root increasing, so the same verdicts come from ca >= cos(angle_shake) and |t|^2 <= dist_shake^2
  // (thresholds prepared once per call) -- without an acos and a square root on the one-lane chain of every Gauss-Newton step
  float ca = (R[0] + R[4] + R[8] - 1.f) * 0.5f;
  ca = fminf(1.f, fmaxf(-1.f, ca));
  const float d2 = x[3] * x[3] + x[4] * x[4] + x[5] * x[5];
  if (!(ca >= cos_shake) || !(d2 <= dist_shake2)) return false;        // a NaN increment counts as shaking: never applied
  const float o[16] = {R[0], R[1], R[2], x[3], R[3], R[4], R[5], x[4], R[6], R[7], R[8], x[5], 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = o[i];
  return true;
}
// sine and cosine of an Euler-angle increment: the angles that survive the shake test are below angle_shake (0.3 rad stock), where
// the Taylor polynomials through x^9 / x^8 are exact to the last bit or two of fp32; larger arguments take the library routine
__device__ __forceinline__ void kf_sincos_small(float x, float* sn, float* cs) {
  if (fabsf(x) < 0.5f) {
    const float x2 = x * x;
    *sn = x * (1.f + x2 * (-1.f / 6.f + x2 * (1.f / 120.f + x2 * (-1.f / 5040.f + x2 * (1.f / 362880.f)))));
    *cs = 1.f + x2 * (-0.5f + x2 * (1.f / 24.f + x2 * (-1.f / 720.f + x2 * (1.f / 40320.f))));
  } else sincosf(x, sn, cs);
}
__device__ __forceinline__ bool vector6_to_transform(const float x[6], float dist_shake2, float cos_shake, float t[16]) {
  float c0, s0, c1, s1, c2, s2;
  kf_sincos_small(x[0], &s0, &c0); kf_sincos_small(x[1], &s1, &c1); kf_sincos_small(x[2], &s2, &c2);
  return transform_from_sincos(x, c0, s0, c1, s1, c2, s2, dist_shake2, cos_shake, t);
}

// The parts' sums of a fold (s_tot[part * 32 + k], all parts written, no barrier yet) -> s_tot[0..26]: four lanes per total add a quarter
// of the parts each, two DPP shifts combine them -- a chain of parts/4 + 2 dependent adds instead of parts, in ONE fixed order that the
// per-step fold (fold_partials) and the persistent loop's fold (fold_partials_tagged) share: the two launch forms of the ICP add the
// same numbers in the same order and therefore arrive at the same bits.  Needs blockDim.x >= 108.
__device__ __forceinline__ void fold_combine_parts(float* s_tot, int parts) {
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x < 27 * 4) {
    const int kk = threadIdx.x >> 2, quarter = threadIdx.x & 3, per = (parts + 3) >> 2;
    for (int p = quarter * per; p < min(parts, (quarter + 1) * per); ++p) t += s_tot[p * 32 + kk];
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x111, 0xf, 0xf, true));   // row_shr:1
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x112, 0xf, 0xf, true));   // row_shr:2
  }
  __syncthreads();
  if (threadIdx.x < 27 * 4 && (threadIdx.x & 3) == 3) s_tot[threadIdx.x >> 2] = t;
  __syncthreads();
}

// ---- fold the previous step's partial sums: identical order in every workgroup ------------------------------------------
// s_tot must hold 8 x 32 floats; on return s_tot[0..26] are the totals (all threads, after the barrier)
// (blockDim/32 interleaved chains per sum; the loads of a chain are issued four at a time so the chain costs
// ceil(n/4) memory round trips instead of n; the additions keep their fixed order.)  s_tot: 32 x 32 floats.
// n_group (0: none): workgroups g, g + n_group, g + 2 n_group, ... are first added up in that order and count as ONE publisher g -- what the batched
// persistent loop's resident workgroup g publishes after playing exactly those workgroups of the dealing (k_icp_loop_batched): the same numbers in the same order
__device__ __forceinline__ float fold_group_sum(const float* __restrict__ partials, int w, int n_wg, int n_group, int k) {
  float v = partials[w * 32 + k];
  if (n_group > 0) for (int u = w + n_group; u < n_wg; u += n_group) v += partials[u * 32 + k];
  return v;
}
__device__ __forceinline__ void fold_partials(const float* __restrict__ partials, int n_wg, float* s_tot, int n_group = 0) {
  const int k = threadIdx.x & 31, part = threadIdx.x >> 5, parts = blockDim.x >> 5;
  const int n_pub = (n_group > 0 && n_group < n_wg) ? n_group : n_wg;
  if (n_pub == n_wg) n_group = 0;
  float s = 0.f;
  if (k < 27) {
    for (int w = part; w < n_pub; w += 4 * parts) {
      const int w1 = w + parts, w2 = w + 2 * parts, w3 = w + 3 * parts;
      const float v0 = fold_group_sum(partials, w, n_wg, n_group, k);
      const float v1 = (w1 < n_pub) ? fold_group_sum(partials, w1, n_wg, n_group, k) : 0.f;
      const float v2 = (w2 < n_pub) ? fold_group_sum(partials, w2, n_wg, n_group, k) : 0.f;
      const float v3 = (w3 < n_pub) ? fold_group_sum(partials, w3, n_wg, n_group, k) : 0.f;
      s += v0; s += v1; s += v2; s += v3;
    }
  }
  s_tot[part * 32 + k] = s;
  fold_combine_parts(s_tot, parts);
}

// direct_exponential_map (eigen_utils.cpp:84-127) in double, then SDF.cpp:92-100: R' = R_exp^T R_cur, t' = t_cur - R_exp^T t_exp
__device__ __forceinline__ void sdf_apply_increment(const float x[6], const float* cur, float ncur[16]) {
  const double u0 = (double)x[0], u1 = (double)x[1], u2 = (double)x[2], t3 = (double)x[3], t4 = (double)x[4], t5 = (double)x[5];
  const double theta = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
  const double si = sin(theta), co = cos(theta);
  const double sinc = fabs(theta) < 1.0e-8 ? 1.0 : si / theta;
  const double mcosc = fabs(theta) < 2.5e-4 ? 0.5 : (1.0 - co) / theta / theta;
  const double msinc = fabs(theta) < 2.5e-4 ? (1. / 6.0) : (1.0 - si / theta) / theta / theta;
  double R[9];
  R[0] = co + mcosc * u0 * u0;         R[1] = -sinc * u2 + mcosc * u0 * u1; R[2] = sinc * u1 + mcosc * u0 * u2;
  R[3] = sinc * u2 + mcosc * u1 * u0;  R[4] = co + mcosc * u1 * u1;         R[5] = -sinc * u0 + mcosc * u1 * u2;
  R[6] = -sinc * u1 + mcosc * u2 * u0; R[7] = sinc * u0 + mcosc * u2 * u1;  R[8] = co + mcosc * u2 * u2;
  double dt[3];
  dt[0] = t3 * (sinc + u0 * u0 * msinc) + t4 * (u0 * u1 * msinc - u2 * mcosc) + t5 * (u0 * u2 * msinc + u1 * mcosc);
  dt[1] = t3 * (u0 * u1 * msinc + u2 * mcosc) + t4 * (sinc + u1 * u1 * msinc) + t5 * (u1 * u2 * msinc - u0 * mcosc);
  dt[2] = t3 * (u0 * u2 * msinc - u1 * mcosc) + t4 * (u1 * u2 * msinc + u0 * mcosc) + t5 * (sinc + u2 * u2 * msinc);
  float Rt[9], tf[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = (float)R[c * 3 + r];
#pragma unroll
  for (int k = 0; k < 3; ++k) tf[k] = (float)dt[k];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ncur[r * 4 + c] = Rt[r * 3] * cur[c] + Rt[r * 3 + 1] * cur[4 + c] + Rt[r * 3 + 2] * cur[8 + c];
    ncur[r * 4 + 3] = cur[r * 4 + 3] - (Rt[r * 3] * tf[0] + Rt[r * 3 + 1] * tf[1] + Rt[r * 3 + 2] * tf[2]);
  }
  ncur[12] = 0.f; ncur[13] = 0.f; ncur[14] = 0.f; ncur[15] = 1.f;
}

enum { STEP_APPLIED = 0, STEP_LOST_DET = 1, STEP_LOST_SHAKE = 2, STEP_CONVERGED = 3 };

// Apply one Gauss-Newton update from the folded sums in s_tot to s_cur (in place).  Runs on lane 0 of every workgroup with
// identical inputs -> identical outputs.  Returns STEP_* through *s_code (LDS).
//   ICP: minimizePointToPlaneErrFunc (ICP.cpp:117-143) + loop body of estimateCameraPose (:71-82)
//   SDF: loop body of CameraPoseFinderSDF::estimateCameraPose (SDF.cpp:62-101)
// s_next (ICP only, may be null): write the new transform THERE instead of over s_cur and return after ONE workgroup barrier -- the caller flips
// between two buffers (the persistent loop: the copy back and its second barrier were 0.33 us of every Gauss-Newton step); the returned code
// is the one *s_code would carry.
__device__ __forceinline__ int apply_step(const TrackArgs& a, const float* s_tot, float* s_cur, int* s_code, float* s_next = nullptr) {
  if (!a.sdf) {
    // ICP: the determinant test (ICP.cpp:138) and the solve + increment (:143, :71-82) do not depend on each other, so two
    // lanes of DIFFERENT waves run them side by side and lane 0 keeps the solve's result only if the determinant passed.
    // LDS scratch behind the 27 totals (s_tot + 32): [0..15] candidate transform, [16] solve verdict, [17] det verdict.
    float* scratch = const_cast<float*>(s_tot) + 32;
    const unsigned det_lane = blockDim.x > 64u ? 64u : 1u;
#ifdef KF_EXPERIMENTS
#define KF_SOLVE_STAMP(i) do { if (a.dbg) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(a.dbg + (i), t_ - t_prev_); t_prev_ = t_; } } while (0)
    unsigned long long t_prev_ = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
#else
#define KF_SOLVE_STAMP(i) do { } while (0)
#endif
#ifdef KF_SOLVE_ONE_LANE
    if (threadIdx.x == det_lane) {
      float A[36], b[6];
      unpack27(s_tot, A, b);
#ifdef KF_EXPERIMENTS
      if (a.exp_nodet) scratch[17] = 0.f; else            // timing experiment (KF_ICP_EXP=12): what does the determinant lane cost the step?
#endif
      scratch[17] = ((double)det6(A) < 1E-10) ? 1.f : 0.f;
      KF_SOLVE_STAMP(6);                                                     // determinant lane: entry -> done
    }
#else
    (void)det_lane;
#endif
    // wave 0 runs the solve, the trigonometry and the increment as ONE chain without touching LDS in between: lane 0 solves; the increment
    // goes to the whole wave as scalars (v_readfirstlane: lane 0 is the wave's first lane); lanes 0-2 take the sine / cosine of one angle
    // each and hand them back the same way (v_readlane); then EVERY lane forms the rotation and the shake verdict from those scalars -- the
    // same expressions, hence the same bits, in all of them -- and twelve lanes compute one element each of T * cur (ICP.cpp:81) instead of
    // lane 0 computing all twelve.  (A third of the instructions of the LDS hand-off form it replaces and no wavefront fences -- and the same
    // 1.2 us per Gauss-Newton step: the phase is the solve's dependent chain plus two workgroup barriers, not instruction issue.)  The
    // determinant (the longer of the two 6x6 factorizations, 1.1 us) runs on a lane of another wave and is joined at the workgroup barrier
    // below; it holds the step back by 0.15 us (KF_ICP_EXP=12 on the experiments build skips it: 145.5 -> 142.7 us per frame).
    if (threadIdx.x < 64) {
      float* vs = scratch;
      float x[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef KF_SOLVE_ONE_LANE
      if (threadIdx.x == 0) {
        float A[36], b[6];
        unpack27(s_tot, A, b);
        llt_solve6(A, b, x);
        KF_SOLVE_STAMP(0);                                                   // unpack + Cholesky solve
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) x[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x[i])));
#else
      {
        bool singular; float det;
        llt_solve6_wave(s_tot, x, singular, det);                            // the whole wave: the increment is wave-uniform, the pivots give the determinant
        if (threadIdx.x == 0) { vs[17] = (singular || (double)det < 1E-10) ? 1.f : 0.f; KF_SOLVE_STAMP(0); }
#ifdef KF_EXPERIMENTS
        if (a.exp_nodet && threadIdx.x == 0) vs[17] = 0.f;
#endif
      }
#endif
      const float ang = threadIdx.x == 0 ? x[0] : (threadIdx.x == 1 ? x[1] : x[2]);
      float sn, cs; kf_sincos_small(ang, &sn, &cs);
      const int sni = __float_as_int(sn), csi = __float_as_int(cs);
      const float c0 = __int_as_float(__builtin_amdgcn_readlane(csi, 0)), s0 = __int_as_float(__builtin_amdgcn_readlane(sni, 0));
      const float c1 = __int_as_float(__builtin_amdgcn_readlane(csi, 1)), s1 = __int_as_float(__builtin_amdgcn_readlane(sni, 1));
      const float c2 = __int_as_float(__builtin_amdgcn_readlane(csi, 2)), s2 = __int_as_float(__builtin_amdgcn_readlane(sni, 2));
      if (threadIdx.x == 0) KF_SOLVE_STAMP(1);                               // increment broadcast + sin/cos + hand-back
      float T[16];
      const bool still = transform_from_sincos(x, c0, s0, c1, s1, c2, s2, a.dist_shake2, a.cos_shake, T);
      if (threadIdx.x == 0) vs[16] = still ? 0.f : 1.f;
      if (still && threadIdx.x < 16) {
        // ICP.cpp:81 cur = T * cur; T's last row is (0, 0, 0, 1): that row of the product is cur's own last row, bit for bit
        const int i = (int)threadIdx.x >> 2, j = (int)threadIdx.x & 3;
        const float t0 = i == 0 ? T[0] : (i == 1 ? T[4] : T[8]), t1 = i == 0 ? T[1] : (i == 1 ? T[5] : T[9]);
        const float t2 = i == 0 ? T[2] : (i == 1 ? T[6] : T[10]), t3 = i == 0 ? T[3] : (i == 1 ? T[7] : T[11]);
        const float prod = t0 * s_cur[j] + t1 * s_cur[4 + j] + t2 * s_cur[8 + j] + t3 * s_cur[12 + j];
        (s_next ? s_next : vs)[threadIdx.x] = i < 3 ? prod : s_cur[12 + j];
      }
      if (threadIdx.x == 0) KF_SOLVE_STAMP(2);                               // rotation, shake test, T * cur
    }
    __syncthreads();
    if (threadIdx.x == 0) KF_SOLVE_STAMP(3);                                 // workgroup barrier (waits for the determinant lane)
    {
      int code = STEP_APPLIED;                                             // every lane reads the two verdicts (LDS broadcast); sixteen lanes copy
      if (scratch[17] != 0.f) code = STEP_LOST_DET;
      else if (scratch[16] != 0.f) code = STEP_LOST_SHAKE;
      if (s_next) return code;                                             // ping-pong form: every lane holds the verdict, the transform lies in s_next
      if (code == STEP_APPLIED && threadIdx.x < 16) s_cur[threadIdx.x] = scratch[threadIdx.x];
      if (threadIdx.x == 0) *s_code = code;
    }
    __syncthreads();
    if (threadIdx.x == 0) KF_SOLVE_STAMP(4);                                 // verdict + copy + barrier
    return *s_code;
  }
  if (threadIdx.x == 0) {
    float A[36], b[6], x[6], T[16], ncur[16];
    unpack27(s_tot, A, b);
    int code = STEP_APPLIED;
    llt_solve6(A, b, x);
    if (!vector6_to_transform(x, a.dist_shake2, a.cos_shake, T)) code = STEP_LOST_SHAKE;
    else {
      const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
      if (nx < 0.001f) code = STEP_CONVERGED;                              // SDF.cpp:87-90: stop before applying x
      else { sdf_apply_increment(x, s_cur, ncur); for (int i = 0; i < 16; ++i) s_cur[i] = ncur[i]; }
    }
    *s_code = code;
  }
  __syncthreads();
  return *s_code;
}

// Common prologue of a step launch: load the running transform, fold + apply the previous step if asked.
// Returns false when the loop has ended (lost / converged); workgroup 0 records that in the track state.
__device__ __forceinline__ bool step_prologue(const TrackArgs& a, float* s_cur, float* s_linv, float* s_tot, int* s_code) {
  KfTrackState* st = a.track;
  if (a.use_state) {
    if (st->status != KF_TRACK_OK || st->converged) return false;          // uniform: written by an earlier launch
    if (threadIdx.x < 16) s_cur[threadIdx.x] = st->cur[a.step & 1][threadIdx.x];
    else if (threadIdx.x < 32) s_linv[threadIdx.x - 16] = st->last_inv[threadIdx.x - 16];
  } else {
    if (threadIdx.x < 16) s_cur[threadIdx.x] = a.cur_val.m[threadIdx.x];
    else if (threadIdx.x < 32) s_linv[threadIdx.x - 16] = a.linv_val.m[threadIdx.x - 16];
  }
  __syncthreads();
  if (!a.consume) {
    if (a.use_state && blockIdx.x == 0 && threadIdx.x < 16) st->cur[(a.step + 1) & 1][threadIdx.x] = s_cur[threadIdx.x];
    return true;
  }
  if (a.ext_prev) fold_partials(a.ext_prev, 1, s_tot);
  else fold_partials(a.partials + (size_t)((a.step + 1) & 1) * KF_ICP_MAX_WG * 32, a.n_prev_wg, s_tot, a.fold_group);
  apply_step(a, s_tot, s_cur, s_code);
  const int code = *s_code;
  if (blockIdx.x == 0) {
    if (threadIdx.x < 27) st->reduced[threadIdx.x] = s_tot[threadIdx.x];
    if (code == STEP_APPLIED) {
      if (threadIdx.x < 16) st->cur[(a.step + 1) & 1][threadIdx.x] = s_cur[threadIdx.x];
      if (threadIdx.x == 0) st->iterations += 1;
    } else if (threadIdx.x == 0) {
      if (code == STEP_CONVERGED) { st->converged = 1; for (int i = 0; i < 16; ++i) st->pose[i] = s_cur[i]; kf_mat44_inverse(s_cur, st->pose_inv); st->tracked = 1; }
      else { st->status = code; st->tracked = 0; }                          // KF_TRACK_LOST_* share the STEP_LOST_* values
    }
  }
  return code == STEP_APPLIED;
}

// workgroup reduction of the 27 sums (DPP row totals -> LDS -> fixed-order adds), written as this workgroup's partial for the
// NEXT launch to fold.  s_rows: 27 x (blockDim / 16) floats.
__device__ __forceinline__ void store_partial(float acc[27], float* partials_out, float* s_rows) {
  const int rows = blockDim.x >> 4, row = threadIdx.x >> 4;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float s = kf_row_scan_sum(acc[k]);
    if ((threadIdx.x & 15) == 15) s_rows[k * rows + row] = s;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    float s = s_rows[threadIdx.x * rows];
    for (int w = 1; w < rows; ++w) s += s_rows[threadIdx.x * rows + w];
    partials_out[blockIdx.x * 32 + threadIdx.x] = s;
  }
}

// ---- ICP ------------------------------------------------------------------------------------------------------------
// findCorrs (:17-60) + buildPointToPlaneSolverRows (:7-16), split so the ICP_PX pixels of a lane overlap their memory round trips:
// stage A (icp_project) needs only the lane's own vertex/normal and yields the model-map index; stage B (icp_finish)
// consumes the gathered model vertex/normal.
__device__ __forceinline__ int icp_project(const TrackArgs& a, const float* cur, const float* linv, float4 iv, float4 in_,
                                           float4& vg, float4& ng) {
  if (kf_is_zero4(in_)) return -1;
  vg = kf_mat_vec(cur, iv);
  ng = kf_mat_vec(cur, in_);
  const float4 vcp = kf_mat_vec(linv, vg);
  const int2 sp = kf_project(kf3(vcp.x, vcp.y, vcp.z), a.cam);
  if (sp.x < 0 || sp.x >= a.cam.cols || sp.y < 0 || sp.y >= a.cam.rows) return -1;
  return sp.y * a.cam.cols + sp.x;
}
__device__ __forceinline__ bool icp_finish(const TrackArgs& a, float4 vg, float4 ng, float4 vt, float4 nt, float row[7]) {
  if (kf_is_zero4(nt)) return false;
  // `norm(p - q) > dist || norm(n_tgt x n_in) > sin` (CalPointToPlaneErrSolverParams.cu:52) on the squares: two square roots fewer per pixel
  const float3 dv = kf3(vt.x - vg.x, vt.y - vg.y, vt.z - vg.z), cr = kf_cross(kf3(nt.x, nt.y, nt.z), kf3(ng.x, ng.y, ng.z));
  if (kf_dot(dv, dv) > a.dist_thres2 || kf_dot(cr, cr) > a.sin_thres2) return false;
  const float3 p = kf3(vt.x, vt.y, vt.z), q = kf3(vg.x, vg.y, vg.z), n = kf3(nt.x, nt.y, nt.z);
  row[0] = q.y * n.z - q.z * n.y; row[1] = q.z * n.x - q.x * n.z; row[2] = q.x * n.y - q.y * n.x;
  row[3] = n.x; row[4] = n.y; row[5] = n.z;
  row[6] = kf_dot(n, kf_sub(p, q));
  return true;
}

// 512 lanes x up to ICP_PX (3) pixels per workgroup: 200 workgroups at VGA level 0 (few partials to fold, no register spills).
#define ICP_THREADS 512
#ifndef ICP_PX
#define ICP_PX 3
#endif
// Pixels per lane (px_l) and publishing workgroups (grid_l) of a pyramid level with npx pixels, under a launch geometry of loop_grid
// workgroups (= the workgroups level 0 needs at ICP_PX pixels per lane).  Two things pull: fewer pixels per lane shorten the pixel
// phase (a coarse step is mostly latency), but every workgroup that holds pixels is one more publisher the step has to wait for (all
// 256 CUs publishing at every level: +36 us per frame).  Measured at VGA on 200 workgroups, tracking stage in us for (level 0, 1, 2)
// pixels per lane, with the fold's two polls in flight: (3,3,1) 162.4, (3,2,2) 161.9, (3,2,1) 159.4, (3,1,1) 156.3 (with one poll in
// flight (3,2,1) was ahead: 172.6 vs 175.5).  Rule: the fewest pixels per lane the launch can hold.  Host and device, both launch forms.
__host__ __device__ static inline void icp_level_geometry(int npx, int loop_grid, int level, int& px_l, int& grid_l) {
  px_l = ICP_PX;
#ifndef KF_ICP_FIXED_PX
  for (int p = 1; p < ICP_PX; ++p) if ((npx + ICP_THREADS * p - 1) / (ICP_THREADS * p) <= loop_grid) { px_l = p; break; }
#ifdef KF_ICP_PX_L0                                                              // tuning overrides (tools/build_variant.sh)
  if (level == 0) px_l = KF_ICP_PX_L0;
#endif
#ifdef KF_ICP_PX_L1
  if (level == 1) px_l = KF_ICP_PX_L1;
#endif
#ifdef KF_ICP_PX_L2
  if (level == 2) px_l = KF_ICP_PX_L2;
#endif
#endif
  (void)level;
  grid_l = (npx + ICP_THREADS * px_l - 1) / (ICP_THREADS * px_l);
}
// index of this lane's j-th pixel: pixels are dealt to the grid_l workgroups in 64-pixel chunks, round robin -- every workgroup sees
// the same mix of surface and background, so they all reach the exchange of partial sums at about the same time (contiguous blocks
// did not)
__device__ __forceinline__ int icp_dealt_pixel_of(int j, int grid_l, int wg) {
  return (((int)(threadIdx.x >> 6) * grid_l + wg) + j * ((ICP_THREADS / 64) * grid_l)) * 64 + (int)(threadIdx.x & 63);
}
__device__ __forceinline__ int icp_dealt_pixel(int j, int grid_l) { return icp_dealt_pixel_of(j, grid_l, (int)blockIdx.x); }
// one Gauss-Newton step's pixel phase for this lane's (up to ICP_PX) pixels: correspondences + the 27 products, fused accumulation
__device__ __forceinline__ void icp_accumulate(const TrackArgs& a, const float* s_cur, const float* s_linv, const float4 iv[ICP_PX], const float4 in_[ICP_PX],
                                               const float4* __restrict__ model_v, const float4* __restrict__ model_n, float acc[27]) {
  float4 vg[ICP_PX], ng[ICP_PX], vt[ICP_PX], nt[ICP_PX]; int mi[ICP_PX];
#pragma unroll
  for (int j = 0; j < ICP_PX; ++j) mi[j] = icp_project(a, s_cur, s_linv, iv[j], in_[j], vg[j], ng[j]);
  // the model-map gathers of all pixels go out together: no branch around them (a pixel without a correspondence reads entry 0 and
  // is dropped below) -- behind a conditional load the compiler parks a register copy and with it a wait, and the ICP_PX gathers
  // became ICP_PX dependent round trips per Gauss-Newton step
#ifndef KF_ICP_COND_GATHER
#pragma unroll
  for (int j = 0; j < ICP_PX; ++j) { const int g = mi[j] >= 0 ? mi[j] : 0; nt[j] = model_n[g]; vt[j] = model_v[g]; }
#else
#pragma unroll
  for (int j = 0; j < ICP_PX; ++j) { nt[j] = make_float4(0.f, 0.f, 0.f, 0.f); vt[j] = nt[j]; if (mi[j] >= 0) { nt[j] = model_n[mi[j]]; vt[j] = model_v[mi[j]]; } }
#endif
#pragma unroll
  for (int j = 0; j < ICP_PX; ++j) {
    float row[7];
    if (mi[j] < 0 || !icp_finish(a, vg[j], ng[j], vt[j], nt[j], row)) continue;
    int s = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 7; ++c) { acc[s] = __builtin_fmaf(row[r], row[c], acc[s]); ++s; }   // :92-105 packing; fused multiply-add: half the instructions, the better rounding (the sums are tolerance-checked)
  }
}
// workgroup total of the 27 sums (512 lanes): 16-lane row totals by DPP, one LDS word per (sum, row), then 4 lanes per sum add 8 row
// totals each and two DPP shifts combine the four partial chains (fixed order).  Returns true on the one lane that ends up holding
// sum k in sw.  s_wave: 27 x 32 floats.
__device__ __forceinline__ bool icp_wg_reduce(float acc[27], float* s_wave, int& k, float& sw) {
  const int row = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = kf_row_scan_sum(acc[i]);      // 27 independent DPP chains, free to interleave
  if ((threadIdx.x & 15) == 15) {
#pragma unroll
    for (int i = 0; i < 27; ++i) s_wave[i * (ICP_THREADS / 16) + row] = acc[i];
  }
  __syncthreads();
  k = threadIdx.x >> 2; sw = 0.f;
  if (threadIdx.x < 27 * 4) {
    const int part = threadIdx.x & 3;
    sw = s_wave[k * (ICP_THREADS / 16) + part * 8];
#pragma unroll
    for (int w = 1; w < 8; ++w) sw += s_wave[k * (ICP_THREADS / 16) + part * 8 + w];
    sw += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sw), 0x111, 0xf, 0xf, true));   // row_shr:1
    sw += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sw), 0x112, 0xf, 0xf, true));   // row_shr:2
    return part == 3;
  }
  return false;
}
__global__ void __launch_bounds__(ICP_THREADS) k_icp_step(TrackArgs a) {
  __shared__ float s_cur[16], s_linv[16];
  __shared__ float s_wave[27 * (ICP_THREADS / 16)], s_tot[(ICP_THREADS / 32) * 32];
  __shared__ int s_code;
  // the lane's own vertices / normals do not depend on the running transform: request them before the fold + solve
  const int npx = a.px_end > 0 ? a.px_end : a.cam.cols * a.cam.rows;
  float4 iv[ICP_PX], in_[ICP_PX];
#pragma unroll
  for (int j = 0; j < ICP_PX; ++j) {
    // px_l > 0: the persistent loop's dealing (same pixels in the same lanes -> same bits); else a contiguous range per workgroup
    const int i = a.px_l > 0 ? icp_dealt_pixel(j, a.deal_grid) : a.px_begin + (int)blockIdx.x * (ICP_THREADS * ICP_PX) + (int)threadIdx.x + j * ICP_THREADS;
    iv[j] = make_float4(0.f, 0.f, 0.f, 0.f); in_[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((a.px_l == 0 || j < a.px_l) && i < npx) { iv[j] = a.new_v[i]; in_[j] = a.new_n[i]; }
  }
  if (!step_prologue(a, s_cur, s_linv, s_tot, &s_code)) return;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  icp_accumulate(a, s_cur, s_linv, iv, in_, a.model_v, a.model_n, acc);
  int k; float sw;
  if (icp_wg_reduce(acc, s_wave, k, sw)) a.partials[(size_t)(a.step & 1) * KF_ICP_MAX_WG * 32 + blockIdx.x * 32 + k] = sw;
}

// ---- persistent ICP: the whole 19-step loop in ONE launch ---------------------------------------------------------------------
// One 512-lane workgroup per CU (no scratch at <= 256 VGPRs), 150 of them at VGA, all co-resident.  Between two Gauss-Newton
// steps nothing but the 27 partial sums of each workgroup has to cross CUs, so there is no grid barrier: a workgroup
// publishes its sums as tagged 64-bit words (one write-through store each) and every workgroup folds all of them, polling
// until the tags are current (fold_partials_tagged below; MI355X_MICROARCH.md "Valid forms": sc1 store -> sc1 load hand-off).
// A step then costs one store->load propagation (~1 us) instead of a dependent dispatch (~5 us measured here) plus a cold
// start of the 30 KB straight-line step code, which is what bounded the one-launch-per-step form at ~22 us per step; the
// two-level counter barrier this replaces cost three dependent atomic round trips (~4 us) per step.
struct IcpLoopArgs {
  const float4* new_v[KF_MAX_LEVELS]; const float4* new_n[KF_MAX_LEVELS];
  const float4* model_v[KF_MAX_LEVELS]; const float4* model_n[KF_MAX_LEVELS];
  KfCam cam[KF_MAX_LEVELS];
  int iters[KF_MAX_LEVELS]; int levels;
  float dist_thres, sin_thres, dist_shake, angle_shake, cos_shake, dist_shake2, dist_thres2, sin_thres2;
  unsigned long long* slots;                     // KF_ICP_LOOP_STEPS x KF_ICP_LOOP_MAX_WG x 32 tagged partial sums, one array per step
  unsigned tag_base;                             // this launch's sequence number (host counter x 64); tag = tag_base + step
  KfTrackState* track;
  unsigned* stall_word;                          // pinned host word: set when the loop gives up waiting, polled by the next kf_icp_track
  int exp_mode;                                  // diagnostics only (KF_ICP_EXP): 1 = skip the solve (timing), 7 = shader-clock stamps per phase
  int play_dead;                                 // fault injection (kf_inject_track_stall): this workgroup exits at once, as if it had never become resident
  // Riders: workgroups [n_loop, gridDim.x) do not take part in the loop -- they carry the NEXT frame's u16 -> f32 + gate + bilateral filter
  // (kf_prefetch_frame, requested before kf_icp_track).  The loop keeps one workgroup on each of ~200 CUs busy for ~0.14 ms at two waves per
  // SIMD and leaves the other CUs idle; the riders are dispatched behind the loop's workgroups (they cannot displace them), wait for
  // nobody, and are long done when the loop ends.  gridDim.x == n_loop: none.
  int n_loop;
  int n_virtual;                                 // workgroups the pixel dealing is defined for (= the workgroups level 0 needs at ICP_PX pixels per lane, what the per-step form
                                                 // launches): n_loop in k_icp_loop; larger in k_icp_loop_batched, whose n_loop resident workgroups each play several of them
  KfBilateralArgs bil; int bil_gx, bil_tiles, bil_fast;
  // The fusion pass's brick cull as the TAIL of this launch (cull.h): it needs nothing but the committed pose, which every workgroup of the loop holds
  // once the last step is applied -- so k_integrate_cull and one kernel boundary leave the stream (k_icp_loop only; the host arms it when the previous
  // kf_integrate_volume left its parameters behind and the volume is small enough: kf_icp_track).
  int cull_on;
  IntegrateArgs cull;
};

// A workgroup gives up waiting for the others' partial sums after ICP_WAIT_LIMIT ticks of the 100 MHz wall clock (s_memrealtime) = 20 ms -- a
// legitimate wait is tens of microseconds; the clock is read every 64th poll (~1-1.5 us each).  Giving up costs latency, not the frame: one
// workgroup then finishes the frame's Gauss-Newton loop alone (icp_solo_finish below), with the same bits.
#define ICP_WAIT_LIMIT 2000000ull
#define ICP_FOLD_BATCH 13
#ifndef KF_ICP_POLL_PIPE
#define KF_ICP_POLL_PIPE 2        // s_sleep between the two polls kept in flight (0 is not a value: -DKF_ICP_POLL_SINGLE selects the one-poll loop)
#endif
// Partial sums of the persistent loop travel as 64-bit (value, tag) words: the tag is the launch's sequence number plus the
// Gauss-Newton step, every step has its own slot array, and a word is published by ONE 8-byte write-through store -- so a
// reader needs no barrier and no flag: it polls the words it is about to add until their tags are current.  The adds run
// in a fixed order (workgroup-major within a part, then the parts): every workgroup of the loop arrives at the same bits.  The
// per-step launch form (k_icp_step) deals pixels to workgroups differently, so the two forms agree to tolerance, not bitwise.
__device__ __forceinline__ void fold_partials_tagged(const unsigned long long* slots, int n_wg, unsigned tag, float* s_tot, int* s_abort, int exp_mode = 0,
                                                     const unsigned* rescue_tag = nullptr, unsigned tag_base = 0u) {
  const int k = threadIdx.x & 31, part = threadIdx.x >> 5, parts = blockDim.x >> 5;
  float s = 0.f;
  unsigned long long t_wait0 = 0ull;
  if (k < 27) {
    for (int w0 = part; w0 < n_wg; w0 += ICP_FOLD_BATCH * parts) {
      float v[ICP_FOLD_BATCH];
      unsigned spins = 0;
#ifndef KF_ICP_POLL_SINGLE
      // several polls in flight: a poll costs a full memory round trip, and words that become current just after a poll has been issued
      // are otherwise only seen one round trip + one sleep later; the oldest set is checked while the younger one is on its way
      // (tracking stage 172.5 -> 157.5 us; three or four sets in flight spill registers: 222 / 369 us)
#ifndef KF_ICP_POLL_DEPTH
#define KF_ICP_POLL_DEPTH 2
#endif
      // the batch's addresses are fixed before the loop and a lane without a publisher for slot j polls its own FIRST word once more
      // (not one shared word: 170 of a coarse level's 208 slots are of this kind and would all hammer one publisher's line; the value is dropped below): no branch around a load, no exec-mask juggling per slot --
      // the poll iteration was 265 instructions, i.e. the loop noticed a publication up to 0.4 us late
      unsigned long long u[KF_ICP_POLL_DEPTH][ICP_FOLD_BATCH];
      const unsigned long long* src[ICP_FOLD_BATCH];
#pragma unroll
      for (int j = 0; j < ICP_FOLD_BATCH; ++j) { const int w = w0 + j * parts; src[j] = slots + (size_t)((w < n_wg ? w : w0) * 32 + k); }
#pragma unroll
      for (int d = 0; d < KF_ICP_POLL_DEPTH; ++d) {
        if (d) __builtin_amdgcn_s_sleep(KF_ICP_POLL_PIPE);
#pragma unroll
        for (int j = 0; j < ICP_FOLD_BATCH; ++j) u[d][j] = __hip_atomic_load(src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (;;) {
        unsigned stale = 0u;
#pragma unroll
        for (int j = 0; j < ICP_FOLD_BATCH; ++j) stale |= (unsigned)(u[0][j] >> 32) ^ tag;
        if (stale == 0u) break;
        if ((++spins & 63u) == 0u) {
          // (somebody else of this launch already gave up and one workgroup is finishing the frame alone: nothing more will be published)
          if (rescue_tag && __hip_atomic_load(rescue_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag_base) { *s_abort = 1; break; }
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (t_wait0 == 0ull) t_wait0 = now; else if (now - t_wait0 > ICP_WAIT_LIMIT) { *s_abort = 1; break; }
        }
#pragma unroll
        for (int d = 0; d + 1 < KF_ICP_POLL_DEPTH; ++d)
#pragma unroll
          for (int j = 0; j < ICP_FOLD_BATCH; ++j) u[d][j] = u[d + 1][j];
        __builtin_amdgcn_s_sleep(KF_ICP_POLL_PIPE);
#pragma unroll
        for (int j = 0; j < ICP_FOLD_BATCH; ++j) u[KF_ICP_POLL_DEPTH - 1][j] = __hip_atomic_load(src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int j = 0; j < ICP_FOLD_BATCH; ++j) v[j] = (w0 + j * parts < n_wg) ? __uint_as_float((unsigned)u[0][j]) : 0.f;
#else
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < ICP_FOLD_BATCH; ++j) {
          const int w = w0 + j * parts;
          v[j] = 0.f;
          if (w < n_wg && !(exp_mode == 3 && w >= 16 && k != 0)) {     // (experiment 3: wait for every workgroup, move 1/27 of the words)
            const unsigned long long u = __hip_atomic_load(&slots[w * 32 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[j] = __uint_as_float((unsigned)u);
            ok = ok && (unsigned)(u >> 32) == tag;
          }
        }
        if (ok) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 63u) == 0u) {                                             // never spin forever
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (t_wait0 == 0ull) t_wait0 = now; else if (now - t_wait0 > ICP_WAIT_LIMIT) { *s_abort = 1; break; }
        }
      }
#endif
#pragma unroll
      for (int j = 0; j < ICP_FOLD_BATCH; ++j) s += v[j];
    }
  }
  s_tot[part * 32 + k] = s;
  fold_combine_parts(s_tot, parts);       // the parts' sums, in the order the per-step fold uses too
}


// ---- a loop that timed out is finished by ONE workgroup ---------------------------------------------------------------------------------
// The loop's workgroups wait for each other, so they must all be resident; when some are not (a foreign process holds CUs -- the cross-process
// registry of ctx.hip only sees processes of this library) a fold times out.  The frame is NOT given up: the first workgroup to time out claims
// the launch (rescue_tag), everybody else leaves (workgroups dispatched later see the claim and leave at once), and the claimant runs the whole
// Gauss-Newton loop of this frame again by itself, playing every workgroup of the launch in turn: the same pixel dealing (icp_dealt_pixel_of),
// the same pixel phase (icp_accumulate), the same workgroup reduction (icp_wg_reduce), its partial sums published into the same tagged slots and
// folded by the same fold -- hence the SAME BITS as the loop or the per-step form would have produced (tests/test_gpu_track_forms.py injects the
// fault).  It costs milliseconds (one CU walks all pixels 19 times), not the frame; the host backs off to per-step launches afterwards.
// Not inlined: the hot kernel's register allocation stays what it was.
__device__ __forceinline__ void icp_solo_finish(const IcpLoopArgs& L, float (*s_pose)[16], float* s_linv, float* s_wave, float* s_tot, int* s_abort) {
  KfTrackState* st = L.track;
  int cur_buf = 0;
  float* s_cur = s_pose[0];
  if (threadIdx.x < 16) s_cur[threadIdx.x] = st->pose[threadIdx.x];            // from the committed pose again (ICP.cpp:62): nothing of the attempt is kept
  if (threadIdx.x == 0) *s_abort = 0;
  __syncthreads();
  if (threadIdx.x == 0) kf_mat44_inverse(s_cur, s_linv);
  __syncthreads();
  TrackArgs a;
  a.dist_thres = L.dist_thres; a.sin_thres = L.sin_thres; a.dist_shake = L.dist_shake; a.angle_shake = L.angle_shake; a.cos_shake = L.cos_shake; a.dist_shake2 = L.dist_shake2;
  a.dist_thres2 = L.dist_thres2; a.sin_thres2 = L.sin_thres2; a.sdf = 0;
#ifdef KF_EXPERIMENTS
  a.dbg = nullptr; a.exp_nodet = false;
#endif
  int step = 0, n_prev = 0, applied = 0, code = STEP_APPLIED;
  for (int l = L.levels - 1; l >= 0 && code == STEP_APPLIED; --l) {
    a.cam = L.cam[l];
    const float4* __restrict__ new_v = L.new_v[l]; const float4* __restrict__ new_n = L.new_n[l];
    const int npx = a.cam.cols * a.cam.rows;
    int px_l, grid_l;
    icp_level_geometry(npx, L.n_virtual, l, px_l, grid_l);
    for (int it = 0; it < L.iters[l]; ++it, ++step) {
      if (step > 0) {
        fold_partials_tagged(L.slots + (size_t)(step - 1) * KF_ICP_LOOP_MAX_WG * 32, n_prev, L.tag_base + (unsigned)(step - 1), s_tot, s_abort);
        code = apply_step(a, s_tot, s_cur, nullptr, s_pose[cur_buf ^ 1]);
        if (code != STEP_APPLIED) break;
        cur_buf ^= 1; s_cur = s_pose[cur_buf];
        ++applied;
      }
      const int n_pub = grid_l < L.n_loop ? grid_l : L.n_loop;                 // publishers: the launch's workgroups (k_icp_loop: one per workgroup of the dealing)
      for (int g = 0; g < n_pub; ++g) {                                        // every workgroup of the launch, one after the other
        float gsum = 0.f;
        for (int w = g; w < grid_l; w += L.n_loop) {                           // ... each with the workgroups of the dealing it plays (k_icp_loop_batched), summed in its order
          float4 iv[ICP_PX], in_[ICP_PX];
#pragma unroll
          for (int j = 0; j < ICP_PX; ++j) {
            const int i = icp_dealt_pixel_of(j, grid_l, w);
            iv[j] = make_float4(0.f, 0.f, 0.f, 0.f); in_[j] = iv[j];
            if (j < px_l && i < npx) { iv[j] = new_v[i]; in_[j] = new_n[i]; }
          }
          float acc[27];
#pragma unroll
          for (int k = 0; k < 27; ++k) acc[k] = 0.f;
          icp_accumulate(a, s_cur, s_linv, iv, in_, L.model_v[l], L.model_n[l], acc);
          int k; float sw;
          const bool holder = icp_wg_reduce(acc, s_wave, k, sw);
          if (holder) gsum = (w == g) ? sw : gsum + sw;
          __syncthreads();                                                     // s_wave is reused by the next turn
          if (holder && w + L.n_loop >= grid_l)
            __hip_atomic_store(L.slots + (size_t)step * KF_ICP_LOOP_MAX_WG * 32 + g * 32 + k,
                               ((unsigned long long)(L.tag_base + (unsigned)step) << 32) | (unsigned long long)__float_as_uint(gsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      n_prev = n_pub;
    }
  }
  if (code == STEP_APPLIED) {                                                  // the last step's system, then commit _pose (ICP.cpp:84)
    fold_partials_tagged(L.slots + (size_t)(step - 1) * KF_ICP_LOOP_MAX_WG * 32, n_prev, L.tag_base + (unsigned)(step - 1), s_tot, s_abort);
    code = apply_step(a, s_tot, s_cur, nullptr, s_pose[cur_buf ^ 1]);
    if (code == STEP_APPLIED) { cur_buf ^= 1; s_cur = s_pose[cur_buf]; }
    if (threadIdx.x < 27) st->reduced[threadIdx.x] = s_tot[threadIdx.x];
  }
  if (code != STEP_APPLIED) { if (threadIdx.x == 0) { st->status = code; st->tracked = 0; st->iterations = applied; st->converged = 0; st->rescued = 1; } return; }
  if (threadIdx.x < 16) st->pose[threadIdx.x] = s_cur[threadIdx.x];
  if (threadIdx.x == 0) { st->status = KF_TRACK_OK; st->tracked = 1; st->iterations = applied + 1; st->converged = 0; st->rescued = 1; }
  __shared__ float s_tinv[16];
  __shared__ unsigned s_cull[17];
  if (threadIdx.x == 64) kf_mat44_inverse(s_cur, s_tinv);
  __syncthreads();
  if (threadIdx.x < 16) st->pose_inv[threadIdx.x] = s_tinv[threadIdx.x];
  if (L.cull_on)                                                               // the launch's tail (cull.h), every workgroup's share in turn: nobody else is left
    for (int w = 0; w < L.n_loop; ++w) cull_tail(L.cull, s_tinv, w, L.n_loop, s_cull);
}
// Who ends a persistent tracking launch?  One compare-and-swap on KfTrackState::commit_word decides for every workgroup: (tag_base | 1) the launch's own
// workgroups, (tag_base | 2) ONE workgroup that finishes it alone after a time-out.  `want`: 1 or 2; `seen0`: the word as this workgroup read it when it
// started (an older launch's value, normally).  Returns the winner (1 or 2); *mine: this call's compare-and-swap set the word.  One lane calls.
__device__ __forceinline__ unsigned track_commit_decide(KfTrackState* st, unsigned tag_base, unsigned want, unsigned seen0, bool* mine) {
  unsigned expect = seen0;
  *mine = false;
  for (;;) {
    if ((expect & ~63u) == tag_base) return expect & 63u;                 // this launch's word is set: somebody decided
    const unsigned seen = atomicCAS(&st->commit_word, expect, tag_base | want);
    if (seen == expect) { *mine = true; return want; }
    expect = seen;                                                          // (an older word replaced by this launch's)
  }
}
// a fold of the loop timed out (uniform: every lane of the workgroup is here): claim the launch or leave
// s_abort on exit of the decision: 2 this workgroup finishes the frame alone; 1 somebody else does (leave); 3 the launch's own workgroups are ending it -- only
// possible when the time-out hit the LAST fold (a workgroup that passed it proves every partial sum of every step published, this one's included): retry that fold
#define ICP_DECIDE_ABORT() do { \
    if (threadIdx.x == 0) { \
      bool mine_; \
      const unsigned who_ = track_commit_decide(st, L.tag_base, 2u, commit_seen0, &mine_); \
      if (who_ == 2u && mine_) { \
        __hip_atomic_store(&st->rescue_tag, L.tag_base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      /* whoever still polls leaves */ \
        __hip_atomic_store(L.stall_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
        s_abort = 2; \
      } else s_abort = who_ == 2u ? 1 : 3; \
    } \
    __syncthreads(); \
  } while (0)
#ifdef KF_ICP_NO_SOLO          /* A/B variant only (tools/build_variant.sh): what the solo path costs the hot loop in registers / code placement */
#define ICP_SOLO_CALL do { if (threadIdx.x == 0) { st->status = KF_TRACK_STALLED; st->tracked = 0; } } while (0)
#else
#define ICP_SOLO_CALL icp_solo_finish(L, s_pose, s_linv, s_wave, s_tot, &s_abort)
#endif

// a rider workgroup of a tracking launch: two 64x4 tiles of the NEXT frame's gate + bilateral filter (IcpLoopArgs::bil)
__device__ __forceinline__ void icp_rider(const IcpLoopArgs& L) {
  __shared__ float s_bil[2 * (BIL_TX + 8) * (BIL_TY + 8)];
  const int half = (int)(threadIdx.x >> 8), t = ((int)blockIdx.x - L.n_loop) * 2 + half;
  const int tt = t < L.bil_tiles ? t : L.bil_tiles;                          // past the last tile: a row below the image, nothing is touched
  float* tile = s_bil + half * ((BIL_TX + 8) * (BIL_TY + 8));
  if (L.bil_fast) kf_bilateral_tile<4, true>(L.bil, tt % L.bil_gx, tt / L.bil_gx, (int)(threadIdx.x & 255), tile);
  else kf_bilateral_tile<4, false>(L.bil, tt % L.bil_gx, tt / L.bil_gx, (int)(threadIdx.x & 255), tile);
}

__global__ void __launch_bounds__(ICP_THREADS) k_icp_loop(IcpLoopArgs L) {
  if ((int)blockIdx.x >= L.n_loop) { icp_rider(L); return; }                   // a rider: two 64x4 filter tiles (bilateral_tile.h), then done
  __shared__ float s_pose[2][16], s_linv[16];                                  // the running transform lives in one of two buffers: a step writes the other one
  __shared__ float s_wave[27 * (ICP_THREADS / 16)], s_tot[(ICP_THREADS / 32) * 32];
  __shared__ int s_abort;
  int s_code = STEP_APPLIED, cur_buf = 0;
  float* s_cur = s_pose[0];
  KfTrackState* st = L.track;
  // a workgroup that only becomes resident after the launch has been given up (icp_solo_finish above) has nothing to do; `play_dead`: fault
  // injection (kf_inject_track_stall) -- the last workgroup behaves as if it never became resident
  // (the pose and the claim word are requested together: one round trip)
  const float pose_e = st->pose[threadIdx.x & 15];                             // ICP.cpp:62 cur_transform = _pose
  const unsigned commit_seen0 = __hip_atomic_load(&st->commit_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (requested with the pose and the claim word: one round trip)
  if (__hip_atomic_load(&st->rescue_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L.tag_base) return;
  if (L.play_dead && (int)blockIdx.x == L.n_loop - 1) return;
  if (threadIdx.x < 16) s_cur[threadIdx.x] = pose_e;
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();
  if (threadIdx.x == 0) kf_mat44_inverse(s_cur, s_linv);                        // ICP.cpp:63 last_transform_inv
  __syncthreads();
  TrackArgs a;
  a.dist_thres = L.dist_thres; a.sin_thres = L.sin_thres; a.dist_shake = L.dist_shake; a.angle_shake = L.angle_shake; a.cos_shake = L.cos_shake; a.dist_shake2 = L.dist_shake2;
  a.dist_thres2 = L.dist_thres2; a.sin_thres2 = L.sin_thres2; a.sdf = 0;
#ifdef KF_EXPERIMENTS
  a.dbg = (KF_EXP_MODE(L) == 11 && blockIdx.x == 0) ? L.slots + (size_t)26 * KF_ICP_LOOP_MAX_WG * 32 : nullptr;
  a.exp_nodet = KF_EXP_MODE(L) == 12;
#endif
  int step = 0, n_prev = 0, applied = 0;
  bool timed_out = false;                        // a fold gave up waiting: leave both loops, see ICP_ON_ABORT (one copy of the solo finish)
  // diagnostic build path (KF_ICP_EXP=7): workgroup 0 accumulates shader-clock ticks per segment into track->reduced
  const bool stamp = KF_EXP_MODE(L) == 7 && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long t_last = 0; float seg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define KF_STAMP(i) do { if (stamp) { unsigned long long t_now = __builtin_amdgcn_s_memtime(); seg[i] += (float)(t_now - t_last); t_last = t_now; } } while (0)
  if (stamp) t_last = __builtin_amdgcn_s_memtime();
  for (int l = L.levels - 1; l >= 0; --l) {                                      // coarse -> fine, ICP.cpp:65
    a.cam = L.cam[l];
    const float4* __restrict__ new_v = L.new_v[l]; const float4* __restrict__ new_n = L.new_n[l];
    const float4* __restrict__ model_v = L.model_v[l]; const float4* __restrict__ model_n = L.model_n[l];
    const int npx = a.cam.cols * a.cam.rows;
    int px_l, grid_l;
    icp_level_geometry(npx, L.n_virtual, l, px_l, grid_l);
    const bool has_px = (int)blockIdx.x < grid_l;
    // this lane's own vertices / normals depend neither on the running transform nor on the iteration: they are loaded once
    // per pyramid level and stay in registers for all of its iterations (the reference re-reads them 4 / 5 / 10 times)
    float4 iv[ICP_PX], in_[ICP_PX];
#pragma unroll
    for (int j = 0; j < ICP_PX; ++j) {
      const int i = icp_dealt_pixel(j, grid_l);
      // (a conditional load on purpose: the branch-free form that helps the per-step gathers in icp_accumulate costs 21 us per frame
      // here -- measured, profiles/r03_icp_step.txt -- and these loads happen once per pyramid level, not once per step)
      iv[j] = make_float4(0.f, 0.f, 0.f, 0.f); in_[j] = iv[j];
      if (has_px && j < px_l && i < npx) { iv[j] = new_v[i]; in_[j] = new_n[i]; }
    }
    for (int it = 0; it < L.iters[l]; ++it, ++step) {
      KF_STAMP(0);
#ifdef KF_EXPERIMENTS
      if (step > 0 && KF_EXP_MODE(L) == 14) {
        // experiment (VERDICT r3 #3b, "leader fold"): workgroup 0 alone folds and solves; everybody else polls ONE line -- the new transform as 16
        // tagged words (+ the verdict) -- instead of folding the 200 x 27 partial sums itself.  Same bits; timing: profiles/r04_icp_exchange.txt
        unsigned long long* bc = L.slots + (size_t)28 * KF_ICP_LOOP_MAX_WG * 32 + (size_t)(step - 1) * 32;
        const unsigned tag = L.tag_base + (unsigned)(step - 1);
        if (blockIdx.x == 0) {
          fold_partials_tagged(L.slots + (size_t)(step - 1) * KF_ICP_LOOP_MAX_WG * 32, n_prev, tag, s_tot, &s_abort);
          s_code = apply_step(a, s_tot, s_cur, nullptr, s_pose[cur_buf ^ 1]);
          if (threadIdx.x < 17) {
            const unsigned bits = threadIdx.x < 16 ? __float_as_uint(s_pose[cur_buf ^ 1][threadIdx.x]) : (unsigned)s_code;
            __hip_atomic_store(bc + threadIdx.x, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else {
          __shared__ int s_bc_code;
          if (threadIdx.x < 17) {
            unsigned long long u;
            do { u = __hip_atomic_load(bc + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((unsigned)(u >> 32) != tag) __builtin_amdgcn_s_sleep(1); } while ((unsigned)(u >> 32) != tag);
            if (threadIdx.x < 16) s_pose[cur_buf ^ 1][threadIdx.x] = __uint_as_float((unsigned)u); else s_bc_code = (int)(unsigned)u;
          }
          __syncthreads();
          s_code = s_bc_code;
        }
        if (s_code != STEP_APPLIED) {
          if (blockIdx.x == 0 && threadIdx.x == 0) { st->status = s_code; st->tracked = 0; st->iterations = applied; st->converged = 0; st->rescued = 0; }
          return;
        }
        cur_buf ^= 1; s_cur = s_pose[cur_buf];
        __syncthreads();
        ++applied;
      } else
#endif
      if (step > 0) {
        fold_partials_tagged(L.slots + (size_t)(step - 1) * KF_ICP_LOOP_MAX_WG * 32, (KF_EXP_MODE(L) == 2 && n_prev > 16) ? 16 : n_prev, L.tag_base + (unsigned)(step - 1), s_tot, &s_abort, KF_EXP_MODE(L), &st->rescue_tag, L.tag_base);
        if (s_abort) { timed_out = true; break; }
        KF_STAMP(1);
#ifdef KF_EXPERIMENTS
        if (KF_EXP_MODE(L) == 9 && blockIdx.x == 5 && (threadIdx.x & 63) == 0)     // per-wave: fold of the previous step done
          L.slots[(size_t)25 * KF_ICP_LOOP_MAX_WG * 32 + (size_t)step * 32 + 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef KF_EXPERIMENTS
        if (KF_EXP_MODE(L) == 8 && threadIdx.x == 0) L.slots[(size_t)24 * KF_ICP_LOOP_MAX_WG * 32 + (size_t)(step - 1) * 1024 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        if (KF_EXP_MODE(L) == 1) { s_code = STEP_APPLIED; __syncthreads(); } else     // timing only: skip the 6x6 solve
        { s_code = apply_step(a, s_tot, s_cur, nullptr, s_pose[cur_buf ^ 1]); if (s_code == STEP_APPLIED) { cur_buf ^= 1; s_cur = s_pose[cur_buf]; } }
        if (s_code != STEP_APPLIED) {                                            // same verdict in every workgroup
          if (blockIdx.x == 0 && threadIdx.x == 0) { st->status = s_code; st->tracked = 0; st->iterations = applied; st->converged = 0; st->rescued = 0; }
          return;
        }
        ++applied;
      }
      KF_STAMP(2);
      float acc[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[k] = 0.f;
      if (has_px) {
        icp_accumulate(a, s_cur, s_linv, iv, in_, model_v, model_n, acc);
        KF_STAMP(3);
#ifdef KF_EXPERIMENTS
        if (KF_EXP_MODE(L) == 9 && blockIdx.x == 5 && (threadIdx.x & 63) == 0)     // per-wave: pixel phase done (workgroup 5)
          L.slots[(size_t)25 * KF_ICP_LOOP_MAX_WG * 32 + (size_t)step * 32 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
#endif
        // workgroup partial, stored write-through (sc1) for the other CUs
        int k; float sw;
        if (icp_wg_reduce(acc, s_wave, k, sw))
          __hip_atomic_store(L.slots + (size_t)step * KF_ICP_LOOP_MAX_WG * 32 + blockIdx.x * 32 + k,
                             ((unsigned long long)(L.tag_base + (unsigned)step) << 32) | (unsigned long long)__float_as_uint(sw),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      KF_STAMP(4);
      KF_STAMP(5);
#ifdef KF_EXPERIMENTS
      if (KF_EXP_MODE(L) == 9 && blockIdx.x == 5 && (threadIdx.x & 63) == 0)       // per-wave: partial published / wave idle again
        L.slots[(size_t)25 * KF_ICP_LOOP_MAX_WG * 32 + (size_t)step * 32 + 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef KF_EXPERIMENTS
      if (KF_EXP_MODE(L) == 8 && threadIdx.x == 0) L.slots[(size_t)24 * KF_ICP_LOOP_MAX_WG * 32 + (size_t)step * 1024 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
      n_prev = grid_l;
    }
    if (timed_out) break;
  }
  if (stamp) { for (int i = 0; i < 6; ++i) st->reduced[20 + i] = seg[i]; }
  // the last step's system, then commit _pose (ICP.cpp:84)
  __shared__ unsigned s_who;
  for (;;) {
    if (!timed_out) fold_partials_tagged(L.slots + (size_t)(step - 1) * KF_ICP_LOOP_MAX_WG * 32, n_prev, L.tag_base + (unsigned)(step - 1), s_tot, &s_abort, 0, &st->rescue_tag, L.tag_base);
    if (!(timed_out || s_abort)) break;
    ICP_DECIDE_ABORT();
    if (s_abort == 2) { ICP_SOLO_CALL; return; }
    if (s_abort != 3 || timed_out) return;                                     // somebody else finishes the launch alone
    __syncthreads();
    if (threadIdx.x == 0) s_abort = 0;                                         // the others are through: every sum is published, the last fold cannot wait again
    __syncthreads();
  }
  // exactly one party ends the launch: a lane of an idle wave settles it while wave 0 solves; s_who is read behind apply_step's barrier
  if (threadIdx.x == 128) { bool mine_; s_who = track_commit_decide(st, L.tag_base, 1u, commit_seen0, &mine_); }
  s_code = apply_step(a, s_tot, s_cur, nullptr, s_pose[cur_buf ^ 1]);
  if (s_who == 2u) return;                                                     // a workgroup claimed the launch after a time-out: it commits and runs every share of the tail
  if (s_code == STEP_APPLIED) { cur_buf ^= 1; s_cur = s_pose[cur_buf]; }
  const bool tail = L.cull_on && s_code == STEP_APPLIED;                       // (uniform over the launch: every workgroup arrives at the same verdict)
  if (blockIdx.x != 0 && !tail) return;
  __shared__ float s_tinv[16];
  __shared__ unsigned s_cull[17];
  if (blockIdx.x == 0) {
    if (threadIdx.x < 27 && KF_EXP_MODE(L) != 7) st->reduced[threadIdx.x] = s_tot[threadIdx.x];
    if (s_code != STEP_APPLIED) { if (threadIdx.x == 0) { st->status = s_code; st->tracked = 0; st->iterations = applied; st->converged = 0; st->rescued = 0; } return; }
    if (threadIdx.x < 16) st->pose[threadIdx.x] = s_cur[threadIdx.x];
    if (threadIdx.x == 0) { st->status = KF_TRACK_OK; st->tracked = 1; st->iterations = applied + 1; st->converged = 0; st->rescued = 0; }    // (the whole verdict: this launch may have run without k_track_begin's reset)
  }
  if (threadIdx.x == 64) kf_mat44_inverse(s_cur, s_tinv);                      // a lane of another wave: world -> camera (integrateVolume.cu:84), for the fusion pass and the tail
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < 16) st->pose_inv[threadIdx.x] = s_tinv[threadIdx.x];
  if (tail) cull_tail(L.cull, s_tinv, (int)blockIdx.x, L.n_loop, s_cull);      // every workgroup holds the same pose bits, hence the same inverse
}


// ---- the same loop for images whose level 0 needs more workgroups than the chip holds at once (1280x960: 800) ------------------------------------------
// The per-step form pays a kernel boundary per Gauss-Newton step and folds 800 x 27 partials from a cold start each time (24 us per step at 1280x960:
// 0.46 ms of BASELINE config C5's frame).  Here n_loop resident workgroups play the n_virtual workgroups of the dealing in turn -- workgroup p plays p,
// p + n_loop, ... -- each turn exactly a workgroup's pixel phase, reduction and publication in the per-step form's dealing, so the tagged slots hold the
// same partial sums and the same fold arrives at the SAME BITS as k_icp_step (tests/test_gpu_track_forms.py).  A lane's pixels cannot stay in registers
// across steps (several turns share them): they are re-read every step, from L2 / the Infinity Cache (78 MB per step at 1280x960 fit the 256 MiB cache).
__global__ void __launch_bounds__(ICP_THREADS) k_icp_loop_batched(IcpLoopArgs L) {
  if ((int)blockIdx.x >= L.n_loop) { icp_rider(L); return; }
  __shared__ float s_pose[2][16], s_linv[16];
  __shared__ float s_wave[27 * (ICP_THREADS / 16)], s_tot[(ICP_THREADS / 32) * 32];
  __shared__ int s_abort;
  int s_code = STEP_APPLIED, cur_buf = 0;
  float* s_cur = s_pose[0];
  KfTrackState* st = L.track;
  const float pose_e = st->pose[threadIdx.x & 15];                             // ICP.cpp:62 cur_transform = _pose
  const unsigned commit_seen0 = __hip_atomic_load(&st->commit_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_load(&st->rescue_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L.tag_base) return;
  if (L.play_dead && (int)blockIdx.x == L.n_loop - 1) return;
  if (threadIdx.x < 16) s_cur[threadIdx.x] = pose_e;
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();
  if (threadIdx.x == 0) kf_mat44_inverse(s_cur, s_linv);                        // ICP.cpp:63 last_transform_inv
  __syncthreads();
  TrackArgs a;
  a.dist_thres = L.dist_thres; a.sin_thres = L.sin_thres; a.dist_shake = L.dist_shake; a.angle_shake = L.angle_shake; a.cos_shake = L.cos_shake; a.dist_shake2 = L.dist_shake2;
  a.dist_thres2 = L.dist_thres2; a.sin_thres2 = L.sin_thres2; a.sdf = 0;
#ifdef KF_EXPERIMENTS
  a.dbg = nullptr; a.exp_nodet = false;
#endif
  int step = 0, n_prev = 0, applied = 0;
  bool timed_out = false;
  for (int l = L.levels - 1; l >= 0; --l) {                                      // coarse -> fine, ICP.cpp:65
    a.cam = L.cam[l];
    const float4* __restrict__ new_v = L.new_v[l]; const float4* __restrict__ new_n = L.new_n[l];
    const int npx = a.cam.cols * a.cam.rows;
    int px_l, grid_l;
    icp_level_geometry(npx, L.n_virtual, l, px_l, grid_l);
    for (int it = 0; it < L.iters[l]; ++it, ++step) {
      // this step's FIRST turn's pixels are requested before the previous step's sums are waited for (they depend on the level only): in flight during the fold;
      // every later turn's pixels are requested one turn ahead (below)
      float4 ivn[ICP_PX], inn[ICP_PX];
#pragma unroll
      for (int j = 0; j < ICP_PX; ++j) {
        const int i = icp_dealt_pixel_of(j, grid_l, (int)blockIdx.x);
        ivn[j] = make_float4(0.f, 0.f, 0.f, 0.f); inn[j] = ivn[j];
        if ((int)blockIdx.x < grid_l && j < px_l && i < npx) { ivn[j] = new_v[i]; inn[j] = new_n[i]; }
      }
      if (step > 0) {
        fold_partials_tagged(L.slots + (size_t)(step - 1) * KF_ICP_LOOP_MAX_WG * 32, n_prev, L.tag_base + (unsigned)(step - 1), s_tot, &s_abort, 0, &st->rescue_tag, L.tag_base);
        if (s_abort) { timed_out = true; break; }
        s_code = apply_step(a, s_tot, s_cur, nullptr, s_pose[cur_buf ^ 1]);
        if (s_code != STEP_APPLIED) {                                            // same verdict in every workgroup
          if (blockIdx.x == 0 && threadIdx.x == 0) { st->status = s_code; st->tracked = 0; st->iterations = applied; st->converged = 0; st->rescued = 0; }
          return;
        }
        cur_buf ^= 1; s_cur = s_pose[cur_buf];
        ++applied;
      }
      // this workgroup's turns: the workgroups blockIdx.x, blockIdx.x + n_loop, ... of the dealing; their partial sums are added up HERE, in that order, and
      // published as ONE set of 27 words -- a step's fold then walks n_loop x 27 tagged words (205 at 1280x960), not grid_l x 27 (800): round 4's batched
      // loop spent four polling passes per step on them.  The per-step form folds its 800 partials in the same groups (fold_partials, TrackArgs::fold_group).
      float gsum = 0.f;
      // the NEXT turn's pixels are requested before this turn's pixel phase: a turn's own loads (L2 / Infinity Cache, ~1 us) would otherwise be exposed three or
      // four times per step -- the workgroups of a step reach the exchange that much earlier
      for (int w = (int)blockIdx.x; w < grid_l; w += L.n_loop) {
        float4 iv[ICP_PX], in_[ICP_PX];
#pragma unroll
        for (int j = 0; j < ICP_PX; ++j) { iv[j] = ivn[j]; in_[j] = inn[j]; }
        if (w + L.n_loop < grid_l) {
#pragma unroll
          for (int j = 0; j < ICP_PX; ++j) {
            const int i = icp_dealt_pixel_of(j, grid_l, w + L.n_loop);
            ivn[j] = make_float4(0.f, 0.f, 0.f, 0.f); inn[j] = ivn[j];
            if (j < px_l && i < npx) { ivn[j] = new_v[i]; inn[j] = new_n[i]; }
          }
        }
        float acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.f;
        icp_accumulate(a, s_cur, s_linv, iv, in_, L.model_v[l], L.model_n[l], acc);
        int k; float sw;
        const bool holder = icp_wg_reduce(acc, s_wave, k, sw);
        if (holder) gsum = (w == (int)blockIdx.x) ? sw : gsum + sw;
        __syncthreads();                                                         // s_wave is reused by the next turn
        if (holder && w + L.n_loop >= grid_l)                                    // the last turn: publish
          __hip_atomic_store(L.slots + (size_t)step * KF_ICP_LOOP_MAX_WG * 32 + blockIdx.x * 32 + k,
                             ((unsigned long long)(L.tag_base + (unsigned)step) << 32) | (unsigned long long)__float_as_uint(gsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      n_prev = grid_l < L.n_loop ? grid_l : L.n_loop;
    }
    if (timed_out) break;
  }
  __shared__ unsigned s_who;
  for (;;) {
    if (!timed_out) fold_partials_tagged(L.slots + (size_t)(step - 1) * KF_ICP_LOOP_MAX_WG * 32, n_prev, L.tag_base + (unsigned)(step - 1), s_tot, &s_abort, 0, &st->rescue_tag, L.tag_base);
    if (!(timed_out || s_abort)) break;
    ICP_DECIDE_ABORT();
    if (s_abort == 2) { ICP_SOLO_CALL; return; }
    if (s_abort != 3 || timed_out) return;
    __syncthreads();
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
  }
  if (threadIdx.x == 128) { bool mine_; s_who = track_commit_decide(st, L.tag_base, 1u, commit_seen0, &mine_); }      // (see k_icp_loop)
  s_code = apply_step(a, s_tot, s_cur, nullptr, s_pose[cur_buf ^ 1]);
  if (s_who == 2u) return;
  if (s_code == STEP_APPLIED) { cur_buf ^= 1; s_cur = s_pose[cur_buf]; }
  if (blockIdx.x != 0) return;
  if (threadIdx.x < 27) st->reduced[threadIdx.x] = s_tot[threadIdx.x];
  if (s_code != STEP_APPLIED) { if (threadIdx.x == 0) { st->status = s_code; st->tracked = 0; st->iterations = applied; st->converged = 0; st->rescued = 0; } return; }
  if (threadIdx.x < 16) st->pose[threadIdx.x] = s_cur[threadIdx.x];
  if (threadIdx.x == 64) kf_mat44_inverse(s_cur, st->pose_inv);
  if (threadIdx.x == 0) { st->status = KF_TRACK_OK; st->tracked = 1; st->iterations = applied + 1; st->converged = 0; st->rescued = 0; }
}

// ---- SDF tracker ------------------------------------------------------------------------------------------------------
// the six perturbed transforms delta * cur for +w1, -w1, +w2, -w2, +w3, -w3 (CalSDFErrSolverParams.cu:118-133) into s_m[1..6]; s_m[0] = cur.  96 lanes.
__device__ __forceinline__ void sdf_perturbed_transforms(float (*s_m)[16], float w_h) {
  if (threadIdx.x < 96) {
    const int mi = threadIdx.x >> 4, e = threadIdx.x & 15, r = e >> 2, cidx = e & 3;
    const int axis = mi >> 1; const float sg = (mi & 1) ? -1.f : 1.f;     // second matrix of a pair flips the sign
    float d[4] = {0.f, 0.f, 0.f, 0.f}; d[r] = 1.f;
    if (axis == 0) { if (r == 1) d[2] = -sg * w_h; if (r == 2) d[1] = sg * w_h; }          // m23 = -w, m32 = +w
    else if (axis == 1) { if (r == 0) d[2] = sg * w_h; if (r == 2) d[0] = -sg * w_h; }     // m13 = +w, m31 = -w
    else { if (r == 0) d[1] = -sg * w_h; if (r == 1) d[0] = sg * w_h; }                    // m12 = -w, m21 = +w
    s_m[1 + mi][e] = d[0] * s_m[0][cidx] + d[1] * s_m[0][4 + cidx] + d[2] * s_m[0][8 + cidx] + d[3] * s_m[0][12 + cidx];
  }
}
// one pixel into the 27 running sums (computeSDFSolverbufKernel, CalSDFErrSolverParams.cu:68-108: row[i] * row[j], i <= j); d = 0: no depth, no row
template <typename AD>
__device__ __forceinline__ void sdf_accumulate_pixel(const KfVolume& vol, const AD& S, const float (*s_m)[16], const KfCam& cam, int i, float d, float w_h, float v_h,
                                                     const KfRecip& rS, float rcell, bool slab, float acc[27]) {
  const float3 p = kf_depth_to_skeleton((unsigned)(i % cam.cols), (unsigned)(i / cam.cols), d, cam);
  float row[7];
  const bool ok = sdf_pixel_row<AD>(vol, S, s_m, p, w_h, v_h, rS, rcell, slab, row) && d != 0.f;
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 7; ++k) row[k] = 0.f;                            // (a rejected pixel adds +0.0 twenty-seven times: the sums keep their bits)
  }
  int s = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 7; ++c) { acc[s] = __builtin_fmaf(row[r], row[c], acc[s]); ++s; }
}
template <typename AD>
__global__ void __launch_bounds__(TRK_THREADS) k_sdf_step(TrackArgs a) {
  __shared__ float s_m[7][16];                 // cur, then delta*cur for +w1,-w1,+w2,-w2,+w3,-w3 (CalSDFErrSolverParams.cu:118-133)
  __shared__ float s_linv[16];
  __shared__ float s_wave[27 * (TRK_THREADS / 16)], s_tot[32 * 32];
  __shared__ int s_code;
  if (!step_prologue(a, s_m[0], s_linv, s_tot, &s_code)) return;
  const float w_h = 0.001f;                                               // :119 `float w_h = 0.001;`
  const float v_h = a.vol.size / (float)a.vol.res;                        // :120
  sdf_perturbed_transforms(s_m, w_h);
  __syncthreads();
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  const KfRecip rS = kf_recip(a.vol.size);
  const float rcell = 1.0f / a.vol.cell;
  const AD S(a.vol);
  // a wave takes 8x8 pixel tiles (see k_sdf_loop: a compact footprint in the volume = few cache lines per gather instruction), grid-stride over the tiles
  const int tiles_x = (a.cam.cols + 7) >> 3, n_tiles = tiles_x * ((a.cam.rows + 7) >> 3);
  for (int tile = (int)blockIdx.x * (TRK_THREADS / 64) + (int)(threadIdx.x >> 6); tile < n_tiles; tile += (int)gridDim.x * (TRK_THREADS / 64)) {
    const int x = (tile % tiles_x) * 8 + (int)(threadIdx.x & 7), y = (tile / tiles_x) * 8 + (int)((threadIdx.x >> 3) & 7);
    const bool in = x < a.cam.cols && y < a.cam.rows;
    const int i = in ? y * a.cam.cols + x : 0;
    const float d = in ? a.depth[i] : 0.f;
    if (__ballot(d != 0.f) == 0ull) continue;                             // a wave without a single depth value: nothing to look up
    // z-slabs: a pixel belongs to the slab that owns the voxel of its world point; every rank sums its own pixels and the
    // 27-float systems are all-reduced.  The 13 lookups reach at most the halo (kf_sdf_partition_step checks it).
    sdf_accumulate_pixel<AD>(a.vol, S, s_m, a.cam, i, d, w_h, v_h, rS, rcell, a.slab_pixels != 0, acc);
  }
  store_partial(acc, a.partials + (size_t)(a.step & 1) * KF_ICP_MAX_WG * 32, s_wave);
}
// volumes whose stored voxels fill less than 4 GB are addressed through a raw buffer descriptor (sdf_rows.h)
static inline bool sdf_buffer_addressing(const KfVolume& v) {
  return (unsigned long long)(v.bz1 - v.bz0) * v.nb * v.nb * KF_BRICK_VOX * sizeof(float2) < 0xFFFFFFFFull && v.nb <= 1024;
}
static inline void launch_sdf_step(kf_ctx* c, int grid, const TrackArgs& a) {
  if (sdf_buffer_addressing(a.vol)) hipLaunchKernelGGL(k_sdf_step<SdfBufAddr>, dim3(grid), dim3(TRK_THREADS), 0, c->stream, a);
  else hipLaunchKernelGGL(k_sdf_step<SdfFlatAddr>, dim3(grid), dim3(TRK_THREADS), 0, c->stream, a);
}

// ---- SDF tracker, persistent: the whole Gauss-Newton loop of CameraPoseFinderSDF::estimateCameraPose (SDF.cpp:57-103) in ONE launch ---------------
// Round 4 launched one k_sdf_step per iteration (+ begin + finish): at VGA 300 workgroups of 4 waves -- about one wave per SIMD -- each walking 4 pixels x 13
// dependent lookups, 60 us per executed iteration (profiles/r05_c3_before_kernel_stats.csv), three of them per frame on Scene S.  Here the iterations share a
// launch the way the ICP loop's do: n_loop co-resident 512-lane workgroups, pixels dealt in 64-pixel chunks round robin (icp_dealt_pixel_of: every workgroup
// sees the same mix of surface and background), the 27 sums per workgroup by DPP + LDS, published as tagged 64-bit words and folded by everybody in one fixed
// order (fold_partials_tagged), the 6x6 system solved by a wave (llt_solve6_wave), the shake / convergence tests and the fp64 exponential map on that wave,
// and the loop leaves as soon as the increment's norm drops below 1e-3 (SDF.cpp:87-90) -- no launches that find `converged` and return.
// A fold that times out is finished by ONE workgroup playing every workgroup in turn (as icp_solo_finish, but the same code: `first` / `stride`), and
// exactly one party ends the launch (KfTrackState::commit_word).
#ifndef SDF_THREADS
#define SDF_THREADS ICP_THREADS
#endif
struct SdfLoopArgs {
  KfVolume vol; const float* depth; KfCam cam;
  int max_iter;
  float dist_shake, angle_shake, cos_shake, dist_shake2;
  unsigned long long* slots; unsigned tag_base; KfTrackState* track; unsigned* stall_word;
  int n_loop;                                    // workgroups of the launch = publishers of a step
  int px_l;                                      // pixels per lane = 8x8 tiles per wave: ceil(tiles / (waves per workgroup * n_loop))
  int slab_pixels;
  int play_dead;                                 // fault injection (kf_inject_track_stall)
  int cull_on; IntegrateArgs cull;               // the fusion pass's brick cull as the launch's tail (cull.h), as in k_icp_loop
};
// direct_exponential_map (eigen_utils.cpp:84-127) + SDF.cpp:92-100 with the three even series in t = theta^2 instead of sqrt / sin / cos / five fp64 divisions
// on the one-lane chain of every iteration: sinc = sum (-t)^k / (2k+1)!, mcosc = sum (-t)^k / (2k+2)!, msinc = sum (-t)^k / (2k+3)!, cos = 1 - t mcosc.
// Nine terms: below 1e-19 for theta <= 0.5 (the shake test admits 0.3 rad with the stock parameters); larger increments take the library routines.  The
// reference's small-angle constants (sinc = 1 below 1e-8, mcosc = 1/2 and msinc = 1/6 below 2.5e-4) are kept.  Tolerance side (pose within 1e-4).
__device__ __forceinline__ void sdf_apply_increment_series(const float x[6], const float* cur, float ncur[16]) {
  const double u0 = (double)x[0], u1 = (double)x[1], u2 = (double)x[2];
  const double t = u0 * u0 + u1 * u1 + u2 * u2;
  if (!(t <= 0.25)) { sdf_apply_increment(x, cur, ncur); return; }
  // three independent Horner chains in -t (they interleave on the one lane that runs them)
  double a = 1.0 / 355687428096000.0, b = 1.0 / 6402373705728000.0, c = 1.0 / 121645100408832000.0;                    // 1/17!, 1/18!, 1/19!
  a = 1.0 / 1307674368000.0 - t * a;  b = 1.0 / 20922789888000.0 - t * b;  c = 1.0 / 355687428096000.0 - t * c;        // 1/15!, 1/16!, 1/17!
  a = 1.0 / 6227020800.0 - t * a;     b = 1.0 / 87178291200.0 - t * b;     c = 1.0 / 1307674368000.0 - t * c;          // 1/13!, 1/14!, 1/15!
  a = 1.0 / 39916800.0 - t * a;       b = 1.0 / 479001600.0 - t * b;       c = 1.0 / 6227020800.0 - t * c;             // 1/11!, 1/12!, 1/13!
  a = 1.0 / 362880.0 - t * a;         b = 1.0 / 3628800.0 - t * b;         c = 1.0 / 39916800.0 - t * c;               // 1/9!, 1/10!, 1/11!
  a = 1.0 / 5040.0 - t * a;           b = 1.0 / 40320.0 - t * b;           c = 1.0 / 362880.0 - t * c;                 // 1/7!, 1/8!, 1/9!
  a = 1.0 / 120.0 - t * a;            b = 1.0 / 720.0 - t * b;             c = 1.0 / 5040.0 - t * c;                   // 1/5!, 1/6!, 1/7!
  a = 1.0 / 6.0 - t * a;              b = 1.0 / 24.0 - t * b;              c = 1.0 / 120.0 - t * c;                    // 1/3!, 1/4!, 1/5!
  a = 1.0 - t * a;                    b = 0.5 - t * b;                     c = 1.0 / 6.0 - t * c;
  const double co = 1.0 - t * b;                                          // cos(theta), from the un-thresholded series
  const double sinc = t < 1.0e-16 ? 1.0 : a;                                           // fabs(theta) < 1e-8
  const double mcosc = t < 6.25e-8 ? 0.5 : b;                                          // fabs(theta) < 2.5e-4
  const double msinc = t < 6.25e-8 ? (1. / 6.0) : c;
  const double t3 = (double)x[3], t4 = (double)x[4], t5 = (double)x[5];
  double R[9];
  R[0] = co + mcosc * u0 * u0;         R[1] = -sinc * u2 + mcosc * u0 * u1; R[2] = sinc * u1 + mcosc * u0 * u2;
  R[3] = sinc * u2 + mcosc * u1 * u0;  R[4] = co + mcosc * u1 * u1;         R[5] = -sinc * u0 + mcosc * u1 * u2;
  R[6] = -sinc * u1 + mcosc * u2 * u0; R[7] = sinc * u0 + mcosc * u2 * u1;  R[8] = co + mcosc * u2 * u2;
  double dt[3];
  dt[0] = t3 * (sinc + u0 * u0 * msinc) + t4 * (u0 * u1 * msinc - u2 * mcosc) + t5 * (u0 * u2 * msinc + u1 * mcosc);
  dt[1] = t3 * (u0 * u1 * msinc + u2 * mcosc) + t4 * (sinc + u1 * u1 * msinc) + t5 * (u1 * u2 * msinc - u0 * mcosc);
  dt[2] = t3 * (u0 * u2 * msinc - u1 * mcosc) + t4 * (u1 * u2 * msinc + u0 * mcosc) + t5 * (sinc + u2 * u2 * msinc);
  float Rt[9], tf[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) Rt[r * 3 + cc] = (float)R[cc * 3 + r];
#pragma unroll
  for (int k = 0; k < 3; ++k) tf[k] = (float)dt[k];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) ncur[r * 4 + cc] = Rt[r * 3] * cur[cc] + Rt[r * 3 + 1] * cur[4 + cc] + Rt[r * 3 + 2] * cur[8 + cc];
    ncur[r * 4 + 3] = cur[r * 4 + 3] - (Rt[r * 3] * tf[0] + Rt[r * 3 + 1] * tf[1] + Rt[r * 3 + 2] * tf[2]);
  }
  ncur[12] = 0.f; ncur[13] = 0.f; ncur[14] = 0.f; ncur[15] = 1.f;
}
// One iteration's update (SDF.cpp:79-100) by wave 0 from the folded sums: solve, shake test, convergence test, exponential map.  Every workgroup runs it on
// identical inputs -> identical transforms.  On STEP_APPLIED s_cur holds the new transform when the function returns (all lanes, after its barrier).
__device__ __forceinline__ int sdf_apply_step_wave(const float* s_tot, float* s_cur, float dist_shake2, float cos_shake, int* s_code) {
  if (threadIdx.x < 64) {
    float x[6]; bool singular; float det;
    llt_solve6_wave(s_tot, x, singular, det);                                // (A.llt().solve(b): no pivot test in the reference -- a non-positive pivot yields NaN there and here)
    const float ang = threadIdx.x == 0 ? x[0] : (threadIdx.x == 1 ? x[1] : x[2]);
    float sn, cs; kf_sincos_small(ang, &sn, &cs);
    const int sni = __float_as_int(sn), csi = __float_as_int(cs);
    const float c0 = __int_as_float(__builtin_amdgcn_readlane(csi, 0)), s0 = __int_as_float(__builtin_amdgcn_readlane(sni, 0));
    const float c1 = __int_as_float(__builtin_amdgcn_readlane(csi, 1)), s1 = __int_as_float(__builtin_amdgcn_readlane(sni, 1));
    const float c2 = __int_as_float(__builtin_amdgcn_readlane(csi, 2)), s2 = __int_as_float(__builtin_amdgcn_readlane(sni, 2));
    float T[16];
    const bool still = transform_from_sincos(x, c0, s0, c1, s1, c2, s2, dist_shake2, cos_shake, T);      // SDF.cpp:25-43 (only its verdict is used, :81-85)
    int code = STEP_APPLIED;
    if (!still) code = STEP_LOST_SHAKE;
    else {
      const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
      if (nx < 0.001f) code = STEP_CONVERGED;                                // SDF.cpp:87-90: stop before applying x
    }
    if (threadIdx.x == 0) {
      if (code == STEP_APPLIED) {
        float ncur[16];
        sdf_apply_increment_series(x, s_cur, ncur);
#pragma unroll
        for (int i = 0; i < 16; ++i) s_cur[i] = ncur[i];
      }
      *s_code = code;
    }
  }
  __syncthreads();
  return *s_code;
}
// icp_wg_reduce for any workgroup size T (a multiple of 64): 16-lane row totals by DPP, one LDS word per (sum, row), then 4 lanes per sum add a quarter of the
// T / 16 row totals each and two DPP shifts combine them (fixed order).  s_wave: 27 x (T / 16) floats.
template <int T>
__device__ __forceinline__ bool sdf_wg_reduce(float acc[27], float* s_wave, int& k, float& sw) {
  constexpr int ROWS = T / 16, PER = ROWS / 4;
  const int row = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = kf_row_scan_sum(acc[i]);
  if ((threadIdx.x & 15) == 15) {
#pragma unroll
    for (int i = 0; i < 27; ++i) s_wave[i * ROWS + row] = acc[i];
  }
  __syncthreads();
  k = threadIdx.x >> 2; sw = 0.f;
  if (threadIdx.x < 27 * 4) {
    const int part = threadIdx.x & 3;
    sw = s_wave[k * ROWS + part * PER];
#pragma unroll
    for (int w = 1; w < PER; ++w) sw += s_wave[k * ROWS + part * PER + w];
    sw += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sw), 0x111, 0xf, 0xf, true));   // row_shr:1
    sw += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sw), 0x112, 0xf, 0xf, true));   // row_shr:2
    return part == 3;
  }
  return false;
}
template <typename AD>
__global__ void __launch_bounds__(SDF_THREADS) k_sdf_loop(SdfLoopArgs L) {
  __shared__ float s_m[7][16];
  __shared__ float s_wave[27 * (SDF_THREADS / 16)], s_tot[(SDF_THREADS / 32) * 32];
  __shared__ int s_abort, s_code, s_role;
  __shared__ float s_tinv[16];
  __shared__ unsigned s_cull[17];
  KfTrackState* st = L.track;
  // (the claim word and the commit word are requested together with the pose: one round trip)
  const unsigned commit_seen0 = __hip_atomic_load(&st->commit_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_load(&st->rescue_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L.tag_base) return;     // the launch was given to one workgroup before this one arrived
  if (L.play_dead && (int)blockIdx.x == L.n_loop - 1) return;
  const float w_h = 0.001f;                                               // CalSDFErrSolverParams.cu:119
  const float v_h = L.vol.size / (float)L.vol.res;                        // :120
  const int grid_l = L.n_loop;
  const int tiles_x = (L.cam.cols + 7) >> 3, tiles_y = (L.cam.rows + 7) >> 3;
  const KfRecip rS = kf_recip(L.vol.size);
  const float rcell = 1.0f / L.vol.cell;
  const AD S(L.vol);
  int first = (int)blockIdx.x, stride = L.n_loop;                         // the workgroups of the dealing this one plays: its own -- or all of them, once it finishes alone
  bool solo = false;
  int code = STEP_APPLIED, applied = 0;
  for (int attempt = 0; ; ++attempt) {
    if (threadIdx.x < 16) s_m[0][threadIdx.x] = st->pose[threadIdx.x];    // SDF.cpp:56 cur_transform = _pose (again from the committed pose: nothing of an attempt is kept)
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    code = STEP_APPLIED; applied = 0;
    bool timed_out = false;
    for (int it = 0; it < L.max_iter; ++it) {                             // SDF.cpp:57
      sdf_perturbed_transforms(s_m, w_h);
      __syncthreads();
#pragma unroll 1
      for (int w = first; w < grid_l; w += stride) {
        float acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.f;
#pragma unroll 1
        for (int j = 0; j < L.px_l; ++j) {
          // a wave takes an 8x8 pixel TILE, not 64 pixels of a row: its 64 world points then fall into a few bricks' worth of cache lines (a scanline
          // of 64 pixels crosses ~30 voxel columns and, on a slanted surface, as many layers: ~22 lines per gather instruction, which is what
          // bounded the pixel phase -- profiles/r05_c3_*); tiles are dealt to the waves round robin like the ICP loop's 64-pixel chunks
          const int tile = ((int)(threadIdx.x >> 6) * grid_l + w) + j * ((SDF_THREADS / 64) * grid_l);
          const int tx = tile % tiles_x, ty = tile / tiles_x;
          const int x = tx * 8 + (int)(threadIdx.x & 7), y = ty * 8 + (int)((threadIdx.x >> 3) & 7);
          const bool in = ty < tiles_y && x < L.cam.cols && y < L.cam.rows;
          const int i = in ? y * L.cam.cols + x : 0;
          const float d = in ? L.depth[i] : 0.f;
          if (__ballot(d != 0.f) == 0ull) continue;                       // a wave without a single depth value
          sdf_accumulate_pixel<AD>(L.vol, S, s_m, L.cam, i, d, w_h, v_h, rS, rcell, L.slab_pixels != 0, acc);
        }
        int k; float sw;
        if (sdf_wg_reduce<SDF_THREADS>(acc, s_wave, k, sw))
          __hip_atomic_store(L.slots + (size_t)it * KF_ICP_LOOP_MAX_WG * 32 + w * 32 + k,
                             ((unsigned long long)(L.tag_base + (unsigned)it) << 32) | (unsigned long long)__float_as_uint(sw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                                   // s_wave is reused by the next turn
      }
      fold_partials_tagged(L.slots + (size_t)it * KF_ICP_LOOP_MAX_WG * 32, grid_l, L.tag_base + (unsigned)it, s_tot, &s_abort, 0, solo ? nullptr : &st->rescue_tag, L.tag_base);
      if (s_abort) { timed_out = true; break; }
      code = sdf_apply_step_wave(s_tot, s_m[0], L.dist_shake2, L.cos_shake, &s_code);      // SDF.cpp:79-100
      if (code != STEP_APPLIED) break;
      ++applied;
    }
    // Who ends the launch?  role 0: leave; 1: the ordinary end (one of the launch's workgroups); 2: finish the launch alone (run the loop again, playing every
    // workgroup); 3: the others went through after all (every sum is published by now): run the loop again in step with them; 4: the solo finisher's end
    if (threadIdx.x == 0) {
      int role;
      if (solo) role = timed_out ? 0 : 4;
      else {
        bool mine;
        const unsigned who = track_commit_decide(st, L.tag_base, timed_out ? 2u : 1u, commit_seen0, &mine);
        if (!timed_out) role = who == 1u ? 1 : 0;
        else if (who == 2u && mine) {                                      // this workgroup's compare-and-swap took the word: it finishes the launch alone
          __hip_atomic_store(&st->rescue_tag, L.tag_base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);       // whoever still polls leaves
          __hip_atomic_store(L.stall_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          role = 2;
        } else role = who == 1u ? 3 : 0;
      }
      s_role = role;
    }
    __syncthreads();
    const int role = s_role;
    if (role == 1 || role == 4) break;
    if (role == 2) { solo = true; first = 0; stride = 1; continue; }
    if (role == 3 && attempt < 3) continue;
    if (solo && threadIdx.x == 0) { st->status = KF_TRACK_STALLED; st->tracked = 0; st->iterations = applied; st->converged = 0; st->rescued = 1; }   // (a solo fold waits for nobody: unreachable)
    return;
  }
  // ---- the launch's end: SDF.cpp:103 _pose = cur_transform (a lost frame keeps the old pose: `return false` at :84), then the tail --------------------
  const bool writer = solo || blockIdx.x == 0;
  const bool good = code != STEP_LOST_SHAKE && code != STEP_LOST_DET;
  if (writer) {
    if (threadIdx.x < 27) st->reduced[threadIdx.x] = s_tot[threadIdx.x];
    if (good && threadIdx.x < 16) st->pose[threadIdx.x] = s_m[0][threadIdx.x];
    if (threadIdx.x == 0) {
      st->status = good ? KF_TRACK_OK : code; st->tracked = good ? 1 : 0; st->iterations = applied;
      st->converged = code == STEP_CONVERGED ? 1 : 0; st->rescued = solo ? 1 : 0;
    }
  }
  if (!good) return;
  const bool tail = L.cull_on != 0;
  if (!writer && !tail) return;
  if (threadIdx.x == 64) kf_mat44_inverse(s_m[0], s_tinv);                // world -> camera (integrateVolume.cu:84), for the fusion pass and the tail
  __syncthreads();
  if (writer && threadIdx.x < 16) st->pose_inv[threadIdx.x] = s_tinv[threadIdx.x];
  if (tail) {
    if (solo) { for (int w = 0; w < L.n_loop; ++w) cull_tail(L.cull, s_tinv, w, L.n_loop, s_cull); }
    else cull_tail(L.cull, s_tinv, (int)blockIdx.x, L.n_loop, s_cull);
  }
}

// ---- loop control ---------------------------------------------------------------------------------------------------
// mode 0: frame 0 (no tracking, "tracked"); mode 1: start of a Gauss-Newton loop
__global__ void k_track_begin(KfTrackState* st, int mode, KfGridBarrier* gb) {
  if (blockIdx.x == 0 && threadIdx.x < 10 && gb) reinterpret_cast<KfPaddedCounter*>(gb)[threadIdx.x].v = 0u;   // 8 groups + top + gen
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->status = KF_TRACK_OK; st->iterations = 0; st->converged = 0; st->arrive = 0u; st->rescued = 0;
  if (mode == 0) { st->tracked = 1; return; }
  st->tracked = 0;
  for (int i = 0; i < 16; ++i) st->cur[0][i] = st->pose[i];                 // ICP.cpp:62
  kf_mat44_inverse(st->pose, st->last_inv);                                // ICP.cpp:63 last_transform_inv = _pose.getInverse()
}

// Last launch of a loop: fold + apply the final step (a.step = number of steps launched), then commit _pose (ICP.cpp:84,
// SDF.cpp:103).  With use_state == 0 it only leaves the 27 sums in track->reduced (per-call wrappers).
__global__ void __launch_bounds__(ICP_THREADS) k_track_finish(TrackArgs a) {
  __shared__ float s_cur[16], s_linv[16], s_tot[32 * 32];
  __shared__ int s_code;
  KfTrackState* st = a.track;
  if (!a.use_state || a.fold_out) {                                       // fold only: per-call wrappers, pixel-partitioned ICP
    fold_partials(a.partials + (size_t)((a.step + 1) & 1) * KF_ICP_MAX_WG * 32, a.n_prev_wg, s_tot);
    if (threadIdx.x < 32) {
      if (a.fold_out) a.fold_out[threadIdx.x] = threadIdx.x < 27 ? s_tot[threadIdx.x] : 0.f;
      else if (threadIdx.x < 27) st->reduced[threadIdx.x] = s_tot[threadIdx.x];
    }
    return;
  }
  if (!step_prologue(a, s_cur, s_linv, s_tot, &s_code)) return;            // lost or converged: already recorded
  if (threadIdx.x < 16) st->pose[threadIdx.x] = s_cur[threadIdx.x];
  if (threadIdx.x == 64) kf_mat44_inverse(s_cur, st->pose_inv);
  if (threadIdx.x == 0) st->tracked = 1;
}

// acos(ca) <= angle  <=>  ca >= cos(angle) for angle in [0, pi]; beyond pi every rotation passes, below 0 none does (NaN stays NaN: none)
static inline float shake_cos(float angle) { return angle >= 3.14159274f ? -2.f : (angle < 0.f ? 2.f : cosf(angle)); }
// norm > t  <=>  norm^2 > t^2 only for t >= 0; for a negative threshold the reference's comparison holds for every norm (every pixel
// is rejected, every increment counts as shaking): -1 keeps that, since a squared norm is never below zero
static inline float gate_square(float t) { return t < 0.f ? -1.f : t * t; }
// the thresholds of one tracking call, in every form the kernels use
static inline void set_thresholds(TrackArgs& a, float dist_thres, float sin_thres, float dist_shake, float angle_shake) {
  a.dist_thres = dist_thres; a.sin_thres = sin_thres; a.dist_shake = dist_shake; a.angle_shake = angle_shake;
  a.cos_shake = shake_cos(angle_shake); a.dist_shake2 = gate_square(dist_shake);
  a.dist_thres2 = gate_square(dist_thres); a.sin_thres2 = gate_square(sin_thres);
}
static inline KfCam to_cam(const kf_camera_params* p) {
  KfCam c; c.cols = (int)p->cols; c.rows = (int)p->rows; c.cx = p->cx; c.cy = p->cy; c.fx = p->fx; c.fy = p->fy; return c;
}
static inline int track_grid(int npx) {           // SDF tracker: 256 lanes, grid-stride
  int g = kf_div_up(npx, TRK_THREADS * TRK_PX);
  return g < 1 ? 1 : (g > KF_ICP_MAX_WG ? KF_ICP_MAX_WG : g);
}
static inline int icp_grid(int npx) { return kf_div_up(npx, ICP_THREADS * ICP_PX); }   // ICP: every pixel exactly once

extern "C" int kf_set_pose(kf_ctx* c, const kf_mat44* pose) {
  if (!c || !pose) return KF_ERR_ARG;
  { const int ds = kf_tail_cull_discard(c); if (ds) return ds; }   // (a cull that ran for the tracked pose does not describe this one)
  memcpy(c->host_pinned, pose->m, 64);
  const int one = 1;
  memcpy((char*)c->host_pinned + 64, &one, sizeof(one));
  kf_mat44_inverse(pose->m, (float*)((char*)c->host_pinned + 128));          // same arithmetic as the device's commits
  KF_CHECK(hipMemcpyAsync(c->track->pose, c->host_pinned, 64, hipMemcpyHostToDevice, c->stream));
  KF_CHECK(hipMemcpyAsync(c->track->pose_inv, (char*)c->host_pinned + 128, 64, hipMemcpyHostToDevice, c->stream));
  // a pose supplied by the caller counts as a successful localisation: the device-predicated integrate (transform == NULL) fuses with it
  KF_CHECK(hipMemcpyAsync(&c->track->tracked, (char*)c->host_pinned + 64, sizeof(one), hipMemcpyHostToDevice, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));     // the pinned staging words are reused
  return 0;
}

extern "C" int kf_cal_point_to_plane_solver_params(kf_ctx* c, uint32_t level, const kf_mat44* cur, const kf_mat44* last_inv,
                                                   const kf_camera_params* cam, float dist_thres, float sin_thres) {
  if (!c || !cur || !last_inv || !cam || level >= (uint32_t)c->levels) return KF_ERR_ARG;
  if ((int)cam->cols != c->lvl_cols[level] || (int)cam->rows != c->lvl_rows[level]) return KF_ERR_ARG;
  TrackArgs a; memset(&a, 0, sizeof(a));
  a.new_v = c->new_v[level]; a.new_n = c->new_n[level]; a.model_v = c->model_v[level]; a.model_n = c->model_n[level];
  a.cam = to_cam(cam);
  for (int i = 0; i < 16; ++i) { a.cur_val.m[i] = cur->m[i]; a.linv_val.m[i] = last_inv->m[i]; }
  set_thresholds(a, dist_thres, sin_thres, 0.f, 0.f);
  a.partials = c->icp_partials; a.track = c->track;
  const int grid = icp_grid(a.cam.cols * a.cam.rows);
  if (grid > KF_ICP_MAX_WG) return KF_ERR_ARG;
  hipLaunchKernelGGL(k_icp_step, dim3(grid), dim3(ICP_THREADS), 0, c->stream, a);
  a.step = 1; a.n_prev_wg = grid;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, a);
  return (int)hipGetLastError();
}

extern "C" int kf_cal_sdf_solver_params(kf_ctx* c, const kf_camera_params* cam, const kf_mat44* cur) {
  if (!c || !cam || !cur) return KF_ERR_ARG;
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  TrackArgs a; memset(&a, 0, sizeof(a));
  a.cam = to_cam(cam); a.vol = c->vol; a.depth = c->trunced_depth; a.sdf = 1;
  for (int i = 0; i < 16; ++i) a.cur_val.m[i] = cur->m[i];
  a.partials = c->icp_partials; a.track = c->track;
  const int grid = track_grid(c->cols * c->rows);
  launch_sdf_step(c, grid, a);
  a.step = 1; a.n_prev_wg = grid;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, a);
  return (int)hipGetLastError();
}

extern "C" int kf_read_solver_params(kf_ctx* c, float out27[27]) {
  if (!c || !out27) return KF_ERR_ARG;
  KF_CHECK(hipMemcpyAsync(c->host_pinned, c->track->reduced, 27 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  memcpy(out27, c->host_pinned, 27 * sizeof(float));
  return 0;
}

#ifndef KF_ICP_COOPERATIVE_DEFAULT
#define KF_ICP_COOPERATIVE_DEFAULT 0
#endif
// a persistent loop timed out and was finished by one workgroup alone (icp_solo_finish: the frame is kept, milliseconds late): back off to
// per-step launches for a while (64 frames, doubling up to 4096 on repeats; 1024 clean loop frames in a row forget the history).  One episode
// is noted once: launches of the loop that were already enqueued when the word was seen raise it again while the back-off is running.
static void kf_note_loop_stall(kf_ctx* c) {
  if (c->persistent_backoff > 0) return;
  c->persistent_backoff_len = c->persistent_backoff_len ? (c->persistent_backoff_len >= 2048 ? 4096 : c->persistent_backoff_len * 2) : 64;
  c->persistent_backoff = c->persistent_backoff_len;
  c->loop_clean_frames = 0;
}

// The fusion pass's cull as the tail of a persistent tracking launch (k_icp_loop, k_sdf_loop): the previous kf_integrate_volume(transform == NULL) left its
// parameters behind, the tile tables describe THIS frame's depth map, no deferred-weight words (their cull retires bricks: side effects nobody could undo) and
// few enough macro cells for the launch's workgroups.  kf_integrate_volume consumes it, or undoes it when it is asked for something else.
static bool kf_arm_tail_cull(kf_ctx* c, const kf_camera_params* cam0, int n_wg, int waves, IntegrateArgs& out) {
  static int tail_env = -1;
  if (tail_env < 0) { const char* e = getenv("KF_CULL_IN_TRACK"); tail_env = e ? atoi(e) : 1; }
  if (!(tail_env && c->cull_hint.valid && !kf_defer_enabled(c) &&
        c->tile_serial != 0 && c->tile_serial == c->trunc_serial && c->tile_built_dist == c->cull_hint.max_dist &&
        memcmp(&c->cull_hint.dcam, cam0, sizeof(*cam0)) == 0 && kf_cull_tail_fits(c, n_wg, waves))) return false;
  kf_fill_cull_args(c, out, &c->cull_hint.dcam, c->cull_hint.sdf_trunc, c->cull_hint.max_dist);
  c->tail_cull.armed = 1; c->tail_cull.parity = out.parity; c->tail_cull.bz0 = c->vol.bz0; c->tail_cull.bz1 = c->vol.bz1;
  c->tail_cull.sdf_trunc = c->cull_hint.sdf_trunc; c->tail_cull.max_dist = c->cull_hint.max_dist; c->tail_cull.dcam = c->cull_hint.dcam;
  c->tail_cull.trunc_serial = c->trunc_serial;
  return true;
}

extern "C" int kf_icp_track(kf_ctx* c, uint32_t frame_id, const kf_icp_params* icp, const kf_camera_params* cam0) {
  if (!c || !icp || !cam0) return KF_ERR_ARG;
  if ((int)icp->pyramid_levels != c->levels || (int)cam0->cols != c->cols || (int)cam0->rows != c->rows) return KF_ERR_ARG;
  c->last_track_form = 0;
  { const int ds = kf_tail_cull_discard(c); if (ds) return ds; }   // (a tail cull of the previous frame that no kf_integrate_volume consumed)
  if (frame_id == 0) {                                       // ICP.cpp:52-55
    hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, 0, c->grid_barrier);
    return (int)hipGetLastError();
  }
  int iters[KF_MAX_LEVELS] = {0, 0, 0};                      // ICP.cpp:14-35
  if (c->levels == 1) iters[0] = 3; else if (c->levels == 2) { iters[0] = 10; iters[1] = 5; } else { iters[0] = 10; iters[1] = 5; iters[2] = 4; }
  int st;
  kf_evt_begin(c, KF_STAGE_TRACK);
  kf_camera_params cams[KF_MAX_LEVELS]; cams[0] = *cam0;
  for (int l = 1; l < c->levels; ++l) {                      // ICP.cpp:36-48
    cams[l].cols = cams[l - 1].cols / 2; cams[l].rows = cams[l - 1].rows / 2;
    cams[l].cx = cams[l - 1].cx / 2; cams[l].cy = cams[l - 1].cy / 2; cams[l].fx = cams[l - 1].fx / 2; cams[l].fy = cams[l - 1].fy / 2;
  }
  const int grid0 = icp_grid(c->cols * c->rows);
  static int persistent_env = -1, coop_env = -1;
  if (persistent_env < 0) { const char* e = getenv("KF_ICP_PERSISTENT"); persistent_env = e ? atoi(e) : 1; }
  if (coop_env < 0) { const char* e = getenv("KF_ICP_COOPERATIVE"); coop_env = e ? atoi(e) : KF_ICP_COOPERATIVE_DEFAULT; }
  // A persistent loop of an EARLIER frame that gave up waiting (its workgroups were not co-resident: a foreign process on the GPU) has
  // set the pinned stall word by now -- no read-back of the verdict needed.  The next frames use one launch per step (same bits, see
  // k_icp_step); after `persistent_backoff_len` of them the loop is tried again, and every further stall doubles that wait.
  unsigned* stall_word = (unsigned*)((char*)c->host_pinned + KF_PINNED_STALL_WORD);
  if (__atomic_load_n(stall_word, __ATOMIC_RELAXED)) { __atomic_store_n(stall_word, 0u, __ATOMIC_RELAXED); kf_note_loop_stall(c); }
  else if (c->persistent_backoff == 0 && c->persistent_backoff_len && ++c->loop_clean_frames >= 1024) { c->persistent_backoff_len = 0; c->loop_clean_frames = 0; }
  bool use_loop = persistent_env && !c->loop_refused && grid0 <= KF_ICP_LOOP_MAX_WG && iters[0] + iters[1] + iters[2] <= KF_ICP_LOOP_STEPS &&
                  kf_live_contexts(c->cfg.device) == 1 && !kf_device_shared(c->cfg.device);
  if (c->loop_occupancy == 0) {                              // (asked once, whatever form this call takes: the per-step form folds in the batched loop's groups)
    // every workgroup must be resident at once (they wait for each other's tagged partial sums): ask the runtime how many 512-lane
    // workgroups of THESE kernels a CU holds (registers, LDS) instead of assuming one
    int per_cu = 0, per_cu_b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_icp_loop, ICP_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, k_icp_loop_batched, ICP_THREADS, 0) != hipSuccess || per_cu_b < 1) per_cu_b = -1;
    c->loop_occupancy = per_cu; c->loop_occupancy_batched = per_cu_b;
  }
  // Images whose level 0 needs more workgroups than the chip holds at once (1280x960: 800): the batched loop -- as many resident workgroups as the
  // VGA loop uses (the CU count less a share for the riders), each playing several workgroups of the dealing in turn.  KF_ICP_BATCHED=0: per step.
  static int batched_env = -1;
  if (batched_env < 0) { const char* e = getenv("KF_ICP_BATCHED"); batched_env = e ? atoi(e) : 1; }
  int n_resident = grid0;                                    // workgroups of the loop launch
  bool batched = false;
  // fold_group: the publishers of a batched launch.  Every launch form of such an image adds the partial sums of the workgroups g, g + fold_group, ... first
  // (the batched loop's resident workgroup g plays exactly those): the forms stay bitwise equal
  int fold_group = 0;
  {
    const bool beyond = c->loop_occupancy < 1 || (long long)grid0 > (long long)c->loop_occupancy * c->num_cus;
    static int room_env = -1;                                // KF_ICP_BATCHED_ROOM=n: resident workgroups of the batched loop (A/B; every launch form folds in groups of n, so the forms stay bitwise equal)
    if (room_env < 0) { const char* e = getenv("KF_ICP_BATCHED_ROOM"); room_env = e ? atoi(e) : 0; }
    const int room = c->loop_occupancy_batched >= 1 ? ((room_env >= 16 && room_env <= c->num_cus * c->loop_occupancy_batched) ? room_env : c->num_cus - c->num_cus / 5) : 0;      // 256 CUs: 205 resident workgroups, 51 CUs left to the riders
    const bool can_batch = beyond && batched_env && !coop_env && room >= 16 && grid0 <= KF_ICP_LOOP_MAX_WG;
    if (can_batch) fold_group = room;
    if (use_loop && beyond) {
      if (can_batch) { batched = true; n_resident = room; }
      else use_loop = false;
    }
  }
  if (use_loop && c->persistent_backoff > 0) { --c->persistent_backoff; use_loop = false; }
  // ICP.cpp:57-63: the four pyramids + the loop's set-up, one launch -- unless every pyramid describes its level 0 already (the raycast launch
  // left the model maps' behind, its riders the prefetched frame's: raycast.hip) AND the persistent loop runs, which reads nothing of that
  // set-up but the committed pose and writes its whole verdict itself: the steady-state frame then has no pyramid launch at all
  const bool pyramids_done = c->levels == 3 && c->new_pyr_ok && c->model_pyr_ok;
  if (!(use_loop && !coop_env && pyramids_done) && (st = kf_launch_pyramids_and_begin(c, 1))) return st;
  if (use_loop) {
    IcpLoopArgs L; memset(&L, 0, sizeof(L));
    for (int l = 0; l < c->levels; ++l) {
      L.new_v[l] = c->new_v[l]; L.new_n[l] = c->new_n[l]; L.model_v[l] = c->model_v[l]; L.model_n[l] = c->model_n[l];
      L.cam[l] = to_cam(&cams[l]); L.iters[l] = iters[l];
    }
    L.levels = c->levels;
    { TrackArgs th; set_thresholds(th, icp->dist_thres, icp->norm_sin_thres, icp->dist_shake, icp->angle_shake);
      L.dist_thres = th.dist_thres; L.sin_thres = th.sin_thres; L.dist_shake = th.dist_shake; L.angle_shake = th.angle_shake;
      L.cos_shake = th.cos_shake; L.dist_shake2 = th.dist_shake2; L.dist_thres2 = th.dist_thres2; L.sin_thres2 = th.sin_thres2; }
    c->icp_loop_seq += 64u;                                  // tags of one launch never collide with an earlier launch's slots
    L.slots = c->icp_loop_slots; L.tag_base = c->icp_loop_seq; L.track = c->track;
    L.stall_word = stall_word;
    if (c->inject_stall > 0) { L.play_dead = 1; --c->inject_stall; }
    { static int em = -1; if (em < 0) em = KF_EXP_ENV("KF_ICP_EXP"); L.exp_mode = em; }
    L.n_loop = n_resident; L.n_virtual = grid0;
    // a pending kf_prefetch_frame: the next frame's filter rides in this launch (see IcpLoopArgs) and leaves the gated + filtered image in
    // the alternate buffers; the raycast launch that follows carries its tile tables (the fusion pass in between clears them) and its
    // vertices / normals.  Not under a cooperative launch, whose whole grid would have to be resident.
    static int ride_env = -1;
    if (ride_env < 0) { const char* e = getenv("KF_PREFETCH_IN_TRACK"); ride_env = e ? atoi(e) : 1; }
    unsigned n_riders = 0;
    if (ride_env && !coop_env && c->fp_pending && c->alt_raw && !c->fp_filtered) {
      bool fast;
      kf_bilateral_args(c, c->fp_src, nullptr, c->alt_raw, c->alt_trunced, c->alt_filtered, c->fp_params[0], c->fp_params[1], c->fp_params[2], c->fp_params[3],
                        false, &L.bil, &fast);
      L.bil_fast = fast ? 1 : 0;
      L.bil_gx = kf_div_up(c->cols, BIL_TX); L.bil_tiles = L.bil_gx * kf_div_up(c->rows, BIL_TY);
      n_riders = (unsigned)((L.bil_tiles + 1) / 2);
      c->fp_filtered = 1;
    }
    if (!batched && !coop_env && L.exp_mode == 0 && kf_arm_tail_cull(c, cam0, grid0, ICP_THREADS / 64, L.cull)) L.cull_on = 1;
    if (coop_env) {
      // a cooperative launch: the runtime itself checks that the whole grid can be resident and refuses otherwise
      void* params[] = {(void*)&L};
      const hipError_t e = hipLaunchCooperativeKernel((const void*)k_icp_loop, dim3(grid0), dim3(ICP_THREADS), params, 0, c->stream);
      if (e == hipSuccess) { c->last_track_form = 1; kf_evt_end(c, KF_STAGE_TRACK); return 0; }
      (void)hipGetLastError();                               // refused: this device / configuration cannot hold the loop -- per-step from now on
      c->loop_refused = 1;
    } else {
      if (batched) hipLaunchKernelGGL(k_icp_loop_batched, dim3((unsigned)n_resident + n_riders), dim3(ICP_THREADS), 0, c->stream, L);
      else hipLaunchKernelGGL(k_icp_loop, dim3((unsigned)grid0 + n_riders), dim3(ICP_THREADS), 0, c->stream, L);
      c->last_track_form = 1;
      kf_evt_end(c, KF_STAGE_TRACK);
      return (int)hipGetLastError();
    }
  }
  // one launch per Gauss-Newton step: the same pixel dealing, the same reduction and fold orders as the loop -> the same pose bits
  TrackArgs a; memset(&a, 0, sizeof(a));
  a.use_state = 1;
  set_thresholds(a, icp->dist_thres, icp->norm_sin_thres, icp->dist_shake, icp->angle_shake);
  a.partials = c->icp_partials; a.track = c->track;
  a.fold_group = fold_group;
  int step = 0, prev_grid = 0;
  for (int l = c->levels - 1; l >= 0; --l)                   // coarse -> fine, ICP.cpp:65
    for (int it = 0; it < iters[l]; ++it) {
      a.new_v = c->new_v[l]; a.new_n = c->new_n[l]; a.model_v = c->model_v[l]; a.model_n = c->model_n[l];
      a.cam = to_cam(&cams[l]);
      a.step = step; a.consume = step > 0; a.n_prev_wg = prev_grid;
      icp_level_geometry(a.cam.cols * a.cam.rows, grid0, l, a.px_l, a.deal_grid);
      const int grid = a.deal_grid;
      if (grid > KF_ICP_MAX_WG) return KF_ERR_ARG;
      hipLaunchKernelGGL(k_icp_step, dim3(grid), dim3(ICP_THREADS), 0, c->stream, a);
      prev_grid = grid; ++step;
    }
  a.step = step; a.consume = 1; a.n_prev_wg = prev_grid;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, a);
  c->last_track_form = 2;
  kf_evt_end(c, KF_STAGE_TRACK);
  return (int)hipGetLastError();
}

// ---- pixel-partitioned ICP for z-slab / multi-GPU runs (all-reduce of the 27-float system) -------------
// The caller owns a 32-float device buffer `dev_sums`.  Per Gauss-Newton step: kf_icp_partition_step sums this rank's pixel
// range and leaves the 27 sums in dev_sums; the caller all-reduces dev_sums (SUM) over the ranks; the next step (or
// kf_icp_partition_finish) consumes it.  Every rank applies the identical all-reduced system, so the poses agree bitwise.
extern "C" int kf_icp_partition_begin(kf_ctx* c, uint32_t frame_id) {
  if (!c) return KF_ERR_ARG;
  { const int ds = kf_tail_cull_discard(c); if (ds) return ds; }
  if (frame_id != 0) kf_evt_begin(c, KF_STAGE_TRACK);       // (the stage timer runs from here to kf_icp_partition_finish: it includes the caller's all-reduces)
  return kf_launch_pyramids_and_begin(c, frame_id == 0 ? 0 : 1);
}
static int icp_iters(int levels, int iters[KF_MAX_LEVELS]) {                // ICP.cpp:14-35
  iters[0] = iters[1] = iters[2] = 0;
  if (levels == 1) iters[0] = 3; else if (levels == 2) { iters[0] = 10; iters[1] = 5; } else if (levels == 3) { iters[0] = 10; iters[1] = 5; iters[2] = 4; } else return 0;
  return iters[0] + iters[1] + iters[2];
}
extern "C" int kf_icp_partition_steps(kf_ctx* c) { int it[KF_MAX_LEVELS]; return c ? icp_iters(c->levels, it) : 0; }
extern "C" int kf_icp_partition_step(kf_ctx* c, uint32_t step, const kf_icp_params* icp, const kf_camera_params* cam0,
                                     uint32_t part, uint32_t parts, float* dev_sums) {
  if (!c || !icp || !cam0 || !dev_sums || parts == 0 || part >= parts) return KF_ERR_ARG;
  int iters[KF_MAX_LEVELS];
  const int total = icp_iters(c->levels, iters);
  if ((int)step >= total) return KF_ERR_ARG;
  int l = c->levels - 1, s = (int)step;                                     // coarse -> fine
  while (s >= iters[l]) { s -= iters[l]; --l; }
  kf_camera_params cam = *cam0;
  for (int k = 0; k < l; ++k) { cam.cols /= 2; cam.rows /= 2; cam.cx /= 2; cam.cy /= 2; cam.fx /= 2; cam.fy /= 2; }
  TrackArgs a; memset(&a, 0, sizeof(a));
  a.use_state = 1;
  a.new_v = c->new_v[l]; a.new_n = c->new_n[l]; a.model_v = c->model_v[l]; a.model_n = c->model_n[l];
  a.cam = to_cam(&cam);
  set_thresholds(a, icp->dist_thres, icp->norm_sin_thres, icp->dist_shake, icp->angle_shake);
  a.partials = c->icp_partials; a.track = c->track;
  a.step = (int)step; a.consume = step > 0; a.ext_prev = step > 0 ? dev_sums : nullptr;
  const int rows_per = kf_div_up(a.cam.rows, (int)parts);                    // whole image rows per rank
  const int y0 = (int)part * rows_per, y1 = (y0 + rows_per < a.cam.rows) ? y0 + rows_per : a.cam.rows;
  a.px_begin = y0 * a.cam.cols; a.px_end = (y1 > y0 ? y1 : y0) * a.cam.cols;
  const int grid = (a.px_end > a.px_begin) ? icp_grid(a.px_end - a.px_begin) : 1;
  if (grid > KF_ICP_MAX_WG) return KF_ERR_ARG;
  if (a.px_end <= a.px_begin) { a.px_begin = 1; a.px_end = 1; }              // empty range: the kernel's bound check skips every pixel
  hipLaunchKernelGGL(k_icp_step, dim3(grid), dim3(ICP_THREADS), 0, c->stream, a);
  TrackArgs f = a; f.step = (int)step + 1; f.n_prev_wg = grid; f.fold_out = dev_sums; f.ext_prev = nullptr;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, f);
  return (int)hipGetLastError();
}
extern "C" int kf_icp_partition_finish(kf_ctx* c, const kf_icp_params* icp, const float* dev_sums) {
  if (!c || !icp || !dev_sums) return KF_ERR_ARG;
  int iters[KF_MAX_LEVELS];
  TrackArgs a; memset(&a, 0, sizeof(a));
  a.use_state = 1; a.consume = 1; a.step = icp_iters(c->levels, iters); a.ext_prev = dev_sums;
  set_thresholds(a, icp->dist_thres, icp->norm_sin_thres, icp->dist_shake, icp->angle_shake); a.partials = c->icp_partials; a.track = c->track;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, a);
  kf_evt_end(c, KF_STAGE_TRACK);
  return (int)hipGetLastError();
}

#ifdef KF_EXPERIMENTS
extern "C" int kf_exp_read_icp_slots(kf_ctx* c, unsigned long long* dst, size_t first, size_t n) {
  KF_CHECK(hipMemcpyAsync(dst, c->icp_loop_slots + first, n * 8, hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}
#endif

extern "C" int kf_sdf_track(kf_ctx* c, uint32_t frame_id, const kf_sdf_tracker_params* sp, const kf_camera_params* cam) {
  if (!c || !sp || !cam) return KF_ERR_ARG;
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  c->last_track_form = 0;
  { const int ds = kf_tail_cull_discard(c); if (ds) return ds; }
  if (frame_id == 0) {                                       // SDF.cpp:46-49
    hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, 0, c->grid_barrier);
    return (int)hipGetLastError();
  }
  kf_evt_begin(c, KF_STAGE_TRACK);
  // The whole loop in one launch (k_sdf_loop) under the conditions of the persistent ICP loop: this context alone on the device, no back-off after a
  // time-out, every workgroup resident.  KF_SDF_PERSISTENT=0 (or KF_ICP_PERSISTENT=0): one launch per iteration, as the z-slab partition runs it.
  {
    static int loop_env = -1;
    if (loop_env < 0) { const char* e = getenv("KF_SDF_PERSISTENT"); const char* e2 = getenv("KF_ICP_PERSISTENT"); loop_env = (e ? atoi(e) : 1) && (e2 ? atoi(e2) : 1); }
    unsigned* stall_word = (unsigned*)((char*)c->host_pinned + KF_PINNED_STALL_WORD);
    if (__atomic_load_n(stall_word, __ATOMIC_RELAXED)) { __atomic_store_n(stall_word, 0u, __ATOMIC_RELAXED); kf_note_loop_stall(c); }
    else if (c->persistent_backoff == 0 && c->persistent_backoff_len && ++c->loop_clean_frames >= 1024) { c->persistent_backoff_len = 0; c->loop_clean_frames = 0; }
    const bool buf = sdf_buffer_addressing(c->vol);
    bool use_loop = loop_env && !c->loop_refused && sp->max_iter_nums >= 1 && sp->max_iter_nums <= KF_ICP_LOOP_STEPS &&
                    kf_live_contexts(c->cfg.device) == 1 && !kf_device_shared(c->cfg.device);
    if (use_loop && c->sdf_loop_occupancy == 0) {
      int per_cu = 0;
      const hipError_t e = buf ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sdf_loop<SdfBufAddr>, SDF_THREADS, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sdf_loop<SdfFlatAddr>, SDF_THREADS, 0);
      c->sdf_loop_occupancy = (e == hipSuccess && per_cu >= 1) ? per_cu : -1;
    }
    if (use_loop && c->sdf_loop_occupancy < 1) use_loop = false;
    if (use_loop && c->persistent_backoff > 0) { --c->persistent_backoff; use_loop = false; }
    if (use_loop) {
      const int npx = c->cols * c->rows;
      static int wg_env = -1;
      if (wg_env < 0) { const char* e = getenv("KF_SDF_LOOP_WG"); wg_env = e ? atoi(e) : 0; }
      int n_loop = wg_env > 0 ? wg_env : c->num_cus;                           // one workgroup per CU
      n_loop = n_loop > c->num_cus * c->sdf_loop_occupancy ? c->num_cus * c->sdf_loop_occupancy : n_loop;
      n_loop = n_loop > KF_ICP_LOOP_MAX_WG ? KF_ICP_LOOP_MAX_WG : n_loop;
      const int need = kf_div_up(kf_div_up(c->cols, 8) * kf_div_up(c->rows, 8), SDF_THREADS / 64);
      if (n_loop > need) n_loop = need;
      SdfLoopArgs L; memset(&L, 0, sizeof(L));
      L.vol = c->vol; L.depth = c->trunced_depth; L.cam = to_cam(cam); L.max_iter = (int)sp->max_iter_nums;
      { TrackArgs th; set_thresholds(th, 0.f, 0.f, sp->dist_shake, sp->angle_shake);
        L.dist_shake = th.dist_shake; L.angle_shake = th.angle_shake; L.cos_shake = th.cos_shake; L.dist_shake2 = th.dist_shake2; }
      c->icp_loop_seq += 64u;
      L.slots = c->icp_loop_slots; L.tag_base = c->icp_loop_seq; L.track = c->track; L.stall_word = stall_word;
      const int n_tiles = kf_div_up(c->cols, 8) * kf_div_up(c->rows, 8);      // 8x8 pixel tiles, one per wave and round
      L.n_loop = n_loop; L.px_l = kf_div_up(n_tiles, (SDF_THREADS / 64) * n_loop);
      if (c->inject_stall > 0) { L.play_dead = 1; --c->inject_stall; }
      if (kf_arm_tail_cull(c, cam, n_loop, SDF_THREADS / 64, L.cull)) L.cull_on = 1;
      if (buf) hipLaunchKernelGGL(k_sdf_loop<SdfBufAddr>, dim3((unsigned)n_loop), dim3(SDF_THREADS), 0, c->stream, L);
      else hipLaunchKernelGGL(k_sdf_loop<SdfFlatAddr>, dim3((unsigned)n_loop), dim3(SDF_THREADS), 0, c->stream, L);
      c->last_track_form = 1;
      kf_evt_end(c, KF_STAGE_TRACK);
      return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, 1, c->grid_barrier);
  TrackArgs a; memset(&a, 0, sizeof(a));
  a.use_state = 1; a.sdf = 1;
  a.cam = to_cam(cam); a.vol = c->vol; a.depth = c->trunced_depth;
  set_thresholds(a, 0.f, 0.f, sp->dist_shake, sp->angle_shake);
  a.partials = c->icp_partials; a.track = c->track;
  const int grid = track_grid(c->cols * c->rows);
  int step = 0;
  for (uint32_t it = 0; it < sp->max_iter_nums; ++it) {
    a.step = step; a.consume = step > 0; a.n_prev_wg = grid;
    launch_sdf_step(c, grid, a);
    ++step;
  }
  a.step = step; a.consume = step > 0; a.n_prev_wg = grid;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, a);
  c->last_track_form = 2;
  kf_evt_end(c, KF_STAGE_TRACK);
  return (int)hipGetLastError();
}

// ---- SDF tracker on z-slabs (pixel -> owning slab by pworld0.z, all-reduce of the 27-float system) -----------
// Same protocol as the pixel-partitioned ICP above: the caller owns a 32-float device buffer `dev_sums`; per Gauss-Newton iteration
// kf_sdf_partition_step sums the pixels whose world point falls into the layers this context owns and leaves the 27 sums there, the
// caller all-reduces it (SUM) over the ranks, the next step (or kf_sdf_partition_finish) applies the reduced system.  Every rank
// applies identical numbers -> identical poses.  All `max_iter_nums` steps are always issued (launches after convergence return at
// once), so every rank makes the same collective calls.
extern "C" int kf_sdf_partition_begin(kf_ctx* c, uint32_t frame_id) {
  if (!c) return KF_ERR_ARG;
  { const int ds = kf_tail_cull_discard(c); if (ds) return ds; }
  c->last_track_form = 0;
  if (frame_id != 0) kf_evt_begin(c, KF_STAGE_TRACK);       // (ends in kf_sdf_partition_finish: the caller's all-reduces are inside)
  hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, frame_id == 0 ? 0 : 1, c->grid_barrier);
  return (int)hipGetLastError();
}
static int sdf_partition_args(kf_ctx* c, const kf_sdf_tracker_params* sp, const kf_camera_params* cam, TrackArgs& a) {
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  // the perturbed lookups (rotations by 0.001 rad about the origin of a volume-sized lever, +- one cell) and their trilinear taps must
  // stay inside the stored layers: refuse a halo thinner than that, as kf_raycast_volume_slab does for its own reach
  const int need = (int)ceilf(0.001f * 1.7320508f * c->vol.size / c->vol.cell) + 3;
  const int lo = c->vol.own_z0 - c->vol.bz0 * KF_BRICK, hi = c->vol.bz1 * KF_BRICK - c->vol.own_z1;
  if ((c->vol.own_z0 > 0 && lo < need) || (c->vol.own_z1 < c->vol.res && hi < need)) return KF_ERR_ARG;
  memset(&a, 0, sizeof(a));
  a.use_state = 1; a.sdf = 1; a.slab_pixels = 1;
  a.cam = to_cam(cam); a.vol = c->vol; a.depth = c->trunced_depth;
  set_thresholds(a, 0.f, 0.f, sp->dist_shake, sp->angle_shake);
  a.partials = c->icp_partials; a.track = c->track;
  return 0;
}
extern "C" int kf_sdf_partition_step(kf_ctx* c, uint32_t step, const kf_sdf_tracker_params* sp, const kf_camera_params* cam, float* dev_sums) {
  if (!c || !sp || !cam || !dev_sums || step >= sp->max_iter_nums) return KF_ERR_ARG;
  TrackArgs a;
  int st = sdf_partition_args(c, sp, cam, a);
  if (st) return st;
  const int grid = track_grid(c->cols * c->rows);
  a.step = (int)step; a.consume = step > 0; a.n_prev_wg = grid; a.ext_prev = step > 0 ? dev_sums : nullptr;
  launch_sdf_step(c, grid, a);
  TrackArgs f = a; f.step = (int)step + 1; f.n_prev_wg = grid; f.fold_out = dev_sums; f.ext_prev = nullptr;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, f);
  return (int)hipGetLastError();
}
extern "C" int kf_sdf_partition_finish(kf_ctx* c, const kf_sdf_tracker_params* sp, const kf_camera_params* cam, const float* dev_sums) {
  if (!c || !sp || !cam || !dev_sums) return KF_ERR_ARG;
  TrackArgs a;
  int st = sdf_partition_args(c, sp, cam, a);
  if (st) return st;
  a.consume = sp->max_iter_nums > 0; a.step = (int)sp->max_iter_nums; a.ext_prev = dev_sums;
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(ICP_THREADS), 0, c->stream, a);
  kf_evt_end(c, KF_STAGE_TRACK);
  return (int)hipGetLastError();
}

// The blocking read-back split in two, so that a caller can put more work behind the tracker before it waits for the verdict:
// kf_request_track_result enqueues the copy of the tracking state (into pinned memory) right where the stream stands and marks
// the spot with an event; kf_wait_track_result blocks on THAT event only -- whatever was enqueued after the request keeps running.
extern "C" int kf_request_track_result(kf_ctx* c) {
  if (!c) return KF_ERR_ARG;
  if (!c->ev_track) KF_CHECK(hipEventCreateWithFlags(&c->ev_track, hipEventDisableTiming));
  KF_CHECK(hipMemcpyAsync((char*)c->host_pinned + 1024, c->track, sizeof(KfTrackState), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipEventRecord(c->ev_track, c->stream));
  c->track_requested = 1;
  return 0;
}
extern "C" int kf_wait_track_result(kf_ctx* c, kf_track_result* out) {
  if (!c || !out) return KF_ERR_ARG;
  if (!c->track_requested) return KF_ERR_STATE;
  KF_CHECK(hipEventSynchronize(c->ev_track));
  c->track_requested = 0;
  const KfTrackState* h = (const KfTrackState*)((const char*)c->host_pinned + 1024);
  memcpy(out->pose.m, h->pose, 64);
  out->tracked = h->tracked; out->status = h->status; out->iterations = h->iterations; out->launch_form = (h->rescued && c->last_track_form == 1) ? 3 : c->last_track_form;
  if (h->rescued && c->last_track_form == 1) kf_note_loop_stall(c);
  return 0;
}

extern "C" int kf_read_track_result(kf_ctx* c, kf_track_result* out) {
  if (!c || !out) return KF_ERR_ARG;
  KfTrackState* h = (KfTrackState*)c->host_pinned;
  KF_CHECK(hipMemcpyAsync(h, c->track, sizeof(KfTrackState), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  memcpy(out->pose.m, h->pose, 64);
  // launch_form 3: the persistent loop timed out waiting for a partial sum -- some of its workgroups were not resident (another process on
  // the GPU?) -- and one workgroup finished the frame alone (same bits, milliseconds late).  The next frames track with one launch per
  // step, then the loop is tried again.
  out->tracked = h->tracked; out->status = h->status; out->iterations = h->iterations; out->launch_form = (h->rescued && c->last_track_form == 1) ? 3 : c->last_track_form;
  if (h->rescued && c->last_track_form == 1) kf_note_loop_stall(c);
  return 0;
}

// Fault injection for tests and rehearsals: in each of the next `launches` launches of the persistent ICP loop one workgroup exits at once, as
// if a foreign process had kept it off the chip -- the others time out (20 ms) and one of them finishes the frame alone.  Results are unchanged.
extern "C" int kf_inject_track_stall(kf_ctx* c, int launches) {
  if (!c || launches < 0) return KF_ERR_ARG;
  c->inject_stall = launches;
  return 0;
}
