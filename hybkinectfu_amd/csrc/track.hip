// track.hip -- camera tracking: projective point-to-plane ICP and the direct SDF tracker, with the whole Gauss-Newton
// loop resident on the device.
//
// Reference: computeGbufKernel / findCorrs / buildPointToPlaneSolverRows (src/cuda/CalPointToPlaneErrSolverParams.cu:7-129),
// reduceGbufKernel (src/cuda/device_functions.h:17-55), computeSDFSolverbufKernel / buildSDFSolverRows
// (src/cuda/CalSDFErrSolverParams.cu:7-138), CameraPoseFinderICP (src/CameraPoseFinderICP.cpp:12-145),
// CameraPoseFinderSDF (src/CameraPoseFinderSDF.cpp:25-106), direct_exponential_map (src/utils/eigen_utils.cpp:60-127).
//
// gfx950 design: the reference builds the 6x6 system with 27 sequential 256-wide shared-memory tree reductions
// (54 barriers) per launch, a second 27-block launch, a device sync and a 108-byte read-back, 19 times per frame.
// Here one launch per Gauss-Newton step does everything:
//   * every lane keeps the 27 partial sums of its pixels in registers (no MFMA: a 6x6 outer product is far too small);
//   * a wave64 __shfl_down tree folds them, LDS holds the four per-wave partials, 27 lanes write the workgroup partial;
//   * the workgroups arrive on an agent-scope ticket (release fence -> relaxed atomic); the last one to arrive acquires,
//     sums the partials in a fixed order (bitwise reproducible run to run), and ONE lane runs the 6x6 determinant test,
//     the Cholesky solve, the shake test and the pose update in place.  The host never sees the 27 floats.
// The sums are fp32 in a different association than the reference's tree, so parity for this stage is by tolerance
// (pose within 1e-4 m / 1e-4 rad), as SURVEY.md section 8a row a6 states.
#include "kf_internal.h"
#include <string.h>

#define TRK_THREADS 256

struct TrackArgs {
  const float4* new_v; const float4* new_n; const float4* model_v; const float4* model_n;
  KfCam cam;
  const float* cur_ptr; const float* linv_ptr;   // device-resident transforms, or null -> *_val
  KfMat cur_val, linv_val;
  float dist_thres, sin_thres, dist_shake, angle_shake;
  float* partials;                               // gridDim.x x 32
  KfTrackState* track;
  int solve;                                     // 1: fused Gauss-Newton step, 0: only leave the 27 sums in track->reduced
  int final_step;                                // 1: commit cur -> pose when the step succeeds
  // SDF tracker only
  KfVolume vol; const float* depth;
};

// ---- workgroup reduction + arrival ticket --------------------------------------------------------------------------
// Returns true in every thread of the LAST workgroup to arrive, after which s_tot[0..26] hold the grand totals.
__device__ __forceinline__ bool reduce27_and_arrive(float acc[27], float* partials, KfTrackState* track, float* s_wave /*[4][32]*/,
                                                    float* s_tot /*[8][32]*/, int* s_last) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    float s = kf_wave_sum(acc[k]);
    if (lane == 0) s_wave[wave * 32 + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    float s = ((s_wave[threadIdx.x] + s_wave[32 + threadIdx.x]) + s_wave[64 + threadIdx.x]) + s_wave[96 + threadIdx.x];
    partials[blockIdx.x * 32 + threadIdx.x] = s;
  }
  // publish: every storing wave drains its stores, the workgroup meets, one lane releases at agent scope, then the ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(&track->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!*s_last) return false;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    track->ticket = 0u;                                     // re-armed for the next launch (kernel boundary orders it)
  }
  __syncthreads();
  const int k = threadIdx.x & 31, part = threadIdx.x >> 5;  // 8 interleaved partial chains per sum, fixed order
  float s = 0.f;
  if (k < 27) for (unsigned w = part; w < gridDim.x; w += 8) s += partials[w * 32 + k];
  s_tot[part * 32 + k] = s;
  __syncthreads();
  if (threadIdx.x < 27) {
    float t = s_tot[threadIdx.x];
#pragma unroll
    for (int p = 1; p < 8; ++p) t += s_tot[p * 32 + threadIdx.x];
    s_tot[threadIdx.x] = t;
    track->reduced[threadIdx.x] = t;                        // rigid_align_buf_reduced
  }
  __syncthreads();
  return true;
}

// ---- 6x6 dense algebra on one lane (stands in for Eigen) ------------------------------------------------------------
// src/CameraPoseFinderICP.cpp:119-136
__device__ static void unpack27(const float* in, float A[36], float b[6]) {
  int s = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      float v = in[s++];
      if (j == 6) b[i] = v; else { A[i * 6 + j] = v; A[j * 6 + i] = v; }
    }
}
// determinant by partial-pivot LU (Eigen's path for a 6x6 `determinant()`), fp32
__device__ static float det6(const float A[36]) {
  float m[36];
  for (int i = 0; i < 36; ++i) m[i] = A[i];
  float det = 1.f;
  for (int k = 0; k < 6; ++k) {
    int p = k; float best = fabsf(m[k * 6 + k]);
    for (int r = k + 1; r < 6; ++r) { float v = fabsf(m[r * 6 + k]); if (v > best) { best = v; p = r; } }
    if (best == 0.f) return 0.f;
    if (p != k) { for (int c = 0; c < 6; ++c) { float t = m[k * 6 + c]; m[k * 6 + c] = m[p * 6 + c]; m[p * 6 + c] = t; } det = -det; }
    float piv = m[k * 6 + k];
    det *= piv;
    for (int r = k + 1; r < 6; ++r) {
      float f = m[r * 6 + k] / piv;
      for (int c = k + 1; c < 6; ++c) m[r * 6 + c] -= f * m[k * 6 + c];
    }
  }
  return det;
}
// x = LLT(A) \ b, fp32 (ICP.cpp:143, SDF.cpp:79)
__device__ static void llt_solve6(const float A[36], const float b[6], float x[6]) {
  float L[36];
  for (int i = 0; i < 36; ++i) L[i] = 0.f;
  for (int j = 0; j < 6; ++j) {
    float s = A[j * 6 + j];
    for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
    float d = sqrtf(s);
    L[j * 6 + j] = d;
    for (int i = j + 1; i < 6; ++i) {
      float t = A[i * 6 + j];
      for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = t / d;
    }
  }
  float y[6];
  for (int i = 0; i < 6; ++i) { float t = b[i]; for (int k = 0; k < i; ++k) t -= L[i * 6 + k] * y[k]; y[i] = t / L[i * 6 + i]; }
  for (int i = 5; i >= 0; --i) { float t = y[i]; for (int k = i + 1; k < 6; ++k) t -= L[k * 6 + i] * x[k]; x[i] = t / L[i * 6 + i]; }
}
__device__ static void mat3_mul(const float a[9], const float b[9], float o[9]) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
    o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}
// Mat.h:240-262
__device__ static void mat44_mul(const float* a, const float* b, float* out) {
  float r[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      r[i * 4 + j] = a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j] + a[i * 4 + 2] * b[8 + j] + a[i * 4 + 3] * b[12 + j];
  for (int i = 0; i < 16; ++i) out[i] = r[i];
}
// vector6ToTransformMatrix (ICP.cpp:95-111, SDF.cpp:25-43): R = Rx Ry Rz, shake test on the rotation angle and |t|
__device__ static bool vector6_to_transform(const float x[6], float dist_shake, float angle_shake, float t[16]) {
  float c0 = cosf(x[0]), s0 = sinf(x[0]), c1 = cosf(x[1]), s1 = sinf(x[1]), c2 = cosf(x[2]), s2 = sinf(x[2]);
  float Rx[9] = {1, 0, 0, 0, c0, -s0, 0, s0, c0};
  float Ry[9] = {c1, 0, s1, 0, 1, 0, -s1, 0, c1};
  float Rz[9] = {c2, -s2, 0, s2, c2, 0, 0, 0, 1};
  float Rxy[9], R[9];
  mat3_mul(Rx, Ry, Rxy); mat3_mul(Rxy, Rz, R);
  float ca = (R[0] + R[4] + R[8] - 1.f) * 0.5f;
  ca = fminf(1.f, fmaxf(-1.f, ca));
  float angle = acosf(ca);                                   // == AngleAxisf(R).angle()
  float d = sqrtf(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
  if (!(angle <= angle_shake) || !(d <= dist_shake)) return false;   // NaN counts as shaking: never applied
  float o[16] = {R[0], R[1], R[2], x[3], R[3], R[4], R[5], x[4], R[6], R[7], R[8], x[5], 0, 0, 0, 1};
  for (int i = 0; i < 16; ++i) t[i] = o[i];
  return true;
}

// ---- ICP ------------------------------------------------------------------------------------------------------------
// findCorrs (:17-60) + buildPointToPlaneSolverRows (:7-16) for one pixel
__device__ __forceinline__ bool icp_row(const TrackArgs& a, const float* cur, const float* linv, int x, int y, float row[7]) {
  const int cols = a.cam.cols, rows = a.cam.rows;
  const float4 iv = a.new_v[y * cols + x], in_ = a.new_n[y * cols + x];
  if (kf_is_zero4(in_)) return false;
  const float4 vg = kf_mat_vec(cur, iv);
  const float4 ng = kf_mat_vec(cur, in_);
  const float4 vcp = kf_mat_vec(linv, vg);
  const int2 sp = kf_project(kf3(vcp.x, vcp.y, vcp.z), a.cam);
  if (sp.x < 0 || sp.x >= cols || sp.y < 0 || sp.y >= rows) return false;
  const float4 nt = a.model_n[sp.y * cols + sp.x];
  if (kf_is_zero4(nt)) return false;
  const float4 vt = a.model_v[sp.y * cols + sp.x];
  const float d = kf_norm(kf3(vt.x - vg.x, vt.y - vg.y, vt.z - vg.z));
  const float s = kf_norm(kf_cross(kf3(nt.x, nt.y, nt.z), kf3(ng.x, ng.y, ng.z)));
  if (d > a.dist_thres || s > a.sin_thres) return false;
  const float3 p = kf3(vt.x, vt.y, vt.z), q = kf3(vg.x, vg.y, vg.z), n = kf3(nt.x, nt.y, nt.z);
  row[0] = q.y * n.z - q.z * n.y; row[1] = q.z * n.x - q.x * n.z; row[2] = q.x * n.y - q.y * n.x;
  row[3] = n.x; row[4] = n.y; row[5] = n.z;
  row[6] = kf_dot(n, kf_sub(p, q));
  return true;
}

__global__ void __launch_bounds__(TRK_THREADS) k_icp_step(TrackArgs a) {
  KfTrackState* st = a.track;
  if (a.solve && st->status != KF_TRACK_OK) return;         // an earlier step lost the camera: the loop has ended
  __shared__ float s_cur[16], s_linv[16];
  __shared__ float s_wave[4 * 32], s_tot[8 * 32];
  __shared__ int s_last;
  if (threadIdx.x < 16) s_cur[threadIdx.x] = a.cur_ptr ? a.cur_ptr[threadIdx.x] : a.cur_val.m[threadIdx.x];
  else if (threadIdx.x < 32) s_linv[threadIdx.x - 16] = a.linv_ptr ? a.linv_ptr[threadIdx.x - 16] : a.linv_val.m[threadIdx.x - 16];
  __syncthreads();
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  const int npx = a.cam.cols * a.cam.rows;
  for (int i = blockIdx.x * TRK_THREADS + threadIdx.x; i < npx; i += gridDim.x * TRK_THREADS) {
    float row[7];
    if (!icp_row(a, s_cur, s_linv, i % a.cam.cols, i / a.cam.cols, row)) continue;
    int s = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 7; ++c) acc[s++] += row[r] * row[c];            // :92-105 packing
  }
  if (!reduce27_and_arrive(acc, a.partials, st, s_wave, s_tot, &s_last)) return;
  if (!a.solve || threadIdx.x != 0) return;
  // minimizePointToPlaneErrFunc (ICP.cpp:117-143) + the loop body of estimateCameraPose (:71-82)
  float A[36], b[6], x[6], T[16];
  unpack27(s_tot, A, b);
  if ((double)det6(A) < 1E-10) { st->status = KF_TRACK_LOST_DET; st->tracked = 0; return; }
  llt_solve6(A, b, x);
  if (!vector6_to_transform(x, a.dist_shake, a.angle_shake, T)) { st->status = KF_TRACK_LOST_SHAKE; st->tracked = 0; return; }
  float ncur[16];
  mat44_mul(T, s_cur, ncur);                                               // cur = T * cur
  for (int i = 0; i < 16; ++i) st->cur[i] = ncur[i];
  st->iterations += 1;
  if (a.final_step) { for (int i = 0; i < 16; ++i) st->pose[i] = ncur[i]; st->tracked = 1; }
}

// ---- SDF tracker ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TRK_THREADS) k_sdf_step(TrackArgs a) {
  KfTrackState* st = a.track;
  if (a.solve && (st->status != KF_TRACK_OK || st->converged)) return;
  __shared__ float s_m[7][16];                 // cur, then delta*cur for +w1,-w1,+w2,-w2,+w3,-w3 (CalSDFErrSolverParams.cu:118-133)
  __shared__ float s_wave[4 * 32], s_tot[8 * 32];
  __shared__ int s_last;
  const float w_h = 0.001f;                                               // :119 `float w_h = 0.001;`
  const float v_h = a.vol.size / (float)a.vol.res;                        // :120
  if (threadIdx.x < 16) s_m[0][threadIdx.x] = a.cur_ptr ? a.cur_ptr[threadIdx.x] : a.cur_val.m[threadIdx.x];
  __syncthreads();
  if (threadIdx.x < 96) {
    const int mi = threadIdx.x >> 4, e = threadIdx.x & 15, r = e >> 2, cidx = e & 3;
    // (row, col) pairs that carry -/+ w_h for axis mi/2 ; second matrix of a pair flips the sign
    const int axis = mi >> 1; const float sg = (mi & 1) ? -1.f : 1.f;
    float d[4] = {0.f, 0.f, 0.f, 0.f}; d[r] = 1.f;
    if (axis == 0) { if (r == 1) d[2] = -sg * w_h; if (r == 2) d[1] = sg * w_h; }          // m23 = -w, m32 = +w
    else if (axis == 1) { if (r == 0) d[2] = sg * w_h; if (r == 2) d[0] = -sg * w_h; }     // m13 = +w, m31 = -w
    else { if (r == 0) d[1] = -sg * w_h; if (r == 1) d[0] = sg * w_h; }                    // m12 = -w, m21 = +w
    s_m[1 + mi][e] = d[0] * s_m[0][cidx] + d[1] * s_m[0][4 + cidx] + d[2] * s_m[0][8 + cidx] + d[3] * s_m[0][12 + cidx];
  }
  __syncthreads();
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  const int npx = a.cam.cols * a.cam.rows;
  for (int i = blockIdx.x * TRK_THREADS + threadIdx.x; i < npx; i += gridDim.x * TRK_THREADS) {
    const float d = a.depth[i];
    if (d == 0.f) continue;
    const float3 p = kf_depth_to_skeleton((unsigned)(i % a.cam.cols), (unsigned)(i / a.cam.cols), d, a.cam);
    const float4 p4 = make_float4(p.x, p.y, p.z, 1.0f);
    // buildSDFSolverRows (:7-66): all 13 lookups must succeed
    bool ok = true;
    float sdf0, sw[6], sv[6];
    const float4 pw0 = kf_mat_vec(s_m[0], p4);
    ok &= kf_interpolate_sdf(a.vol, kf3(pw0.x, pw0.y, pw0.z), sdf0);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (!ok) break;
      const float4 pr = kf_mat_vec(s_m[1 + k], p4);
      ok &= kf_interpolate_sdf(a.vol, kf3(pr.x, pr.y, pr.z), sw[k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (!ok) break;
      float3 q = kf3(pw0.x, pw0.y, pw0.z);
      if ((k >> 1) == 0) q.x = (k & 1) ? pw0.x - v_h : pw0.x + v_h;
      else if ((k >> 1) == 1) q.y = (k & 1) ? pw0.y - v_h : pw0.y + v_h;
      else q.z = (k & 1) ? pw0.z - v_h : pw0.z + v_h;
      ok &= kf_interpolate_sdf(a.vol, q, sv[k]);
    }
    if (!ok) continue;
    float row[7];
    row[0] = (sw[0] - sw[1]) / (2 * w_h); row[1] = (sw[2] - sw[3]) / (2 * w_h); row[2] = (sw[4] - sw[5]) / (2 * w_h);
    row[3] = (sv[0] - sv[1]) / (2 * v_h); row[4] = (sv[2] - sv[3]) / (2 * v_h); row[5] = (sv[4] - sv[5]) / (2 * v_h);
    row[6] = sdf0;
    int s = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 7; ++c) acc[s++] += row[r] * row[c];
  }
  if (!reduce27_and_arrive(acc, a.partials, st, s_wave, s_tot, &s_last)) return;
  if (!a.solve || threadIdx.x != 0) return;
  // CameraPoseFinderSDF::estimateCameraPose loop body (SDF.cpp:62-101)
  float A[36], b[6], x[6], T[16];
  unpack27(s_tot, A, b);
  llt_solve6(A, b, x);                                                     // no determinant test (:79)
  if (!vector6_to_transform(x, a.dist_shake, a.angle_shake, T)) { st->status = KF_TRACK_LOST_SHAKE; st->tracked = 0; return; }
  const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
  const float* cur = s_m[0];
  if (nx < 0.001f) {                                                       // :87-90 stop before applying x
    st->converged = 1;
    for (int i = 0; i < 16; ++i) st->pose[i] = cur[i];
    st->tracked = 1;
    return;
  }
  // direct_exponential_map (eigen_utils.cpp:84-127) in double
  const double u0 = (double)x[0], u1 = (double)x[1], u2 = (double)x[2], t3 = (double)x[3], t4 = (double)x[4], t5 = (double)x[5];
  const double theta = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
  const double si = sin(theta), co = cos(theta);
  const double sinc = fabs(theta) < 1.0e-8 ? 1.0 : si / theta;
  const double mcosc = fabs(theta) < 2.5e-4 ? 0.5 : (1.0 - co) / theta / theta;
  const double msinc = fabs(theta) < 2.5e-4 ? (1. / 6.0) : (1.0 - si / theta) / theta / theta;
  double R[9];
  R[0] = co + mcosc * u0 * u0;        R[1] = -sinc * u2 + mcosc * u0 * u1; R[2] = sinc * u1 + mcosc * u0 * u2;
  R[3] = sinc * u2 + mcosc * u1 * u0; R[4] = co + mcosc * u1 * u1;         R[5] = -sinc * u0 + mcosc * u1 * u2;
  R[6] = -sinc * u1 + mcosc * u2 * u0; R[7] = sinc * u0 + mcosc * u2 * u1; R[8] = co + mcosc * u2 * u2;
  double dt[3];
  dt[0] = t3 * (sinc + u0 * u0 * msinc) + t4 * (u0 * u1 * msinc - u2 * mcosc) + t5 * (u0 * u2 * msinc + u1 * mcosc);
  dt[1] = t3 * (u0 * u1 * msinc + u2 * mcosc) + t4 * (sinc + u1 * u1 * msinc) + t5 * (u1 * u2 * msinc - u0 * mcosc);
  dt[2] = t3 * (u0 * u2 * msinc - u1 * mcosc) + t4 * (u1 * u2 * msinc + u0 * mcosc) + t5 * (sinc + u2 * u2 * msinc);
  // SDF.cpp:96-97: R' = R_exp^T * R_cur ; t' = t_cur - R_exp^T * t_exp  (fp32 after the casts)
  float Rt[9], tf[3], ncur[16];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = (float)R[c * 3 + r];
  for (int k = 0; k < 3; ++k) tf[k] = (float)dt[k];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) ncur[r * 4 + c] = Rt[r * 3] * cur[c] + Rt[r * 3 + 1] * cur[4 + c] + Rt[r * 3 + 2] * cur[8 + c];
    ncur[r * 4 + 3] = cur[r * 4 + 3] - (Rt[r * 3] * tf[0] + Rt[r * 3 + 1] * tf[1] + Rt[r * 3 + 2] * tf[2]);
  }
  ncur[12] = 0.f; ncur[13] = 0.f; ncur[14] = 0.f; ncur[15] = 1.f;
  for (int i = 0; i < 16; ++i) st->cur[i] = ncur[i];
  st->iterations += 1;
  if (a.final_step) { for (int i = 0; i < 16; ++i) st->pose[i] = ncur[i]; st->tracked = 1; }
}

// ---- loop control ---------------------------------------------------------------------------------------------------
// mode 0: frame 0 (no tracking, "tracked"); mode 1: start of a Gauss-Newton loop
__global__ void k_track_begin(KfTrackState* st, int mode) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->status = KF_TRACK_OK; st->iterations = 0; st->converged = 0; st->ticket = 0u;
  if (mode == 0) { st->tracked = 1; return; }
  st->tracked = 0;
  for (int i = 0; i < 16; ++i) st->cur[i] = st->pose[i];                    // ICP.cpp:62
  kf_mat44_inverse(st->pose, st->last_inv);                                // ICP.cpp:63 last_transform_inv = _pose.getInverse()
}

// SDF loop that ran out of iterations without converging still commits (SDF.cpp:103-105)
__global__ void k_sdf_end(KfTrackState* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (st->status == KF_TRACK_OK && !st->tracked) { for (int i = 0; i < 16; ++i) st->pose[i] = st->cur[i]; st->tracked = 1; }
}

static inline KfCam to_cam(const kf_camera_params* p) {
  KfCam c; c.cols = (int)p->cols; c.rows = (int)p->rows; c.cx = p->cx; c.cy = p->cy; c.fx = p->fx; c.fy = p->fy; return c;
}
static inline int track_grid(int npx) {
  int g = kf_div_up(npx, TRK_THREADS * 4);
  return g < 1 ? 1 : (g > KF_ICP_MAX_WG ? KF_ICP_MAX_WG : g);
}

extern "C" int kf_set_pose(kf_ctx* c, const kf_mat44* pose) {
  if (!c || !pose) return KF_ERR_ARG;
  memcpy(c->host_pinned, pose->m, 64);
  KF_CHECK(hipMemcpyAsync(c->track->pose, c->host_pinned, 64, hipMemcpyHostToDevice, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));     // the pinned staging word is reused
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
  a.dist_thres = dist_thres; a.sin_thres = sin_thres;
  a.partials = c->icp_partials; a.track = c->track; a.solve = 0; a.final_step = 0;
  hipLaunchKernelGGL(k_icp_step, dim3(track_grid(a.cam.cols * a.cam.rows)), dim3(TRK_THREADS), 0, c->stream, a);
  return (int)hipGetLastError();
}

extern "C" int kf_cal_sdf_solver_params(kf_ctx* c, const kf_camera_params* cam, const kf_mat44* cur) {
  if (!c || !cam || !cur) return KF_ERR_ARG;
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  TrackArgs a; memset(&a, 0, sizeof(a));
  a.cam = to_cam(cam); a.vol = c->vol; a.depth = c->trunced_depth;
  for (int i = 0; i < 16; ++i) a.cur_val.m[i] = cur->m[i];
  a.partials = c->icp_partials; a.track = c->track; a.solve = 0;
  hipLaunchKernelGGL(k_sdf_step, dim3(track_grid(c->cols * c->rows)), dim3(TRK_THREADS), 0, c->stream, a);
  return (int)hipGetLastError();
}

extern "C" int kf_read_solver_params(kf_ctx* c, float out27[27]) {
  if (!c || !out27) return KF_ERR_ARG;
  KF_CHECK(hipMemcpyAsync(c->host_pinned, c->track->reduced, 27 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  memcpy(out27, c->host_pinned, 27 * sizeof(float));
  return 0;
}

extern "C" int kf_icp_track(kf_ctx* c, uint32_t frame_id, const kf_icp_params* icp, const kf_camera_params* cam0) {
  if (!c || !icp || !cam0) return KF_ERR_ARG;
  if ((int)icp->pyramid_levels != c->levels || (int)cam0->cols != c->cols || (int)cam0->rows != c->rows) return KF_ERR_ARG;
  if (frame_id == 0) {                                       // ICP.cpp:52-55
    hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, 0);
    return (int)hipGetLastError();
  }
  int iters[KF_MAX_LEVELS] = {0, 0, 0};                      // ICP.cpp:14-35
  if (c->levels == 1) iters[0] = 3; else if (c->levels == 2) { iters[0] = 10; iters[1] = 5; } else { iters[0] = 10; iters[1] = 5; iters[2] = 4; }
  int st;
  kf_evt_begin(c, KF_STAGE_TRACK);
  if ((st = kf_launch_pyramids(c, false, true, true))) return st;       // ICP.cpp:57-60
  if ((st = kf_launch_pyramids(c, true, true, true))) return st;
  hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, 1);
  kf_camera_params cams[KF_MAX_LEVELS]; cams[0] = *cam0;
  for (int l = 1; l < c->levels; ++l) {                      // ICP.cpp:36-48
    cams[l].cols = cams[l - 1].cols / 2; cams[l].rows = cams[l - 1].rows / 2;
    cams[l].cx = cams[l - 1].cx / 2; cams[l].cy = cams[l - 1].cy / 2; cams[l].fx = cams[l - 1].fx / 2; cams[l].fy = cams[l - 1].fy / 2;
  }
  for (int l = c->levels - 1; l >= 0; --l)
    for (int it = 0; it < iters[l]; ++it) {
      TrackArgs a; memset(&a, 0, sizeof(a));
      a.new_v = c->new_v[l]; a.new_n = c->new_n[l]; a.model_v = c->model_v[l]; a.model_n = c->model_n[l];
      a.cam = to_cam(&cams[l]);
      a.cur_ptr = c->track->cur; a.linv_ptr = c->track->last_inv;
      a.dist_thres = icp->dist_thres; a.sin_thres = icp->norm_sin_thres; a.dist_shake = icp->dist_shake; a.angle_shake = icp->angle_shake;
      a.partials = c->icp_partials; a.track = c->track; a.solve = 1; a.final_step = (l == 0 && it == iters[0] - 1) ? 1 : 0;
      hipLaunchKernelGGL(k_icp_step, dim3(track_grid(a.cam.cols * a.cam.rows)), dim3(TRK_THREADS), 0, c->stream, a);
    }
  kf_evt_end(c, KF_STAGE_TRACK);
  return (int)hipGetLastError();
}

extern "C" int kf_sdf_track(kf_ctx* c, uint32_t frame_id, const kf_sdf_tracker_params* sp, const kf_camera_params* cam) {
  if (!c || !sp || !cam) return KF_ERR_ARG;
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  if (frame_id == 0) {                                       // SDF.cpp:46-49
    hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, 0);
    return (int)hipGetLastError();
  }
  kf_evt_begin(c, KF_STAGE_TRACK);
  hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, c->stream, c->track, 1);
  for (uint32_t it = 0; it < sp->max_iter_nums; ++it) {
    TrackArgs a; memset(&a, 0, sizeof(a));
    a.cam = to_cam(cam); a.vol = c->vol; a.depth = c->trunced_depth;
    a.cur_ptr = c->track->cur;
    a.dist_shake = sp->dist_shake; a.angle_shake = sp->angle_shake;
    a.partials = c->icp_partials; a.track = c->track; a.solve = 1; a.final_step = (it + 1 == sp->max_iter_nums) ? 1 : 0;
    hipLaunchKernelGGL(k_sdf_step, dim3(track_grid(c->cols * c->rows)), dim3(TRK_THREADS), 0, c->stream, a);
  }
  hipLaunchKernelGGL(k_sdf_end, dim3(1), dim3(64), 0, c->stream, c->track);
  kf_evt_end(c, KF_STAGE_TRACK);
  return (int)hipGetLastError();
}

extern "C" int kf_read_track_result(kf_ctx* c, kf_track_result* out) {
  if (!c || !out) return KF_ERR_ARG;
  KfTrackState* h = (KfTrackState*)c->host_pinned;
  KF_CHECK(hipMemcpyAsync(h, c->track, sizeof(KfTrackState), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  memcpy(out->pose.m, h->pose, 64);
  out->tracked = h->tracked; out->status = h->status; out->iterations = h->iterations; out->reserved = 0;
  return 0;
}
