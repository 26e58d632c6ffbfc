/*
 * kf_oracle.cpp -- CPU restatement of the reference's per-frame path (see kf_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product library.
 *
 * Build: g++ -O2 -ffp-contract=off -fopenmp (oracle/Makefile).  Every fp32 source operation of the
 * reference is one rounded fp32 operation here; double promotions caused by unsuffixed literals in the
 * reference are reproduced explicitly and commented.
 *
 * How it is pinned (details in DESIGN.md section 2.2): the header arithmetic (Mat44, voxel <-> world, nearest voxel, trilinear
 * lookup, updateVoxel, camera projection) is checked bit for bit against the reference's own headers compiled as they lie
 * (oracle/ref_harness.cpp -> oracle/_ref/libkfref.so, tests/test_oracle_vs_ref.py); the integrate / raycast / marching-cubes
 * bodies reproduce the numbers the reference's kernels produced in the survey session (SURVEY.md section 8c,
 * tests/test_oracle_golden.py).  PARITY UNPINNED for: the bilateral filter body, the ICP / SDF row builders and the host
 * Gauss-Newton solves (Eigen is not vendored: LLT, determinant and AngleAxis are restated by hand) -- closed-form tests only.
 */
#include "kf_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <omp.h>

namespace {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct i3 { int x, y, z; };

inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
inline f4 mk4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
inline f4 ld4(const float* p, int idx) { return mk4(p[4 * idx], p[4 * idx + 1], p[4 * idx + 2], p[4 * idx + 3]); }
inline void st4(float* p, int idx, f4 v) { p[4 * idx] = v.x; p[4 * idx + 1] = v.y; p[4 * idx + 2] = v.z; p[4 * idx + 3] = v.w; }

/* src/cuda/cuda_declar.h:29-105 */
inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline f3 mul3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float norm3(f3 v) { return sqrtf(dot3(v, v)); }
/* cuda_declar.h:89-94: `v*(1.0/len)` -- reciprocal in double, narrowed to float, then fp32 multiply */
inline f3 normalize3(f3 v) {
  float len = norm3(v);
  if (len < 1e-8) return mk3(0, 0, 0);        /* float vs double literal compare */
  float r = (float)(1.0 / (double)len);
  return mul3(v, r);
}
inline bool is_zero4(f4 v) { return v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0; }

/* src/cuda/Mat.h:230-238: row-major, left-to-right sums */
inline f4 mat_vec(const float* m, f4 v) {
  return mk4(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w,
             m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * v.w,
             m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * v.w,
             m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w);
}

/* (int) of a double as the reference's CUDA path does it (cvt.rzi.s32.f64): truncation, SATURATING, NaN -> 0.  The north
 * star names the CUDA path as the one to match; a host build of the same headers (x86-64 cvttsd2si) would return INT_MIN for
 * NaN / out-of-range instead.  The two differ only for NaN (0 vs INT_MIN: findCorrs' `>= 0` bounds test,
 * CalPointToPlaneErrSolverParams.cu:39-43, accepts pixel 0) and for +overflow (INT_MAX vs INT_MIN: rejected by every bounds
 * test either way).  gfx950's v_cvt_i32_f32 / v_cvt_i32_f64 have the CUDA semantics natively (kf_internal.h: kf_to_int). */
inline int to_int(double v) {
  if (v != v) return 0;
  if (v >= 2147483648.0) return 2147483647;
  if (v <= -2147483649.0) return (int)0x80000000;
  return (int)v;
}
inline int to_int_f(float v) { return to_int((double)v); }

/* src/cuda/DepthCamera.h:19-29 */
inline f3 depth_to_skeleton(unsigned ux, unsigned uy, float depth, const okf_cam& c) {
  float z = depth;
  float vx = z * ((float)ux - c.cx) / c.fx;
  float vy = z * ((float)uy - c.cy) / c.fy;
  return mk3(vx, vy, z);
}
/* src/cuda/DepthCamera.h:30-43: `(int)(pImage.x+0.5)` adds a double literal */
inline void project_to_screen(f3 v, const okf_cam& c, int& sx, int& sy) {
  float px = v.x * c.fx / v.z + c.cx;
  float py = v.y * c.fy / v.z + c.cy;
  sx = to_int((double)px + 0.5);
  sy = to_int((double)py + 0.5);
}

/* ---- volume helpers: src/cuda/tsdfVolume.h ---- */
static uint64_t g_band_violations = 0;
static const okf_voxel g_unobserved = {0.f, 0.f, {0, 0, 0}, 0};
inline int first_layer(const okf_volume* v) { return v->z_layers ? v->z_base : 0; }
inline int stored_layers(const okf_volume* v) { return v->z_layers ? v->z_layers : v->res; }
inline const okf_voxel& vox_at(const okf_volume* v, int x, int y, int z) {
  const int zl = z - first_layer(v);
  if ((unsigned)zl >= (unsigned)stored_layers(v)) {       /* only a z-band volume can get here */
    __atomic_fetch_add(&g_band_violations, (uint64_t)1, __ATOMIC_RELAXED);
    return g_unobserved;
  }
  return v->data[((size_t)zl * v->res + y) * v->res + x];
}
/* tsdfVolume.h:38-49 */
inline f3 voxel_to_world(const okf_volume* v, int x, int y, int z) {
  f3 c = mk3((float)x, (float)y, (float)z);
  c.x += 0.5f; c.y += 0.5f; c.z += 0.5f;
  float cell = v->size / (float)v->res;
  c.x *= cell; c.y *= cell; c.z *= cell;
  return c;
}
/* tsdfVolume.h:50-56 */
inline i3 world_to_voxel(const okf_volume* v, f3 p) {
  i3 r;
  r.x = to_int_f(p.x * (float)v->res / v->size);
  r.y = to_int_f(p.y * (float)v->res / v->size);
  r.z = to_int_f(p.z * (float)v->res / v->size);
  return r;
}
/* tsdfVolume.h:81-97: nearest voxel, index clamped */
inline const okf_voxel& voxel_nearest(const okf_volume* v, f3 p) {
  i3 g = world_to_voxel(v, p);
  int R = v->res;
  g.x = g.x < 0 ? 0 : (g.x > R - 1 ? R - 1 : g.x);
  g.y = g.y < 0 ? 0 : (g.y > R - 1 ? R - 1 : g.y);
  g.z = g.z < 0 ? 0 : (g.z > R - 1 ? R - 1 : g.z);
  return vox_at(v, g.x, g.y, g.z);
}
/* tsdfVolume.h:151-172 */
inline bool interp_params(const okf_volume* v, f3 pos, i3& base, float& a, float& b, float& c) {
  i3 g = world_to_voxel(v, pos);
  int R = v->res;
  if (g.x <= 0 || g.x >= R - 1) return false;
  if (g.y <= 0 || g.y >= R - 1) return false;
  if (g.z <= 0 || g.z >= R - 1) return false;
  float cell = v->size / (float)R;
  float vx = ((float)g.x + 0.5f) * cell;
  float vy = ((float)g.y + 0.5f) * cell;
  float vz = ((float)g.z + 0.5f) * cell;
  g.x = (pos.x < vx) ? (g.x - 1) : g.x;
  g.y = (pos.y < vy) ? (g.y - 1) : g.y;
  g.z = (pos.z < vz) ? (g.z - 1) : g.z;
  base = g;
  a = (pos.x - ((float)g.x + 0.5f) * cell) / cell;
  b = (pos.y - ((float)g.y + 0.5f) * cell) / cell;
  c = (pos.z - ((float)g.z + 0.5f) * cell) / cell;
  return true;
}
/* tsdfVolume.h:98-122 */
inline bool interpolate_sdf(const okf_volume* v, f3 pos, float& dist) {
  i3 g; float a, b, c;
  if (!interp_params(v, pos, g, a, b, c)) return false;
  const okf_voxel& v000 = vox_at(v, g.x, g.y, g.z);             if (v000.weight == 0) return false;
  const okf_voxel& v001 = vox_at(v, g.x, g.y, g.z + 1);         if (v001.weight == 0) return false;
  const okf_voxel& v010 = vox_at(v, g.x, g.y + 1, g.z);         if (v010.weight == 0) return false;
  const okf_voxel& v011 = vox_at(v, g.x, g.y + 1, g.z + 1);     if (v011.weight == 0) return false;
  const okf_voxel& v100 = vox_at(v, g.x + 1, g.y, g.z);         if (v100.weight == 0) return false;
  const okf_voxel& v101 = vox_at(v, g.x + 1, g.y, g.z + 1);     if (v101.weight == 0) return false;
  const okf_voxel& v110 = vox_at(v, g.x + 1, g.y + 1, g.z);     if (v110.weight == 0) return false;
  const okf_voxel& v111 = vox_at(v, g.x + 1, g.y + 1, g.z + 1); if (v111.weight == 0) return false;
  float ia = 1 - a, ib = 1 - b, ic = 1 - c;
  dist = v000.tsdf * ia * ib * ic +
         v001.tsdf * ia * ib * c +
         v010.tsdf * ia * b * ic +
         v011.tsdf * ia * b * c +
         v100.tsdf * a * ib * ic +
         v101.tsdf * a * ib * c +
         v110.tsdf * a * b * ic +
         v111.tsdf * a * b * c;
  return true;
}
/* tsdfVolume.h:123-148; float->uchar conversion truncates */
inline bool interpolate_color(const okf_volume* v, f3 pos, uint8_t out[3]) {
  i3 g; float a, b, c;
  if (!interp_params(v, pos, g, a, b, c)) return false;
  const okf_voxel* q[8] = {
    &vox_at(v, g.x, g.y, g.z), &vox_at(v, g.x, g.y, g.z + 1), &vox_at(v, g.x, g.y + 1, g.z), &vox_at(v, g.x, g.y + 1, g.z + 1),
    &vox_at(v, g.x + 1, g.y, g.z), &vox_at(v, g.x + 1, g.y, g.z + 1), &vox_at(v, g.x + 1, g.y + 1, g.z), &vox_at(v, g.x + 1, g.y + 1, g.z + 1)};
  for (int k = 0; k < 8; ++k) if (q[k]->weight == 0) return false;
  float ia = 1 - a, ib = 1 - b, ic = 1 - c;
  float wa[8] = {ia, ia, ia, ia, a, a, a, a};
  float wb[8] = {ib, ib, b, b, ib, ib, b, b};
  float wc[8] = {ic, c, ic, c, ic, c, ic, c};
  for (int ch = 0; ch < 3; ++ch) {
    float acc = 0;
    for (int k = 0; k < 8; ++k) {
      float term = (float)q[k]->color[ch] * wa[k] * wb[k] * wc[k];
      acc = (k == 0) ? term : acc + term;
    }
    out[ch] = (uint8_t)acc;
  }
  return true;
}

/* tests only, see okf_set_perturbation below; 0 = the restatement proper */
int g_perturb = 0;
/* a quotient as a product with the divisor's rounded reciprocal: at most an ulp or so off the IEEE quotient (perturbation bit 3) */
inline float quot(float a, float b) { return (g_perturb & 8) ? a * (1.0f / b) : a / b; }

/* ---- small dense linear algebra in fp32 (stands in for Eigen, which the image lacks) ---- */
/* determinant by LU with partial pivoting (Eigen's path for 6x6: PartialPivLU) */
float det6(const float A[36]) {
  float m[36]; memcpy(m, A, sizeof(m));
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
/* Cholesky A = L L^T, solve; like Eigen LLT it does not fail on a non-positive pivot (result is then NaN) */
void llt_solve6(const float A[36], const float b[6], float x[6]) {
  float L[36]; memset(L, 0, sizeof(L));
  for (int j = 0; j < 6; ++j) {
    float s = A[j * 6 + j];
    for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
    float d = sqrtf(s);
    L[j * 6 + j] = d;
    for (int i = j + 1; i < 6; ++i) {
      float t = A[i * 6 + j];
      for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = quot(t, d);
    }
  }
  float y[6];
  for (int i = 0; i < 6; ++i) { float t = b[i]; for (int k = 0; k < i; ++k) t -= L[i * 6 + k] * y[k]; y[i] = quot(t, L[i * 6 + i]); }
  for (int i = 5; i >= 0; --i) { float t = y[i]; for (int k = i + 1; k < 6; ++k) t -= L[k * 6 + i] * x[k]; x[i] = quot(t, L[i * 6 + i]); }
}
/* 27 packed sums -> symmetric 6x6 + rhs : src/CameraPoseFinderICP.cpp:119-136 */
void unpack27(const float in[27], float A[36], float b[6]) {
  int s = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      float v = in[s++];
      if (j == 6) b[i] = v; else { A[i * 6 + j] = v; A[j * 6 + i] = v; }
    }
}
void mat3_mul(const float a[9], const float b[9], float o[9]) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
    o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}

const uint64_t kTriWords[256] = {
#include "mc_tables.inc"
};

}  // namespace

extern "C" {

/* Last-bit perturbations of the tracker's inputs (tests only: how far does the oracle diverge from a 1-ulp-perturbed copy of
 * itself over a sequence?  tests/test_tracking_floor.py).  bit 0: the 27 ICP / SDF sums are accumulated in the reversed pixel order;
 * bit 1: bilateral weights by exp2f(x * log2(e)) instead of expf(x); bit 2: the products of a row are accumulated with fused
 * multiply-adds; bit 3: the Cholesky solve's quotients as products with a rounded reciprocal.  0 (default) = the restatement proper. */
void okf_set_perturbation(int mode) { g_perturb = mode; }

int okf_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  int got = 1;
#pragma omp parallel
  {
#pragma omp single
    got = omp_get_num_threads();
  }
  return got;
}

/* a1: src/HybKinectfu.cpp:73 `float v = mat.at<ushort>(r,c)*0.001;` -- int*double, narrowed to float */
void okf_depth_mm_to_m(const uint16_t* mm, int n, float* out) {
  for (int i = 0; i < n; ++i) out[i] = (float)((double)mm[i] * 0.001);
}

/* a2: DataPreprocesser.cu:25-33 */
void okf_trunc_depth(const float* in, int cols, int rows, float tmin, float tmax, float* out) {
  int n = cols * rows;
  for (int i = 0; i < n; ++i) { float d = in[i]; out[i] = (d < tmax && d > tmin) ? d : 0.f; }
}

/* a3: DataPreprocesser.cu:37-79 ; wrapper constants :95-96 (0.5 is a double literal) */
void okf_bilateral(const float* in, int cols, int rows, float sigma_pixel, float sigma_depth, float* out) {
  float sd_inv = (float)(0.5 / (double)(sigma_depth * sigma_depth));
  float ss_inv = (float)(0.5 / (double)(sigma_pixel * sigma_pixel));
  const int radius = (int)ceil(2.0 * (double)sigma_pixel);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float value = in[y * cols + x];
      out[y * cols + x] = value;
      if (value == 0) continue;
      int xs = x - radius < 0 ? 0 : x - radius, xe = x + radius > cols - 1 ? cols - 1 : x + radius;
      int ys = y - radius < 0 ? 0 : y - radius, ye = y + radius > rows - 1 ? rows - 1 : y + radius;
      float sum1 = 0, sum2 = 0; bool abort_px = false;
      for (int cy = ys; cy <= ye && !abort_px; ++cy)
        for (int cx = xs; cx <= xe; ++cx) {
          float tmp = in[cy * cols + cx];
          if (tmp == 0) continue;
          if (fabsf(tmp - value) > 5 * sigma_depth) { abort_px = true; break; }   /* :66-69 keeps the raw value */
          float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
          float data2 = (value - tmp) * (value - tmp);
          float w = (g_perturb & 2) ? exp2f(-(space2 * ss_inv + data2 * sd_inv) * 1.44269504f)
                                    : expf(-(space2 * ss_inv + data2 * sd_inv));     /* __expf on the device */
          sum1 += tmp * w; sum2 += w;
        }
      if (abort_px) continue;
      if (sum2 > 0) out[y * cols + x] = sum1 / sum2;
    }
}

/* a4: VerticesNormalsCalculater.cu:15-33 */
void okf_depth_to_vertices(const float* depth, const okf_cam* cam, float* v4) {
  int cols = cam->cols, rows = cam->rows;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float d = depth[y * cols + x];
      if (d == 0) st4(v4, y * cols + x, mk4(0, 0, 0, 0));
      else { f3 v = depth_to_skeleton(x, y, d, *cam); st4(v4, y * cols + x, mk4(v.x, v.y, v.z, 1.0f)); }
    }
}
/* a4: VerticesNormalsCalculater.cu:35-66 */
void okf_vertices_to_normals(const float* v4, int cols, int rows, float* n4) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int i = y * cols + x;
      st4(n4, i, mk4(0, 0, 0, 0));
      if (x == cols - 1 || y == rows - 1 || x == 0 || y == 0) continue;
      f4 v0 = ld4(v4, i); if (v0.z == 0) continue;
      f4 r = ld4(v4, i + 1); if (r.z == 0) continue;
      f4 u = ld4(v4, i + cols); if (u.z == 0) continue;
      f4 l = ld4(v4, i - 1); if (l.z == 0) continue;
      f4 d = ld4(v4, i - cols); if (d.z == 0) continue;
      f3 c = normalize3(cross3(sub3(mk3(u.x, u.y, u.z), mk3(d.x, d.y, d.z)), sub3(mk3(r.x, r.y, r.z), mk3(l.x, l.y, l.z))));
      st4(n4, i, mk4(c.x, c.y, c.z, 0));
    }
}

/* a5: sample.cu:37-61 -- `*0.25` is a double literal narrowed by operator*(float4,const float&) */
void okf_pyrdown_vertices(const float* in4, int in_cols, int in_rows, float* out4) {
  int oc = in_cols / 2, orow = in_rows / 2;
  for (int y = 0; y < orow; ++y)
    for (int x = 0; x < oc; ++x) {
      f4 p00 = ld4(in4, (2 * y) * in_cols + 2 * x), p01 = ld4(in4, (2 * y) * in_cols + 2 * x + 1);
      f4 p10 = ld4(in4, (2 * y + 1) * in_cols + 2 * x), p11 = ld4(in4, (2 * y + 1) * in_cols + 2 * x + 1);
      if (p00.z == 0 || p01.z == 0 || p10.z == 0 || p11.z == 0) { st4(out4, y * oc + x, mk4(0, 0, 0, 0)); continue; }
      float q = 0.25f;
      st4(out4, y * oc + x, mk4((p00.x + p01.x + p10.x + p11.x) * q, (p00.y + p01.y + p10.y + p11.y) * q,
                                (p00.z + p01.z + p10.z + p11.z) * q, (p00.w + p01.w + p10.w + p11.w) * q));
    }
}
/* a5: sample.cu:16-36 */
void okf_pyrdown_normals(const float* in4, int in_cols, int in_rows, float* out4) {
  int oc = in_cols / 2, orow = in_rows / 2;
  for (int y = 0; y < orow; ++y)
    for (int x = 0; x < oc; ++x) {
      st4(out4, y * oc + x, mk4(0, 0, 0, 0));
      f4 p00 = ld4(in4, (2 * y) * in_cols + 2 * x), p01 = ld4(in4, (2 * y) * in_cols + 2 * x + 1);
      f4 p10 = ld4(in4, (2 * y + 1) * in_cols + 2 * x), p11 = ld4(in4, (2 * y + 1) * in_cols + 2 * x + 1);
      if (is_zero4(p01) || is_zero4(p10) || is_zero4(p00) || is_zero4(p11)) continue;
      float q = 0.25f;
      f3 n = mk3((p00.x + p01.x + p10.x + p11.x) * q, (p00.y + p01.y + p10.y + p11.y) * q, (p00.z + p01.z + p10.z + p11.z) * q);
      f3 nn = normalize3(n);
      st4(out4, y * oc + x, mk4(nn.x, nn.y, nn.z, 0));
    }
}

/* a6: findCorrs (CalPointToPlaneErrSolverParams.cu:17-60) + buildPointToPlaneSolverRows (:7-16) */
static bool icp_row(int x, int y, const float* new_v, const float* new_n, const float* model_v, const float* model_n,
                    const okf_cam& cam, const float* cur, const float* last_inv, float dist_thres, float sin_thres, float row[7]) {
  int cols = cam.cols, rows = cam.rows;
  f4 iv = ld4(new_v, y * cols + x), in_ = ld4(new_n, y * cols + x);
  if (is_zero4(in_)) return false;
  f4 vg = mat_vec(cur, iv);
  f4 ng = mat_vec(cur, in_);
  f4 vcp = mat_vec(last_inv, vg);
  int sx, sy;
  project_to_screen(mk3(vcp.x, vcp.y, vcp.z), cam, sx, sy);
  if (sx < 0 || sx >= cols || sy < 0 || sy >= rows) return false;
  f4 nt = ld4(model_n, sy * cols + sx);
  if (is_zero4(nt)) return false;
  f4 vt = ld4(model_v, sy * cols + sx);
  f3 delta = mk3(vt.x - vg.x, vt.y - vg.y, vt.z - vg.z);
  float d = norm3(delta);
  float s = norm3(cross3(mk3(nt.x, nt.y, nt.z), mk3(ng.x, ng.y, ng.z)));
  if (d > dist_thres || s > sin_thres) return false;
  f3 p = mk3(vt.x, vt.y, vt.z), q = mk3(vg.x, vg.y, vg.z), n = mk3(nt.x, nt.y, nt.z);
  row[0] = q.y * n.z - q.z * n.y;
  row[1] = q.z * n.x - q.x * n.z;
  row[2] = q.x * n.y - q.y * n.x;
  row[3] = n.x; row[4] = n.y; row[5] = n.z;
  row[6] = dot3(n, sub3(p, q));
  return true;
}

void okf_icp_system(const float* new_v, const float* new_n, const float* model_v, const float* model_n,
                    const okf_cam* cam, const float cur[16], const float last_inv[16],
                    float dist_thres, float sin_thres, double* out27d, float* out27f, int* valid) {
  int cols = cam->cols, rows = cam->rows;
  /* per-row partial sums (rows run in parallel), combined in row order: deterministic for any thread count */
  double* rowd = (double*)calloc((size_t)rows * 27, sizeof(double));
  float* rowf = (float*)calloc((size_t)rows * 27, sizeof(float));
  int* rown = (int*)calloc((size_t)rows, sizeof(int));
  const bool rev = (g_perturb & 1) != 0, fused = (g_perturb & 4) != 0;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < rows; ++y)
    for (int xx = 0; xx < cols; ++xx) {
      const int x = rev ? cols - 1 - xx : xx;
      float row[7];
      if (!icp_row(x, y, new_v, new_n, model_v, model_n, *cam, cur, last_inv, dist_thres, sin_thres, row)) continue;
      ++rown[y];
      int s = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {                                                                     /* :92-105 */
          float pr = row[i] * row[j]; rowd[y * 27 + s] += (double)pr;
          rowf[y * 27 + s] = fused ? fmaf(row[i], row[j], rowf[y * 27 + s]) : rowf[y * 27 + s] + pr; ++s;
        }
    }
  double accd[27]; float accf[27]; int nvalid = 0;
  for (int k = 0; k < 27; ++k) { accd[k] = 0; accf[k] = 0; }
  for (int yy = 0; yy < rows; ++yy) { const int y = rev ? rows - 1 - yy : yy; nvalid += rown[y]; for (int k = 0; k < 27; ++k) { accd[k] += rowd[y * 27 + k]; accf[k] += rowf[y * 27 + k]; } }
  free(rowd); free(rowf); free(rown);
  if (out27d) memcpy(out27d, accd, sizeof(accd));
  if (out27f) memcpy(out27f, accf, sizeof(accf));
  if (valid) *valid = nvalid;
}

/* src/cuda/Mat.h:319-440 cofactor inverse, expression order preserved */
void okf_mat44_inverse(const float e[16], float out[16]) {
  float inv[16];
  inv[0] = e[5] * e[10] * e[15] - e[5] * e[11] * e[14] - e[9] * e[6] * e[15] + e[9] * e[7] * e[14] + e[13] * e[6] * e[11] - e[13] * e[7] * e[10];
  inv[4] = -e[4] * e[10] * e[15] + e[4] * e[11] * e[14] + e[8] * e[6] * e[15] - e[8] * e[7] * e[14] - e[12] * e[6] * e[11] + e[12] * e[7] * e[10];
  inv[8] = e[4] * e[9] * e[15] - e[4] * e[11] * e[13] - e[8] * e[5] * e[15] + e[8] * e[7] * e[13] + e[12] * e[5] * e[11] - e[12] * e[7] * e[9];
  inv[12] = -e[4] * e[9] * e[14] + e[4] * e[10] * e[13] + e[8] * e[5] * e[14] - e[8] * e[6] * e[13] - e[12] * e[5] * e[10] + e[12] * e[6] * e[9];
  inv[1] = -e[1] * e[10] * e[15] + e[1] * e[11] * e[14] + e[9] * e[2] * e[15] - e[9] * e[3] * e[14] - e[13] * e[2] * e[11] + e[13] * e[3] * e[10];
  inv[5] = e[0] * e[10] * e[15] - e[0] * e[11] * e[14] - e[8] * e[2] * e[15] + e[8] * e[3] * e[14] + e[12] * e[2] * e[11] - e[12] * e[3] * e[10];
  inv[9] = -e[0] * e[9] * e[15] + e[0] * e[11] * e[13] + e[8] * e[1] * e[15] - e[8] * e[3] * e[13] - e[12] * e[1] * e[11] + e[12] * e[3] * e[9];
  inv[13] = e[0] * e[9] * e[14] - e[0] * e[10] * e[13] - e[8] * e[1] * e[14] + e[8] * e[2] * e[13] + e[12] * e[1] * e[10] - e[12] * e[2] * e[9];
  inv[2] = e[1] * e[6] * e[15] - e[1] * e[7] * e[14] - e[5] * e[2] * e[15] + e[5] * e[3] * e[14] + e[13] * e[2] * e[7] - e[13] * e[3] * e[6];
  inv[6] = -e[0] * e[6] * e[15] + e[0] * e[7] * e[14] + e[4] * e[2] * e[15] - e[4] * e[3] * e[14] - e[12] * e[2] * e[7] + e[12] * e[3] * e[6];
  inv[10] = e[0] * e[5] * e[15] - e[0] * e[7] * e[13] - e[4] * e[1] * e[15] + e[4] * e[3] * e[13] + e[12] * e[1] * e[7] - e[12] * e[3] * e[5];
  inv[14] = -e[0] * e[5] * e[14] + e[0] * e[6] * e[13] + e[4] * e[1] * e[14] - e[4] * e[2] * e[13] - e[12] * e[1] * e[6] + e[12] * e[2] * e[5];
  inv[3] = -e[1] * e[6] * e[11] + e[1] * e[7] * e[10] + e[5] * e[2] * e[11] - e[5] * e[3] * e[10] - e[9] * e[2] * e[7] + e[9] * e[3] * e[6];
  inv[7] = e[0] * e[6] * e[11] - e[0] * e[7] * e[10] - e[4] * e[2] * e[11] + e[4] * e[3] * e[10] + e[8] * e[2] * e[7] - e[8] * e[3] * e[6];
  inv[11] = -e[0] * e[5] * e[11] + e[0] * e[7] * e[9] + e[4] * e[1] * e[11] - e[4] * e[3] * e[9] - e[8] * e[1] * e[7] + e[8] * e[3] * e[5];
  inv[15] = e[0] * e[5] * e[10] - e[0] * e[6] * e[9] - e[4] * e[1] * e[10] + e[4] * e[2] * e[9] + e[8] * e[1] * e[6] - e[8] * e[2] * e[5];
  float det = e[0] * inv[0] + e[1] * inv[4] + e[2] * inv[8] + e[3] * inv[12];
  float detr = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * detr;
}

/* src/cuda/Mat.h:240-262 */
void okf_mat44_mul(const float a[16], const float b[16], float out[16]) {
  float r[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      r[i * 4 + j] = a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j] + a[i * 4 + 2] * b[8 + j] + a[i * 4 + 3] * b[12 + j];
  memcpy(out, r, sizeof(r));
}

/* src/CameraPoseFinderICP.cpp:117-143: unpack, det(ATA) < 1e-10 -> fail, x = LLT solve */
int okf_solve6(const float in27[27], int check_det, float x[6]) {
  float A[36], b[6];
  unpack27(in27, A, b);
  if (check_det && det6(A) < 1E-10) return 0;
  llt_solve6(A, b, x);
  return 1;
}

/* src/CameraPoseFinderICP.cpp:95-111: R = Rx(x0) Ry(x1) Rz(x2), t = x[3:6]; angle-axis angle / |t| shake test.
 * AngleAxisf(R).angle() is 2*atan2(|q.vec|, |q.w|) of the quaternion of R; acos((tr-1)/2) is the same angle. */
int okf_vector6_to_transform(const float x[6], float dist_shake, float angle_shake, float t[16]) {
  float c0 = cosf(x[0]), s0 = sinf(x[0]), c1 = cosf(x[1]), s1 = sinf(x[1]), c2 = cosf(x[2]), s2 = sinf(x[2]);
  float Rx[9] = {1, 0, 0, 0, c0, -s0, 0, s0, c0};
  float Ry[9] = {c1, 0, s1, 0, 1, 0, -s1, 0, c1};
  float Rz[9] = {c2, -s2, 0, s2, c2, 0, 0, 0, 1};
  float Rxy[9], R[9];
  mat3_mul(Rx, Ry, Rxy); mat3_mul(Rxy, Rz, R);
  float tr = R[0] + R[4] + R[8];
  float ca = (tr - 1.f) * 0.5f; ca = ca > 1.f ? 1.f : (ca < -1.f ? -1.f : ca);
  float angle = acosf(ca);
  float d = sqrtf(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
  if (angle > angle_shake || d > dist_shake) return 0;
  float o[16] = {R[0], R[1], R[2], x[3], R[3], R[4], R[5], x[4], R[6], R[7], R[8], x[5], 0, 0, 0, 1};
  memcpy(t, o, sizeof(o));
  return 1;
}

/* a7: src/CameraPoseFinderICP.cpp:12-94 */
int okf_icp_estimate(const float* const* new_v, const float* const* new_n,
                     const float* const* model_v, const float* const* model_n,
                     int levels, const okf_cam* cam0, float dist_thres, float sin_thres,
                     float dist_shake, float angle_shake, float pose[16]) {
  int iters[3];
  if (levels == 1) { iters[0] = 3; }
  else if (levels == 2) { iters[0] = 10; iters[1] = 5; }
  else if (levels == 3) { iters[0] = 10; iters[1] = 5; iters[2] = 4; }
  else return 0;
  okf_cam cams[3]; cams[0] = *cam0;
  for (int l = 1; l < levels; ++l) {                      /* :36-48 */
    cams[l].cols = cams[l - 1].cols / 2; cams[l].rows = cams[l - 1].rows / 2;
    cams[l].cx = cams[l - 1].cx / 2; cams[l].cy = cams[l - 1].cy / 2;
    cams[l].fx = cams[l - 1].fx / 2; cams[l].fy = cams[l - 1].fy / 2;
  }
  float cur[16], last_inv[16];
  memcpy(cur, pose, sizeof(cur));
  okf_mat44_inverse(pose, last_inv);
  for (int l = levels - 1; l >= 0; --l)
    for (int it = 0; it < iters[l]; ++it) {
      float sums[27], x[6], t[16];
      okf_icp_system(new_v[l], new_n[l], model_v[l], model_n[l], &cams[l], cur, last_inv, dist_thres, sin_thres, 0, sums, 0);
      if (!okf_solve6(sums, 1, x)) return 0;
      if (!okf_vector6_to_transform(x, dist_shake, angle_shake, t)) return 0;
      okf_mat44_mul(t, cur, cur);                          /* :81 cur = T * cur */
    }
  memcpy(pose, cur, sizeof(cur));
  return 1;
}

/* a8: buildSDFSolverRows CalSDFErrSolverParams.cu:7-66 */
static bool sdf_row(const okf_volume* vol, f3 p, const float* cur, const float pm[6][16], float w_h, float v_h, float out[7]) {
  bool ret = true;
  float sdf0; f4 p4 = mk4(p.x, p.y, p.z, 1.0f);
  f4 pw0 = mat_vec(cur, p4);
  if (!interpolate_sdf(vol, mk3(pw0.x, pw0.y, pw0.z), sdf0)) ret = false;
  float sw[6];
  for (int k = 0; k < 6; ++k) {
    f4 pr = mat_vec(pm[k], p4);
    if (!interpolate_sdf(vol, mk3(pr.x, pr.y, pr.z), sw[k])) ret = false;
  }
  float sv[6];
  f3 off[6] = {mk3(pw0.x + v_h, pw0.y, pw0.z), mk3(pw0.x - v_h, pw0.y, pw0.z), mk3(pw0.x, pw0.y + v_h, pw0.z),
               mk3(pw0.x, pw0.y - v_h, pw0.z), mk3(pw0.x, pw0.y, pw0.z + v_h), mk3(pw0.x, pw0.y, pw0.z - v_h)};
  for (int k = 0; k < 6; ++k) if (!interpolate_sdf(vol, off[k], sv[k])) ret = false;
  if (!ret) return false;
  out[0] = (sw[0] - sw[1]) / (2 * w_h);
  out[1] = (sw[2] - sw[3]) / (2 * w_h);
  out[2] = (sw[4] - sw[5]) / (2 * w_h);
  out[3] = (sv[0] - sv[1]) / (2 * v_h);
  out[4] = (sv[2] - sv[3]) / (2 * v_h);
  out[5] = (sv[4] - sv[5]) / (2 * v_h);
  out[6] = sdf0;
  return true;
}

/* wrapper CalSDFErrSolverParams.cu:110-138: six perturbed transforms delta*cur, fp32 products */
static void sdf_perturbed(const float cur[16], float w_h, float pm[6][16]) {
  /* (row,col) 0-based of the +w_h / -w_h entries: w1: m23=-w,m32=+w ; w2: m13=+w,m31=-w ; w3: m12=-w,m21=+w */
  const int idx[3][2] = {{1 * 4 + 2, 2 * 4 + 1}, {0 * 4 + 2, 2 * 4 + 0}, {0 * 4 + 1, 1 * 4 + 0}};
  const float sgn[3] = {-1.f, 1.f, -1.f};
  for (int a = 0; a < 3; ++a)
    for (int pmn = 0; pmn < 2; ++pmn) {
      float d[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      float s = (pmn == 0) ? sgn[a] : -sgn[a];
      d[idx[a][0]] = s * w_h; d[idx[a][1]] = -s * w_h;
      okf_mat44_mul(d, cur, pm[a * 2 + pmn]);
    }
}

void okf_sdf_system(const okf_volume* vol, const float* depth, const okf_cam* cam, const float cur[16],
                    double* out27d, float* out27f, int* valid) {
  float w_h = 0.001f;                                     /* :119 `float w_h = 0.001;` */
  float v_h = vol->size / (float)vol->res;                /* :120 */
  float pm[6][16];
  sdf_perturbed(cur, w_h, pm);
  int cols = cam->cols, rows = cam->rows;
  double accd[27]; float accf[27]; int nvalid = 0;
  for (int k = 0; k < 27; ++k) { accd[k] = 0; accf[k] = 0; }
  const bool rev = (g_perturb & 1) != 0, fused = (g_perturb & 4) != 0;      /* tests only: okf_set_perturbation */
  for (int yy = 0; yy < rows; ++yy)
    for (int xx = 0; xx < cols; ++xx) {
      const int y = rev ? rows - 1 - yy : yy, x = rev ? cols - 1 - xx : xx;
      float d = depth[y * cols + x];
      if (d == 0) continue;
      float row[7];
      f3 p = depth_to_skeleton(x, y, d, *cam);
      if (!sdf_row(vol, p, cur, pm, w_h, v_h, row)) continue;
      ++nvalid;
      int s = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) { float pr = row[i] * row[j]; accd[s] += (double)pr; accf[s] = fused ? fmaf(row[i], row[j], accf[s]) : accf[s] + pr; ++s; }
    }
  if (out27d) memcpy(out27d, accd, sizeof(accd));
  if (out27f) memcpy(out27f, accf, sizeof(accf));
  if (valid) *valid = nvalid;
}

/* src/utils/eigen_utils.cpp:42-127 (double) */
void okf_exp_map(const double v[6], double rd[9], double dt[3]) {
  const double ang_min_sinc = 1.0e-8, ang_min_mc = 2.5e-4;
  double u0 = v[0], u1 = v[1], u2 = v[2];
  double theta = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
  double si = sin(theta), co = cos(theta);
  double sinc = fabs(theta) < ang_min_sinc ? 1.0 : si / theta;
  double mcosc = fabs(theta) < ang_min_mc ? 0.5 : (1.0 - co) / theta / theta;
  double msinc = fabs(theta) < ang_min_mc ? (1. / 6.0) : (1.0 - si / theta) / theta / theta;
  rd[0] = co + mcosc * u0 * u0;        rd[1] = -sinc * u2 + mcosc * u0 * u1; rd[2] = sinc * u1 + mcosc * u0 * u2;
  rd[3] = sinc * u2 + mcosc * u1 * u0; rd[4] = co + mcosc * u1 * u1;         rd[5] = -sinc * u0 + mcosc * u1 * u2;
  rd[6] = -sinc * u1 + mcosc * u2 * u0; rd[7] = sinc * u0 + mcosc * u2 * u1; rd[8] = co + mcosc * u2 * u2;
  dt[0] = v[3] * (sinc + u0 * u0 * msinc) + v[4] * (u0 * u1 * msinc - u2 * mcosc) + v[5] * (u0 * u2 * msinc + u1 * mcosc);
  dt[1] = v[3] * (u0 * u1 * msinc + u2 * mcosc) + v[4] * (sinc + u1 * u1 * msinc) + v[5] * (u1 * u2 * msinc - u0 * mcosc);
  dt[2] = v[3] * (u0 * u2 * msinc - u1 * mcosc) + v[4] * (u1 * u2 * msinc + u0 * mcosc) + v[5] * (sinc + u2 * u2 * msinc);
}

/* a9: src/CameraPoseFinderSDF.cpp:44-106 */
int okf_sdf_estimate(const okf_volume* vol, const float* depth, const okf_cam* cam, int max_iter,
                     float dist_shake, float angle_shake, float pose[16], int* iters_done) {
  float cur[16]; memcpy(cur, pose, sizeof(cur));
  int iter = 0;
  const float e = 0.001f;
  while (iter < max_iter) {
    float sums[27], x[6], t[16];
    okf_sdf_system(vol, depth, cam, cur, 0, sums, 0);
    okf_solve6(sums, 0, x);                                           /* :79 no determinant check */
    if (!okf_vector6_to_transform(x, dist_shake, angle_shake, t)) { if (iters_done) *iters_done = iter; return 0; }
    float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
    if (nx < e) break;                                                 /* :87-90 */
    double xd[6], R[9], tr[3];
    for (int k = 0; k < 6; ++k) xd[k] = (double)x[k];
    okf_exp_map(xd, R, tr);
    float Rt[9], tf[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = (float)R[c * 3 + r];   /* rotation().transpose().cast<float>() */
    for (int k = 0; k < 3; ++k) tf[k] = (float)tr[k];
    float nr[9], nt[3];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        nr[r * 3 + c] = Rt[r * 3] * cur[c] + Rt[r * 3 + 1] * cur[4 + c] + Rt[r * 3 + 2] * cur[8 + c];   /* :96 */
      float rt = Rt[r * 3] * tf[0] + Rt[r * 3 + 1] * tf[1] + Rt[r * 3 + 2] * tf[2];
      nt[r] = cur[r * 4 + 3] - rt;                                                                      /* :97 */
    }
    for (int r = 0; r < 3; ++r) { cur[r * 4] = nr[r * 3]; cur[r * 4 + 1] = nr[r * 3 + 1]; cur[r * 4 + 2] = nr[r * 3 + 2]; cur[r * 4 + 3] = nt[r]; }
    cur[12] = 0; cur[13] = 0; cur[14] = 0; cur[15] = 1;
    ++iter;
  }
  if (iters_done) *iters_done = iter;
  memcpy(pose, cur, sizeof(cur));
  return 1;
}

/* a10: integrateKernel integrateVolume.cu:15-77 + updateVoxel tsdfVolume.h:57-75 */
uint64_t okf_integrate(okf_volume* vol, int z0, int z1, const float* depth, const float* normals4,
                       const uint8_t* rgb, int has_color, int color_angled, const float pose[16],
                       float sdf_trunc, float max_dist, const okf_cam* dc, const okf_cam* rc) {
  float tinv[16];
  okf_mat44_inverse(pose, tinv);                                   /* :84 */
  const int R = vol->res;
  uint64_t n_upd = 0;
  if (z0 < first_layer(vol) || z1 > first_layer(vol) + stored_layers(vol)) {     /* a z-band volume asked for layers it does not hold */
    __atomic_fetch_add(&g_band_violations, (uint64_t)1, __ATOMIC_RELAXED);
    return 0;
  }
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_upd)
  for (int y = 0; y < R; ++y)
    for (int x = 0; x < R; ++x)
      for (int z = z0; z < z1; ++z) {
        f3 pw = voxel_to_world(vol, x, y, z);
        f4 pf4 = mat_vec(tinv, mk4(pw.x, pw.y, pw.z, 1.0f));
        f3 pf = mk3(pf4.x, pf4.y, pf4.z);
        if (pf.z <= 0) continue;
        int sx, sy;
        project_to_screen(pf, *dc, sx, sy);
        if (sx >= (int)dc->cols - 1 || sy >= (int)dc->rows - 1 || sx < 1 || sy < 1) continue;      /* :43 */
        float d = depth[sy * dc->cols + sx];
        if (d == 0) continue;
        float normalz = normals4[4 * (sy * dc->cols + sx) + 2];
        uint8_t col[3] = {0, 0, 0};
        if (has_color) {                                                                          /* :54-63 hard-coded 525/320/240 */
          int cxp = to_int_f(pf.x * 525 / pf.z + 320);
          int cyp = to_int_f(pf.y * 525 / pf.z + 240);
          if (cxp >= (int)rc->cols - 1 || cyp >= (int)rc->rows - 1 || cxp < 1 || cyp < 1) continue;
          const uint8_t* px = rgb + 3 * ((size_t)cyp * rc->cols + cxp);
          col[0] = px[0]; col[1] = px[1]; col[2] = px[2];
        }
        if (d < max_dist) {
          float sdf = d - pf.z;
          if (sdf > -sdf_trunc) {
            float tsdf = fminf(1.0f, sdf / sdf_trunc);
            okf_voxel& v = vol->data[((size_t)(z - first_layer(vol)) * R + y) * R + x];
            float ow = v.weight, ot = v.tsdf;
            float w = 1.f;
            float nw = fminf(ow + w, vol->max_weight);
            float nt = (ot * ow + tsdf * w) / (ow + w);
            if (has_color) {
              /* :72 `(color_angled?fminf(1.0,abs(normalz)/0.75):1.0)*2.0` -- division and *2 in double */
              float wc = color_angled ? (float)((double)fminf(1.0f, (float)((double)fabsf(normalz) / 0.75)) * 2.0) : 2.0f;
              for (int c = 0; c < 3; ++c) {
                float oc = (float)v.color[c];
                float nc = fminf(255.0f, (oc * ow + (float)col[c] * wc) / (ow + wc));
                v.color[c] = (uint8_t)nc;
              }
            }
            v.weight = nw; v.tsdf = nt;
            ++n_upd;
          }
        }
      }
  return n_upd;
}

uint64_t okf_band_violations(void) { return __atomic_exchange_n(&g_band_violations, (uint64_t)0, __ATOMIC_RELAXED); }

uint64_t okf_count_weight_gt0(const okf_volume* vol) {
  size_t n = (size_t)vol->res * vol->res * stored_layers(vol); uint64_t c = 0;
#pragma omp parallel for reduction(+ : c)
  for (size_t i = 0; i < n; ++i) c += vol->data[i].weight > 0 ? 1 : 0;
  return c;
}

/* a11: gradientForPoint raycastingVolume.cu:16-42 */
static bool gradient_for_point(const okf_volume* vol, f3 samplepos, f3 vtx, f3& grad) {
  i3 g = world_to_voxel(vol, samplepos);
  int R = vol->res;
  if (g.x <= 1 || g.x >= R - 2) return false;
  if (g.y <= 1 || g.y >= R - 2) return false;
  if (g.z <= 1 || g.z >= R - 2) return false;
  float cell = vol->size / (float)R;
  f3 n; float f1, f2;
  if (!interpolate_sdf(vol, mk3(vtx.x + cell, vtx.y, vtx.z), f1)) return false;
  if (!interpolate_sdf(vol, mk3(vtx.x - cell, vtx.y, vtx.z), f2)) return false;
  n.x = f1 - f2;
  if (!interpolate_sdf(vol, mk3(vtx.x, vtx.y + cell, vtx.z), f1)) return false;
  if (!interpolate_sdf(vol, mk3(vtx.x, vtx.y - cell, vtx.z), f2)) return false;
  n.y = f1 - f2;
  if (!interpolate_sdf(vol, mk3(vtx.x, vtx.y, vtx.z + cell), f1)) return false;
  if (!interpolate_sdf(vol, mk3(vtx.x, vtx.y, vtx.z - cell), f2)) return false;
  n.z = f1 - f2;
  float len = norm3(n);
  if (len < 1e-8) return false;
  grad = mul3(n, 1 / len);                 /* fp32 reciprocal, unlike normalize() */
  return true;
}

/* a11: raycastKernel :121-156, raySample :65-119, getMinTime/getMaxTime :44-63 */
void okf_raycast(const okf_volume* vol, int has_color, const float pose[16], float inc, const okf_cam* cam,
                 float near_plane, float far_plane, float* v4, float* n4, uint8_t* rgb, uint32_t* steps) {
  int cols = cam->cols, rows = cam->rows;
  f3 vmax = mk3(vol->size, vol->size, vol->size);
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int i = y * cols + x;
      st4(v4, i, mk4(0, 0, 0, 0)); st4(n4, i, mk4(0, 0, 0, 0));
      if (has_color && rgb) { rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 0; }
      if (steps) steps[i] = 0;
      f3 cam_dir = normalize3(depth_to_skeleton(x, y, 1.0f, *cam));
      f3 org = mk3(pose[3], pose[7], pose[11]);
      f4 wd4 = mat_vec(pose, mk4(cam_dir.x, cam_dir.y, cam_dir.z, 0.0f));
      f3 dir = mk3(wd4.x, wd4.y, wd4.z);
      dir.x = (dir.x == 0.f) ? (float)1e-15 : dir.x;
      dir.y = (dir.y == 0.f) ? (float)1e-15 : dir.y;
      dir.z = (dir.z == 0.f) ? (float)1e-15 : dir.z;
      float txmin = ((dir.x > 0 ? 0.f : vmax.x) - org.x) / dir.x;
      float tymin = ((dir.y > 0 ? 0.f : vmax.y) - org.y) / dir.y;
      float tzmin = ((dir.z > 0 ? 0.f : vmax.z) - org.z) / dir.z;
      float tmin = fmaxf(fmaxf(txmin, tymin), tzmin);
      float txmax = ((dir.x > 0 ? vmax.x : 0.f) - org.x) / dir.x;
      float tymax = ((dir.y > 0 ? vmax.y : 0.f) - org.y) / dir.y;
      float tzmax = ((dir.z > 0 ? vmax.z : 0.f) - org.z) / dir.z;
      float tmax = fminf(fminf(txmax, tymax), tzmax);
      tmin = fmaxf(tmin, near_plane / cam_dir.z);
      tmax = fminf(tmax, far_plane / cam_dir.z);
      if (tmin >= tmax) continue;
      float t = tmin, last_sdf = 0; f3 last_pos = mk3(0, 0, 0);
      uint32_t nsteps = 0;
      while (t < tmax) {
        f3 pos = add3(org, mul3(dir, t));
        float sdf = voxel_nearest(vol, pos).tsdf;
        ++nsteps;
        if (last_sdf > 0.0f && sdf < 0.0f) {
          float ftdt, ft;
          if (!interpolate_sdf(vol, pos, ftdt)) break;
          if (!interpolate_sdf(vol, last_pos, ft)) break;
          float alpha = t - inc * ftdt / (ftdt - ft);
          f3 vtx = add3(org, mul3(dir, alpha));
          if (has_color && rgb) {
            uint8_t c[3] = {0, 0, 0};
            interpolate_color(vol, vtx, c);
            rgb[3 * i] = c[0]; rgb[3 * i + 1] = c[1]; rgb[3 * i + 2] = c[2];
          }
          f3 grad;
          if (!gradient_for_point(vol, last_pos, vtx, grad)) break;
          st4(v4, i, mk4(vtx.x, vtx.y, vtx.z, 1.0f));
          st4(n4, i, mk4(grad.x, grad.y, grad.z, 0));
          break;
        }
        last_sdf = sdf; last_pos = pos;
        t += inc;
      }
      if (steps) steps[i] = nsteps;
    }
}

int okf_interpolate_sdf(const okf_volume* vol, const float pos[3], float* dist) {
  float d = 0; bool ok = interpolate_sdf(vol, mk3(pos[0], pos[1], pos[2]), d);
  if (ok) *dist = d;
  return ok ? 1 : 0;
}

/* helper arithmetic exposed for tests/test_oracle_vs_ref.py (pinned against the reference's headers) */
int okf_interpolate_color(const okf_volume* vol, const float pos[3], uint8_t out[3]) {
  return interpolate_color(vol, mk3(pos[0], pos[1], pos[2]), out) ? 1 : 0;
}
void okf_world_to_voxel(const okf_volume* vol, const float pos[3], int out[3]) {
  i3 g = world_to_voxel(vol, mk3(pos[0], pos[1], pos[2])); out[0] = g.x; out[1] = g.y; out[2] = g.z;
}
float okf_norm(const float v[3]) { return norm3(mk3(v[0], v[1], v[2])); }
int okf_to_int(double v) { return to_int(v); }

/* a12: vertexInterp marchingcube.cu:5-26 */
static okf_vertex vertex_interp(float iso, f3 p1, f3 p2, float d1, float d2, const uint8_t c1[3], const uint8_t c2[3]) {
  okf_vertex r1, r2, res;
  float inv255 = (float)(1.0 / (double)255.f);
  r1.pos[0] = p1.x; r1.pos[1] = p1.y; r1.pos[2] = p1.z;
  r2.pos[0] = p2.x; r2.pos[1] = p2.y; r2.pos[2] = p2.z;
  for (int k = 0; k < 3; ++k) { r1.color[k] = (float)c1[k] * inv255; r2.color[k] = (float)c2[k] * inv255; }
  if (fabsf(iso - d1) < 0.00001f) return r1;
  if (fabsf(iso - d2) < 0.00001f) return r2;
  if (fabsf(d1 - d2) < 0.00001f) return r1;
  float mu = (iso - d1) / (d2 - d1);
  res.pos[0] = p1.x + mu * (p2.x - p1.x);
  res.pos[1] = p1.y + mu * (p2.y - p1.y);
  res.pos[2] = p1.z + mu * (p2.z - p1.z);
  for (int k = 0; k < 3; ++k) res.color[k] = ((float)c1[k] + mu * (float)((int)c2[k] - (int)c1[k])) / 255.f;
  return res;
}

/* a12: extractIsoSurfaceAtPosition marchingcube.cu:41-137.  Emits into tris[count..], canonical order (z,y,x,k). */
uint32_t okf_marching_cubes(const okf_volume* vol, int z0, int z1, int has_color, float thr,
                            okf_triangle* tris, uint32_t max_tris) {
  const int R = vol->res;
  uint32_t count = 0;
  float cell = vol->size / (float)R;
  float P = cell * 0.5f, M = cell * (-0.5f);
  /* corner order of the reference's early-outs: 000,100,010,001,110,011,101,111 (x,y,z bits) */
  const int cb[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 1}};
  for (int z = z0; z < z1; ++z)
    for (int y = 0; y < R; ++y)
      for (int x = 0; x < R; ++x) {
        f3 wp = voxel_to_world(vol, x, y, z);
        f3 p[8]; float d[8]; uint8_t c[8][3];
        bool ok = true;
        for (int k = 0; k < 8 && ok; ++k) {
          p[k] = add3(wp, mk3(cb[k][0] ? P : M, cb[k][1] ? P : M, cb[k][2] ? P : M));
          if (!interpolate_sdf(vol, p[k], d[k])) { ok = false; break; }
          c[k][0] = c[k][1] = c[k][2] = 0;
          if (has_color) interpolate_color(vol, p[k], c[k]);
        }
        if (!ok) continue;
        /* names: k0=000 k1=100 k2=010 k3=001 k4=110 k5=011 k6=101 k7=111 */
        unsigned ci = 0;                                         /* :77-85 */
        if (d[2] < 0.f) ci += 1;
        if (d[4] < 0.f) ci += 2;
        if (d[1] < 0.f) ci += 4;
        if (d[0] < 0.f) ci += 8;
        if (d[5] < 0.f) ci += 16;
        if (d[7] < 0.f) ci += 32;
        if (d[6] < 0.f) ci += 64;
        if (d[3] < 0.f) ci += 128;
        bool skip = false;
        for (int k = 0; k < 8; ++k) if (fabsf(d[k]) > thr) skip = true;     /* :101-108 */
        if (skip) continue;
        uint64_t w = kTriWords[ci];
        unsigned emask = 0;
        for (int i = 0; i < 16; ++i) { unsigned e = (unsigned)((w >> (4 * i)) & 0xF); if (e != 0xF) emask |= 1u << e; }
        if (emask == 0 || emask == 255) continue;                          /* :110 */
        /* edge e joins corners (a,b): :116-127 */
        const int ea[12] = {2, 4, 1, 0, 5, 7, 6, 3, 2, 4, 1, 0};
        const int eb[12] = {4, 1, 0, 2, 7, 6, 3, 5, 5, 7, 6, 3};
        okf_vertex vl[12];
        for (int e = 0; e < 12; ++e)
          if (emask & (1u << e)) vl[e] = vertex_interp(0.f, p[ea[e]], p[eb[e]], d[ea[e]], d[eb[e]], c[ea[e]], c[eb[e]]);
        for (int i = 0; i < 15; i += 3) {
          unsigned e0 = (unsigned)((w >> (4 * i)) & 0xF);
          if (e0 == 0xF) break;
          unsigned e1 = (unsigned)((w >> (4 * (i + 1))) & 0xF), e2 = (unsigned)((w >> (4 * (i + 2))) & 0xF);
          if (count >= max_tris) continue;                                  /* :29-31 */
          okf_triangle& t = tris[count++];
          t.v[0] = vl[e0]; t.v[1] = vl[e1]; t.v[2] = vl[e2];
        }
      }
  return count;
}

}  /* extern "C" */
