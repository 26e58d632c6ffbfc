// bilateral_tile.h -- the fused u16 -> f32 + depth gate + bilateral filter of one 64x4 pixel tile (DataPreprocesser.cu:17-79,
// HybKinectfu.cpp:63-96), shared by k_gate_bilateral (preprocess.hip) and by the raycast launch that carries the NEXT frame's filter along
// (raycast.hip: k_raycast_prefetch).
#pragma once
#include "kf_internal.h"

#define BIL_TX 64
#define BIL_TY 4
#define BIL_MAXR 8

// Bilateral tap weight exp(-(space2 * ss_inv + diff^2 * sd_inv)) (DataPreprocesser.cu:70-73, __expf there) as ONE hardware exp2: the two
// factors carry log2(e) already, the sum is a fused multiply-add.  The filter is tolerance-checked (2e-6 relative: the reference's fast
// intrinsic is not reproducible on any other target anyway); the per-call kernel and the fused one share this function and their tap
// order, so they still agree bit for bit.
#define KF_LOG2E 1.44269504f
__device__ __forceinline__ float kf_bilateral_weight(float diff, float space2, float c_ss, float c_sd) {
  return __builtin_amdgcn_exp2f(__builtin_fmaf(diff * diff, c_sd, -(space2 * c_ss)));
}


// Fused front end of HybKinectfu::processNewFrame (src/HybKinectfu.cpp:63-110): u16 mm -> f32 m (:73), range gate
// (DataPreprocesser.cu:17-36) and bilateral filter (:37-79) in ONE launch.  The (64+2R)x(4+2R) gated tile lives in LDS; the
// R x R tap loop is fully unrolled and branch-free: a zero (invalid or out-of-image) tap gets weight 0 -- adding +0 leaves the
// reference's running sums bit-identical -- and the reference's early `return` (any tap further than 5 sigma_d away keeps the
// unfiltered value, :66-69) becomes a flag tested once at the end.
// The kernel also leaves the TSDF integration's tile maxima behind (integrate.hip: the max, over every 8x8 and 16x16 pixel tile, of
// the gated depth that can integrate, d < max_dist): the gated tile is in LDS anyway, a wave is one 64-pixel row, so three DPP
// steps give the max of each 8-pixel group and one more that of each 16-pixel group, and the group leaders merge them into the
// (cleared) tables with fire-and-forget integer atomic maxima -- non-negative floats order like their bit patterns.  That replaces
// a launch of its own (k_integrate_prepare) in front of every integrate.  acc.tile == nullptr: tables not wanted.
struct KfTileAccum { int* tile; int off0, w0, off1, w1, n; float max_dist; };   // n: entries of both maxima tables = offset of the minima (0: minima not wanted)
// FAST (every sane parameter set; the host decides): an invalid pixel sits in the LDS tile as a huge sentinel instead of 0, so its tap needs
// no special case -- the squared difference sends the exponent to -inf, exp2 gives exactly +0, and 0 * sentinel adds exactly +0 to the
// weighted sum -- and the difference / square / exponent of two taps are one packed instruction each.  Tap order and every rounded
// operation of a valid tap are those of the plain loop: the same bits (per-call kernel == fused kernel stays a bit-exact test), a
// quarter fewer vector instructions in a kernel that is bound by them (81 taps per pixel).
#define BIL_SENTINEL 1e18f
#define BIL_SENTINEL_CUT 1e17f
struct KfBilateralArgs {
  const uint16_t* mm; const float* raw_in;       // the frame as u16 millimetres (converted here), or an f32 metre image
  float* raw_out; float* trunced; float* filtered;
  int cols, rows;
  float tmin, tmax, ss_inv, sd_inv, sigma_depth;
  KfTileAccum acc;
};
// The TSDF integration's tile maxima (and, once saturated free space can exist, minima) of one 64-pixel image row held by a wave: `value` is
// the lane's gated depth (0: invalid or outside the image).  Shared by the filter below and by kf_tiles_from_gated.
__device__ __forceinline__ void kf_tile_accumulate(const KfTileAccum& acc, float value, bool inside, int x, int y, int lx, int cols, int rows) {
  if (acc.tile) {                                          // uniform; every lane of the wave takes part in the DPP steps
    float d = (value < acc.max_dist) ? value : 0.f;
    d = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0xB1, 0xf, 0xf, true)));    // quad_perm:[1,0,3,2]
    d = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x4E, 0xf, 0xf, true)));    // quad_perm:[2,3,0,1]
    d = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x141, 0xf, 0xf, true)));   // row_half_mirror: 8-pixel groups
    const float d16 = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x140, 0xf, 0xf, true)));   // row_mirror: 16-pixel groups
    // a group whose leader lies outside the image holds no pixel at all (d == 0): the indices below stay inside the tables
    if ((lx & 7) == 0 && d > 0.f) atomicMax(acc.tile + acc.off0 + (y >> 3) * acc.w0 + (x >> 3), __float_as_int(d));
    if ((lx & 15) == 0 && d16 > 0.f) atomicMax(acc.tile + acc.off1 + (y >> 4) * acc.w1 + (x >> 4), __float_as_int(d16));
    // the tile MINIMA (second half of the buffer, cleared to +inf): 0 as soon as one pixel of the tile cannot integrate (invalid or
    // beyond max_dist), else the smallest depth -- what lets the cull prove "every voxel of this brick sees free space" (integrate.hip)
    if (acc.n) {                                             // uniform
    float m = inside ? ((value != 0.f && value < acc.max_dist) ? value : 0.f) : __builtin_huge_valf();
    m = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(0x7F800000, __float_as_int(m), 0xB1, 0xf, 0xf, false)));
    m = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(0x7F800000, __float_as_int(m), 0x4E, 0xf, 0xf, false)));
    m = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(0x7F800000, __float_as_int(m), 0x141, 0xf, 0xf, false)));
    const float m16 = fminf(m, __int_as_float(__builtin_amdgcn_update_dpp(0x7F800000, __float_as_int(m), 0x140, 0xf, 0xf, false)));
    if ((lx & 7) == 0 && x < cols && y < rows) atomicMin(acc.tile + acc.n + acc.off0 + (y >> 3) * acc.w0 + (x >> 3), __float_as_int(m));
    if ((lx & 15) == 0 && x < cols && y < rows) atomicMin(acc.tile + acc.n + acc.off1 + (y >> 4) * acc.w1 + (x >> 4), __float_as_int(m16));
    }
  }
}

// tile (bx, by) by 256 threads (tid 0..255) with `tile` their (BIL_TX + 2R) x (BIL_TY + 2R) floats of LDS; every thread of the calling
// workgroup must come through here (one workgroup barrier inside), whether or not its tile lies in the image
template <int R, bool FAST>
__device__ __forceinline__ void kf_bilateral_tile(const KfBilateralArgs& b, int bx, int by, int tid, float* tile) {
  const uint16_t* __restrict__ mm = b.mm; const float* __restrict__ raw_in = b.raw_in;
  float* __restrict__ raw_out = b.raw_out; float* __restrict__ trunced = b.trunced; float* __restrict__ filtered = b.filtered;
  const int cols = b.cols, rows = b.rows;
  const float tmin = b.tmin, tmax = b.tmax, ss_inv = b.ss_inv, sd_inv = b.sd_inv, sigma_depth = b.sigma_depth;
  const KfTileAccum acc = b.acc;
  constexpr int TW = BIL_TX + 2 * R, TH = BIL_TY + 2 * R;
  const int x0 = bx * BIL_TX - R, y0 = by * BIL_TY - R;
  for (int i = tid; i < TW * TH; i += 256) {
    const int lx = i % TW, ly = i / TW, gx = x0 + lx, gy = y0 + ly;
    float g = 0.f;
    if (gx >= 0 && gx < cols && gy >= 0 && gy < rows) {
      const float d = mm ? (float)((double)mm[gy * cols + gx] * 0.001) : raw_in[gy * cols + gx];   // HybKinectfu.cpp:73
      g = (d < tmax && d > tmin) ? d : 0.f;                                                        // DataPreprocesser.cu:25-33
      const bool interior = lx >= R && lx < R + BIL_TX && ly >= R && ly < R + BIL_TY;              // written exactly once
      if (interior) { if (mm) raw_out[gy * cols + gx] = d; trunced[gy * cols + gx] = g; }
    }
    tile[i] = (FAST && g == 0.f) ? BIL_SENTINEL : g;
  }
  __syncthreads();
  const int lx = tid & 63, ly = tid >> 6;
  const int x = bx * BIL_TX + lx, y = by * BIL_TY + ly;
  const bool inside = x < cols && y < rows;
  float value = inside ? tile[(ly + R) * TW + lx + R] : 0.f;
  if (FAST && value > BIL_SENTINEL_CUT) value = 0.f;       // (the centre pixel itself is invalid)
  kf_tile_accumulate(acc, value, inside, x, y, lx, cols, rows);
  if (!inside) return;
  float result = value;
  if (value != 0.f) {
    float sum1 = 0.f, sum2 = 0.f;
    const float thr = 5 * sigma_depth;
    const float c_ss = ss_inv * KF_LOG2E, c_sd = -(sd_inv * KF_LOG2E);
    if (FAST) {
      const kf_f2 v2 = f2_splat(value), csd2 = f2_splat(c_sd);
      bool over = false;                                   // some VALID tap differs from the centre by more than 5 sigma (:66-69)
#pragma unroll
      for (int dy = -R; dy <= R; ++dy) {
        float row[2 * R + 2], dif[2 * R + 2], ex[2 * R + 2];
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) row[dx + R] = tile[(ly + R + dy) * TW + lx + R + dx];
        row[2 * R + 1] = value;
#pragma unroll
        for (int k = 0; k <= 2 * R; k += 2) {              // two taps per packed instruction: difference, square, exponent
          const kf_f2 t2 = {row[k], row[k + 1]};
          const kf_f2 sp = {-((float)((k - R) * (k - R) + dy * dy) * c_ss), -((float)((k + 1 - R) * (k + 1 - R) + dy * dy) * c_ss)};
          const kf_f2 d2 = v2 - t2;
          const kf_f2 e2 = f2_fma(d2 * d2, csd2, sp);
          dif[k] = d2.x; dif[k + 1] = d2.y; ex[k] = e2.x; ex[k + 1] = e2.y;
        }
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {                 // the sums in the plain loop's tap order
          const float ad = fabsf(dif[k]);
          over = over || (ad > thr && ad < BIL_SENTINEL_CUT);
          const float w = __builtin_amdgcn_exp2f(ex[k]);
          sum1 = __builtin_fmaf(row[k], w, sum1); sum2 += w;
        }
      }
      if (!over && sum2 > 0.f) result = sum1 / sum2;
    } else {
      float max_diff = 0.f;                                // largest |tap - centre| over the valid taps: the 5-sigma test, once
#pragma unroll
      for (int dy = -R; dy <= R; ++dy) {
        float row[2 * R + 1];
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) row[dx + R] = tile[(ly + R + dy) * TW + lx + R + dx];
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) {
          const float tmp = row[dx + R];
          const bool valid = tmp != 0.f;
          const float diff = value - tmp;
          max_diff = fmaxf(max_diff, valid ? fabsf(diff) : 0.f);
          const float wv = kf_bilateral_weight(diff, (float)(dx * dx + dy * dy), c_ss, c_sd);
          const float w = valid ? wv : 0.f;                 // an invalid tap adds +0 to both sums: the loop stays straight-line
          sum1 = __builtin_fmaf(tmp, w, sum1); sum2 += w;
        }
      }
      if (!(max_diff > thr) && sum2 > 0.f) result = sum1 / sum2;
    }
  }
  filtered[y * cols + x] = result;
}


// The tile tables alone, from a depth map that has been gated already (b.trunced): what is left for the raycast launch to do when the
// filter itself rode in the tracking launch (track.hip: k_icp_loop) -- the tables may only be written once the fusion pass has cleared them.
__device__ __forceinline__ void kf_tiles_from_gated(const KfBilateralArgs& b, int bx, int by, int tid) {
  const int lx = tid & 63, ly = tid >> 6;
  const int x = bx * BIL_TX + lx, y = by * BIL_TY + ly;
  const bool inside = x < b.cols && y < b.rows;
  const float value = inside ? b.trunced[y * b.cols + x] : 0.f;
  kf_tile_accumulate(b.acc, value, inside, x, y, lx, b.cols, b.rows);
}

// Vertex and normal of pixel (x, y) from the filtered depth map (VerticesNormalsCalculater.cu:15-33 and :35-64 in one pass): the body of
// k_vertices_normals (preprocess.hip), also run by rider workgroups of the raycast launch (raycast.hip) for the NEXT frame.
__device__ __forceinline__ void kf_vertex_normal_pixel(const float* __restrict__ depth, float4* __restrict__ out_v, float4* __restrict__ out_n, const KfCam& cam, int x, int y,
                                                       float4* keep_v = nullptr, float4* keep_n = nullptr) {
  const int i = y * cam.cols + x;
  // DepthCamera.h:19-29 `depth*(x - cx)/fx`: ten quotients by the two focal lengths per pixel -> their reciprocals are refined once
  const KfRecip rfx = kf_recip(cam.fx), rfy = kf_recip(cam.fy);
  auto skeleton = [&](int px, int py, float d) { return kf3(kf_div(d * ((float)(unsigned)px - cam.cx), rfx), kf_div(d * ((float)(unsigned)py - cam.cy), rfy), d); };
  const float d0 = depth[i];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f), n = v;
  float3 v0 = kf3(0.f, 0.f, 0.f);
  if (d0 != 0.f) { v0 = skeleton(x, y, d0); v = make_float4(v0.x, v0.y, v0.z, 1.0f); }
  if (d0 != 0.f && !(x == cam.cols - 1 || y == cam.rows - 1 || x == 0 || y == 0)) {
    const float dr = depth[i + 1], du = depth[i + cam.cols], dl = depth[i - 1], dd = depth[i - cam.cols];
    if (dr != 0.f && du != 0.f && dl != 0.f && dd != 0.f) {                       // a vertex's z is its depth: z == 0 <=> depth == 0
      const float3 vr = skeleton(x + 1, y, dr), vu = skeleton(x, y + 1, du);
      const float3 vl = skeleton(x - 1, y, dl), vd = skeleton(x, y - 1, dd);
      const float3 c = kf_normalize(kf_cross(kf_sub(vu, vd), kf_sub(vr, vl)));
      n = make_float4(c.x, c.y, c.z, 0.f);
    }
  }
  out_v[i] = v; out_n[i] = n;
  if (keep_v) { *keep_v = v; *keep_n = n; }
}

// sample.cu:37-61 (vertices) and :16-36 (normals), intended semantics: each output pixel exactly once.
__device__ __forceinline__ float4 pyr_vertex(float4 p00, float4 p01, float4 p10, float4 p11) {
  if (p00.z == 0.f || p01.z == 0.f || p10.z == 0.f || p11.z == 0.f) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float q = 0.25f;                                  // `*0.25`: double literal narrowed by operator*(float4, const float&)
  return make_float4((p00.x + p01.x + p10.x + p11.x) * q, (p00.y + p01.y + p10.y + p11.y) * q,
                     (p00.z + p01.z + p10.z + p11.z) * q, (p00.w + p01.w + p10.w + p11.w) * q);
}
__device__ __forceinline__ float4 pyr_normal(float4 p00, float4 p01, float4 p10, float4 p11) {
  if (kf_is_zero4(p01) || kf_is_zero4(p10) || kf_is_zero4(p00) || kf_is_zero4(p11)) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float q = 0.25f;
  float3 n = kf_normalize(kf3((p00.x + p01.x + p10.x + p11.x) * q, (p00.y + p01.y + p10.y + p11.y) * q, (p00.z + p01.z + p10.z + p11.z) * q));
  return make_float4(n.x, n.y, n.z, 0.f);
}

// Levels 1 and 2 of a vertex / normal map pair for one TW x TH tile of level 0 (TW, TH multiples of 4, tile origin (x0, y0) a multiple of them):
// every 2x2 and 4x4 block lies inside the tile, so the pyramid needs nothing from other workgroups.  s_v / s_n hold the tile's level-0
// texels (row-major, TW wide; zeros outside the image), s1_v / s1_n take its (TW/2) x (TH/2) level-1 texels.  All `nthreads` threads
// (ids 0 .. nthreads-1) of the calling group come through here together, between them two barriers of `sync` (a callable: __syncthreads).
// Same functions, same operand order as k_pyramid (preprocess.hip): the same bits.
struct KfPyrOut { float4* v1; float4* n1; float4* v2; float4* n2; int c1, r1, c2, r2; };       // v1 null: no pyramid wanted
template <int TW, int TH, typename Sync>
__device__ __forceinline__ void kf_tile_pyramid(const KfPyrOut& o, int x0, int y0, int tid, const float4* s_v, const float4* s_n, float4* s1_v, float4* s1_n, Sync sync) {
  constexpr int W1 = TW / 2, H1 = TH / 2, W2 = TW / 4, H2 = TH / 4;
  sync();
  if (tid < W1 * H1) {
    const int lx = tid % W1, ly = tid / W1, x1 = x0 / 2 + lx, y1 = y0 / 2 + ly;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), qn = qv;
    if (x1 < o.c1 && y1 < o.r1) {
      const int k = (2 * ly) * TW + 2 * lx;
      qv = pyr_vertex(s_v[k], s_v[k + 1], s_v[k + TW], s_v[k + TW + 1]);
      qn = pyr_normal(s_n[k], s_n[k + 1], s_n[k + TW], s_n[k + TW + 1]);
      o.v1[(size_t)y1 * o.c1 + x1] = qv; o.n1[(size_t)y1 * o.c1 + x1] = qn;
    }
    s1_v[tid] = qv; s1_n[tid] = qn;
  }
  sync();
  if (o.v2 && tid < W2 * H2) {
    const int lx = tid % W2, ly = tid / W2, x2 = x0 / 4 + lx, y2 = y0 / 4 + ly;
    if (x2 < o.c2 && y2 < o.r2) {
      const int k = (2 * ly) * W1 + 2 * lx;
      o.v2[(size_t)y2 * o.c2 + x2] = pyr_vertex(s1_v[k], s1_v[k + 1], s1_v[k + W1], s1_v[k + W1 + 1]);
      o.n2[(size_t)y2 * o.c2 + x2] = pyr_normal(s1_n[k], s1_n[k + 1], s1_n[k + W1], s1_n[k + W1 + 1]);
    }
  }
}



// host side (preprocess.hip): the launch arguments for a buffer set, and the vertices + normals launch that follows the filter
void kf_bilateral_args(kf_ctx* c, const uint16_t* mm, const float* raw_in, float* raw_out, float* trunced, float* filtered,
                       float tmin, float tmax, float sigma_pixel, float sigma_depth, bool build_tiles, KfBilateralArgs* b, bool* fast);
int kf_launch_vertices_normals(kf_ctx* c, hipStream_t stream, const float* filtered, float4* v0, float4* n0, const kf_camera_params* cam);
