// preprocess.hip -- depth gate, bilateral filter, back-projection, normals and the 2x2 box pyramids.
// Reference: src/cuda/DataPreprocesser.cu, src/cuda/VerticesNormalsCalculater.cu, src/cuda/sample.cu.
//
// gfx950 notes: every map here (<= 4.9 MB) lives in the XCD L2 / Infinity Cache for the whole frame, so these kernels are
// launch- and latency-bound, not HBM-bound.  Lanes run along x (64 consecutive pixels = 256 B / 1 KiB per wave row),
// and the pyramid kernel writes level 1 and level 2 of vertices AND normals in one launch (the reference uses 8).
#include "kf_internal.h"
#include "bilateral_tile.h"
#include <string.h>
#include <stdlib.h>

static inline KfCam to_cam(const kf_camera_params* p) {
  KfCam c; c.cols = (int)p->cols; c.rows = (int)p->rows; c.cx = p->cx; c.cy = p->cy; c.fx = p->fx; c.fy = p->fy; return c;
}

// DataPreprocesser.cu:17-36: keep d iff trunc_min < d < trunc_max (strict both sides)
__global__ void __launch_bounds__(256) k_trunc_depth(const float* __restrict__ in, float* __restrict__ out, int n, float tmin, float tmax) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float d = in[i];
  out[i] = (d < tmax && d > tmin) ? d : 0.f;
}

// DataPreprocesser.cu:37-79.  64x4 pixel tile per workgroup, (64+2r)x(4+2r) depth halo staged through LDS.
__global__ void __launch_bounds__(256) k_bilateral(const float* __restrict__ in, float* __restrict__ out, int cols, int rows,
                                                   float ss_inv, float sd_inv, float sigma_depth, int radius) {
  __shared__ float tile[(BIL_TY + 2 * BIL_MAXR) * (BIL_TX + 2 * BIL_MAXR)];
  const int tw = BIL_TX + 2 * radius, th = BIL_TY + 2 * radius;
  const int x0 = blockIdx.x * BIL_TX - radius, y0 = blockIdx.y * BIL_TY - radius;
  for (int i = threadIdx.x; i < tw * th; i += 256) {
    int lx = i % tw, ly = i / tw, gx = x0 + lx, gy = y0 + ly;
    tile[i] = (gx >= 0 && gx < cols && gy >= 0 && gy < rows) ? in[gy * cols + gx] : 0.f;   // outside = skipped like a zero tap
  }
  __syncthreads();
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int x = blockIdx.x * BIL_TX + lx, y = blockIdx.y * BIL_TY + ly;
  if (x >= cols || y >= rows) return;
  const float value = tile[(ly + radius) * tw + lx + radius];
  float result = value;
  if (value != 0.f) {
    // the reference clamps the window to the image; out-of-image taps are zeros in the tile and zeros are skipped (:61-64)
    float sum1 = 0.f, sum2 = 0.f;
    bool aborted = false;
    const float thr = 5 * sigma_depth;
    const float c_ss = ss_inv * KF_LOG2E, c_sd = -(sd_inv * KF_LOG2E);               // the weight as one exp2: see kf_bilateral_weight
    for (int dy = -radius; dy <= radius && !aborted; ++dy) {
      const float* rowp = &tile[(ly + radius + dy) * tw + lx + radius];
      for (int dx = -radius; dx <= radius; ++dx) {
        float tmp = rowp[dx];
        if (tmp == 0.f) continue;
        if (fabsf(tmp - value) > thr) { aborted = true; break; }                      // :66-69 keeps the unfiltered value
        const float w = kf_bilateral_weight(value - tmp, (float)(dx * dx + dy * dy), c_ss, c_sd);
        sum1 = __builtin_fmaf(tmp, w, sum1); sum2 += w;
      }
    }
    if (!aborted && sum2 > 0.f) result = sum1 / sum2;
  }
  out[y * cols + x] = result;
}

// kf_preprocess's first launch: u16 -> f32 + gate + bilateral of one 64x4 tile per workgroup (bilateral_tile.h)
template <int R, bool FAST>
__global__ void __launch_bounds__(256) k_gate_bilateral(KfBilateralArgs b) {
  __shared__ float tile[(BIL_TX + 2 * R) * (BIL_TY + 2 * R)];
  kf_bilateral_tile<R, FAST>(b, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, tile);
}

// depthToVerticesKernel + verticesToNormalsKernel (VerticesNormalsCalculater.cu:15-66) in one launch: the four neighbour
// vertices a normal needs are recomputed from the filtered depth with the same arithmetic instead of being re-read.
__global__ void __launch_bounds__(256) k_vertices_normals(const float* __restrict__ depth, float4* __restrict__ out_v, float4* __restrict__ out_n, KfCam cam) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cam.cols || y >= cam.rows) return;
  kf_vertex_normal_pixel(depth, out_v, out_n, cam, x, y);                      // bilateral_tile.h
}

// VerticesNormalsCalculater.cu:15-33
__global__ void __launch_bounds__(256) k_depth_to_vertices(const float* __restrict__ depth, float4* __restrict__ out, KfCam cam) {
  int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cam.cols || y >= cam.rows) return;
  float d = depth[y * cam.cols + x];
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d != 0.f) { float3 v = kf_depth_to_skeleton((unsigned)x, (unsigned)y, d, cam); r = make_float4(v.x, v.y, v.z, 1.0f); }
  out[y * cam.cols + x] = r;
}

// VerticesNormalsCalculater.cu:35-66
__global__ void __launch_bounds__(256) k_vertices_to_normals(const float4* __restrict__ in, float4* __restrict__ out, int cols, int rows) {
  int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cols || y >= rows) return;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(x == cols - 1 || y == rows - 1 || x == 0 || y == 0)) {
    int i = y * cols + x;
    float4 v0 = in[i], vr = in[i + 1], vu = in[i + cols], vl = in[i - 1], vd = in[i - cols];
    if (v0.z != 0.f && vr.z != 0.f && vu.z != 0.f && vl.z != 0.f && vd.z != 0.f) {
      float3 c = kf_normalize(kf_cross(kf_sub(kf3(vu.x, vu.y, vu.z), kf3(vd.x, vd.y, vd.z)), kf_sub(kf3(vr.x, vr.y, vr.z), kf3(vl.x, vl.y, vl.z))));
      r = make_float4(c.x, c.y, c.z, 0.f);
    }
  }
  out[y * cols + x] = r;
}

// One thread per 2x2 block of level-1 pixels (= one level-2 pixel): reads 16 level-0 texels, writes 4 level-1 and 1 level-2.
// blockIdx.z: 0 = vertices, 1 = normals.
// slots 0/1 = new vertices/normals, 2/3 = model vertices/normals (odd slot = normal averaging rule)
struct PyrMaps { const float4* in[4]; float4* l1[4]; float4* l2[4]; };
// begin_mode >= 0: block (0,0,0) also runs the start-of-tracking bookkeeping (k_track_begin) so a frame needs no extra launch
__global__ void __launch_bounds__(256) k_pyramid(PyrMaps m, int cols0, int rows0, int levels, int kind_base,
                                                 KfTrackState* st, KfGridBarrier* gb, int begin_mode) {
  if (begin_mode >= 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    if (threadIdx.x < 10 && gb) reinterpret_cast<KfPaddedCounter*>(gb)[threadIdx.x].v = 0u;      // 8 groups + top + gen
    if (threadIdx.x == 0) {
      st->status = KF_TRACK_OK; st->iterations = 0; st->converged = 0; st->arrive = 0u;
      if (begin_mode == 0) st->tracked = 1;
      else {
        st->tracked = 0;
        for (int i = 0; i < 16; ++i) st->cur[0][i] = st->pose[i];            // ICP.cpp:62
        kf_mat44_inverse(st->pose, st->last_inv);                           // ICP.cpp:63
      }
    }
  }
  const int kind = kind_base + blockIdx.z;
  const float4* __restrict__ in = m.in[kind];
  float4* __restrict__ o1 = m.l1[kind];
  float4* __restrict__ o2 = m.l2[kind];
  const int c1 = cols0 >> 1, r1 = rows0 >> 1, c2 = c1 >> 1, r2 = r1 >> 1;
  const int bx = blockIdx.x * 32 + (threadIdx.x & 31), by = blockIdx.y * 8 + (threadIdx.x >> 5);
  float4 q[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int x1 = bx * 2 + i, y1 = by * 2 + j;
      q[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x1 < c1 && y1 < r1) {
        const float4* p = in + (size_t)(2 * y1) * cols0 + 2 * x1;
        float4 p00 = p[0], p01 = p[1], p10 = p[cols0], p11 = p[cols0 + 1];
        q[j][i] = (kind & 1) ? pyr_normal(p00, p01, p10, p11) : pyr_vertex(p00, p01, p10, p11);
        o1[(size_t)y1 * c1 + x1] = q[j][i];
      }
    }
  if (levels > 2 && bx < c2 && by < r2)
    o2[(size_t)by * c2 + bx] = (kind & 1) ? pyr_normal(q[0][0], q[0][1], q[1][0], q[1][1]) : pyr_vertex(q[0][0], q[0][1], q[1][0], q[1][1]);
}

static void pyr_maps(kf_ctx* c, PyrMaps& m) {
  float4** src[4] = {c->new_v, c->new_n, c->model_v, c->model_n};
  for (int k = 0; k < 4; ++k) { m.in[k] = src[k][0]; m.l1[k] = src[k][1]; m.l2[k] = c->levels > 2 ? src[k][2] : nullptr; }
}
int kf_launch_pyramids(kf_ctx* c, bool model, bool vertices, bool normals) {
  if (c->levels < 2 || (!vertices && !normals)) return 0;
  PyrMaps m; pyr_maps(c, m);
  int c1 = c->cols >> 1, r1 = c->rows >> 1;
  dim3 grid(kf_div_up(kf_div_up(c1, 2), 32), kf_div_up(kf_div_up(r1, 2), 8), (vertices && normals) ? 2 : 1);
  hipLaunchKernelGGL(k_pyramid, grid, dim3(256), 0, c->stream, m, c->cols, c->rows, c->levels, (model ? 2 : 0) + (vertices ? 0 : 1),
                     (KfTrackState*)nullptr, (KfGridBarrier*)nullptr, -1);
  return (int)hipGetLastError();
}
// the pyramids (ICP.cpp:57-60) that do not describe their level 0 yet -- the raycast launch leaves the model maps' behind, its riders the
// prefetched frame's (raycast.hip) -- plus the start-of-tracking bookkeeping in ONE launch
int kf_launch_pyramids_and_begin(kf_ctx* c, int begin_mode) {
  PyrMaps m; pyr_maps(c, m);
  int c1 = c->cols >> 1, r1 = c->rows >> 1;
  const bool need_new = !c->new_pyr_ok, need_model = !c->model_pyr_ok;
  const bool none = c->levels < 2 || (!need_new && !need_model);
  const int kinds = none ? 1 : (need_new && need_model ? 4 : 2), kind_base = (!none && !need_new) ? 2 : 0;
  dim3 grid(none ? 1 : kf_div_up(kf_div_up(c1, 2), 32), none ? 1 : kf_div_up(kf_div_up(r1, 2), 8), kinds);
  hipLaunchKernelGGL(k_pyramid, grid, dim3(256), 0, c->stream, m, none ? 0 : c->cols, none ? 0 : c->rows, c->levels, kind_base,
                     c->track, c->grid_barrier, begin_mode);
  c->new_pyr_ok = 1; c->model_pyr_ok = 1;
  return (int)hipGetLastError();
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------
extern "C" int kf_trunc_depth(kf_ctx* c, float tmin, float tmax) {
  if (!c) return KF_ERR_ARG;
  int st = kf_materialize_raw_depth(c);
  if (st) return st;
  int n = c->cols * c->rows;
  hipLaunchKernelGGL(k_trunc_depth, dim3(kf_div_up(n, 256)), dim3(256), 0, c->stream, c->raw_depth, c->trunced_depth, n, tmin, tmax);
  c->trunc_serial++;
  return (int)hipGetLastError();
}

extern "C" int kf_bilateral_filter_depth(kf_ctx* c, float sigma_pixel, float sigma_depth) {
  if (!c) return KF_ERR_ARG;
  // DataPreprocesser.cu:95-96: 0.5 is a double literal; the quotient is narrowed to float
  float sd_inv = (float)(0.5 / (double)(sigma_depth * sigma_depth));
  float ss_inv = (float)(0.5 / (double)(sigma_pixel * sigma_pixel));
  int radius = (int)ceil(2.0 * (double)sigma_pixel);                                  // :45
  if (radius < 0 || radius > BIL_MAXR) return KF_ERR_ARG;
  dim3 grid(kf_div_up(c->cols, BIL_TX), kf_div_up(c->rows, BIL_TY));
  hipLaunchKernelGGL(k_bilateral, grid, dim3(256), 0, c->stream, c->trunced_depth, c->filtered_depth, c->cols, c->rows,
                     ss_inv, sd_inv, sigma_depth, radius);
  return (int)hipGetLastError();
}

extern "C" int kf_calculate_new_vertices(kf_ctx* c, const kf_camera_params* cam) {
  if (!c || !cam || (int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  dim3 grid(kf_div_up(c->cols, 64), kf_div_up(c->rows, 4));
  c->new_pyr_ok = 0;
  hipLaunchKernelGGL(k_depth_to_vertices, grid, dim3(256), 0, c->stream, c->filtered_depth, c->new_v[0], to_cam(cam));
  return (int)hipGetLastError();
}

extern "C" int kf_calculate_new_normals(kf_ctx* c) {
  if (!c) return KF_ERR_ARG;
  dim3 grid(kf_div_up(c->cols, 64), kf_div_up(c->rows, 4));
  c->new_pyr_ok = 0;
  hipLaunchKernelGGL(k_vertices_to_normals, grid, dim3(256), 0, c->stream, c->new_v[0], c->new_n[0], c->cols, c->rows);
  return (int)hipGetLastError();
}

// arguments of the fused gate + bilateral launch for a buffer set; *fast: the sentinel form applies (bilateral_tile.h)
void kf_bilateral_args(kf_ctx* c, const uint16_t* mm, const float* raw_in, float* raw_out, float* trunced, float* filtered,
                       float tmin, float tmax, float sigma_pixel, float sigma_depth, bool build_tiles, KfBilateralArgs* b, bool* fast) {
  const float sd_inv = (float)(0.5 / (double)(sigma_depth * sigma_depth));
  const float ss_inv = (float)(0.5 / (double)(sigma_pixel * sigma_pixel));
  KfTileAccum acc; memset(&acc, 0, sizeof(acc));
  if (build_tiles) {                                       // layout of the two tables: kf_integrate_volume (integrate.hip)
    acc.tile = reinterpret_cast<int*>(c->tile_max_depth); acc.max_dist = c->fuse_max_dist;
    acc.off0 = 0; acc.w0 = kf_div_up(c->cols, 8); acc.off1 = acc.w0 * kf_div_up(c->rows, 8); acc.w1 = kf_div_up(c->cols, 16);
    acc.n = kf_defer_enabled(c) ? c->n_tile_floats : 0;      // the minima only matter to the cull's whole-brick retirement (deferred free-space weights)
  }
  // the sentinel form needs every valid depth far below the sentinel and the sentinel's tap weight to underflow to exactly zero
  *fast = tmax < 1e15f && sd_inv * KF_LOG2E > 1e-30f && sd_inv * KF_LOG2E < 1e30f;
  b->mm = mm; b->raw_in = raw_in; b->raw_out = raw_out; b->trunced = trunced; b->filtered = filtered; b->cols = c->cols; b->rows = c->rows;
  b->tmin = tmin; b->tmax = tmax; b->ss_inv = ss_inv; b->sd_inv = sd_inv; b->sigma_depth = sigma_depth; b->acc = acc;
}
int kf_launch_vertices_normals(kf_ctx* c, hipStream_t stream, const float* filtered, float4* v0, float4* n0, const kf_camera_params* cam) {
  dim3 grid2(kf_div_up(c->cols, 64), kf_div_up(c->rows, 4));
  hipLaunchKernelGGL(k_vertices_normals, grid2, dim3(256), 0, stream, filtered, v0, n0, to_cam(cam));
  return (int)hipGetLastError();
}
// the two fused launches of kf_preprocess on a given stream and buffer set
static int launch_fused_preprocess(kf_ctx* c, hipStream_t stream, const uint16_t* mm, const float* raw_in, float* raw_out, float* trunced, float* filtered,
                                   float4* v0, float4* n0, float tmin, float tmax, float sigma_pixel, float sigma_depth, const kf_camera_params* cam,
                                   bool build_tiles) {
  dim3 grid(kf_div_up(c->cols, BIL_TX), kf_div_up(c->rows, BIL_TY));
  KfBilateralArgs b; bool fast;
  kf_bilateral_args(c, mm, raw_in, raw_out, trunced, filtered, tmin, tmax, sigma_pixel, sigma_depth, build_tiles, &b, &fast);
  if (fast) hipLaunchKernelGGL((k_gate_bilateral<4, true>), grid, dim3(256), 0, stream, b);
  else hipLaunchKernelGGL((k_gate_bilateral<4, false>), grid, dim3(256), 0, stream, b);
  return kf_launch_vertices_normals(c, stream, filtered, v0, n0, cam);
}

extern "C" int kf_preprocess(kf_ctx* c, float tmin, float tmax, float sigma_pixel, float sigma_depth, const kf_camera_params* cam) {
  if (!c || !cam || (int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  int st;
  kf_evt_begin(c, KF_STAGE_PREPROCESS);
  const int radius = (int)ceil(2.0 * (double)sigma_pixel);
  const float want[4] = {tmin, tmax, sigma_pixel, sigma_depth};
  c->fp_pending = 0; c->fp_filtered = 0;                    // a recorded request no raycast has picked up is void now
  if (c->prefetch_valid && c->pending_mm && c->pending_mm == c->prefetch_src && memcmp(want, c->prefetch_params, sizeof(want)) == 0 &&
      memcmp(cam, &c->prefetch_cam, sizeof(*cam)) == 0) {
    // this very frame was preprocessed ahead of time on the side stream (kf_prefetch_frame): adopt its buffers
    float* t;
    t = c->raw_depth; c->raw_depth = c->alt_raw; c->alt_raw = t;
    t = c->trunced_depth; c->trunced_depth = c->alt_trunced; c->alt_trunced = t; c->trunc_serial++;
    t = c->filtered_depth; c->filtered_depth = c->alt_filtered; c->alt_filtered = t;
    float4* q;
    q = c->new_v[0]; c->new_v[0] = c->alt_v0; c->alt_v0 = q;
    q = c->new_n[0]; c->new_n[0] = c->alt_n0; c->alt_n0 = q;
    c->new_pyr_ok = 0;
    if (c->alt_pyr_ok && c->levels == 3 && c->alt_v12[0]) {     // the set came with its pyramids (riders of the raycast launch): they are adopted too
      for (int l = 0; l < 2; ++l) {
        q = c->new_v[l + 1]; c->new_v[l + 1] = c->alt_v12[l]; c->alt_v12[l] = q;
        q = c->new_n[l + 1]; c->new_n[l + 1] = c->alt_n12[l]; c->alt_n12[l] = q;
      }
      c->new_pyr_ok = 1;
    }
    c->alt_pyr_ok = 0;
    if (c->fp_done) {
      // produced by the previous frame's raycast launch on this very stream: nothing to wait for.  If that launch also built the integrate
      // tile tables for this depth map (and no integrate has cleared them since), they become current now.
      st = 0;
      if (c->fp_tiles && !c->tiles_clear) {
        c->tile_serial = c->trunc_serial; c->tile_built_dist = c->fp_tiles_dist;
        if (c->fp_tiles_min) c->tile_min_serial = c->trunc_serial;
      }
      c->fp_done = 0; c->fp_tiles = 0;
    } else st = (int)hipStreamWaitEvent(c->stream, c->ev_prefetched, 0);
    c->pending_mm = nullptr; c->prefetch_valid = 0;
    if (st == 0) st = kf_pending_depth_consumed(c);
  } else if (radius == 4) {                                    // stock sigma_pixel = 2: two fused launches instead of five
    c->prefetch_valid = 0;
    // the tile tables can ride along when they are clear (the last fusion pass cleared them) and an integration distance is known
    const bool build_tiles = c->tiles_clear && c->fuse_max_dist > 0.f;
    c->new_pyr_ok = 0;
    st = launch_fused_preprocess(c, c->stream, c->pending_mm, c->raw_depth, c->raw_depth, c->trunced_depth, c->filtered_depth, c->new_v[0], c->new_n[0],
                                 tmin, tmax, sigma_pixel, sigma_depth, cam, build_tiles);
    c->trunc_serial++;
    if (build_tiles) { c->tile_serial = c->trunc_serial; c->tile_built_dist = c->fuse_max_dist; c->tiles_clear = 0; if (kf_defer_enabled(c)) c->tile_min_serial = c->trunc_serial; }
    c->pending_mm = nullptr;
    if (st == 0) st = kf_pending_depth_consumed(c);
  } else {
    c->prefetch_valid = 0;
    if ((st = kf_trunc_depth(c, tmin, tmax))) return st;
    if ((st = kf_bilateral_filter_depth(c, sigma_pixel, sigma_depth))) return st;
    if ((st = kf_calculate_new_vertices(c, cam))) return st;
    st = kf_calculate_new_normals(c);
  }
  kf_evt_end(c, KF_STAGE_PREPROCESS);
  // whoever prefetches the next frame may start once this frame's maps exist and the previous frame's readers are behind us
  if (c->prefetch_in_use && st == 0) st = (int)hipEventRecord(c->ev_preprocessed, c->stream);
  return st;
}

// Preprocess the NEXT frame (device-resident u16 millimetres) on a side stream into the alternate buffer set, concurrently with
// whatever the main stream does next -- meant to be called right after kf_icp_track / kf_sdf_track has been enqueued: the
// persistent tracking loop occupies 150 of the 256 CUs with one workgroup each, the rest of the chip is idle for ~0.2 ms.
// The following kf_set_depth_mm_device(same pointer) + kf_preprocess(same parameters) then costs a pointer swap and an event
// wait.  Any other call sequence simply ignores the prefetched set.  Only the fused (sigma_pixel -> radius 4) path prefetches.
// Measured on MI355X (round 1, rocprofv3 two-queue trace): the overlap happens (the 23 us of preprocess run inside the tracking
// loop) but the two cross-stream dependencies cost ~12 us of bubbles on the main stream and the loop itself runs ~13 us longer
// with a neighbour on the chip -- a wash at VGA, so bench.py keeps it off by default (--prefetch turns it on).
extern "C" int kf_prefetch_frame(kf_ctx* c, const uint16_t* dev_mm, uint32_t cols, uint32_t rows, float tmin, float tmax,
                                 float sigma_pixel, float sigma_depth, const kf_camera_params* cam) {
  if (!c || !dev_mm || !cam || (int)cols != c->cols || (int)rows != c->rows || (int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  c->prefetch_valid = 0; c->fp_pending = 0; c->fp_done = 0; c->fp_filtered = 0;
  if ((int)ceil(2.0 * (double)sigma_pixel) != 4) return 0;
  KF_CHECK(hipSetDevice(c->cfg.device));
  const size_t npx = (size_t)c->cols * c->rows;
  if (!c->alt_raw) {
    // the alternate buffer set, all or nothing: a context whose set is incomplete must look like one without a set
    void* p[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t bytes[9] = {npx * 4, npx * 4, npx * 4, npx * sizeof(float4), npx * sizeof(float4), 0, 0, 0, 0};
    if (c->levels == 3)
      for (int l = 0; l < 2; ++l) {
        const size_t n = (size_t)(c->cols >> (l + 1)) * (c->rows >> (l + 1)) * sizeof(float4);
        bytes[5 + 2 * l] = bytes[6 + 2 * l] = n ? n : 16;
      }
    hipError_t e = hipSuccess;
    for (int i = 0; i < 9 && e == hipSuccess; ++i) if (bytes[i]) e = hipMalloc(&p[i], bytes[i]);
    if (e != hipSuccess) { for (int i = 0; i < 9; ++i) if (p[i]) hipFree(p[i]); return (int)e; }
    c->alt_raw = (float*)p[0]; c->alt_trunced = (float*)p[1]; c->alt_filtered = (float*)p[2]; c->alt_v0 = (float4*)p[3]; c->alt_n0 = (float4*)p[4];
    c->alt_v12[0] = (float4*)p[5]; c->alt_n12[0] = (float4*)p[6]; c->alt_v12[1] = (float4*)p[7]; c->alt_n12[1] = (float4*)p[8];
  }
  c->alt_pyr_ok = 0;
  { const int st = kf_upload_wait_for(c, dev_mm); if (st) return st; }   // a frame staged by kf_upload_depth_mm_next: its readers follow on this stream
  static int fused = -1;                                   // KF_PREFETCH_FUSED=0: the side-stream form (events between two streams)
  if (fused < 0) { const char* e = getenv("KF_PREFETCH_FUSED"); fused = e ? atoi(e) : 1; }
  if (fused) {
    // only a note: the next kf_raycast_volume* launch carries the filter along and leaves the preprocessed set in the alternate buffers
    // (they were last read by the frame BEFORE the current one, which lies behind us on the stream)
    c->fp_src = dev_mm; c->fp_params[0] = tmin; c->fp_params[1] = tmax; c->fp_params[2] = sigma_pixel; c->fp_params[3] = sigma_depth;
    c->fp_cam = *cam; c->fp_pending = 1;
    return 0;
  }
  if (!c->side_stream) {
    KF_CHECK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    KF_CHECK(hipEventCreateWithFlags(&c->ev_preprocessed, hipEventDisableTiming));
    KF_CHECK(hipEventCreateWithFlags(&c->ev_prefetched, hipEventDisableTiming));
    KF_CHECK(hipEventRecord(c->ev_preprocessed, c->stream));   // first use: everything enqueued so far
    c->prefetch_in_use = 1;
  }
  // the alternate set was last read by the frame BEFORE the current one; all of that precedes the current frame's preprocess
  KF_CHECK(hipStreamWaitEvent(c->side_stream, c->ev_preprocessed, 0));
  for (int i = 0; i < KF_UP_SLOTS; ++i) if (c->up_used[i] && dev_mm == c->up_dev[i]) KF_CHECK(hipStreamWaitEvent(c->side_stream, c->up_copied[i], 0));
  int st = launch_fused_preprocess(c, c->side_stream, dev_mm, nullptr, c->alt_raw, c->alt_trunced, c->alt_filtered, c->alt_v0, c->alt_n0,
                                   tmin, tmax, sigma_pixel, sigma_depth, cam, false);
  if (st) return st;
  KF_CHECK(hipEventRecord(c->ev_prefetched, c->side_stream));
  c->prefetch_src = dev_mm;
  c->prefetch_params[0] = tmin; c->prefetch_params[1] = tmax; c->prefetch_params[2] = sigma_pixel; c->prefetch_params[3] = sigma_depth;
  c->prefetch_cam = *cam;
  c->prefetch_valid = 1;
  return 0;
}

extern "C" int kf_downsample_new_vertices(kf_ctx* c) { return c ? kf_launch_pyramids(c, false, true, false) : KF_ERR_ARG; }
extern "C" int kf_downsample_new_normals(kf_ctx* c) { return c ? kf_launch_pyramids(c, false, false, true) : KF_ERR_ARG; }
extern "C" int kf_downsample_model_vertices(kf_ctx* c) { return c ? kf_launch_pyramids(c, true, true, false) : KF_ERR_ARG; }
extern "C" int kf_downsample_model_normals(kf_ctx* c) { return c ? kf_launch_pyramids(c, true, false, true) : KF_ERR_ARG; }
