// integrate.hip -- projective TSDF fusion (reference: integrateKernel src/cuda/integrateVolume.cu:15-77,
// tsdfvolume::updateVoxel src/cuda/tsdfVolume.h:57-75, wrapper :78-96).
//
// gfx950 design (HBM-bound kernel, see DESIGN.md "integrate"):
//   * the volume is bricked: an 8x8x8 brick is 4 KiB of contiguous (tsdf, weight) pairs, so one 256-thread workgroup
//     owns one brick and each lane moves one 16-byte pair-of-voxels -> every wave instruction is a full 1 KiB burst;
//   * only voxels that pass the reference's update predicate may touch memory.  A cull pass classifies bricks against
//     the camera frustum, the [0, max_dist + trunc) depth range and a 16x16-pixel tile max of the gated depth image, and
//     queues the survivors; the fusion pass walks that queue.  The tests are conservative (a brick is dropped only
//     when no voxel of it can pass the predicate), so the result is identical to visiting every voxel;
//   * the per-voxel arithmetic is the reference's, one rounded fp32 op per source op (-ffp-contract=off), hence the
//     updated-voxel count and the tsdf/weight bits match the CPU oracle exactly;
//   * per-brick flags (observed / has-negative) are maintained here; raycast and marching cubes use them to skip space.
#include "kf_internal.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>

#include "cull.h"                 // IntegrateArgs, the cull's test of a macro cell (shared with the tracking launch's tail)

// Retire the OTHER parity's counters (nobody touches them during this launch) and clear the tile tables for the next frame's fused
// preprocess kernel: run by workgroup 0 of the fusion pass, so a frame needs no bookkeeping launch.
__device__ __forceinline__ void integrate_maintenance(const IntegrateArgs& a) {
  const int o = a.parity ^ 1;
  if (threadIdx.x < 64) {
    a.cnt->upd_total_shard[threadIdx.x] += a.cnt->upd_shard[o][threadIdx.x * 16];
    a.cnt->upd_shard[o][threadIdx.x * 16] = 0ull;
  }
  if (threadIdx.x == 0) {
    a.cnt->n_active[o] = 0u;
    // one fusion pass = one frame fused, or one frame skipped because tracking failed (HybKinectfu.cpp:123): counted here, once per pass, so that the
    // cull stays free of side effects other than the queue -- it may have run as the tail of the tracking launch (track.hip)
    if (a.track && !a.track->tracked) a.cnt->frames_lost += 1; else a.cnt->frames_fused += 1;
  }
  if (a.clear_tiles) for (int i = threadIdx.x; i < a.n_tile_floats; i += blockDim.x) { a.tile_max[i] = 0.f; a.tile_max[a.n_tile_floats + i] = __builtin_huge_valf(); }
}


// pass 0: one workgroup per 16x16 pixel block -> max of the depth values that can integrate (0 < d < max_dist) over its four
// 8x8 tiles and over the block: the cull tests a brick against the finest of the two tables in which its footprint spans at most
// 4 x 4 tiles, so distant bricks (small footprints) get a much tighter maximum.  Coarser levels were tried (atomic maxima, or
// built per cull workgroup in LDS): they only matter for the few bricks next to the eye and cost more than they save.
// (Fallback: normally the fused preprocess kernel has built the tables already -- preprocess.hip, KfTileAccum.)
__global__ void __launch_bounds__(256) k_integrate_prepare(IntegrateArgs a) {
  __shared__ float s_q[4][2][2];                             // [wave][left / right half][max, min]
  const int bw = a.tile_w[1];
  const int tx = blockIdx.x % bw, ty = blockIdx.x / bw;
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x = tx * 16 + lx, y = ty * 16 + ly;
  float d = 0.f, mn = __builtin_huge_valf();
  if (x < a.dcam.cols && y < a.dcam.rows) { float v = a.depth[y * a.dcam.cols + x]; d = (v < a.max_dist) ? v : 0.f; mn = d; }   // minima: 0 = some pixel cannot integrate
  {                                                          // the tile minima, same reductions with fminf (pixels outside the image: +inf)
    mn = fminf(mn, __int_as_float(__builtin_amdgcn_update_dpp(0x7F800000, __float_as_int(mn), 0xB1, 0xf, 0xf, false)));
    mn = fminf(mn, __int_as_float(__builtin_amdgcn_update_dpp(0x7F800000, __float_as_int(mn), 0x4E, 0xf, 0xf, false)));
    mn = fminf(mn, __int_as_float(__builtin_amdgcn_update_dpp(0x7F800000, __float_as_int(mn), 0x141, 0xf, 0xf, false)));
  }
  const int mi = __float_as_int(mn);
  const float mleft = fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(mi, 0)), __int_as_float(__builtin_amdgcn_readlane(mi, 16))),
                            fminf(__int_as_float(__builtin_amdgcn_readlane(mi, 32)), __int_as_float(__builtin_amdgcn_readlane(mi, 48))));
  const float mright = fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(mi, 8)), __int_as_float(__builtin_amdgcn_readlane(mi, 24))),
                             fminf(__int_as_float(__builtin_amdgcn_readlane(mi, 40)), __int_as_float(__builtin_amdgcn_readlane(mi, 56))));
  // a wave holds four image rows of 16 pixels (one DPP row each): two quad permutes and a half-row mirror leave the maximum of
  // every 8-pixel half row in all of its lanes; eight readlanes then combine the four rows (no LDS shuffles)
  d = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0xB1, 0xf, 0xf, true)));    // quad_perm:[1,0,3,2]
  d = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x4E, 0xf, 0xf, true)));    // quad_perm:[2,3,0,1]
  d = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x141, 0xf, 0xf, true)));   // row_half_mirror
  const int di = __float_as_int(d);
  const float left = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(di, 0)), __int_as_float(__builtin_amdgcn_readlane(di, 16))),
                           fmaxf(__int_as_float(__builtin_amdgcn_readlane(di, 32)), __int_as_float(__builtin_amdgcn_readlane(di, 48))));
  const float right = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(di, 8)), __int_as_float(__builtin_amdgcn_readlane(di, 24))),
                            fmaxf(__int_as_float(__builtin_amdgcn_readlane(di, 40)), __int_as_float(__builtin_amdgcn_readlane(di, 56))));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { s_q[wave][0][0] = left; s_q[wave][1][0] = right; s_q[wave][0][1] = mleft; s_q[wave][1][1] = mright; }
  __syncthreads();
  float* tile_min = a.tile_max + a.n_tile_floats;
  if (threadIdx.x < 4) {                                     // level 0: quadrant (qx, qy) = waves 2*qy and 2*qy + 1, half qx
    const int hx = threadIdx.x & 1, hy = threadIdx.x >> 1;
    const float q = fmaxf(s_q[2 * hy][hx][0], s_q[2 * hy + 1][hx][0]);
    const float qm = fminf(s_q[2 * hy][hx][1], s_q[2 * hy + 1][hx][1]);
    const int qx = tx * 2 + hx, qy = ty * 2 + hy;
    if (qx < a.tile_w[0] && qy < a.tile_h[0]) { a.tile_max[a.tile_off[0] + qy * a.tile_w[0] + qx] = q; tile_min[a.tile_off[0] + qy * a.tile_w[0] + qx] = qm; }
  }
  if (threadIdx.x == 0) {
    float m16 = 0.f, n16 = __builtin_huge_valf();
#pragma unroll
    for (int w = 0; w < 4; ++w) { m16 = fmaxf(m16, fmaxf(s_q[w][0][0], s_q[w][1][0])); n16 = fminf(n16, fminf(s_q[w][0][1], s_q[w][1][1])); }
    a.tile_max[a.tile_off[1] + blockIdx.x] = m16;
    tile_min[a.tile_off[1] + blockIdx.x] = n16;
  }
}

// pass 1 (cull.h: cull_test): one WAVE per 32^3-voxel macro cell, sixteen per workgroup, ONE queue atomic per workgroup
#define CULL_WAVES 16
// DEFER: the deferred-weight words are in use -> whole free-space bricks in a deferred state can be retired here (see below); a separate
// instantiation because the extra test costs registers the plain cull needs for two workgroups per CU
template <bool DEFER>
__global__ void __launch_bounds__(CULL_WAVES * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) k_integrate_cull(IntegrateArgs a) {
  if (a.track && !a.track->tracked) return;                  // HybKinectfu.cpp:123: integrate only when tracking succeeded (counted by the fusion pass: integrate_maintenance)
  __shared__ unsigned s_cnt[CULL_WAVES], s_noop[CULL_WAVES];
  __shared__ unsigned s_base;
  const KfVolume& v = a.vol;
  const int nmxy = (v.nb + 3) >> 2;                          // macro cells per x / y
  const int mz0 = v.bz0 >> 2, mz1 = (v.bz1 + 3) >> 2;        // macro layers touching the stored bricks
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_macro = nmxy * nmxy * (mz1 - mz0);
  // (Several groups of cells per workgroup -- fewer, longer waves for the 262 144 macro cells of 2048^3 -- were measured and lost: the walking form needs
  // 112 registers, one workgroup per CU, and a block barrier pair per group: integrate stage 767 -> 801 / 817 / 889 us at 2, 4, 8 groups, 159 -> 177 us at
  // 1024^3, profiles/r04_deferred_weights.txt.)
  const int wave = blockIdx.x * CULL_WAVES + wid;
  int bx, by, bz;
  bool noop;
  const bool keep = cull_test<DEFER>(a, a.tinv ? a.tinv : a.tinv_val.m, wave, lane, n_macro, nmxy, mz0, bx, by, bz, noop);
  // compaction: ONE atomic per 16 macro cells (an address takes ~11 ns per atomic; per-wave atomics made this pass
  // cost more than the fusion itself at 1024^3)
  const unsigned long long mask = __ballot(keep);
  const unsigned n_noop = DEFER ? (unsigned)__popcll(__ballot(noop)) : 0u;
  if (lane == 0) { s_cnt[wid] = (unsigned)__popcll(mask); s_noop[wid] = n_noop; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0, retired = 0;
#pragma unroll
    for (int w = 0; w < CULL_WAVES; ++w) { const unsigned n = s_cnt[w]; s_cnt[w] = total; total += n; retired += s_noop[w]; }
    s_base = total ? atomicAdd(&a.cnt->n_active[a.parity], total) : 0u;
    if (retired) atomicAdd(&a.cnt->upd_shard[a.parity][(blockIdx.x & 63) * 16], (unsigned long long)retired * KF_BRICK_VOX);   // N_upd stays exact
  }
  __syncthreads();
  if (keep) {
    const unsigned pos = s_base + s_cnt[wid] + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    a.queue[pos] = (unsigned)bx | ((unsigned)by << 10) | ((unsigned)(bz - v.bz0) << 20);   // packed brick coordinates: no div/mod to decode
  }
}

// pass 1 for the LARGEST volumes (2048^3: 262 144 macro cells, of which the frustum meets a few per cent).  k_integrate_cull spends a wave, and a share of a
// workgroup's start-up chain, on every macro cell just to fail its sphere test.  Here a workgroup SIFTS 64 macro cells first, one per lane -- cells dealt
// round robin over the workgroups, so each sees the same mix of visible and invisible space -- and only the cells whose bounding sphere meets the frustum and
// the depth range get a wave's brick tests (cull_test_cell, two cells in flight per wave).  Survivors are staged in LDS and leave with ONE queue atomic per
// workgroup.  The same conservative tests as k_integrate_cull, hence the same survivors (in another order).  Measured (profiles/r04_cull_forms.txt): the
// cull's time is mostly the brick tests of the VISIBLE cells, which want every wave the chip has -- 2048^3 198 -> 138 us, but 1024^3 (40 % of the cells
// visible) 25 -> 33 us, so only volumes of >= 100 000 macro cells take this form; one workgroup per 128^3 super cell that leaves at once when its sphere fails
// was measured too (1024^3 25 -> 45 us: the visible workgroups walk 64 cells each on a fifth of the chip).
#define SIFT_THREADS 256
#ifndef SIFT_CELLS
#define SIFT_CELLS 64                                       // macro cells sifted per workgroup (one per lane of its first waves)
#endif
#define SIFT_STAGE (SIFT_CELLS * 64)                        // staged survivors per workgroup: every brick of every sifted cell fits (16 KiB)
template <bool DEFER>
__global__ void __launch_bounds__(SIFT_THREADS) k_integrate_cull_sift(IntegrateArgs a, int n_wg) {
  if (a.track && !a.track->tracked) return;
  __shared__ unsigned s_vis[SIFT_CELLS], s_stage[SIFT_STAGE];
  __shared__ unsigned s_nvis, s_nkept, s_base;
  const KfVolume& v = a.vol;
  const int nmxy = (v.nb + 3) >> 2, mz0 = v.bz0 >> 2, mz1 = (v.bz1 + 3) >> 2;
  const int n_macro = nmxy * nmxy * (mz1 - mz0);
  const float* m = a.tinv ? a.tinv : a.tinv_val.m;
  const float cell = v.cell;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { s_nvis = 0u; s_nkept = 0u; }
  __syncthreads();
  {
    const int i = (int)threadIdx.x * n_wg + (int)blockIdx.x;               // this lane's macro cell
    bool vis = false;
    if ((int)threadIdx.x < SIFT_CELLS && i < n_macro) {
      const int mx = i % nmxy, my = (i / nmxy) % nmxy, mz = i / (nmxy * nmxy) + mz0;
      float px, py, pz;
      vis = cull_sphere_visible(a, m, (float)(mx * 32 + 16) * cell, (float)(my * 32 + 16) * cell, (float)(mz * 32 + 16) * cell, 27.0f * cell + 1e-4f * v.size, px, py, pz);
    }
    const unsigned long long vm = __ballot(vis);
    unsigned base = 0;
    if (lane == 0 && vm) base = atomicAdd(&s_nvis, (unsigned)__popcll(vm));
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (vis) s_vis[base + (unsigned)__popcll(vm & ((1ull << lane) - 1ull))] = (unsigned)i;
  }
  __syncthreads();
  const unsigned nvis = s_nvis;
  unsigned retired = 0;
  for (unsigned k0 = (unsigned)wid * 2u; k0 < nvis; k0 += (SIFT_THREADS / 64) * 2u) {
    unsigned packed[2]; unsigned long long mask[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool have = k0 + u < nvis;
      const int i = have ? (int)s_vis[k0 + u] : 0;
      int bx, by, bz; bool noop;
      const bool keep = cull_test_cell<DEFER>(a, m, i % nmxy, (i / nmxy) % nmxy, i / (nmxy * nmxy) + mz0, have, lane, bx, by, bz, noop);
      packed[u] = (unsigned)bx | ((unsigned)by << 10) | ((unsigned)(bz - v.bz0) << 20);
      mask[u] = __ballot(keep);
      if (DEFER) retired += (unsigned)__popcll(__ballot(noop));
    }
    const unsigned n0 = (unsigned)__popcll(mask[0]), n1 = (unsigned)__popcll(mask[1]);
    if (n0 + n1) {
      unsigned pos = 0;
      if (lane == 0) pos = atomicAdd(&s_nkept, n0 + n1);                      // (cannot overflow: SIFT_CELLS x 64 bricks fit the stage)
      pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
      if ((mask[0] >> lane) & 1ull) s_stage[pos + (unsigned)__popcll(mask[0] & ((1ull << lane) - 1ull))] = packed[0];
      if ((mask[1] >> lane) & 1ull) s_stage[pos + n0 + (unsigned)__popcll(mask[1] & ((1ull << lane) - 1ull))] = packed[1];
    }
  }
  if (DEFER && retired && lane == 0) atomicAdd(&a.cnt->upd_shard[a.parity][((blockIdx.x * 4u + (unsigned)wid) & 63u) * 16], (unsigned long long)retired * KF_BRICK_VOX);   // N_upd stays exact
  __syncthreads();
  const unsigned staged = s_nkept;
  if (threadIdx.x == 0) s_base = staged ? atomicAdd(&a.cnt->n_active[a.parity], staged) : 0u;
  __syncthreads();
  for (unsigned p = threadIdx.x; p < staged; p += SIFT_THREADS) a.queue[s_base + p] = s_stage[p];
}

// pass 2: one workgroup walks the queue BR bricks at a time, one lane per x-adjacent voxel pair (16 contiguous bytes) of
// each brick.  The phases of the BR bricks are interleaved (all projections, all depth gathers, all predicates, all 16-byte
// loads, all updates) so a workgroup keeps BR x 4 KiB of HBM requests in flight instead of one dependent chain at a time.
template <bool HAS_COLOR, int BR>
__global__ void __launch_bounds__(256) k_integrate_bricks(IntegrateArgs a) {
  const KfVolume& v = a.vol;
  const unsigned n_active = a.cnt->n_active[a.parity];
  if (blockIdx.x == 0) integrate_maintenance(a);
  const float* m = a.tinv ? a.tinv : a.tinv_val.m;
  const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7], m8 = m[8], m9 = m[9], m10 = m[10], m11 = m[11];
  const float cell = v.cell;
  const int lx = (threadIdx.x & 3) * 2, ly = (threadIdx.x >> 2) & 7, lz = threadIdx.x >> 5;
  const KfRecip rtrunc = kf_recip(a.sdf_trunc);
  __shared__ unsigned s_upd;
  unsigned upd_total = 0;
  if (threadIdx.x == 0) s_upd = 0;
  __syncthreads();
  for (unsigned q0 = blockIdx.x * BR; q0 < n_active; q0 += gridDim.x * BR) {
    unsigned slot[BR], fold[BR]; int bx[BR], by[BR], bz[BR];
    float pfx[BR][2], pfy[BR][2], pfz[BR][2], d[BR][2]; int pix[BR][2]; bool ok[BR][2];
    // Phase A: project every voxel (tsdfVolume.h:38-49 voxel centre; Mat.h:230-238 row*vector summed left to right)
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      const bool live = q0 + b < n_active;
      const unsigned packed = live ? a.queue[q0 + b] : 0u;
      bx[b] = (int)(packed & 1023u); by[b] = (int)((packed >> 10) & 1023u); bz[b] = (int)(packed >> 20) + v.bz0;
      slot[b] = ((unsigned)(bz[b] - v.bz0) * (unsigned)v.nb + (unsigned)by[b]) * (unsigned)v.nb + (unsigned)bx[b];
      fold[b] = v.flags[slot[b]];                           // current flags: the atomic below is only issued when a bit is new
      const float wy = ((float)(by[b] * 8 + ly) + 0.5f) * cell, wz = ((float)(bz[b] * 8 + lz) + 0.5f) * cell;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float wx = ((float)(bx[b] * 8 + lx + k) + 0.5f) * cell;
        pfx[b][k] = m0 * wx + m1 * wy + m2 * wz + m3 * 1.0f;
        pfy[b][k] = m4 * wx + m5 * wy + m6 * wz + m7 * 1.0f;
        pfz[b][k] = m8 * wx + m9 * wy + m10 * wz + m11 * 1.0f;
        ok[b][k] = live && pfz[b][k] > 0.f;                                                 // :39 `if (pf.z <= 0) continue`
        int2 sp = make_int2(0, 0);
        if (ok[b][k]) {
          // DepthCamera.h:30-43 `v.x*fx/v.z + cx`, `(int)(p + 0.5)`: both quotients share the divisor pf.z
          const KfRecip rz = kf_recip(pfz[b][k]);
          sp.x = kf_round_px(kf_div(pfx[b][k] * a.dcam.fx, rz) + a.dcam.cx);
          sp.y = kf_round_px(kf_div(pfy[b][k] * a.dcam.fy, rz) + a.dcam.cy);
        }
        ok[b][k] = ok[b][k] && !(sp.x >= a.dcam.cols - 1 || sp.y >= a.dcam.rows - 1 || sp.x < 1 || sp.y < 1);   // :43
        pix[b][k] = ok[b][k] ? sp.y * a.dcam.cols + sp.x : 0;
      }
    }
    // depth gathers of all BR x 2 voxels back to back (L2-resident image)
#pragma unroll
    for (int b = 0; b < BR; ++b)
#pragma unroll
      for (int k = 0; k < 2; ++k) d[b][k] = ok[b][k] ? a.depth[pix[b][k]] : 0.f;
    // Phase B: the reference's remaining predicates (:50-67)
    uchar4 col[BR][2]; float normalz[BR][2]; bool upd[BR][2];
#pragma unroll
    for (int b = 0; b < BR; ++b)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        upd[b][k] = ok[b][k] && d[b][k] != 0.f;                                             // :50
        col[b][k] = make_uchar4(0, 0, 0, 0); normalz[b][k] = 0.f;
        if (HAS_COLOR) {
          if (upd[b][k]) {
            normalz[b][k] = a.normals[pix[b][k]].z;
            const int cxp = kf_to_int((double)(pfx[b][k] * 525 / pfz[b][k] + 320)), cyp = kf_to_int((double)(pfy[b][k] * 525 / pfz[b][k] + 240));   // :56-57
            upd[b][k] = !(cxp >= a.rcam.cols - 1 || cyp >= a.rcam.rows - 1 || cxp < 1 || cyp < 1);
            if (upd[b][k]) col[b][k] = a.rgb[(size_t)cyp * a.rcam.cols + cxp];
          }
        }
        upd[b][k] = upd[b][k] && (d[b][k] < a.max_dist) && ((d[b][k] - pfz[b][k]) > -a.sdf_trunc);   // :64, :67
      }
    // one 16-byte read-modify-write per lane and brick, only where a voxel of the pair passed; the loads go out together
    float4* p[BR]; float4 q[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      p[b] = reinterpret_cast<float4*>(v.tw + (size_t)slot[b] * KF_BRICK_VOX) + threadIdx.x;
      q[b] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((upd[b][0] || upd[b][1]) && KF_EXP_MODE(a) < 2) q[b] = *p[b];
    }
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      unsigned flags = 0;
      if (upd[b][0] || upd[b][1]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (!upd[b][k]) continue;
          const float sdf = d[b][k] - pfz[b][k];
          const float tsdf = fminf(1.0f, kf_div(sdf, rtrunc));
          const float ot = k ? q[b].z : q[b].x, ow = k ? q[b].w : q[b].y;
          const float nw = fminf(ow + 1.f, v.max_weight);                                   // tsdfVolume.h:65
          const float nt = kf_div(ot * ow + tsdf * 1.f, kf_recip(ow + 1.f));                // tsdfVolume.h:66
          if (k) { q[b].z = nt; q[b].w = nw; } else { q[b].x = nt; q[b].y = nw; }
          if (HAS_COLOR) {
            // :72 `(color_angled?fminf(1.0,abs(normalz)/0.75):1.0)*2.0` -- the division and the doubling run in double
            const float wc = a.color_angled ? (float)((double)fminf(1.0f, (float)((double)fabsf(normalz[b][k]) / 0.75)) * 2.0) : 2.0f;
            uchar4* cp = v.color + (size_t)slot[b] * KF_BRICK_VOX + threadIdx.x * 2 + k;
            const uchar4 oc = *cp;
            const float c0 = fminf(255.0f, ((float)oc.x * ow + (float)col[b][k].x * wc) / (ow + wc));   // tsdfVolume.h:68-70
            const float c1 = fminf(255.0f, ((float)oc.y * ow + (float)col[b][k].y * wc) / (ow + wc));
            const float c2 = fminf(255.0f, ((float)oc.z * ow + (float)col[b][k].z * wc) / (ow + wc));
            *cp = make_uchar4((unsigned char)c0, (unsigned char)c1, (unsigned char)c2, 0);
          }
          ++upd_total;
          flags |= KF_FLAG_OBSERVED | (nt < 0.f ? KF_FLAG_HASNEG : 0u);
        }
        if (KF_EXP_MODE(a) < 1) *p[b] = q[b];
        else if (q[b].x == 123.456f) *p[b] = q[b];
      }
      // brick flags: each wave ORs its NEW bits into the brick's byte with a fire-and-forget 32-bit atomic -- no barrier and
      // no dependent read-modify-write on the loop's critical path.  Memory-side atomics cost a CU ~50 ns per wave
      // instruction (MI355X_MICROARCH.md 'Global float atomics'), so the flags read at the top of the iteration gates
      // them: after the first frames almost no brick changes its flags and no atomic is issued at all.
      const unsigned wflags = ((__ballot(flags & KF_FLAG_OBSERVED) ? KF_FLAG_OBSERVED : 0u) | (__ballot(flags & KF_FLAG_HASNEG) ? KF_FLAG_HASNEG : 0u)) & ~fold[b];
      if (wflags && (threadIdx.x & 63) == 0) {
        atomicOr(reinterpret_cast<unsigned*>(v.flags) + (slot[b] >> 2), wflags << (8u * (slot[b] & 3u)));
        if (wflags & KF_FLAG_HASNEG) {
          kf_mark_macro(v, bx[b], by[b], bz[b]);
          atomicOr(&v.negbits[slot[b] >> 5], 1u << (slot[b] & 31u));
        }
      }
      // (this kernel knows nothing of the deferred-weight words of k_integrate_pairs<.., DEFER>: kf_integrate_volume flushes them before it
      // runs and clears them behind it)
    }
  }
  // N_upd: wave sum -> LDS -> ONE atomic per workgroup, spread over 64 counter lines
  float s = kf_wave_sum((float)upd_total);          // < 2^24 per wave: exact
  if ((threadIdx.x & 63) == 0 && s > 0.f) atomicAdd(&s_upd, (unsigned)s);
  __syncthreads();
  if (threadIdx.x == 0 && s_upd) atomicAdd(&a.cnt->upd_shard[a.parity][(blockIdx.x & 63) * 16], (unsigned long long)s_upd);
}

// ---- the same fusion pass with the lane's two x-adjacent voxels carried as one 2-vector ------------------------------------------
// The pass is VALU-bound, not HBM-bound (~240 vector instructions per lane and brick at 8 waves per SIMD; tools/bench_integrate.py
// gives the same time for 1, 2 or 4 bricks in flight and for 2048..8192 workgroups), so the instruction count is what the kernel's
// distance to the HBM roofline is made of.  gfx950 issues v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on two fp32 values at the cost
// of one: every multiply, add and fused step of the reference's per-voxel arithmetic -- still one rounded operation per source
// operation, so the bits do not change -- runs once for the pair.  The code is straight-line (no per-voxel branches: a voxel behind
// the camera gets a harmless divisor and is masked at the end), and the pixel rounding uses floor(p + 0.5f), which equals the
// reference's (int)(p + 0.5) in double for every p the bounds test can accept (exact sum below 2^23; checked exhaustively near every
// binade boundary by the CPU test test_pixel_rounding_floor_form_equals_reference_double_form).
struct KfRecip2 { kf_f2 den, r; };
__device__ __forceinline__ KfRecip2 kf_recip2(kf_f2 b) {                   // kf_recip on both halves
  kf_f2 r0 = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
  const kf_f2 e0 = f2_fma(-b, r0, f2_splat(1.0f));
  KfRecip2 k; k.den = b; k.r = f2_fma(e0, r0, r0);
  return k;
}
__device__ __forceinline__ kf_f2 kf_div2(kf_f2 a, const KfRecip2& k) {     // kf_div on both halves
  const kf_f2 q0 = a * k.r;
  const kf_f2 e1 = f2_fma(-k.den, q0, a);
  const kf_f2 q1 = f2_fma(e1, k.r, q0);
  const kf_f2 e2 = f2_fma(-k.den, q1, a);
  return f2_fma(e2, k.r, q1);
}

// tsdfVolume.h:68-70 on the pair's two colour words (c0 | c1 << 8 | c2 << 16), observed colour words `col`, OLD weights `ow`; integrateVolume.cu:72
// `(color_angled ? fminf(1.0, abs(normalz) / 0.75) : 1.0) * 2.0`.  Both voxels at once, packed like the rest of the kernel.
__device__ __forceinline__ uint2 integrate_color_update2(uint2 oc, unsigned col0, unsigned col1, kf_f2 ow, kf_f2 nz, bool u0, bool u1, int color_angled, const KfRecip2& r075) {
  // The reference forms the weight in double: fminf(1.0, abs(normalz) / 0.75) * 2.0.  Narrowed to float, that quotient equals the
  // correctly rounded FP32 quotient |nz| / 0.75f for every float |nz|: 0.75 is a float, the exact quotient 4|nz|/3 is either exact or
  // has the repeating tail 0101.. / 1010.., never within double rounding's reach of a float midpoint (checked exhaustively over a
  // binade and the denormals: the CPU test test_color_weight_fp32_form_equals_reference_double_form); the doubling
  // is exact.  (A denormal |nz| gives a weight the divisor test below sends down the compiler's full division anyway.)
  kf_f2 wc = f2_splat(2.0f);
  if (color_angled) {
    const kf_f2 an = {fabsf(nz.x), fabsf(nz.y)};
    kf_f2 qn = kf_div2(an, r075);
    if (__ballot((u0 && !(an.x >= 0x1p-60f || an.x == 0.f)) || (u1 && !(an.y >= 0x1p-60f || an.y == 0.f))) != 0ull) { qn.x = an.x / 0.75f; qn.y = an.y / 0.75f; }   // tiny or NaN: full division
    wc.x = 2.0f * fminf(1.0f, qn.x); wc.y = 2.0f * fminf(1.0f, qn.y);
  }
  const kf_f2 den = ow + wc;
  // the three IEEE quotients of a voxel share their divisor: one refined reciprocal per voxel (kf_div2, exact in the normal range); a
  // divisor outside it (first observation at a grazing normal: ow = 0, wc tiny or 0) takes the compiler's full division -- wave-uniform, rare
  const bool slow = __ballot((u0 && !(den.x >= 0x1p-60f && den.x <= 0x1p60f)) || (u1 && !(den.y >= 0x1p-60f && den.y <= 0x1p60f))) != 0ull;
  const kf_f2 dsafe = {u0 ? den.x : 1.0f, u1 ? den.y : 1.0f};
  const KfRecip2 rd = kf_recip2(dsafe);
  unsigned w0 = 0u, w1 = 0u;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const kf_f2 o = {(float)((oc.x >> (8 * ch)) & 255u), (float)((oc.y >> (8 * ch)) & 255u)}, cn = {(float)((col0 >> (8 * ch)) & 255u), (float)((col1 >> (8 * ch)) & 255u)};
    const kf_f2 num = o * ow + cn * wc;
    kf_f2 q = kf_div2(num, rd);
    if (slow) { q.x = num.x / den.x; q.y = num.y / den.y; }
    w0 |= (unsigned)(unsigned char)fminf(255.0f, q.x) << (8 * ch);
    w1 |= (unsigned)(unsigned char)fminf(255.0f, q.y) << (8 * ch);
  }
  return make_uint2(u0 ? w0 : oc.x, u1 ? w1 : oc.y);
}

// DEFER (the default without colour): free space the camera keeps looking through is observed as tsdf 1 frame after frame, and for a voxel
// that already holds tsdf 1 the update of tsdfVolume.h:65-66 is (1 * w + 1) / (w + 1) = 1 exactly and w' = min(w + 1, max_weight) -- only the
// weight moves, by one, for every voxel alike.  A 16-bit word per quarter brick (KfVolume::pend, kf_internal.h: the 128 voxels one wave owns)
// says "all 128 hold tsdf 1 and a weight >= 1, and k whole-quarter free-space observations are pending"; a wave whose 128 voxels ALL pass
// the predicate and ALL lie in front of the truncation band then neither reads nor writes them -- it bumps k (KF_PEND_SAT once every weight
// must have reached max_weight: from then on ANY free-space wave over the quarter is the identity and is only counted).  Whoever has to
// write into such a quarter (a surface band entering it, a partial wave at the frustum's edge or around a depth hole) first applies the
// pending count to all 128 voxels: w <- fminf(w + k, max_weight), which equals k applications of min(w + 1, max) on these small integers bit
// for bit.  The word is (re)established by a wave that knows all 128 voxels and leaves nothing but (tsdf 1, weight >= 1) behind.  Readers of
// the volume (raycast, marching cubes, SDF tracker) look at tsdf and at weight != 0 only -- a deferred quarter's stored weights are >= 1 --
// and kf_download_volume applies the pending count on the fly.  From the second frame of a stream on, two thirds (512^3) to three quarters
// (1024^3) of the waves that would touch memory are of this kind.
// COLOR (the reference's use_color, optionally color_angle_weight: its stock switches): the colour plane rides along -- the voxel's colour
// projection (integrateVolume.cu:56-63) joins the update predicate, the pair's two colour words are one more 8-byte read-modify-write,
// the running average of tsdfVolume.h:68-70 keeps the reference's double-precision weight expression and its IEEE quotients.  Colour
// changes even where (tsdf, weight) no longer do and blends with the weight, so COLOR excludes DEFER (pending counts are flushed before a
// colour frame and the words cleared behind it: kf_integrate_volume).
// COUNT: the launch also counts the voxels it observes for the FIRST time (weight 0 -> > 0, owned layers) into KfCounters::wgt0_shard -- the running count behind
// kf_get_volume_stats.  An instantiation of its own, launched only while a host keeps asking for the count (kf_ctx::wgt0_tracking): two compares and two
// ballots per brick turn out to cost the kernel 3.6 us at 512^3 and 9 us at 1024^3 (profiles/r05_observed_count.txt) -- not something every frame should pay.
template <int BR, bool DEFER, bool COLOR = false, bool LAYERS = false, bool COUNT = false>
__global__ void __launch_bounds__(256) k_integrate_pairs(IntegrateArgs a) {
  static_assert(!(DEFER && COLOR), "colour changes where (tsdf, weight) do not, and blends with the weight: no deferral");
  const KfVolume& v = a.vol;
  const unsigned n_active = a.cnt->n_active[a.parity] >> ((KF_EXP_MODE(a) == 8 || KF_EXP_MODE(a) == 9) ? KF_EXP_MODE(a) - 7 : 0);     // exp_mode 8 / 9: half / quarter of the queue (timing only)
  if (blockIdx.x == 0) integrate_maintenance(a);
  const float* m = a.tinv ? a.tinv : a.tinv_val.m;
  const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7], m8 = m[8], m9 = m[9], m10 = m[10], m11 = m[11];
  const float cell = v.cell;
  const int lx = (threadIdx.x & 3) * 2, ly = (threadIdx.x >> 2) & 7, lz = threadIdx.x >> 5;
  const KfRecip rt = kf_recip(a.sdf_trunc);
  KfRecip2 rtrunc; rtrunc.den = f2_splat(rt.den); rtrunc.r = f2_splat(rt.r);
  const unsigned xlim = (unsigned)(a.dcam.cols - 2), ylim = (unsigned)(a.dcam.rows - 2);
  const unsigned quarter = threadIdx.x >> 6;                            // this wave's quarter of the brick (z layers 2w, 2w + 1)
  const unsigned sat_w = __float_as_uint(v.max_weight), one_f = __float_as_uint(1.0f);
  unsigned short* const pend16 = reinterpret_cast<unsigned short*>(v.pend);      // quarter q of brick slot s: pend16[4 s + q]
  const KfRecip2 r075 = kf_recip2(f2_splat(0.75f));                     // COLOR: the angle weight's |nz| / 0.75
  __shared__ unsigned s_upd, s_new;
  __shared__ unsigned s_layer[LAYERS ? 1024 : 1];                        // LAYERS (sampled frames: a launch of its own instantiation): this workgroup's update counts per brick layer (kf_create: <= 1024 brick layers)
  unsigned upd_total = 0, new_wave = 0;
  if (threadIdx.x == 0) { s_upd = 0; s_new = 0; }
  if (LAYERS) for (int i = threadIdx.x; i < v.nb; i += 256) s_layer[i] = 0u;
  __syncthreads();
  // The queue entries of an iteration are requested one iteration ahead.  On gfx9-family hardware loads and stores share one in-order
  // counter (vmcnt): a queue load issued AFTER the previous iteration's voxel stores can only be waited for together with them, which
  // puts the stores' completion on the chain queue -> projection -> depth gather -> voxel load of every iteration.  Requested before
  // the stores, the entries are older than them and the next iteration starts projecting while its predecessor's stores drain.
  // The loads are made to look lane-varying (kf_opaque) on purpose: for a wave-uniform load the compiler moves the result into a scalar
  // register with v_readfirstlane right behind the load, i.e. waits for it on the spot -- which serialised the queue and flag loads of
  // the BR bricks into 2 x BR dependent round trips at the top of every iteration.  Here they all go out together and are made
  // scalar (readfirstlane) only where they are used.
#ifndef KF_INT_NO_QPREFETCH
  unsigned ahead[BR];
#pragma unroll
  // (the first entries are requested without waiting for the queue length -- every index below the queue's capacity is readable, and an
  // entry at or beyond n_active is discarded below -- so the two loads of a workgroup's start-up travel together: at 512^3 a workgroup
  // lives for one or two iterations and its start-up round trips are a visible part of the kernel)
  for (int b = 0; b < BR; ++b) { const unsigned i = kf_opaque(blockIdx.x * BR + b); ahead[b] = a.queue[i < a.queue_cap ? i : 0u]; }
  // (waited for here rather than at the loop's head, where the wait would be repeated in every iteration -- see the note in front of the stores below)
#pragma unroll
  for (int b = 0; b < BR; ++b) asm volatile("" : "+v"(ahead[b]));
#endif
  for (unsigned q0 = blockIdx.x * BR; q0 < n_active; q0 += gridDim.x * BR) {
    unsigned slot[BR], fold[BR], ent[BR];                        // ent: the packed brick coordinates (decoded where needed: scalar registers are scarce here)
    kf_f2 pfz[BR], d[BR]; int pix0[BR], pix1[BR]; bool ok0[BR], ok1[BR];
    int cpix0[BR], cpix1[BR]; bool okc0[BR], okc1[BR];          // COLOR: the voxels' pixels in the colour image, and whether they lie inside its window
#ifndef KF_INT_NO_QPREFETCH
    unsigned entry[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) entry[b] = (q0 + b < n_active) ? (unsigned)__builtin_amdgcn_readfirstlane((int)ahead[b]) : 0u;
    {
      const unsigned qn = q0 + gridDim.x * BR;
#pragma unroll
      for (int b = 0; b < BR; ++b) { const unsigned i = kf_opaque(qn + b); ahead[b] = (i < n_active) ? a.queue[i] : 0u; }
    }
#endif
    // the bricks' flag bytes (the atomics below are only issued when a bit is new) and, DEFER, the quarter's deferred-weight word: requested together
    unsigned fold_v[BR], pend_v[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
#ifndef KF_INT_NO_QPREFETCH
      const unsigned packed = entry[b];
#else
      const unsigned packed = (q0 + b < n_active) ? a.queue[q0 + b] : 0u;
#endif
      const unsigned sl = ((packed >> 20) * (unsigned)v.nb + ((packed >> 10) & 1023u)) * (unsigned)v.nb + (packed & 1023u);
      fold_v[b] = v.flags[kf_opaque(sl)];
      pend_v[b] = DEFER ? (unsigned)pend16[kf_opaque(sl * 4u + quarter)] : 0u;
    }
    // Phase A: project both voxels (tsdfVolume.h:38-49 voxel centre; Mat.h:230-238 row * vector summed left to right)
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      const bool live = q0 + b < n_active;
#ifndef KF_INT_NO_QPREFETCH
      const unsigned packed = entry[b];
#else
      const unsigned packed = live ? a.queue[q0 + b] : 0u;
#endif
      ent[b] = packed;
      const int bx = (int)(packed & 1023u), by = (int)((packed >> 10) & 1023u), bz = (int)(packed >> 20) + v.bz0;
      slot[b] = ((unsigned)(bz - v.bz0) * (unsigned)v.nb + (unsigned)by) * (unsigned)v.nb + (unsigned)bx;
      const float x0 = (float)(bx * 8 + lx);
      kf_f2 xi = {x0, x0 + 1.0f};                           // (float)(x + 1) == (float)x + 1 for these small integers
      const kf_f2 wx = (xi + f2_splat(0.5f)) * f2_splat(cell);
      const float wy = ((float)(by * 8 + ly) + 0.5f) * cell, wz = ((float)(bz * 8 + lz) + 0.5f) * cell;
      const kf_f2 pfx = ((f2_splat(m0) * wx + f2_splat(m1 * wy)) + f2_splat(m2 * wz)) + f2_splat(m3 * 1.0f);
      const kf_f2 pfy = ((f2_splat(m4) * wx + f2_splat(m5 * wy)) + f2_splat(m6 * wz)) + f2_splat(m7 * 1.0f);
      pfz[b] = ((f2_splat(m8) * wx + f2_splat(m9 * wy)) + f2_splat(m10 * wz)) + f2_splat(m11 * 1.0f);
      const bool z0 = live && pfz[b].x > 0.f, z1 = live && pfz[b].y > 0.f;                   // :39 `if (pf.z <= 0) continue`
      kf_f2 zs = {z0 ? pfz[b].x : 1.0f, z1 ? pfz[b].y : 1.0f};                               // masked voxels: any finite divisor
      const KfRecip2 rz = kf_recip2(zs);
      // DepthCamera.h:30-43 `v.x*fx/v.z + cx`, `(int)(p + 0.5)`: both quotients share the divisor pf.z
      const kf_f2 px = kf_div2(pfx * f2_splat(a.dcam.fx), rz) + f2_splat(a.dcam.cx) + f2_splat(0.5f);
      const kf_f2 py = kf_div2(pfy * f2_splat(a.dcam.fy), rz) + f2_splat(a.dcam.cy) + f2_splat(0.5f);
      const int sx0 = (int)floorf(px.x), sx1 = (int)floorf(px.y), sy0 = (int)floorf(py.x), sy1 = (int)floorf(py.y);
      // :43 `sp.x >= cols-1 || sp.y >= rows-1 || sp.x < 1 || sp.y < 1` rejects: one unsigned compare per coordinate
      ok0[b] = z0 && (unsigned)(sx0 - 1) < xlim && (unsigned)(sy0 - 1) < ylim;
      ok1[b] = z1 && (unsigned)(sx1 - 1) < xlim && (unsigned)(sy1 - 1) < ylim;
      pix0[b] = ok0[b] ? sy0 * a.dcam.cols + sx0 : 0;
      pix1[b] = ok1[b] ? sy1 * a.dcam.cols + sx1 : 0;
      if (COLOR) {
        // :56-57 `int(pf.x * 525 / pf.z + 320)`, `int(pf.y * 525 / pf.z + 240)` -- the reference's literal intrinsics, fp32, truncated;
        // :59 the window test against the colour camera's size
        const kf_f2 cx = kf_div2(pfx * f2_splat(525.f), rz) + f2_splat(320.f), cy = kf_div2(pfy * f2_splat(525.f), rz) + f2_splat(240.f);
        const int cx0 = kf_f2i(cx.x), cx1 = kf_f2i(cx.y), cy0 = kf_f2i(cy.x), cy1 = kf_f2i(cy.y);
        const unsigned cxlim = (unsigned)(a.rcam.cols - 2), cylim = (unsigned)(a.rcam.rows - 2);
        okc0[b] = (unsigned)(cx0 - 1) < cxlim && (unsigned)(cy0 - 1) < cylim;
        okc1[b] = (unsigned)(cx1 - 1) < cxlim && (unsigned)(cy1 - 1) < cylim;
        cpix0[b] = okc0[b] ? cy0 * a.rcam.cols + cx0 : 0;
        cpix1[b] = okc1[b] ? cy1 * a.rcam.cols + cx1 : 0;
      }
    }
    // depth gathers of all BR x 2 voxels back to back (L2-resident image)
#pragma unroll
    for (int b = 0; b < BR; ++b) { d[b].x = ok0[b] ? a.depth[pix0[b]] : 0.f; d[b].y = ok1[b] ? a.depth[pix1[b]] : 0.f; }
    // Phase B: the reference's remaining predicates (:50, :64, :67)
    bool upd0[BR], upd1[BR]; kf_f2 sdf[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      sdf[b] = d[b] - pfz[b];
      upd0[b] = ok0[b] && d[b].x != 0.f && d[b].x < a.max_dist && sdf[b].x > -a.sdf_trunc;
      upd1[b] = ok1[b] && d[b].y != 0.f && d[b].y < a.max_dist && sdf[b].y > -a.sdf_trunc;
      if (COLOR) { upd0[b] = upd0[b] && okc0[b]; upd1[b] = upd1[b] && okc1[b]; }              // :59-62 `continue` when the colour pixel is outside
    }
    if (LAYERS) {                                                          // a sampled frame: every wave adds its quarter's updates to the brick layer's count
#pragma unroll
      for (int b = 0; b < BR; ++b) {
        const unsigned n = (unsigned)__popcll(__ballot(upd0[b])) + (unsigned)__popcll(__ballot(upd1[b]));
        if (n && (threadIdx.x & 63) == 0) atomicAdd(&s_layer[(ent[b] >> 20) + (unsigned)v.bz0], n);
      }
    }
    // Free space, decided per wave before the voxels are even requested: when no updating voxel of the wave lies inside the truncation
    // band (sdf >= trunc for all of them) every one observes tsdf = fminf(1, sdf / trunc) = 1 EXACTLY -- x >= t > 0 implies RN(x / t) >= 1
    // because rounding is monotone -- so the quotient is never formed (FREE; a non-positive or NaN truncation distance turns this off).
    // DEFER: a free-space wave over a saturated quarter, or one that updates ALL 128 voxels of a quarter in a deferred state, skips the memory
    // side altogether (skip); any other wave that updates something in a quarter with pending steps applies them to all 128 voxels (all_lanes).
    bool free_wave[BR], skip[BR], all_lanes[BR]; unsigned pnd[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) { fold[b] = (unsigned)__builtin_amdgcn_readfirstlane((int)fold_v[b]); pnd[b] = DEFER ? (unsigned)__builtin_amdgcn_readfirstlane((int)pend_v[b]) : 0u; }
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      const bool band = (upd0[b] && sdf[b].x < a.sdf_trunc) || (upd1[b] && sdf[b].y < a.sdf_trunc);
      const bool no_band = __ballot(band) == 0ull;
      free_wave[b] = a.free_ok && no_band;
      skip[b] = false; all_lanes[b] = false;
      if (DEFER) {
        const bool free_exact = no_band && a.sdf_trunc > 0.f;                                  // (what free_wave says when the shortcut is not switched off)
        const bool whole = __ballot(upd0[b] && upd1[b]) == ~0ull;                              // all 128 voxels of the quarter pass the predicate
        skip[b] = free_exact && (pnd[b] == KF_PEND_SAT || (pnd[b] != 0u && whole));
        all_lanes[b] = !skip[b] && pnd[b] >= 2u && __ballot(upd0[b] || upd1[b]) != 0ull;      // pending steps + a writer: the quarter is flushed
      }
    }
#ifdef KF_EXPERIMENTS
    // what the fusion pass's waves do (tools/exp_wave_kinds.py); packed counts per word
    if (KF_EXP_MODE(a) == 13) {
#pragma unroll
      for (int b = 0; b < BR; ++b) {
        const bool live = q0 + b < n_active;
        const bool any = __ballot(upd0[b] || upd1[b]) != 0ull, whole = __ballot(upd0[b] && upd1[b]) == ~0ull;
        if (live && (threadIdx.x & 63) == 0) {
          const unsigned sh = (blockIdx.x & 63) * 16;
          if (skip[b]) atomicAdd(&a.cnt->rc_steps[sh], pnd[b] == KF_PEND_SAT ? 1ull : (1ull << 32));           // skipped: saturated | deferred
          else if (!any) ;                                                                                       // nothing to update (= queued waves - the rest)
          else if (all_lanes[b]) atomicAdd(&a.cnt->rc_hits[sh], 1ull);                                           // flush + write
          else if (!free_wave[b]) atomicAdd(&a.cnt->rc_hits[sh], 1ull << 32);                                    // written: some voxel in the band
          else atomicAdd(&a.cnt->mc_blocks[sh], whole ? 1ull : (1ull << 32));                                    // written: whole free space | partial free space
        }
      }
    }
#endif
#ifdef KF_EXPERIMENTS
    // lane census (KF_INTEGRATE_EXP=17, tools/exp_lane_census.py): of a queued brick's 256 pair lanes, how many load their 16 bytes, how many of those update
    // ONE voxel of the pair and how many both -- what the fetched bytes are made of (FETCH 1.7 x the algorithmic read bytes at C2)
    if (KF_EXP_MODE(a) == 17) {
#pragma unroll
      for (int b = 0; b < BR; ++b) {
        const unsigned long long m_any = __ballot(upd0[b] || upd1[b]), m_both = __ballot(upd0[b] && upd1[b]);
        if (q0 + b < n_active && (threadIdx.x & 63) == 0) {
          const unsigned sh = (blockIdx.x & 63) * 16;
          atomicAdd(&a.cnt->rc_steps[sh], (unsigned long long)__popcll(m_any) | (1ull << 32));                        // low: lanes that load; high: waves
          atomicAdd(&a.cnt->rc_hits[sh], (unsigned long long)__popcll(m_both) | ((m_any ? 1ull : 0ull) << 32));     // low: lanes with both voxels; high: waves that load anything
          atomicAdd(&a.cnt->mc_blocks[sh], (m_any == ~0ull ? 1ull : 0ull) | ((m_any && m_any != ~0ull ? 1ull : 0ull) << 32));   // low: waves whose 64 lanes all load; high: partial waves
        }
      }
    }
#endif
    // one 16-byte read-modify-write per lane and brick, only where a voxel of the pair passed; the loads go out together
    float4* p[BR]; float4 q[BR]; bool rw[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      p[b] = reinterpret_cast<float4*>(v.tw + (size_t)slot[b] * KF_BRICK_VOX) + threadIdx.x;
      // no branch around the load: a lane that updates nothing reads one shared, never-written 16-byte word instead (an L1 hit; its
      // value is not used) -- with a conditional load the compiler parks a register copy, and with it a wait, behind EACH of the BR
      // loads, and they no longer travel together
      const bool touch = upd0[b] || upd1[b];
      rw[b] = (touch || all_lanes[b]) && !skip[b];
      const float4* src = (rw[b] && KF_EXP_MODE(a) != 2) ? p[b] : reinterpret_cast<const float4*>(a.queue_pad);
      q[b] = *src;
    }
    // COLOR: the pair's stored colours (8 bytes), the observed colours and -- for the angle weight -- the normals' z, all branch-free
    uint2* pc[BR]; uint2 qc[BR]; unsigned rgb0[BR], rgb1[BR]; float nz0[BR], nz1[BR];
    if (COLOR) {
#pragma unroll
      for (int b = 0; b < BR; ++b) {
        pc[b] = reinterpret_cast<uint2*>(v.color + (size_t)slot[b] * KF_BRICK_VOX) + threadIdx.x;
        qc[b] = *(rw[b] ? pc[b] : reinterpret_cast<const uint2*>(a.queue_pad));
        rgb0[b] = reinterpret_cast<const unsigned*>(a.rgb)[upd0[b] ? cpix0[b] : 0];
        rgb1[b] = reinterpret_cast<const unsigned*>(a.rgb)[upd1[b] ? cpix1[b] : 0];
        nz0[b] = a.normals[upd0[b] ? pix0[b] : 0].z;        // (read whether or not the angle weight is on: a conditional load would be waited for on the spot)
        nz1[b] = a.normals[upd1[b] ? pix1[b] : 0].z;
      }
    }
#ifndef KF_INT_NO_QPREFETCH
    // The next iteration's queue entries were requested at the top of this one and are older than the voxel loads above: they have arrived.  They are
    // "used" HERE, in front of this iteration's stores and flag atomics -- those are conditional, the compiler cannot count them, and its wait for the
    // entries at the head of the next iteration was a wait for everything outstanding: the stores' completion sat on every iteration's chain.
#pragma unroll
    for (int b = 0; b < BR; ++b) asm volatile("" : "+v"(ahead[b]));
#endif
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      if (DEFER && skip[b]) {                                               // uniform: counted; one more pending step unless the quarter is saturated
        upd_total += (upd0[b] ? 1u : 0u) + (upd1[b] ? 1u : 0u);
        if (pnd[b] != KF_PEND_SAT && (threadIdx.x & 63) == 0) pend16[slot[b] * 4u + quarter] = (unsigned short)kf_pend_step(pnd[b], v.max_weight);
        continue;
      }
      // tsdfVolume.h:63-66 on both voxels; a voxel that failed the predicate keeps its stored value
      const kf_f2 ot = {q[b].x, q[b].z};
      kf_f2 ow = {q[b].y, q[b].w};
      if (DEFER && all_lanes[b]) { const float k = (float)(pnd[b] - 1u); ow.x = fminf(ow.x + k, v.max_weight); ow.y = fminf(ow.y + k, v.max_weight); }   // the quarter's pending steps, applied
      const kf_f2 ow1 = ow + f2_splat(1.f);
      if (COUNT) { const int bzv = ((int)(ent[b] >> 20) + v.bz0) * KF_BRICK;  // voxels observed for the first time (owned layers): the running count of kf_get_volume_stats
        new_wave += kf_new_voxels(bzv >= v.own_z0 && bzv < v.own_z1 && v.max_weight > 0.f, upd0[b], ow.x, upd1[b], ow.y); }
      // A free-space wave whose updating voxels all hold tsdf 1 already (weight any value in [0, 2^24]): (1 * w + 1) / (w + 1) has the
      // SAME rounded sum RN(w + 1) above and below the line (1 * w is exact), a finite non-zero number divided by itself: nt = 1
      // exactly -- neither the reciprocal nor the second quotient is formed, only the weight moves.  Wave-uniform branch.
      bool unit = false;
      if (free_wave[b]) {
        const bool k0 = !upd0[b] || (__float_as_uint(ot.x) == one_f && __float_as_uint(ow.x) <= 0x4B800000u);
        const bool k1 = !upd1[b] || (__float_as_uint(ot.y) == one_f && __float_as_uint(ow.y) <= 0x4B800000u);
        unit = __ballot(!(k0 && k1)) == 0ull;
      }
      kf_f2 nt = f2_splat(1.0f);
      if (!unit) {
        kf_f2 tsdf = f2_splat(1.0f);
        if (!free_wave[b]) { tsdf = kf_div2(sdf[b], rtrunc); tsdf.x = fminf(1.0f, tsdf.x); tsdf.y = fminf(1.0f, tsdf.y); }
        nt = kf_div2(ot * ow + tsdf, kf_recip2(ow1));                       // `tsdf * 1.f` is the identity, bit for bit
      }
      const float nw0 = fminf(ow1.x, v.max_weight), nw1 = fminf(ow1.y, v.max_weight);
      unsigned flags = 0;
      float4 r = make_float4(ot.x, ow.x, ot.y, ow.y);                       // the stored pair (with the pending steps applied, if any)
      if (upd0[b]) { r.x = nt.x; r.y = nw0; }
      if (upd1[b]) { r.z = nt.y; r.w = nw1; }
      if (rw[b]) {
        if (KF_EXP_MODE(a) == 0 || KF_EXP_MODE(a) >= 8 || r.x == 123.456f) *p[b] = r;     // experiments 1 / 2: no store
        if (COLOR) {
          const kf_f2 nz = {nz0[b], nz1[b]};
          *pc[b] = integrate_color_update2(qc[b], rgb0[b], rgb1[b], ow, nz, upd0[b], upd1[b], a.color_angled, r075);
        }
      }
      if (upd0[b] || upd1[b]) {
        upd_total += (upd0[b] ? 1u : 0u) + (upd1[b] ? 1u : 0u);
        flags = KF_FLAG_OBSERVED | (((upd0[b] && nt.x < 0.f) || (upd1[b] && nt.y < 0.f)) ? KF_FLAG_HASNEG : 0u);
#ifdef KF_EXPERIMENTS
        if (KF_EXP_MODE(a) == 10) {                                                        // how many waves write back exactly what they read?
          const float4 r = make_float4(upd0[b] ? nt.x : q[b].x, upd0[b] ? nw0 : q[b].y, upd1[b] ? nt.y : q[b].z, upd1[b] ? nw1 : q[b].w);
          const bool same = __float_as_uint(r.x) == __float_as_uint(q[b].x) && __float_as_uint(r.y) == __float_as_uint(q[b].y) &&
                            __float_as_uint(r.z) == __float_as_uint(q[b].z) && __float_as_uint(r.w) == __float_as_uint(q[b].w);
          if (!same) flags |= 0x80u;
        }
#endif
      }
#ifdef KF_EXPERIMENTS
      if (KF_EXP_MODE(a) == 10) {
        const unsigned long long anyupd = __ballot(flags & KF_FLAG_OBSERVED), changed = __ballot(flags & 0x80u);
        if ((threadIdx.x & 63) == 0 && anyupd) {
          // low word: waves that touched memory; high word: those of them that changed nothing (read back as the marching-cubes counter)
          atomicAdd(&a.cnt->mc_blocks[(blockIdx.x & 63) * 16], 1ull + (changed ? 0ull : (1ull << 32)));
        }
      }
#endif
      // brick flags: see k_integrate_bricks
      const unsigned long long wrote = __ballot(flags & KF_FLAG_OBSERVED);
      const unsigned wflags = ((wrote ? KF_FLAG_OBSERVED : 0u) | (__ballot(flags & KF_FLAG_HASNEG) ? KF_FLAG_HASNEG : 0u)) & ~fold[b];
      if (wflags && (threadIdx.x & 63) == 0) {
        atomicOr(reinterpret_cast<unsigned*>(v.flags) + (slot[b] >> 2), wflags << (8u * (slot[b] & 3u)));
        if (wflags & KF_FLAG_HASNEG) {
          kf_mark_macro(v, (int)(ent[b] & 1023u), (int)((ent[b] >> 10) & 1023u), (int)(ent[b] >> 20) + v.bz0);
          atomicOr(&v.negbits[slot[b] >> 5], 1u << (slot[b] & 31u));
        }
      }
      if (DEFER && wrote && (pnd[b] != 0u || __ballot(rw[b]) == ~0ull)) {
        // The quarter's word follows what this wave left behind -- when that can be known: every lane loaded its pair, or the word vouches
        // for the lanes that did not (p >= 1: tsdf 1, weight >= 1; whether saturated is not known).  All 128 voxels (tsdf 1, weight >= 1) ->
        // deferred with nothing pending (saturated when every weight is max_weight); else plain.  (A wave of a plain quarter that loads
        // only some of its lanes -- most band and silhouette waves -- skips all of this.)
        bool lane_unit = pnd[b] != 0u, lane_sat = false;
        if (rw[b]) {
          lane_unit = __float_as_uint(r.x) == one_f && __float_as_uint(r.z) == one_f && r.y >= 1.f && r.w >= 1.f;
          lane_sat = lane_unit && __float_as_uint(r.y) == sat_w && __float_as_uint(r.w) == sat_w;
        }
        const bool all_unit = __ballot(lane_unit) == ~0ull, all_sat = __ballot(lane_sat) == ~0ull;
        const unsigned np = all_unit ? (all_sat ? KF_PEND_SAT : 1u) : 0u;
        if (np != pnd[b] && (threadIdx.x & 63) == 0) pend16[slot[b] * 4u + quarter] = (unsigned short)np;
      }
    }
  }
  float s = kf_wave_sum((float)upd_total);          // < 2^24 per wave: exact
  if ((threadIdx.x & 63) == 0 && s > 0.f) atomicAdd(&s_upd, (unsigned)s);
  if (COUNT && (threadIdx.x & 63) == 0 && new_wave) atomicAdd(&s_new, new_wave);
  __syncthreads();
  if (threadIdx.x == 0 && s_upd) atomicAdd(&a.cnt->upd_shard[a.parity][(blockIdx.x & 63) * 16], (unsigned long long)s_upd);
  if (COUNT && threadIdx.x == 0 && s_new) atomicAdd(&a.cnt->wgt0_shard[(blockIdx.x & 63) * 16], (unsigned long long)s_new);
  if (LAYERS) for (int i = threadIdx.x; i < v.nb; i += 256) { const unsigned n = s_layer[i]; if (n) atomicAdd(&a.layer_work[i], (unsigned long long)n); }
}

// ---- the one-brick form as a two-stage pipeline ------------------------------------------------------------------------------------------------------
// k_integrate_pairs<1, DEFER> walks a brick's chain  queue entry -> flags / deferred-weight word -> projection -> depth gather -> predicate -> 16-byte voxel
// load -> update -> store  once per iteration, and a wave's iteration lasts as long as that chain (~3 us; 32 waves per CU: 1024^3 with deferred weights =
// 1484 wave iterations per CU = 130 us whatever the bytes).  Here the FRONT of brick i + 1's chain (entry, flags / word, projection, depth gather) is issued
// behind brick i's voxel load and before the wait for it, so it travels in the shadow of that load: an iteration costs the longer of the two halves.  Loads
// return in order, and everything of brick i + 1 is younger than brick i's voxel load: the wait for that load leaves them outstanding.  The per-voxel
// arithmetic, the predicate, the deferred-weight rules and the flag updates are those of k_integrate_pairs, statement for statement: same bits.
// A brick's flag byte and deferred-weight word are read one iteration early: within a launch nobody but this wave writes them (a brick is queued once; the
// flag word's other bytes belong to other bricks and only ever gain bits).
struct PipeFront {                       // what stage A leaves behind for stage B
  unsigned ent, slot, fold_v, pend_v;
  kf_f2 pfz, d;
  bool ok0, ok1;
};
template <bool DEFER, bool COUNT = false>
__global__ void __launch_bounds__(256) k_integrate_pairs_pipe(IntegrateArgs a) {
  const KfVolume& v = a.vol;
  const unsigned n_active = a.cnt->n_active[a.parity];
  if (blockIdx.x == 0) integrate_maintenance(a);
  const float* m = a.tinv ? a.tinv : a.tinv_val.m;
  const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7], m8 = m[8], m9 = m[9], m10 = m[10], m11 = m[11];
  const float cell = v.cell;
  const int lx = (threadIdx.x & 3) * 2, ly = (threadIdx.x >> 2) & 7, lz = threadIdx.x >> 5;
  const KfRecip rt = kf_recip(a.sdf_trunc);
  KfRecip2 rtrunc; rtrunc.den = f2_splat(rt.den); rtrunc.r = f2_splat(rt.r);
  const unsigned xlim = (unsigned)(a.dcam.cols - 2), ylim = (unsigned)(a.dcam.rows - 2);
  const unsigned quarter = threadIdx.x >> 6;
  const unsigned sat_w = __float_as_uint(v.max_weight), one_f = __float_as_uint(1.0f);
  unsigned short* const pend16 = reinterpret_cast<unsigned short*>(v.pend);
  __shared__ unsigned s_upd, s_new;
  unsigned upd_total = 0, new_wave = 0;
  if (threadIdx.x == 0) { s_upd = 0; s_new = 0; }
  __syncthreads();
  // stage A of one brick: everything up to (and including) the request of its depth values
  auto front = [&](unsigned packed, bool live) {
    PipeFront f;
    f.ent = packed;
    const int bx = (int)(packed & 1023u), by = (int)((packed >> 10) & 1023u), bz = (int)(packed >> 20) + v.bz0;
    f.slot = ((unsigned)(bz - v.bz0) * (unsigned)v.nb + (unsigned)by) * (unsigned)v.nb + (unsigned)bx;
    f.fold_v = v.flags[kf_opaque(f.slot)];
    f.pend_v = DEFER ? (unsigned)pend16[kf_opaque(f.slot * 4u + quarter)] : 0u;
    const float x0 = (float)(bx * 8 + lx);
    kf_f2 xi = {x0, x0 + 1.0f};
    const kf_f2 wx = (xi + f2_splat(0.5f)) * f2_splat(cell);
    const float wy = ((float)(by * 8 + ly) + 0.5f) * cell, wz = ((float)(bz * 8 + lz) + 0.5f) * cell;
    const kf_f2 pfx = ((f2_splat(m0) * wx + f2_splat(m1 * wy)) + f2_splat(m2 * wz)) + f2_splat(m3 * 1.0f);
    const kf_f2 pfy = ((f2_splat(m4) * wx + f2_splat(m5 * wy)) + f2_splat(m6 * wz)) + f2_splat(m7 * 1.0f);
    f.pfz = ((f2_splat(m8) * wx + f2_splat(m9 * wy)) + f2_splat(m10 * wz)) + f2_splat(m11 * 1.0f);
    const bool z0 = live && f.pfz.x > 0.f, z1 = live && f.pfz.y > 0.f;
    kf_f2 zs = {z0 ? f.pfz.x : 1.0f, z1 ? f.pfz.y : 1.0f};
    const KfRecip2 rz = kf_recip2(zs);
    const kf_f2 px = kf_div2(pfx * f2_splat(a.dcam.fx), rz) + f2_splat(a.dcam.cx) + f2_splat(0.5f);
    const kf_f2 py = kf_div2(pfy * f2_splat(a.dcam.fy), rz) + f2_splat(a.dcam.cy) + f2_splat(0.5f);
    const int sx0 = (int)floorf(px.x), sx1 = (int)floorf(px.y), sy0 = (int)floorf(py.x), sy1 = (int)floorf(py.y);
    f.ok0 = z0 && (unsigned)(sx0 - 1) < xlim && (unsigned)(sy0 - 1) < ylim;
    f.ok1 = z1 && (unsigned)(sx1 - 1) < xlim && (unsigned)(sy1 - 1) < ylim;
    const int pix0 = f.ok0 ? sy0 * a.dcam.cols + sx0 : 0, pix1 = f.ok1 ? sy1 * a.dcam.cols + sx1 : 0;
    // no branch around the gathers (a voxel outside the window reads pixel 0; its value is never looked at: the predicate starts with ok0 / ok1) -- behind a
    // conditional load the compiler parks a wait, and the next brick's front would be waited for before the current brick is updated
    f.d.x = a.depth[pix0]; f.d.y = a.depth[pix1];
    return f;
  };
  const unsigned stride = gridDim.x;
  unsigned q0 = blockIdx.x;
  // start-up: the first two entries are requested without waiting for the queue length (every index below the queue's capacity is readable)
  unsigned e_cur, ahead;
  { const unsigned i0 = kf_opaque(q0), i1 = kf_opaque(q0 + stride);
    e_cur = a.queue[i0 < a.queue_cap ? i0 : 0u]; ahead = a.queue[i1 < a.queue_cap ? i1 : 0u]; }
  if (q0 >= n_active) return;                                             // (uniform; nothing was counted)
  PipeFront cur = front((unsigned)__builtin_amdgcn_readfirstlane((int)e_cur), true);
  // (the first brick's front is waited for here, not at the loop's head: a wait at the head is a wait in EVERY iteration, behind the previous one's stores)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur.d.x), "+v"(cur.d.y), "+v"(cur.fold_v), "+v"(cur.pend_v), "+v"(ahead) :: "memory");
  for (; q0 < n_active; q0 += stride) {
    // ---- stage B, first half: the predicate of brick `cur`, its voxel load goes out
    const unsigned slot = cur.slot, ent = cur.ent;
    const kf_f2 sdf = cur.d - cur.pfz;
    const bool upd0 = cur.ok0 && cur.d.x != 0.f && cur.d.x < a.max_dist && sdf.x > -a.sdf_trunc;
    const bool upd1 = cur.ok1 && cur.d.y != 0.f && cur.d.y < a.max_dist && sdf.y > -a.sdf_trunc;
    const unsigned fold = (unsigned)__builtin_amdgcn_readfirstlane((int)cur.fold_v);
    const unsigned pnd = DEFER ? (unsigned)__builtin_amdgcn_readfirstlane((int)cur.pend_v) : 0u;
    const bool band = (upd0 && sdf.x < a.sdf_trunc) || (upd1 && sdf.y < a.sdf_trunc);
    const bool no_band = __ballot(band) == 0ull;
    const bool free_wave = a.free_ok && no_band;
    bool skip = false, all_lanes = false;
    if (DEFER) {
      const bool free_exact = no_band && a.sdf_trunc > 0.f;
      const bool whole = __ballot(upd0 && upd1) == ~0ull;
      skip = free_exact && (pnd == KF_PEND_SAT || (pnd != 0u && whole));
      all_lanes = !skip && pnd >= 2u && __ballot(upd0 || upd1) != 0ull;
    }
    float4* const p = reinterpret_cast<float4*>(v.tw + (size_t)slot * KF_BRICK_VOX) + threadIdx.x;
    const bool touch = upd0 || upd1;
    const bool rw = (touch || all_lanes) && !skip;
    const float4 q = *(rw ? p : reinterpret_cast<const float4*>(a.queue_pad));
    // ---- stage A of the NEXT brick, in the shadow of that load
    const unsigned qn = q0 + stride;
    const bool next_live = qn < n_active;
    const unsigned e_next = next_live ? (unsigned)__builtin_amdgcn_readfirstlane((int)ahead) : 0u;
    { const unsigned i2 = kf_opaque(qn + stride); ahead = a.queue[i2 < n_active ? i2 : 0u]; }
    PipeFront nxt = cur;
    if (next_live) nxt = front(e_next, true);                              // (uniform; the wait below is explicit, so a branch around these loads costs nothing)
    // ONE wait per iteration, here: brick `cur`'s voxels (requested first, the slowest) and brick `nxt`'s front (requested behind them) have all arrived
    // when the update below starts, and nothing is outstanding but loads -- the stores and flag atomics below are conditional, and the in-order counter
    // cannot wait for a load that was issued behind an unknown number of them without waiting for all (which put the store's completion on the chain)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt.d.x), "+v"(nxt.d.y), "+v"(nxt.fold_v), "+v"(nxt.pend_v), "+v"(ahead) :: "memory");
    // ---- stage B, second half: brick `cur` is updated and stored
    if (DEFER && skip) {
      upd_total += (upd0 ? 1u : 0u) + (upd1 ? 1u : 0u);
      if (pnd != KF_PEND_SAT && (threadIdx.x & 63) == 0) pend16[slot * 4u + quarter] = (unsigned short)kf_pend_step(pnd, v.max_weight);
    } else {
      const kf_f2 ot = {q.x, q.z};
      kf_f2 ow = {q.y, q.w};
      if (DEFER && all_lanes) { const float k = (float)(pnd - 1u); ow.x = fminf(ow.x + k, v.max_weight); ow.y = fminf(ow.y + k, v.max_weight); }
      const kf_f2 ow1 = ow + f2_splat(1.f);
      if (COUNT) { const int bzv = ((int)(ent >> 20) + v.bz0) * KF_BRICK;     // voxels observed for the first time (owned layers): kf_get_volume_stats' running count
        new_wave += kf_new_voxels(bzv >= v.own_z0 && bzv < v.own_z1 && v.max_weight > 0.f, upd0, ow.x, upd1, ow.y); }
      bool unit = false;
      if (free_wave) {
        const bool k0 = !upd0 || (__float_as_uint(ot.x) == one_f && __float_as_uint(ow.x) <= 0x4B800000u);
        const bool k1 = !upd1 || (__float_as_uint(ot.y) == one_f && __float_as_uint(ow.y) <= 0x4B800000u);
        unit = __ballot(!(k0 && k1)) == 0ull;
      }
      kf_f2 nt = f2_splat(1.0f);
      if (!unit) {
        kf_f2 tsdf = f2_splat(1.0f);
        if (!free_wave) { tsdf = kf_div2(sdf, rtrunc); tsdf.x = fminf(1.0f, tsdf.x); tsdf.y = fminf(1.0f, tsdf.y); }
        nt = kf_div2(ot * ow + tsdf, kf_recip2(ow1));
      }
      const float nw0 = fminf(ow1.x, v.max_weight), nw1 = fminf(ow1.y, v.max_weight);
      unsigned flags = 0;
      float4 r = make_float4(ot.x, ow.x, ot.y, ow.y);
      if (upd0) { r.x = nt.x; r.y = nw0; }
      if (upd1) { r.z = nt.y; r.w = nw1; }
      if (rw) *p = r;
      if (touch) {
        upd_total += (upd0 ? 1u : 0u) + (upd1 ? 1u : 0u);
        flags = KF_FLAG_OBSERVED | (((upd0 && nt.x < 0.f) || (upd1 && nt.y < 0.f)) ? KF_FLAG_HASNEG : 0u);
      }
      const unsigned long long wrote = __ballot(flags & KF_FLAG_OBSERVED);
      const unsigned wflags = ((wrote ? KF_FLAG_OBSERVED : 0u) | (__ballot(flags & KF_FLAG_HASNEG) ? KF_FLAG_HASNEG : 0u)) & ~fold;
      if (wflags && (threadIdx.x & 63) == 0) {
        atomicOr(reinterpret_cast<unsigned*>(v.flags) + (slot >> 2), wflags << (8u * (slot & 3u)));
        if (wflags & KF_FLAG_HASNEG) {
          kf_mark_macro(v, (int)(ent & 1023u), (int)((ent >> 10) & 1023u), (int)(ent >> 20) + v.bz0);
          atomicOr(&v.negbits[slot >> 5], 1u << (slot & 31u));
        }
      }
      if (DEFER && wrote && (pnd != 0u || __ballot(rw) == ~0ull)) {
        bool lane_unit = pnd != 0u, lane_sat = false;
        if (rw) {
          lane_unit = __float_as_uint(r.x) == one_f && __float_as_uint(r.z) == one_f && r.y >= 1.f && r.w >= 1.f;
          lane_sat = lane_unit && __float_as_uint(r.y) == sat_w && __float_as_uint(r.w) == sat_w;
        }
        const bool all_unit = __ballot(lane_unit) == ~0ull, all_sat = __ballot(lane_sat) == ~0ull;
        const unsigned np = all_unit ? (all_sat ? KF_PEND_SAT : 1u) : 0u;
        if (np != pnd && (threadIdx.x & 63) == 0) pend16[slot * 4u + quarter] = (unsigned short)np;
      }
    }
    cur = nxt;
  }
  float s = kf_wave_sum((float)upd_total);
  if ((threadIdx.x & 63) == 0 && s > 0.f) atomicAdd(&s_upd, (unsigned)s);
  if (COUNT && (threadIdx.x & 63) == 0 && new_wave) atomicAdd(&s_new, new_wave);
  __syncthreads();
  if (threadIdx.x == 0 && s_upd) atomicAdd(&a.cnt->upd_shard[a.parity][(blockIdx.x & 63) * 16], (unsigned long long)s_upd);
  if (COUNT && threadIdx.x == 0 && s_new) atomicAdd(&a.cnt->wgt0_shard[(blockIdx.x & 63) * 16], (unsigned long long)s_new);
}

#ifdef KF_EXPERIMENTS
// experiments 4-7: the memory side of the fusion pass alone -- every queued brick is read and / or written back (16 bytes per
// lane, all lanes), no arithmetic: what the brick-queue access pattern can reach on this chip (tools/bench_integrate.py).
// MODE 0: load + store; 1: non-temporal load + store; 2: read only (sum kept alive); 3: write only; 4: load + store of as many bricks
// as the queue holds but CONTIGUOUS in memory (slots 0 .. n_active-1): what the scattering of the queue itself costs
template <int BR, int MODE>
__global__ void __launch_bounds__(256) k_exp_brick_rmw(IntegrateArgs a) {
  const KfVolume& v = a.vol;
  const unsigned n_active = a.cnt->n_active[a.parity];
  if (blockIdx.x == 0) integrate_maintenance(a);
  float acc = 0.f;
  for (unsigned q0 = blockIdx.x * BR; q0 < n_active; q0 += gridDim.x * BR) {
    float4* p[BR]; float4 q[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      const unsigned packed = (q0 + b < n_active) ? a.queue[q0 + b] : a.queue[q0];
      const unsigned slot = MODE == 4 ? ((q0 + b < n_active) ? q0 + b : q0) : ((packed >> 20) * (unsigned)v.nb + ((packed >> 10) & 1023u)) * (unsigned)v.nb + (packed & 1023u);
      p[b] = reinterpret_cast<float4*>(v.tw + (size_t)slot * KF_BRICK_VOX) + threadIdx.x;
      if (MODE == 1) { typedef float v4 __attribute__((ext_vector_type(4))); const v4 t = __builtin_nontemporal_load(reinterpret_cast<v4*>(p[b])); q[b] = make_float4(t.x, t.y, t.z, t.w); }
      else if (MODE == 3) q[b] = make_float4(0.f, 0.f, 0.f, 0.f);
      else q[b] = *p[b];
    }
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      q[b].y += 0.f;
      if (MODE == 1) { typedef float v4 __attribute__((ext_vector_type(4))); v4 t = {q[b].x, q[b].y, q[b].z, q[b].w}; __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p[b])); }
      else if (MODE == 2) acc += q[b].x;
      else *p[b] = q[b];
    }
  }
  if (MODE == 2 && acc == 123.456f) a.tile_max[0] = acc;
}
#endif

static inline KfCam to_cam(const kf_camera_params* p) {
  KfCam c; c.cols = (int)p->cols; c.rows = (int)p->rows; c.cx = p->cx; c.cy = p->cy; c.fx = p->fx; c.fy = p->fy; return c;
}

// Does this context defer whole-quarter free-space weight updates?  kf_set_defer(ctx, 0 | 1) decides when called; otherwise KF_INTEGRATE_SAT
// (0: never, 2: always) and, by default, the volume's resolution: 768^3 and finer.  Measured on Scene S (profiles/r04_deferred_weights.txt, same
// box): 1024^3 @ 6 m 1.83 k -> 2.67 k frames/s, 2048^3 @ 8 m 356 -> 663, but 512^3 @ 4 m with the stock 2 m integration gate 4.98 k -> 4.89 k --
// there the fusion pass is start-up + instruction issue, not memory, two thirds of its waves are silhouette / band waves that
// cannot defer, and the deferred form's bookkeeping (+2 us kernel, +1.4 us cull) is all that shows.  Results are bit-identical either way.
bool kf_defer_enabled(const kf_ctx* c) {
  static int sat_env = -1, pairs_env = -1;
  if (sat_env < 0) { const char* e = getenv("KF_INTEGRATE_SAT"); sat_env = e ? atoi(e) : 1; }
  if (pairs_env < 0) { const char* e = getenv("KF_INTEGRATE_PAIRS"); pairs_env = e ? atoi(e) : 1; }
  if (!pairs_env || !c->vol.pend || !(c->vol.max_weight >= 1.f && c->vol.max_weight <= KF_PEND_MAX_WEIGHT)) return false;
  if (c->defer_override >= 0) return c->defer_override != 0;
  return sat_env == 2 || (sat_env == 1 && c->vol.res >= 768);
}

// every pending count applied to its 128 voxels (one wave per brick); the words drop to "nothing pending" and stay valid
__global__ void __launch_bounds__(256) k_flush_pending(KfVolume v, unsigned n_bricks) {
  const unsigned lane = threadIdx.x & 63u, n_waves = gridDim.x * 4u;
  for (unsigned s = blockIdx.x * 4u + (threadIdx.x >> 6); s < n_bricks; s += n_waves) {
    const unsigned long long pp = v.pend[s];
    unsigned long long np = pp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned pq = (unsigned)((pp >> (16 * q)) & 0xFFFFull);
      if (pq < 2u) continue;                                  // uniform
      float4* ptr = reinterpret_cast<float4*>(v.tw + (size_t)s * KF_BRICK_VOX + q * 128) + lane;
      float4 r = *ptr;
      r.y = kf_pend_weight(r.y, pq, v.max_weight); r.w = kf_pend_weight(r.w, pq, v.max_weight);
      *ptr = r;
      if (pq != KF_PEND_SAT) np = (np & ~(0xFFFFull << (16 * q))) | (1ull << (16 * q));
    }
    if (lane == 0u && np != pp) v.pend[s] = np;
  }
}
int kf_flush_pending(kf_ctx* c) {
  if (!c->pend_live || !c->vol.pend) return 0;
  const unsigned n = (unsigned)c->n_stored_bricks;
  hipLaunchKernelGGL(k_flush_pending, dim3(n / 4u + 1u > 8192u ? 8192u : n / 4u + 1u), dim3(256), 0, c->stream, c->vol, n);
  return (int)hipGetLastError();
}
extern "C" int kf_set_defer(kf_ctx* c, int mode) {
  if (!c || mode < -1 || mode > 1) return KF_ERR_ARG;
  c->defer_override = mode;
  return 0;
}

// What the cull reads of IntegrateArgs, for the context's current depth map and the device-resident pose (shared with the tracking launch's tail, track.hip)
void kf_fill_cull_args(kf_ctx* c, IntegrateArgs& a, const kf_camera_params* dcam, float sdf_trunc, float max_dist) {
  memset(&a, 0, sizeof(a));
  a.vol = c->vol; a.dcam = to_cam(dcam); a.rcam = a.dcam;
  a.depth = c->trunced_depth;
  a.tile_max = c->tile_max_depth; a.queue = c->active_bricks; a.cnt = c->counters;
  a.queue_cap = (unsigned)c->n_stored_bricks;
  a.queue_pad = c->active_bricks + ((c->n_stored_bricks + 3) & ~(size_t)3);     // 16-byte aligned spare words behind the queue (allocated in ctx.hip)
  a.sdf_trunc = sdf_trunc; a.max_dist = max_dist;
  for (int l = 0, off = 0; l < 2; ++l) {
    a.tile_w[l] = kf_div_up(c->cols, 8 << l); a.tile_h[l] = kf_div_up(c->rows, 8 << l);
    a.tile_off[l] = off; off += a.tile_w[l] * a.tile_h[l];
  }
  // The 8-pixel table is four times larger and colder in the cull's caches (+2 us per launch at VGA); it pays when bricks are
  // small on screen -- 1024^3 @ 6 m: 4 px at the far end, queue -7 %, fusion -10 us -- not at 512^3 @ 4 m (16 px, -3 %, -0.4 us).
  a.fine_tiles = (8.f * c->vol.cell * a.dcam.fx / (a.max_dist > 0.f ? a.max_dist : 1.f)) < 12.f ? 1 : 0;
  { static int fe = -2; if (fe == -2) { const char* e = getenv("KF_CULL_FINE"); fe = e ? atoi(e) : -1; } if (fe >= 0) a.fine_tiles = fe; }      // A/B
  a.fr_slope[0] = (-1.f - a.dcam.cx) / a.dcam.fx; a.fr_slope[1] = ((float)a.dcam.cols - a.dcam.cx) / a.dcam.fx;
  a.fr_slope[2] = (-1.f - a.dcam.cy) / a.dcam.fy; a.fr_slope[3] = ((float)a.dcam.rows - a.dcam.cy) / a.dcam.fy;
  for (int i = 0; i < 4; ++i) a.fr_norm[i] = sqrtf(1.f + a.fr_slope[i] * a.fr_slope[i]);
  a.tinv = c->track->pose_inv; a.track = c->track;       // kept current by whoever commits the device-resident pose
  a.parity = c->int_parity;
  a.n_tile_floats = c->n_tile_floats;
  { static int md = -1; if (md < 0) { const char* e = getenv("KF_CULL_MACRO_DEPTH"); md = e ? atoi(e) : 1; } a.macro_depth = md; }
}
bool kf_cull_tail_fits(const kf_ctx* c, int n_wg, int waves) {
  const int nmxy = (c->vol.nb + 3) >> 2, nmz = ((c->vol.bz1 + 3) >> 2) - (c->vol.bz0 >> 2);
  return (long long)nmxy * nmxy * nmz <= (long long)n_wg * waves * CULL_TAIL_ROUNDS;
}
extern "C" int kf_cull_tail_counts(kf_ctx* c, uint32_t* consumed, uint32_t* undone) {
  if (!c) return KF_ERR_ARG;
  if (consumed) *consumed = c->tail_cull.consumed;
  if (undone) *undone = c->tail_cull.undone;
  return 0;
}
int kf_tail_cull_discard(kf_ctx* c) {
  if (!c->tail_cull.armed) return 0;
  c->tail_cull.armed = 0; c->tail_cull.undone++;
  KF_CHECK(hipMemsetAsync(&c->counters->n_active[c->tail_cull.parity], 0, sizeof(unsigned), c->stream));
  return 0;
}

extern "C" int kf_integrate_volume(kf_ctx* c, int has_color, int use_angle_weight_color, const kf_mat44* transform,
                                   const kf_integrate_params* ip, const kf_camera_params* dcam, const kf_camera_params* rcam) {
  if (!c || !ip || !dcam) return KF_ERR_ARG;
  if ((int)dcam->cols != c->cols || (int)dcam->rows != c->rows) return KF_ERR_ARG;
  if (has_color && (!c->vol.color || !c->raw_rgb || !rcam)) return KF_ERR_STATE;
  IntegrateArgs a;
  kf_fill_cull_args(c, a, dcam, ip->sdf_truncation, ip->max_integrate_dist);
  a.rcam = rcam ? to_cam(rcam) : a.dcam;
  a.normals = c->new_n[0]; a.rgb = c->raw_rgb;
  a.has_color = has_color; a.color_angled = use_angle_weight_color;
  { static int fs = -1; if (fs < 0) { const char* e = getenv("KF_INTEGRATE_FREESPACE"); fs = e ? atoi(e) : 1; }      // 0: always form the quotients (A/B)
    a.free_ok = (fs && a.sdf_trunc > 0.f) ? 1 : 0; }
  { static int em = -1; if (em < 0) em = KF_EXP_ENV("KF_INTEGRATE_EXP"); a.exp_mode = em; }
  if (transform) {
    kf_mat44_inverse(transform->m, a.tinv_val.m);        // integrateVolume.cu:84, same arithmetic as on the device
    a.tinv = nullptr; a.track = nullptr;
  } else {
    a.tinv = c->track->pose_inv; a.track = c->track;     // kept current by whoever commits the device-resident pose
  }
  a.layer_work = nullptr;
  if (c->layer_work_frames > 0 && c->layer_work) { a.layer_work = c->layer_work; --c->layer_work_frames; }
  a.parity = c->int_parity; c->last_parity = c->int_parity; c->int_parity ^= 1;
  a.clear_tiles = 1; a.n_tile_floats = c->n_tile_floats;
  // deferred free-space weights (k_integrate_pairs<.., DEFER>): the packed-pair kernel without colour.  Any other fusion kernel knows nothing of
  // the deferred-weight words: pending counts are applied before it runs and the words cleared behind it.
  const bool defer = !has_color && kf_defer_enabled(c);
  const bool legacy_over_words = !defer && c->pend_live;
  if (legacy_over_words) { const int fs = kf_flush_pending(c); if (fs) return fs; }
  a.defer_cull = 0;                                      // decided below, once it is known whether the tile minima describe this depth map
  kf_evt_begin(c, KF_STAGE_INTEGRATE);
  // tile maxima: normally left behind by the fused preprocess kernel for exactly this depth map and distance
  const bool tiles_ready = c->tile_serial != 0 && c->tile_serial == c->trunc_serial && c->tile_built_dist == a.max_dist;
  if (!tiles_ready) { hipLaunchKernelGGL(k_integrate_prepare, dim3(a.tile_w[1] * a.tile_h[1]), dim3(256), 0, c->stream, a); c->tile_min_serial = c->trunc_serial; }
  a.defer_cull = (defer && c->tile_min_serial == c->trunc_serial) ? 1 : 0;     // whole-brick retirement needs the minima of THIS depth map
  c->fuse_max_dist = a.max_dist;                         // what the next preprocess builds the tables for
  c->tile_serial = 0; c->tiles_clear = 1;                // the fusion pass below clears the tables behind the cull
  c->fp_tiles = 0;                                       // (tables a raycast launch may have built for a prefetched frame are cleared with them)
  // The cull may have run already, as the tail of this frame's tracking launch (kf_icp_track, persistent loop): consumed when it saw what this call
  // would have shown it -- the device-resident pose, the same parameters, depth map, tile tables, slab and counter set; otherwise undone.
  bool culled = false;
  if (c->tail_cull.armed) {
    const auto& t = c->tail_cull;
    culled = !transform && !defer && tiles_ready && t.parity == a.parity && t.sdf_trunc == a.sdf_trunc && t.max_dist == a.max_dist &&
             memcmp(&t.dcam, dcam, sizeof(*dcam)) == 0 && t.trunc_serial == c->trunc_serial && t.bz0 == c->vol.bz0 && t.bz1 == c->vol.bz1;
    if (culled) { c->tail_cull.armed = 0; c->tail_cull.consumed++; }
    else { const int ds = kf_tail_cull_discard(c); if (ds) return ds; }
  }
  c->cull_hint.valid = transform ? 0 : 1;                // what the next tracking launch may cull for
  c->cull_hint.sdf_trunc = a.sdf_trunc; c->cull_hint.max_dist = a.max_dist; c->cull_hint.dcam = *dcam;
  if (!culled) {
    const int nmxy = (c->vol.nb + 3) >> 2, nmz = ((c->vol.bz1 + 3) >> 2) - (c->vol.bz0 >> 2);
    const unsigned n_macro = (unsigned)nmxy * nmxy * nmz;                      // one wave per macro cell, sixteen per workgroup
    const unsigned cgrid = (n_macro + CULL_WAVES - 1) / CULL_WAVES;
    // large volumes: the macro cells are sifted one per lane first (k_integrate_cull_sift); KF_CULL_SIFT=0 / 1 forces either form
    static int sift_env = -2;
    if (sift_env == -2) { const char* e = getenv("KF_CULL_SIFT"); sift_env = e ? atoi(e) : -1; }
    const bool sift = sift_env >= 0 ? sift_env != 0 : n_macro >= 100000u;
    const int n_wg = (int)((n_macro + SIFT_CELLS - 1) / SIFT_CELLS);
    if (sift && a.defer_cull) hipLaunchKernelGGL(k_integrate_cull_sift<true>, dim3(n_wg), dim3(SIFT_THREADS), 0, c->stream, a, n_wg);
    else if (sift) hipLaunchKernelGGL(k_integrate_cull_sift<false>, dim3(n_wg), dim3(SIFT_THREADS), 0, c->stream, a, n_wg);
    else if (a.defer_cull) hipLaunchKernelGGL(k_integrate_cull<true>, dim3(cgrid), dim3(CULL_WAVES * 64), 0, c->stream, a);
    else hipLaunchKernelGGL(k_integrate_cull<false>, dim3(cgrid), dim3(CULL_WAVES * 64), 0, c->stream, a);
  }
  // Workgroups walking the queue.  Large volumes (>= 2^20 stored bricks: the queue holds >~100k bricks): four bricks in flight per workgroup and
  // EIGHT TIMES as many workgroups as the chip holds at once (6 per CU x 256 = 1536 -> 12288: whole rounds; 8192 = 5.33 rounds ended on a third
  // of the chip: 305.6 -> 297 us at 1024^3, every multiple of 1536 from 7680 to 15360 within 1 % of that).  Smaller ones: ONE brick in flight and exactly as many workgroups as the chip holds at once (8 per CU) -- at 512^3 the queue
  // is ~9 k bricks, i.e. 2.2 rounds of 4.5 k two-brick workgroups of which the last is a fifth full; 2048 resident workgroups that each walk
  // 4-5 bricks with the look-ahead end together: 20.0 -> 18.5 us at 512^3, 11.8 -> 9.6 us at 256^3, one box (at 1024^3 the same form loses: 324 vs
  // 304 us).  KF_INTEGRATE_GRID / KF_INTEGRATE_BR override.
  static unsigned grid_env = 0, resident = 0, resident4 = 0;
  if (!grid_env) { const char* e = getenv("KF_INTEGRATE_GRID"); grid_env = e ? (unsigned)atoi(e) : 1u; if (grid_env != 1u && (grid_env < 64u || grid_env > 65536u)) grid_env = 1u; }
  if (!resident) {
    int per_cu = 0; hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_integrate_pairs<1, false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 8;
    const unsigned cus = (hipGetDeviceProperties(&prop, c->cfg.device) == hipSuccess && prop.multiProcessorCount > 0) ? (unsigned)prop.multiProcessorCount : 256u;
    resident = cus * (unsigned)per_cu;
    int per_cu4 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, k_integrate_pairs<4, false>, 256, 0) != hipSuccess || per_cu4 < 1) per_cu4 = 6;
    resident4 = cus * (unsigned)per_cu4;
  }
  const bool big = c->n_stored_bricks >= ((size_t)1 << 20);
  // DEFER: one brick in flight whatever the size -- most waves end without touching a voxel, so there is little memory latency to cover and
  // the four-brick form's extra registers (85 VGPRs + scalar spills) only cost: 1024^3 181 -> 130 us, 2048^3 795 -> 577 us (BR 4 x 12288 vs
  // BR 1 x 8192 workgroups; BR 1 x 2048 157, x 4096 135, x 16384 132 us at 1024^3: profiles/r04_deferred_weights.txt)
  const unsigned grid_cap = grid_env != 1u ? grid_env : (defer ? (big ? 4u * resident : resident) : (big && !has_color ? 8u * resident4 : (has_color ? 8192u : resident)));
  unsigned grid = (unsigned)(c->n_stored_bricks < grid_cap ? c->n_stored_bricks : grid_cap);
  // the roofline kernel's live timer: the event pair rides on the dispatch itself (kf_evt_attach), so what is measured is the kernel, as rocprofv3 sees it
  hipEvent_t ke0 = nullptr, ke1 = nullptr;
  const bool timed = kf_evt_attach(c, KF_STAGE_INTEGRATE_KERNEL, &ke0, &ke1);
  const bool count = c->wgt0_tracking && c->wgt0_valid;   // the COUNT instantiations, while a host keeps asking kf_get_volume_stats for the observed-voxel count
  bool counted = false;
#define FUSE_LAUNCH(K) do { if (timed) hipExtLaunchKernelGGL(K, dim3(grid), dim3(256), 0, c->stream, ke0, ke1, 0, a); \
                            else hipLaunchKernelGGL(K, dim3(grid), dim3(256), 0, c->stream, a); } while (0)
  static int color_pairs = -1;                            // 1 (default): colour through the packed-pair kernel; 0: the scalar kernel (A/B)
  if (color_pairs < 0) { const char* e = getenv("KF_INTEGRATE_COLOR_PAIRS"); color_pairs = e ? atoi(e) : 1; }
  if (has_color && color_pairs) {
    static int cbr = -1;                                   // bricks in flight per workgroup of the colour variant (KF_INTEGRATE_BR overrides)
    if (cbr < 0) { const char* e = getenv("KF_INTEGRATE_BR"); cbr = e ? atoi(e) : 2; if (cbr != 1 && cbr != 2 && cbr != 4) cbr = 2; }
    if (cbr == 1) FUSE_LAUNCH((k_integrate_pairs<1, false, true>));
    else if (cbr == 2) FUSE_LAUNCH((k_integrate_pairs<2, false, true>));
    else FUSE_LAUNCH((k_integrate_pairs<4, false, true>));
  }
  else if (has_color) FUSE_LAUNCH((k_integrate_bricks<true, 1>));
  else {
    // bricks in flight per workgroup (see the grid above): 1, or 4 for large volumes.  KF_INTEGRATE_BR overrides.
    static int br_env = -1;
    if (br_env < 0) { const char* e = getenv("KF_INTEGRATE_BR"); br_env = e ? atoi(e) : 0; if (br_env != 1 && br_env != 2 && br_env != 4) br_env = 0; }
    const int br = br_env ? br_env : ((big && !defer) ? 4 : 1);
    static int pipe_env = -1;
    if (pipe_env == -1) { const char* e = getenv("KF_INTEGRATE_PIPE"); pipe_env = e ? atoi(e) : -2; }     // -2: decided per call below
    static int pairs = -1;                               // 1 (default): the packed-pair kernel; 0: the scalar one (A/B and colour path)
    if (pairs < 0) { const char* e = getenv("KF_INTEGRATE_PAIRS"); pairs = e ? atoi(e) : 1; }
#ifdef KF_EXPERIMENTS
    if (a.exp_mode == 4) FUSE_LAUNCH((k_exp_brick_rmw<4, 0>));
    else if (a.exp_mode == 5) FUSE_LAUNCH((k_exp_brick_rmw<4, 1>));
    else if (a.exp_mode == 6) FUSE_LAUNCH((k_exp_brick_rmw<4, 2>));
    else if (a.exp_mode == 7) FUSE_LAUNCH((k_exp_brick_rmw<4, 3>));
    else if (a.exp_mode == 12) FUSE_LAUNCH((k_exp_brick_rmw<4, 4>));
    else
#endif
    if (pairs && a.layer_work) {                           // a sampled frame (kf_count_layer_work): the one-brick form that also counts per brick layer
      if (defer) FUSE_LAUNCH((k_integrate_pairs<1, true, false, true>));
      else FUSE_LAUNCH((k_integrate_pairs<1, false, false, true>));
    } else if (pairs && br == 1 && (pipe_env < 0 ? defer : pipe_env != 0)) {     // the one-brick form as a two-stage pipeline: by default where workgroups walk many bricks (KF_INTEGRATE_PIPE=0 / 1 forces)
      if (defer && count) { FUSE_LAUNCH((k_integrate_pairs_pipe<true, true>)); counted = true; }
      else if (defer) FUSE_LAUNCH((k_integrate_pairs_pipe<true>));
      else if (count) { FUSE_LAUNCH((k_integrate_pairs_pipe<false, true>)); counted = true; }
      else FUSE_LAUNCH((k_integrate_pairs_pipe<false>));
    } else if (pairs) {
      if (defer) {
        if (br == 1 && count) { FUSE_LAUNCH((k_integrate_pairs<1, true, false, false, true>)); counted = true; }
        else if (br == 1) FUSE_LAUNCH((k_integrate_pairs<1, true>));
        else if (br == 2) FUSE_LAUNCH((k_integrate_pairs<2, true>));
        else FUSE_LAUNCH((k_integrate_pairs<4, true>));
      } else if (br == 1 && count) { FUSE_LAUNCH((k_integrate_pairs<1, false, false, false, true>)); counted = true; }
      else if (br == 1) FUSE_LAUNCH((k_integrate_pairs<1, false>));
      else if (br == 2) FUSE_LAUNCH((k_integrate_pairs<2, false>));
      else if (count) { FUSE_LAUNCH((k_integrate_pairs<4, false, false, false, true>)); counted = true; }
      else FUSE_LAUNCH((k_integrate_pairs<4, false>));
    } else if (br == 1) FUSE_LAUNCH((k_integrate_bricks<false, 1>));
    else if (br == 2) FUSE_LAUNCH((k_integrate_bricks<false, 2>));
    else FUSE_LAUNCH((k_integrate_bricks<false, 4>));
  }
#undef FUSE_LAUNCH
  // the running count of observed voxels (kf_get_volume_stats): a fusion launch that did not count leaves it behind the volume
  if (!counted) c->wgt0_valid = 0;
  if (c->wgt0_frames_unasked < (1 << 30)) ++c->wgt0_frames_unasked;
  if (c->wgt0_tracking && c->wgt0_frames_unasked > 64) c->wgt0_tracking = 0;        // nobody has asked for 64 frames: the plain kernels again
  if (timed) kf_evt_attached_done(c, KF_STAGE_INTEGRATE_KERNEL);
  kf_evt_end(c, KF_STAGE_INTEGRATE);
  if (defer) c->pend_live = 1;
  if (legacy_over_words) { KF_CHECK(hipMemsetAsync(c->vol.pend, 0, c->n_stored_bricks * sizeof(unsigned long long), c->stream)); c->pend_live = 0; }
  return (int)hipGetLastError();
}

// ---- self-test of kf_div against the compiler's IEEE division (exported for the GPU test-suite) ----------------------------
__global__ void __launch_bounds__(256) k_selftest_div(unsigned n, unsigned seed, int mode, unsigned* mismatches) {
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // two xorshift draws -> operands in the ranges the kernels use
  unsigned s = seed ^ (i * 2654435761u); s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  unsigned t = s * 1664525u + 1013904223u; t ^= t << 13; t ^= t >> 17; t ^= t << 5;
  if (mode == 10 || mode == 11) {
    // the one-instruction float / double -> int conversion against its spelled-out definition: random BIT PATTERNS (every exponent, NaNs,
    // infinities, denormals) and, in the first lanes, the special values and the neighbours of the saturation thresholds
    const float specials[16] = {0.f, -0.f, 1.f, -1.f, 0.5f, -0.5f, 2147483520.f, 2147483648.f, -2147483648.f, -2147483904.f, 4294967296.f,
                                __builtin_huge_valf(), -__builtin_huge_valf(), __builtin_nanf(""), 1e-45f, 16777217.f};
    bool bad;
    if (mode == 10) {
      const float x = i < 16 ? specials[i] : __uint_as_float(s);
      bad = kf_f2i(x) != kf_f2i_spelled(x);
    } else {
      const double specials_d[8] = {2147483647.0, 2147483647.5, 2147483648.0, -2147483648.0, -2147483648.5, -2147483649.0, 1e300, -1e300};
      double x = i < 16 ? (double)specials[i] : (i < 24 ? specials_d[i - 16] : __longlong_as_double(((long long)s << 32) | (long long)t));
      if ((i & 3) == 1 && i >= 24) x = (double)__uint_as_float(s) * 1.0000001;      // values near the float range as well
      bad = kf_to_int(x) != kf_to_int_spelled(x);
    }
    if (bad) atomicAdd(mismatches, 1u);
    return;
  }
  const float u = (float)(s >> 8) * (1.0f / 16777216.0f), w = (float)(t >> 8) * (1.0f / 16777216.0f);
  if (mode == 12) {
    // the closed-form walk of the ray parameter against the chain of additions it replaces: starts anywhere in [2^-4, 2^4) (and right below
    // powers of two), increments with full mantissas, with few mantissa bits (ties with the grid of t) and tiny ones, walks of 1 .. ~400 steps
    unsigned q = t * 22695477u + 1u; q ^= q << 13; q ^= q >> 17; q ^= q << 5;
    float t0 = exp2f(-4.f + 8.f * u);
    if ((i & 7u) == 3u) t0 = __uint_as_float((__float_as_uint(t0) | 0x007FFF00u) - (q & 0xFFu));          // a few ulps below the next binade
    float inc = 1e-3f + 0.08f * w;
    if ((i & 3u) == 1u) inc = __uint_as_float(__float_as_uint(inc) & 0xFFFFF000u);                       // short mantissa
    if ((i & 15u) == 2u) inc = __uint_as_float((__float_as_uint(t0) & 0x7F800000u) - (24u << 23)) * (float)(3 + 2 * (q & 1023u));   // odd multiple of ulp(t)/2: every step a tie
    if ((i & 63u) == 5u) inc = t0 * 1e-8f;                                                                 // below half an ulp: t never moves -> bounded below
    const float steps = (float)((q >> 10) % 400u) + 0.5f * (float)((q >> 20) & 3u);
    const float t_exit = (i & 63u) == 5u ? t0 : t0 + inc * steps;
    float ta = t0, pa = -1.f, tb2 = t0, pb = -1.f;
    kf_ray_advance(ta, pa, inc, t_exit);
    kf_ray_advance_plain(tb2, pb, inc, t_exit);
    if (__float_as_uint(ta) != __float_as_uint(tb2) || __float_as_uint(pa) != __float_as_uint(pb)) atomicAdd(mismatches, 1u);
    return;
  }
  float a, b;
  if (mode == 0) { a = (u * 2.f - 1.f) * 8000.f; b = 1e-4f + w * 20.f; }            // pf.x*fx / pf.z
  else if (mode == 1) { a = (u * 2.f - 1.f) * 3.f; b = 0.005f + w * 0.5f; }          // sdf / trunc
  else if (mode == 2) { a = (u * 2.f - 1.f) * 300.f; b = (float)(1 + (t % 300u)); }  // (t*w + tsdf) / (w + 1)
  else { a = u * 4096.f * (0.5f + w); b = 0.5f + w * 16.f; }                          // pos*R / size
  const float ref = a / b;
  const float got = kf_div(a, kf_recip(b));
  if (__float_as_uint(ref) != __float_as_uint(got)) atomicAdd(mismatches, 1u);
}
extern "C" int kf_selftest_div(kf_ctx* c, unsigned n, unsigned seed, int mode, unsigned* mismatches) {
  if (!c || !mismatches) return KF_ERR_ARG;
  unsigned* d = nullptr;
  KF_CHECK(hipMalloc((void**)&d, 4));
  hipError_t e = hipMemsetAsync(d, 0, 4, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_selftest_div, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, seed, mode, d);
    e = hipMemcpyAsync(mismatches, d, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  hipFree(d);
  return (int)e;
}
