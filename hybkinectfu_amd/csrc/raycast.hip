// raycast.hip -- surface prediction for the next frame's ICP (reference: raycastKernel / raySample / gradientForPoint /
// getMinTime / getMaxTime, src/cuda/raycastingVolume.cu:16-176; tsdfvolume::getVoxel(world), interpolateSDF,
// interpolateColor, src/cuda/tsdfVolume.h:81-172).
//
// gfx950 design (gather / latency bound):
//   * one lane per pixel, a wave covers an 8x8 pixel patch so its 64 rays stay spatially coherent through the volume;
//   * the reference gathers one 12-byte voxel per step (~106 dependent gathers per ray).  Here a step first reads the
//     1-byte flag of the 8^3 brick it lands in (a 256 KiB table at 512^3: L2 resident).  A brick that never held a
//     negative tsdf cannot contain the `sdf < 0` sample of a +/- crossing, so the voxel gather is skipped; the tsdf of
//     the previous sample is fetched lazily only when a negative sample is actually met.  The parameter t is still
//     advanced by the same repeated fp32 addition, so every sample position, the first crossing and all the
//     interpolation inputs are bit-identical to the reference march;
//   * bricks are 4 KiB contiguous, so the trilinear taps at the hit (2 + 6 lookups x 8 voxels) touch 1-2 bricks.
#include "kf_internal.h"
#include "bilateral_tile.h"
#include "grad_shared.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>

// KF_RAYCAST_SHARED_GRAD (0 / 1 / 2, see RaycastArgs::shared_grad) in bits 0-7, KF_RAYCAST_VIEW_HALF (tests: brick layers of the gathers' view on either side, 0 = most) above.
// Read at every launch (two getenv calls), unlike the other switches: a test can then compare the forms on ONE context -- the only way at 2048^3, where a second process
// would have to fuse 69 GB again.
static int rc_shared_grad_env() {
  const char* e = getenv("KF_RAYCAST_SHARED_GRAD"); const char* h = getenv("KF_RAYCAST_VIEW_HALF");
  const int mode = e ? atoi(e) : 1, half = h ? atoi(h) : 0;
  return mode <= 0 ? 0 : ((mode & 255) | ((half > 0 && half < 4096 ? half : 0) << 8));
}
// the gathers' view spans at least two brick layers: they must fit 32-bit offsets (they do up to 724 bricks per axis; beyond: six separate lookups)
static int rc_shared_grad_for(const KfVolume& v) { return (unsigned long long)v.nb * v.nb * KF_BRICK_VOX * sizeof(float2) <= (1ull << 31) ? rc_shared_grad_env() : 0; }

struct RaycastArgs {
  KfVolume vol;
  KfCam cam;
  const float* pose;             // device pointer or null -> pose_val
  KfMat pose_val;
  float4* out_v; float4* out_n; uchar4* out_rgb;
  float* out_t;                  // optional: ray parameter of the first crossing this context detected (+inf: none) -- z-slab merge
  unsigned long long* own_ta;    // z-slab merge, speculative normals (kf_raycast_volume_slab_cross_spec): a second copy of this context's words (the caller all-reduces out_ta in place) ...
  float* out_spec;               // ... and, three floats per pixel, the gradient at this context's own crossing's vertex when that vertex lies in the layers it OWNS (else zeros):
                                 // what kf_slab_ray_normals would compute for the pixel if this crossing wins the MIN all-reduce -- evaluated here, in the shadow of the march
  unsigned long long* out_ta;    // z-slab merge (what SlabPipeline runs): per pixel (bits of the crossing's ray parameter) << 32 | bits of the VERTEX's ray parameter
                                 // alpha -- +inf / 0 without a crossing, alpha 0 where the reference gives up at the crossing (raycastingVolume.cu:87-88).  Only the two
                                 // interpolations at the crossing are evaluated here; the gradient around the vertex is the job of whoever owns the vertex (k_slab_ray_normals)
  KfPyrOut pyr;                  // v1 non-null: the workgroups also leave levels 1 and 2 of the two maps' pyramids (bilateral_tile.h: kf_tile_pyramid)
  float inc, near_plane, far_plane;
  int has_color;
  int neg_words;                 // words of vol.negbits to keep in LDS (0: the table does not fit -> brick flags are read from global memory)
  int meso_words;                // words of the meso table (16^3-voxel cells) to keep in LDS behind the macro / super tables (0: it does not fit)
  int shared_grad;               // 1: a crossing's six gradient taps come from one 32-voxel neighbourhood (grad_shared.h; KF_RAYCAST_SHARED_GRAD=0: six separate lookups; 2: shared, with every other wave forced down the fallback -- tests)
  int bounds_meso;               // 1: where the volume has at most 8192 meso cells (up to 256^3), the tile bounds come from the meso table (KF_RAYCAST_BOUNDS_MESO=0: from the macro table)
  int tile_bounds;               // 1: every workgroup first bounds its tile's rays by the non-empty macro cells its frustum meets (rc_tile_bounds; KF_RAYCAST_BOUNDS=0: off)
  int exp_mode;                  // timing experiments only (KF_RAYCAST_EXP): 1 = stop at the crossing without evaluating it
  KfCounters* work;              // measurement passes only (kf_stage_timers bit 16): count the reference march's samples and the hits
};

// gradientForPoint raycastingVolume.cu:16-42: bounds tested on the LAST sample's voxel, taps taken around the vertex.
// The six taps (+x, -x, +y, -y, +z, -z) are looked up BATCH at a time (BATCH x 8 gathers in flight: 2 = the +/- pair of an axis, 3 = two round trips for the
// six, 6 = one); the reference's early-outs are pure, so loading a batch and then testing the verdicts in its order (+ then -, x then y then z) is equivalent.
// The interpolation itself is the reference's, operation for operation (kf_interp_prepare / kf_interp_finish).
template <int BATCH>
__device__ __forceinline__ bool gradient_for_point(const KfVolume& v, float3 samplepos, float3 vtx, const KfRecip& rS, const KfRecip& rcell, float3& grad) {
  static_assert(BATCH == 2 || BATCH == 3 || BATCH == 6, "six taps in whole batches");
  const float rf = (float)v.res;
  const int3 g = make_int3(kf_f2i(kf_div(samplepos.x * rf, rS)), kf_f2i(kf_div(samplepos.y * rf, rS)), kf_f2i(kf_div(samplepos.z * rf, rS)));
  const int R = v.res;
  if (g.x <= 1 || g.x >= R - 2) return false;
  if (g.y <= 1 || g.y >= R - 2) return false;
  if (g.z <= 1 || g.z >= R - 2) return false;
  const float cell = v.cell;
  float f[6]; bool o[6];
#pragma unroll
  for (int k0 = 0; k0 < 6; k0 += BATCH) {
    KfInterp it[BATCH]; float2 q[BATCH][8];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int t = k0 + k; const float d = (t & 1) ? -cell : cell;
      const float3 p = kf3(vtx.x + ((t >> 1) == 0 ? d : 0.f), vtx.y + ((t >> 1) == 1 ? d : 0.f), vtx.z + ((t >> 1) == 2 ? d : 0.f));
      it[k] = BATCH == 2 ? kf_interp_prepare(v, p, rS, rcell) : kf_interp_prepare_nb(v, p, rS, rcell);
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) { if (BATCH == 2) kf_interp_load(v, it[k], q[k]); else kf_interp_load_nb(v, it[k], q[k]); }     // (larger batches: no branch around the loads, so that they travel together)
#pragma unroll
    for (int k = 0; k < BATCH; ++k) { f[k0 + k] = 0.f; o[k0 + k] = kf_interp_finish(it[k], q[k], f[k0 + k]); }
    // (:22-37: the reference returns at the first failing tap; every tap of this batch and the earlier ones must have succeeded to go on)
#pragma unroll
    for (int k = 0; k < BATCH; ++k) if (!o[k0 + k]) return false;
  }
  float3 n;
  n.x = f[0] - f[1]; n.y = f[2] - f[3]; n.z = f[4] - f[5];
  float len = kf_norm(n);
  if ((double)len < 1e-8) return false;
  grad = kf_scale(n, 1 / len);                               // fp32 reciprocal (:40), unlike normalize()
  return true;
}

// The same verdict and gradient from the shared neighbourhood (grad_shared.h): nine cell computations and 32 gathers instead of 18 and 48.  A wave in which
// some lane's tap cell is not "vtx's cell moved by one" -- and whose verdict is not false already -- evaluates the generic way, all lanes together.
template <int BATCH, int ROUNDS>
__device__ __forceinline__ bool gradient_for_point_either(int shared, bool odd_wave, const KfVolume& v, float3 samplepos, float3 vtx, const KfRecip& rS, const KfRecip& rcell, float3& grad) {
  int verdict = 2;
  if (shared) verdict = rc_gradient_shared<ROUNDS>(v, samplepos, vtx, rS, rcell, (shared & 255) == 2 && odd_wave, shared >> 8, grad);       // (`shared` is a launch argument: uniform; 2: every other wave is sent down the generic path)
  if (verdict == 2) verdict = gradient_for_point<BATCH>(v, samplepos, vtx, rS, rcell, grad) ? 1 : 0;
  return verdict == 1;
}

// the walk of the ray parameter: the chain of additions itself.  Its closed form (kf_ray_advance: exact, self-tested) is 3-4 % SLOWER here --
// a walk is 1-12 additions at the stock increment (4.5 voxels) and the chain costs three instructions per step (-DKF_RAY_ADVANCE_CLOSED for the A/B)
#ifdef KF_RAY_ADVANCE_CLOSED
#define RC_ADVANCE kf_ray_advance
#else
#define RC_ADVANCE kf_ray_advance_plain
#endif
#define RAYCAST_THREADS 512
#ifndef RC_GRAD_BATCH
#define RC_GRAD_BATCH 2           // taps of the crossing's gradient looked up together (2 / 3 / 6); 3 (two round trips) measured slower under the launch's 80-register cap: 46.0 vs 45.1 us at C2, 85.0 vs 83.6 at C4
#endif
#ifndef RC_GRAD_ROUNDS
#define RC_GRAD_ROUNDS 4          // the shared-neighbourhood form of those taps (grad_shared.h): gathers in 1 / 2 / 4 dependent rounds of 32 / 16 / 8
#endif
#ifndef RC_SLAB_GRAD_ROUNDS
#define RC_SLAB_GRAD_ROUNDS 1     // the same in k_slab_ray_normals (no register cap there)
#endif
#define RAYCAST_LDS_BYTES 49152   // budget for the two bit tables: 3 workgroups x 8 waves stay resident per CU (160 KiB LDS)

// bit `i` of a packed table
__device__ __forceinline__ bool rc_bit(const unsigned* words, unsigned i) { return (words[i >> 5] >> (i & 31u)) & 1u; }

// The march of one ray over the samples t_k in [t, t_end): the reference's loop (raySample :65-119) with the two empty-space levels.
// `t`, `t_prev`, `have_last`, `last_sdf` carry the reference's per-ray state in; on return t_cross < +inf names the first crossing of
// the range (with t_cross_prev the sample before it).  Whether sample k is a crossing depends on samples k-1 and k only (the previous
// sample's tsdf is fetched on demand), so a ray's range may be marched in pieces by different lanes: the first crossing of the ray
// is the first piece's that has one.
struct RcRay { float3 org, dir, inv_dir; };
__device__ __forceinline__ void rc_march(const RaycastArgs& a, const KfVolume& v, const unsigned* s_macro, const unsigned* s_super, const unsigned* s_meso, const unsigned* s_neg, bool neg_in_lds, const RcRay& ray,
                                         const KfRecip& rS, float t_end, float& t, float& t_prev, bool& have_last, float& last_sdf,
                                         float& t_cross, float& t_cross_prev, int& n_iter, int& n_samp, int& n_macro) {
  const float3 org = ray.org, dir = ray.dir;
  const int R = v.res;
  const float rf = (float)R;
  const int zs0 = v.bz0 * KF_BRICK, zs1 = v.bz1 * KF_BRICK;
  const int nm = v.nm, ns = v.ns, nq = v.nq;
  const bool meso_in_lds = a.meso_words != 0;
  // the empty-space walk measures in VOXEL units along each axis: with q the sample's (unrounded) voxel coordinate and `base` the first voxel
  // of its cell (32 voxels wide for a macro cell, 8 for a brick), the cell's far face lies E - (q - base) voxels ahead for a ray going up
  // the axis and q - base voxels for one going down: one fused multiply-add per axis with the ray's constants sgn / up, times cell / |dir|
  const float3 sgn = kf3(dir.x > 0.f ? -1.f : 1.f, dir.y > 0.f ? -1.f : 1.f, dir.z > 0.f ? -1.f : 1.f);
  const float3 up = kf3(dir.x > 0.f ? 1.f : 0.f, dir.y > 0.f ? 1.f : 0.f, dir.z > 0.f ? 1.f : 0.f);
  const float3 per_vox = kf3(v.cell * fabsf(ray.inv_dir.x), v.cell * fabsf(ray.inv_dir.y), v.cell * fabsf(ray.inv_dir.z));
  // `worldPos * resolution / size` (tsdfVolume.h:50-56): for a power-of-two size the quotient is an exact scaling -- the same bits from a product
  const float inv_size = 1.0f / v.size;
  const bool pow2 = (__float_as_uint(v.size) & 0x007FFFFFu) == 0u;
  while (t < t_end) {
    ++n_iter;
    const float3 pos = kf_add(org, kf_scale(dir, t));
    // the sample's own voxel -- tsdfvolume::getVoxel(world) tsdfVolume.h:81-97: nearest voxel, index clamped
    const float qx = pow2 ? (pos.x * rf) * inv_size : kf_div(pos.x * rf, rS), qy = pow2 ? (pos.y * rf) * inv_size : kf_div(pos.y * rf, rS),
                qz = pow2 ? (pos.z * rf) * inv_size : kf_div(pos.z * rf, rS);
    int gx = kf_f2i(qx), gy = kf_f2i(qy), gz = kf_f2i(qz);
    gx = max(0, min(gx, R - 1)); gy = max(0, min(gy, R - 1)); gz = max(0, min(gz, R - 1));
    // level 1: a 32^3-voxel macro cell without any negative voxel -> none of the samples inside it can be the negative
    // side of a crossing: walk to its far side.  The cell is the VOXEL's macro cell (g >> 5), like the brick level below:
    // picking it from the position with a different rounding could disagree with the voxel index at a cell face and skip
    // a sample whose voxel lies in the neighbouring (non-empty) cell.
    // level 2: only samples whose voxel this context OWNS can be its crossing candidates (the whole volume on one GPU; with
    // z-slabs the neighbour's layers are stored as halo and serve the previous-sample / trilinear / gradient reads only); a brick
    // that never held a negative tsdf cannot hold the negative sample of a crossing either.
    // Both levels take ONE code path with selected parameters: the lanes of
    // a wave sit in different states, and a wave executes the union of the paths its lanes take on every trip.
    const int mx = gx >> 5, my = gy >> 5, mz = gz >> 5;
    const bool macro_empty = !rc_bit(s_macro, __umul24(__umul24((unsigned)mz, (unsigned)nm) + (unsigned)my, (unsigned)nm) + (unsigned)mx);
    // level 0: the 128^3-voxel super cell (4 x 4 x 4 macro cells) the same way -- two thirds of a ray's trips were macro cells of open space
    const bool super_empty = !rc_bit(s_super, __umul24(__umul24((unsigned)(mz >> KF_SUPER_SHIFT), (unsigned)ns) + (unsigned)(my >> KF_SUPER_SHIFT), (unsigned)ns) + (unsigned)(mx >> KF_SUPER_SHIFT));
    const bool owned = gz >= v.own_z0 && gz < v.own_z1;
    // level 1.5 (round 5): the 16^3-voxel meso cell (2 x 2 x 2 bricks) -- inside a non-empty macro cell most bricks are still empty, and at 1024^3 the per-brick bits do
    // not fit into LDS (every brick-level trip then reads the flag byte from global memory): the meso bits do (32 KiB), and an empty meso cell is walked in one trip
    const bool meso_empty = meso_in_lds && !macro_empty &&
                            !rc_bit(s_meso, __umul24(__umul24((unsigned)(gz >> 4), (unsigned)nq) + (unsigned)(gy >> 4), (unsigned)nq) + (unsigned)(gx >> 4));
    size_t slot = 0; bool has_neg = false;
    if (!macro_empty && !meso_empty && owned) {
      slot = kf_brick_slot(v, gx >> 3, gy >> 3, gz >> 3);
      has_neg = neg_in_lds ? rc_bit(s_neg, (unsigned)slot) : (v.flags[slot] & KF_FLAG_HASNEG) != 0;
    }
    if (!has_neg) {
      n_macro += macro_empty ? (super_empty ? 0x10000 : 1) : 0;
      // macro cell: walk to its far side; owned brick with the table an LDS read away: brick by brick is cheaper than sample by
      // sample (the cell is the VOXEL's brick: if rounding put pos a hair outside it, the walk is merely shorter, never past the
      // far face); otherwise one sample
      const bool walk = macro_empty || meso_empty || (owned && neg_in_lds);
      if (walk) {
        const int sup = (KF_MACRO << KF_SUPER_SHIFT);
        const int mask = super_empty ? ~(sup - 1) : macro_empty ? ~31 : meso_empty ? ~15 : ~7;
        const float edge = super_empty ? (float)sup : macro_empty ? 32.f : meso_empty ? 16.f : 8.f;
        const float eps = super_empty ? 1e-4f * (float)sup : macro_empty ? 3.2e-3f : meso_empty ? 4e-3f : 8e-3f;       // eps: 1e-4 / 1e-4 / 2.5e-4 / 1e-3 of the cell edge, in voxels
        // the exit parameter only has to be conservative (eps and the 1e-6 t margin absorb a few ulps)
        const float dx = __builtin_fmaf(qx - (float)(gx & mask), sgn.x, up.x * edge) - eps;
        const float dy = __builtin_fmaf(qy - (float)(gy & mask), sgn.y, up.y * edge) - eps;
        const float dz = __builtin_fmaf(qz - (float)(gz & mask), sgn.z, up.z * edge) - eps;
        const float dt = fminf(fminf(dx * per_vox.x, dy * per_vox.y), dz * per_vox.z);
        const float t_exit = fminf(t + dt - 1e-6f * t, t_end);
        RC_ADVANCE(t, t_prev, a.inc, t_exit);                   // the reference's own repeated addition: identical sample parameters
      } else { t_prev = t; t += a.inc; }
      have_last = false;
      continue;
    }
    ++n_samp;
    const float sdf = v.tw[(size_t)slot * KF_BRICK_VOX + (size_t)(((gz & 7) << 6) | ((gy & 7) << 3) | (gx & 7))].x;
    if (sdf < 0.0f) {
      if (!have_last) {                                                    // the previous sample's tsdf was never fetched: fetch it now
        const float3 last_pos = kf_add(org, kf_scale(dir, t_prev));        // recomputed exactly as the march computed it
        int lx = kf_f2i(pow2 ? (last_pos.x * rf) * inv_size : kf_div(last_pos.x * rf, rS)), ly = kf_f2i(pow2 ? (last_pos.y * rf) * inv_size : kf_div(last_pos.y * rf, rS)),
            lz = kf_f2i(pow2 ? (last_pos.z * rf) * inv_size : kf_div(last_pos.z * rf, rS));
        lx = max(0, min(lx, R - 1)); ly = max(0, min(ly, R - 1)); lz = max(0, min(lz, R - 1));
        last_sdf = (lz >= zs0 && lz < zs1) ? v.tw[kf_vox_index(v, lx, ly, lz)].x : 0.f;
        have_last = true;
      }
      if (last_sdf > 0.0f) { t_cross = t; t_cross_prev = t_prev; break; }  // zero crossing :83
    }
    last_sdf = sdf; have_last = true; t_prev = t;
    t += a.inc;
  }
}

// the ray of pixel (x, y) -- raycastKernel :136-150.  One function for the march and for k_slab_rays_unpack, which rebuilds a vertex from
// its ray parameter: the same operations in the same order give the same bits.
__device__ __forceinline__ void rc_pixel_ray(const KfCam& cam, const float* T, int x, int y, float3& org, float3& dir, float3& cam_dir) {
  cam_dir = kf_normalize(kf_depth_to_skeleton((unsigned)x, (unsigned)y, 1.0f, cam));
  org = kf3(T[3], T[7], T[11]);
  const float4 wd = kf_mat_vec(T, make_float4(cam_dir.x, cam_dir.y, cam_dir.z, 0.0f));
  dir = kf3(wd.x, wd.y, wd.z);
  dir.x = (dir.x == 0.f) ? (float)1e-15 : dir.x;
  dir.y = (dir.y == 0.f) ? (float)1e-15 : dir.y;
  dir.z = (dir.z == 0.f) ? (float)1e-15 : dir.z;
}

// [t_min, t_max) of a ray: getMinTime / getMaxTime (raycastingVolume.cu:44-63) clipped by the near / far planes (:152-153)
__device__ __forceinline__ void rc_ray_interval(float S, float near_plane, float far_plane, float3 org, float3 dir, float3 cam_dir, float& tmin, float& tmax) {
  tmin = fmaxf(fmaxf(((dir.x > 0 ? 0.f : S) - org.x) / dir.x, ((dir.y > 0 ? 0.f : S) - org.y) / dir.y), ((dir.z > 0 ? 0.f : S) - org.z) / dir.z);
  tmax = fminf(fminf(((dir.x > 0 ? S : 0.f) - org.x) / dir.x, ((dir.y > 0 ? S : 0.f) - org.y) / dir.y), ((dir.z > 0 ? S : 0.f) - org.z) / dir.z);
  tmin = fmaxf(tmin, near_plane / cam_dir.z);
  tmax = fminf(tmax, far_plane / cam_dir.z);
}

// ---- where along its rays can a 32x16 pixel tile meet a surface at all? ---------------------------------------------------------------------------------
// A crossing's negative sample lies in a macro cell (32^3 voxels) whose bit is set.  Before the march the workgroup intersects the tile's frustum (four planes
// through the camera centre, half a pixel of margin) with the bounding spheres of the non-empty SUPER cells, then of the non-empty MACRO cells inside those that
// pass, and keeps the smallest / largest distance from the camera centre any of them reaches: [t_lo, t_hi].  A ray parameter IS the distance from the camera
// centre (unit directions), so every sample outside that interval lies in an empty cell and can be skipped exactly like the march skips one empty cell at a
// time -- only all at once: the parameter is brought to t_lo by the reference's own chain of additions (kf_ray_advance, the closed form of that chain) and the
// march ends at t_hi.  At 512^3 the walk to the first surface was 5.8 of a ray's 11.7 loop trips (profiles/r03_raycast_levels.txt), each ~340 instructions for
// the whole wave.  Conservative by construction: spheres are inflated by 1 % + 1e-3 of the volume (the pose's rotation block is orthonormal to ~1e-6 only);
// more candidates than RC_BOUNDS_MAX: no bounds.  s_rb: RC_BOUNDS_WORDS words of LDS, s_rb[0..3] initialised (0, +inf bits, 0, 0) before the tables' barrier.
#define RC_BOUNDS_MAX 512
#define RC_BOUNDS_WORDS (4 + RC_BOUNDS_MAX)
struct RcFrustum { float ox, oy, oz; float r0, r1, r2, r3, r4, r5, r6, r7, r8; float aL, aR, bT, bB, nL, nR, nT, nB; };
__device__ __forceinline__ bool rc_sphere_in_frustum(const RcFrustum& f, float cx, float cy, float cz, float r, float& dist) {
  const float dx = cx - f.ox, dy = cy - f.oy, dz = cz - f.oz;
  const float px = f.r0 * dx + f.r3 * dy + f.r6 * dz, py = f.r1 * dx + f.r4 * dy + f.r7 * dz, pz = f.r2 * dx + f.r5 * dy + f.r8 * dz;     // R^T (c - org)
  dist = sqrtf(dx * dx + dy * dy + dz * dz);
  return pz + r > 0.f && (px - f.aL * pz) >= -r * f.nL && (f.aR * pz - px) >= -r * f.nR && (py - f.bT * pz) >= -r * f.nT && (f.bB * pz - py) >= -r * f.nB;
}
// the frustum of the pixel rectangle [x0, x1] x [y0, y1] (pixel centres, half a pixel of margin)
__device__ __forceinline__ void rc_set_window(RcFrustum& f, const KfCam& cam, int x0, int y0, int x1, int y1) {
  f.aL = ((float)x0 - 0.5f - cam.cx) / cam.fx; f.aR = ((float)x1 + 0.5f - cam.cx) / cam.fx;
  f.bT = ((float)y0 - 0.5f - cam.cy) / cam.fy; f.bB = ((float)y1 + 0.5f - cam.cy) / cam.fy;
  f.nL = sqrtf(1.f + f.aL * f.aL); f.nR = sqrtf(1.f + f.aR * f.aR); f.nT = sqrtf(1.f + f.bT * f.bT); f.nB = sqrtf(1.f + f.bB * f.bB);
}
// s_rb: [0] super-cell candidates, [1] / [2] the tile's bounds as bits, [3] spare, then the candidate list.
// (A third level -- every wave testing the 64 bricks of each candidate macro cell against its own 8x8 patch's frustum -- was built and measured: the tests cost
// more than the shorter march saves, 48.7 vs 46.2 us at 512^3 and 103 vs 85 us at 1024^3 where the bricks' bits do not fit into LDS; profiles/r05_raycast_bounds.txt)
__device__ __forceinline__ void rc_tile_bounds(const RaycastArgs& a, const unsigned* s_macro, const unsigned* s_super, const unsigned* s_meso, const unsigned* s_neg, bool neg_in_lds, const float* T,
                                               int tile_x, int tile_y, unsigned* s_rb, float& t_lo, float& t_hi) {
  const KfVolume& v = a.vol;
  RcFrustum f;
  f.ox = T[3]; f.oy = T[7]; f.oz = T[11];
  f.r0 = T[0]; f.r1 = T[1]; f.r2 = T[2]; f.r3 = T[4]; f.r4 = T[5]; f.r5 = T[6]; f.r6 = T[8]; f.r7 = T[9]; f.r8 = T[10];      // rows of R (camera -> world): R^T applied column-wise above
  const int x0 = tile_x * 32, y0 = tile_y * 16;
  rc_set_window(f, a.cam, x0, y0, min(x0 + 31, a.cam.cols - 1), min(y0 + 15, a.cam.rows - 1));
  const int nm = v.nm, ns = v.ns;
  const float cell = v.cell, slack = 1e-3f * v.size;
  const float r_super = 0.8660254f * 1.01f * (float)(KF_MACRO << KF_SUPER_SHIFT) * cell + slack, r_macro = 0.8660254f * 1.01f * (float)KF_MACRO * cell + slack;
  unsigned* list = s_rb + 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float lo = __builtin_huge_valf(), hi = 0.f;
  unsigned n = 0u;
  // small volumes (up to 256^3: 16^3 = 4096 meso cells of 16^3 voxels): the same scan over the MESO table -- cells half as wide, bounds twice as tight, where
  // a 256^3 volume has only 8^3 macro cells (RaycastArgs::bounds_meso; the table is in LDS)
  const int nq = v.nq;
  const bool by_meso = a.bounds_meso && a.meso_words != 0 && nq * nq * nq <= RAYCAST_THREADS * 16;
  if (by_meso || nm * nm * nm <= RAYCAST_THREADS * 16) {
    // few cells (macro: up to 512^3 = 4096): every thread tests its share of them directly -- one barrier instead of two, no list.  A thread takes whole BYTES of the
    // table (one LDS read per eight cells, then only the set bits), cells numbered x-fastest
    const unsigned* tbl = by_meso ? s_meso : s_macro;
    const int nc = by_meso ? nq : nm;
    const float edge = by_meso ? 16.f : (float)KF_MACRO;
    const float r_cell = by_meso ? 0.8660254f * 1.01f * 16.f * cell + slack : r_macro;
    const int n_cells = nc * nc * nc;
    const float inv_nc = 1.0f / (float)nc;
    for (int byte = (int)threadIdx.x; byte * 8 < n_cells; byte += RAYCAST_THREADS) {
      unsigned bits = (tbl[byte >> 2] >> ((byte & 3) * 8)) & 0xFFu;
      while (bits) {
        const int m = byte * 8 + (int)__builtin_ctz(bits);
        bits &= bits - 1u;
        if (m >= n_cells) break;
        // m = (mz * nc + my) * nc + mx by two exact float quotients (m < 2^13, nc <= 20: the products are far from the rounding boundary after the +0.5)
        const int q = (int)(((float)m + 0.5f) * inv_nc), mx = m - q * nc, mz = (int)(((float)q + 0.5f) * inv_nc), my = q - mz * nc;
        float dist;
        if (rc_sphere_in_frustum(f, ((float)mx * edge + 0.5f * edge) * cell, ((float)my * edge + 0.5f * edge) * cell, ((float)mz * edge + 0.5f * edge) * cell, r_cell, dist)) {
          lo = fminf(lo, fmaxf(dist - r_cell, 0.f)); hi = fmaxf(hi, dist + r_cell);
        }
      }
    }
  } else {
    // level 0: non-empty super cells that meet the tile's frustum -> the list
    for (int s = (int)threadIdx.x; s < ns * ns * ns; s += RAYCAST_THREADS) {
      if (!rc_bit(s_super, (unsigned)s)) continue;
      const int sx = s % ns, sy = (s / ns) % ns, sz = s / (ns * ns);
      const float h = 0.5f * (float)(KF_MACRO << KF_SUPER_SHIFT);
      float dist;
      if (rc_sphere_in_frustum(f, ((float)(sx * (KF_MACRO << KF_SUPER_SHIFT)) + h) * cell, ((float)(sy * (KF_MACRO << KF_SUPER_SHIFT)) + h) * cell,
                               ((float)(sz * (KF_MACRO << KF_SUPER_SHIFT)) + h) * cell, r_super, dist)) {
        const unsigned k = atomicAdd(&s_rb[0], 1u);
        if (k < RC_BOUNDS_MAX) list[k] = (unsigned)s;
      }
    }
    __syncthreads();
    n = s_rb[0];
    if (n <= RC_BOUNDS_MAX) {
      // level 1: the macro cells of the listed super cells, one super cell per wave and round, one macro cell per lane
      for (unsigned k = (unsigned)wave; k < n; k += RAYCAST_THREADS / 64) {
        const int s = (int)list[k];
        const int mx = ((s % ns) << KF_SUPER_SHIFT) + (lane & 3), my = (((s / ns) % ns) << KF_SUPER_SHIFT) + ((lane >> 2) & 3), mz = ((s / (ns * ns)) << KF_SUPER_SHIFT) + (lane >> 4);
        if (mx >= nm || my >= nm || mz >= nm) continue;
        const unsigned mi = __umul24(__umul24((unsigned)mz, (unsigned)nm) + (unsigned)my, (unsigned)nm) + (unsigned)mx;
        if (!rc_bit(s_macro, mi)) continue;
        float dist;
        if (rc_sphere_in_frustum(f, ((float)(mx * KF_MACRO) + 0.5f * KF_MACRO) * cell, ((float)(my * KF_MACRO) + 0.5f * KF_MACRO) * cell, ((float)(mz * KF_MACRO) + 0.5f * KF_MACRO) * cell, r_macro, dist)) {
          lo = fminf(lo, fmaxf(dist - r_macro, 0.f)); hi = fmaxf(hi, dist + r_macro);
        }
      }
    }
  }
  // the few lanes that found a cell in sight put their extremes into LDS themselves (non-negative floats order like their bits): a dozen atomics per workgroup
  // are cheaper than twelve cross-lane shuffles per wave
  (void)lane;
  if (hi > 0.f) { atomicMin(&s_rb[1], __float_as_uint(lo)); atomicMax(&s_rb[2], __float_as_uint(hi)); }
  __syncthreads();
  if (n > RC_BOUNDS_MAX) { t_lo = 0.f; t_hi = __builtin_huge_valf(); return; }
  t_lo = __uint_as_float(s_rb[1]); t_hi = __uint_as_float(s_rb[2]);
}

// one 32x16 pixel tile (tile_x, tile_y) by the 512 threads of a workgroup; s_tables: the workgroup's dynamic LDS
__device__ __forceinline__ void raycast_tile(const RaycastArgs& a, int tile_x, int tile_y, unsigned* s_tables) {
  const KfVolume& v = a.vol;
  // Packed bit tables live in LDS so that the empty-space walk costs LDS reads instead of dependent L2 round trips: one bit
  // per 32^3-voxel macro cell and per 128^3-voxel super cell of the whole volume (KfVolume::macrobits, kept current by the
  // fusion pass with atomicOr: copied as they are), and -- when it fits -- one bit per stored 8^3 brick (KfVolume::negbits).
#ifdef KF_EXPERIMENTS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
  const int skip_words = v.macro_words + v.super_words + a.meso_words;   // multiples of 4: everything is moved as uint4 (the meso table lies behind the other two: one copy)
  const unsigned* s_macro = s_tables;
  const unsigned* s_super = s_tables + v.macro_words;
  const unsigned* s_meso = s_tables + v.macro_words + v.super_words;
  const unsigned* s_neg = s_tables + skip_words;
  __shared__ unsigned s_rb[RC_BOUNDS_WORDS];
  if (threadIdx.x == 0) { s_rb[0] = 0u; s_rb[1] = 0x7F800000u; s_rb[2] = 0u; s_rb[3] = 0u; }
  {
    const uint4* msrc = reinterpret_cast<const uint4*>(v.macrobits);
    uint4* mdst = reinterpret_cast<uint4*>(s_tables);
    for (int i = threadIdx.x; i < skip_words / 4; i += RAYCAST_THREADS) mdst[i] = msrc[i];
    if (a.neg_words) {
      const uint4* nsrc = reinterpret_cast<const uint4*>(v.negbits);
      uint4* ndst = reinterpret_cast<uint4*>(s_tables + skip_words);
      for (int i = threadIdx.x; i < a.neg_words / 4; i += RAYCAST_THREADS) ndst[i] = nsrc[i];
    }
    __syncthreads();
  }
  const bool neg_in_lds = a.neg_words != 0;
  float tile_lo = 0.f, tile_hi = __builtin_huge_valf();
  if (a.tile_bounds) rc_tile_bounds(a, s_macro, s_super, s_meso, s_neg, neg_in_lds, a.pose ? a.pose : a.pose_val.m, tile_x, tile_y, s_rb, tile_lo, tile_hi);     // (uniform)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // a workgroup is a 32x16 pixel tile, a wave an 8x8 patch of it
  const int x = tile_x * 32 + (wave & 3) * 8 + (lane & 7), y = tile_y * 16 + (wave >> 2) * 8 + (lane >> 3);
  // (pixels outside the image and the timing experiment "half the rays" -- latency- or throughput-bound? -- march nothing but stay for the epilogue)
  const bool live = x < a.cam.cols && y < a.cam.rows && !(KF_EXP_MODE(a) == 2 && ((tile_x + tile_y) & 1));
  float4 out_v = make_float4(0.f, 0.f, 0.f, 0.f), out_n = make_float4(0.f, 0.f, 0.f, 0.f);
  auto march_pixel = [&]() {
  const int pix = y * a.cam.cols + x;
  const float inf = __builtin_huge_valf();
  uchar4 out_c = make_uchar4(0, 0, 0, 0);
  float t_cross = inf, t_cross_prev = 0.f, out_alpha = 0.f;
  const float* T = a.pose ? a.pose : a.pose_val.m;
  float3 org, dir, cam_dir;
  rc_pixel_ray(a.cam, T, x, y, org, dir, cam_dir);
  const float S = v.size;
  float tmin, tmax;
  rc_ray_interval(S, a.near_plane, a.far_plane, org, dir, cam_dir, tmin, tmax);
  const float ref_tmin = tmin, ref_tmax = tmax;
#ifdef KF_EXPERIMENTS
  unsigned long long st1 = __builtin_amdgcn_s_memtime(), st2 = st1;
#endif
  int n_iter = 0, n_samp = 0, n_macro = 0;
  if (tmin < tmax) {
    // raySample :65-119
    const int R = v.res;
    const KfRecip rS = kf_recip(S);                                       // `worldPos.x*_resolution.x/_size.x`: shared divisor
    const KfRecip rcell = kf_recip(v.cell);
    float t = tmin, t_prev = tmin;
    float last_sdf = 0.f; bool have_last = true;
    RcRay ray; ray.org = org; ray.dir = dir; ray.inv_dir = kf3(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
    // z-slabs: only samples inside the owned layers can be this context's candidates, so the march is clipped to the ray's
    // passage through them (one cell of margin on either side); the parameter still reaches the first such sample by the
    // reference's repeated addition, and the previous sample is fetched lazily like after any other skip
    float t_end = tmax, t_first = tmin;
    if (v.own_z0 > 0 || v.own_z1 < R) {
      const float za = ((float)(v.own_z0 - 1) * v.cell - org.z) * ray.inv_dir.z, zb = ((float)(v.own_z1 + 1) * v.cell - org.z) * ray.inv_dir.z;
      const float t_in = fminf(za, zb), t_out = fmaxf(za, zb);
      t_end = fminf(tmax, t_out);
      t_first = fmaxf(tmin, fminf(t_in - 1e-6f * fabsf(t_in), t_end));
    }
    if (t < t_first) { RC_ADVANCE(t, t_prev, a.inc, t_first); have_last = false; }
    // the tile's bounds (rc_tile_bounds): nothing before tile_lo or from tile_hi on can be a crossing's negative sample; the long first walk in closed form
    t_end = fminf(t_end, tile_hi);
    if (t < tile_lo && t < t_end) { kf_ray_advance(t, t_prev, a.inc, fminf(tile_lo, t_end)); have_last = false; }
    rc_march(a, v, s_macro, s_super, s_meso, s_neg, neg_in_lds, ray, rS, t_end, t, t_prev, have_last, last_sdf, t_cross, t_cross_prev, n_iter, n_samp, n_macro);
#ifdef KF_EXPERIMENTS
    st2 = __builtin_amdgcn_s_memtime();
#endif
    // The crossing is evaluated HERE, after the march loop, not inside it: lanes of a wave meet their crossings at
    // different iterations, and inside the loop the 64-gather evaluation would run once per distinct iteration with a
    // handful of active lanes each time.  After the loop every lane that found a crossing evaluates together.
    if (t_cross < inf && KF_EXP_MODE(a) != 1) {
      const float3 pos = kf_add(org, kf_scale(dir, t_cross)), last_pos = kf_add(org, kf_scale(dir, t_cross_prev));
      float ftdt, ft; bool ok_cur, ok_last;
      kf_interpolate_sdf_pair(v, pos, last_pos, rS, rcell, ok_cur, ftdt, ok_last, ft);
      // :87-88 `break` on either failure.  One gradient evaluation serves both outputs: the model maps (vertex, normal) -- or, z-slabs, the speculative normal.
      // z-slabs: the vertex org + dir * alpha may lie ANYWHERE along the ray -- alpha = t - inc * f(t) / (f(t) - f(t - inc)) extrapolates without bound when the
      // two interpolated values nearly agree (an isolated negative voxel at a silhouette) -- so its gradient taps are not this slab's to read unless the vertex
      // lies in the layers it OWNS (the owner's part of k_slab_ray_normals, for this context's own crossing: same vertex, same layer test, same gradient;
      // last_pos is the march's own previous sample -- what that kernel finds by replaying the chain of additions); alpha leaves either way.
      if (ok_cur && ok_last) {
        const float alpha = t_cross - a.inc * ftdt / (ftdt - ft);
        const float3 vtx = kf_add(org, kf_scale(dir, alpha));
        bool want = true;
        if (a.out_ta) {
          out_alpha = alpha;
          int gzv = kf_f2i(kf_div(vtx.z * (float)v.res, rS));
          gzv = max(0, min(gzv, v.res - 1));
          want = a.out_spec != nullptr && __float_as_uint(alpha) != 0u && gzv >= v.own_z0 && gzv < v.own_z1;
        } else if (a.has_color) { uchar4 c = make_uchar4(0, 0, 0, 0); kf_interpolate_color(v, vtx, c); out_c = c; }
        float3 grad;
        if (want && gradient_for_point_either<RC_GRAD_BATCH, RC_GRAD_ROUNDS>(a.shared_grad, ((tile_x + tile_y + wave) & 1) != 0, v, last_pos, vtx, rS, rcell, grad)) {
          out_n = make_float4(grad.x, grad.y, grad.z, 0.f);
          if (!a.out_ta) { out_v = make_float4(vtx.x, vtx.y, vtx.z, 1.0f); out_alpha = alpha; }
        }
      }
    } else if (t_cross < inf) out_v = make_float4(t_cross, 0.f, 0.f, 1.f);
  }
#ifdef KF_EXPERIMENTS
  if (KF_EXP_MODE(a) == 3) {                                                // diagnostics: shader-clock ticks of the three phases, loop trips
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    out_v = make_float4((float)(st1 - st0), (float)(st2 - st1), (float)(st3 - st2), (float)n_iter);
    out_n = make_float4((float)n_samp, (float)(n_macro & 0xFFFF), (float)(n_macro >> 16), 0.f);
  }
#endif
  if (a.out_ta) {
    const unsigned long long word = ((unsigned long long)__float_as_uint(t_cross) << 32) | (unsigned long long)(t_cross < inf ? __float_as_uint(out_alpha) : 0u);
    a.out_ta[pix] = word;
    if (a.out_spec) { a.own_ta[pix] = word; a.out_spec[3 * pix] = out_n.x; a.out_spec[3 * pix + 1] = out_n.y; a.out_spec[3 * pix + 2] = out_n.z; }
  }
  else { a.out_v[pix] = out_v; a.out_n[pix] = out_n; }
  if (a.work) {
    // what the REFERENCE's march reads for this ray (raycastingVolume.cu:65-119): one voxel per sample from t_min up to the
    // crossing (or t_max); a crossing is evaluated with 2 + 6 trilinear look-ups of 8 voxels each
    const float t_stop = t_cross < inf ? t_cross : ref_tmax;
    const float n_s = (ref_tmin < ref_tmax) ? floorf((t_stop - ref_tmin) / a.inc) + 1.f : 0.f;
    const float steps = kf_wave_sum(n_s), hits = kf_wave_sum(t_cross < inf ? 1.f : 0.f);
    if ((threadIdx.x & 63) == 0) {
      const unsigned sh = ((unsigned)(tile_y * 61 + tile_x) * 8u + (threadIdx.x >> 6)) & 63u;
      atomicAdd(&a.work->rc_steps[sh * 16], (unsigned long long)steps);
      atomicAdd(&a.work->rc_hits[sh * 16], (unsigned long long)hits);
    }
  }
  if (a.out_t) a.out_t[pix] = t_cross;
  if (a.has_color) a.out_rgb[pix] = out_c;
  };
  if (live) march_pixel();
  // Levels 1 and 2 of the model maps' pyramids, which the NEXT frame's tracker reads first (ICP.cpp:57-60): a 32x16 tile holds whole 2x2 and
  // 4x4 blocks, so the workgroup that produced the texels averages them itself -- through the LDS the bit tables occupied until its last wave
  // left the march -- and the tracker's pyramid launch (a pass over four maps, ~8 us) has nothing left to do.
  if (a.pyr.v1) {                                                            // uniform
    __syncthreads();                                                         // every wave is done with the tables
    float4* s_v = reinterpret_cast<float4*>(s_tables);
    float4* s_n = s_v + 32 * 16, *s1_v = s_n + 32 * 16, *s1_n = s1_v + 16 * 8;
    const int li = ((wave >> 2) * 8 + (lane >> 3)) * 32 + (wave & 3) * 8 + (lane & 7);
    s_v[li] = out_v; s_n[li] = out_n;
    kf_tile_pyramid<32, 16>(a.pyr, tile_x * 32, tile_y * 16, (int)threadIdx.x, s_v, s_n, s1_v, s1_n, [] { __syncthreads(); });
  }
}

__global__ void __launch_bounds__(RAYCAST_THREADS) k_raycast(RaycastArgs a) {
  extern __shared__ unsigned s_dyn[];
  raycast_tile(a, (int)blockIdx.x, (int)blockIdx.y, s_dyn);
}

// The raycast with the NEXT frame's depth conversion + gate + bilateral filter riding along (kf_prefetch_frame, fused form).  The raycast
// lasts as long as its slowest waves while the average SIMD is busy for less than half of that; the filter of the frame that comes next
// depends on nothing this frame computes.  Workgroups [0, n_rc) are the raycast's tiles (dispatched first, higher wave priority), the rest
// filter two 64x4 tiles each (bilateral_tile.h) and fill the chip as the raycast drains -- no second stream, no events.
// behind == 1: the filter itself has run already (riders of the tracking launch, track.hip); what is left are the integrate tile tables, which
// may only be written now that the fusion pass has cleared them (one gated-depth read per pixel; b.acc.tile null: not wanted), and the frame's
// vertices + normals (out_v non-null), which then need no launch of their own.
// ... and with them levels 1 and 2 of their pyramids (pyr.v1 non-null): a 64x4 tile holds whole 2x2 and 4x4 blocks, so the riders need nothing
// from each other and the tracker's pyramid launch finds the new maps' pyramids done.
struct KfFrontTail { int behind; float4* out_v; float4* out_n; KfCam cam; KfPyrOut pyr; };
#define RIDER_LDS_BYTES (2 * (BIL_TX * BIL_TY + BIL_TX * BIL_TY / 4) * 2 * (int)sizeof(float4))      // per half: the tile's vertices + normals and their level 1
// (six waves per SIMD = three 8-wave workgroups per CU: all 600 ray tiles of a VGA frame resident at once, as in k_raycast -- the filter code would
// otherwise take 90 registers and push the last 88 tiles into a second round)
template <bool FAST>
__global__ void __launch_bounds__(RAYCAST_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) k_raycast_prefetch(RaycastArgs a, KfBilateralArgs b, KfFrontTail ft, int rc_gx, int n_rc, int bil_gx, int bil_tiles) {
  extern __shared__ unsigned s_dyn[];
  if ((int)blockIdx.x < n_rc) {
    __builtin_amdgcn_s_setprio(2);
    raycast_tile(a, (int)blockIdx.x % rc_gx, (int)blockIdx.x / rc_gx, s_dyn);
  } else {
    const int half = (int)(threadIdx.x >> 8), t = ((int)blockIdx.x - n_rc) * 2 + half, tid = (int)(threadIdx.x & 255);
    // a tile index past the last one names a row below the image: its threads touch nothing but still meet the barrier
    const int tt = t < bil_tiles ? t : bil_tiles;
    if (ft.behind) {
      kf_tiles_from_gated(b, tt % bil_gx, tt / bil_gx, tid);
      const int x0 = (tt % bil_gx) * BIL_TX, y0 = (tt / bil_gx) * BIL_TY, x = x0 + (tid & 63), y = y0 + (tid >> 6);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), n = v;
      if (ft.out_v && x < ft.cam.cols && y < ft.cam.rows) kf_vertex_normal_pixel(b.filtered, ft.out_v, ft.out_n, ft.cam, x, y, &v, &n);
      if (ft.pyr.v1) {                                                       // uniform
        float4* s_v = reinterpret_cast<float4*>(s_dyn) + half * (RIDER_LDS_BYTES / 2 / (int)sizeof(float4));
        float4* s_n = s_v + BIL_TX * BIL_TY, *s1_v = s_n + BIL_TX * BIL_TY, *s1_n = s1_v + BIL_TX * BIL_TY / 4;
        s_v[tid] = v; s_n[tid] = n;
        kf_tile_pyramid<BIL_TX, BIL_TY>(ft.pyr, x0, y0, tid, s_v, s_n, s1_v, s1_n, [] { __syncthreads(); });
      }
    } else kf_bilateral_tile<4, FAST>(b, tt % bil_gx, tt / bil_gx, tid, reinterpret_cast<float*>(s_dyn) + half * ((BIL_TX + 8) * (BIL_TY + 8)));
  }
}

static int raycast_launch(kf_ctx* c, int has_color, const kf_mat44* transform, const kf_raycast_params* rp, const kf_camera_params* cam,
                          float near_plane, float far_plane, float* out_t, float4* out_v, float4* out_n, unsigned long long* out_ta = nullptr,
                          unsigned long long* own_ta = nullptr, float* out_spec = nullptr) {
  if (!c || !rp || !cam) return KF_ERR_ARG;
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  if (has_color && (!c->vol.color || !c->raycast_rgb)) return KF_ERR_STATE;
  RaycastArgs a;
  a.vol = c->vol;
  a.cam.cols = (int)cam->cols; a.cam.rows = (int)cam->rows; a.cam.cx = cam->cx; a.cam.cy = cam->cy; a.cam.fx = cam->fx; a.cam.fy = cam->fy;
  if (transform) { for (int i = 0; i < 16; ++i) a.pose_val.m[i] = transform->m[i]; a.pose = nullptr; }
  else a.pose = c->track->pose;
  if (!out_ta && (!out_v || !out_n)) c->model_pyr_ok = 0;             // the model maps' level 0 is rewritten
  memset(&a.pyr, 0, sizeof(a.pyr));
  static int pyr_env = -1;
  if (pyr_env < 0) { const char* e = getenv("KF_RAYCAST_PYRAMID"); pyr_env = e ? atoi(e) : 1; }
  const bool model_pyr = pyr_env && !out_ta && !out_v && !out_n && c->levels == 3;      // the model maps themselves, stock pyramid depth
  if (model_pyr) {
    a.pyr.v1 = c->model_v[1]; a.pyr.n1 = c->model_n[1]; a.pyr.v2 = c->model_v[2]; a.pyr.n2 = c->model_n[2];
    a.pyr.c1 = c->cols >> 1; a.pyr.r1 = c->rows >> 1; a.pyr.c2 = a.pyr.c1 >> 1; a.pyr.r2 = a.pyr.r1 >> 1;
  }
  a.out_v = out_v ? out_v : c->model_v[0]; a.out_n = out_n ? out_n : c->model_n[0]; a.out_rgb = c->raycast_rgb; a.out_t = out_t; a.out_ta = out_ta;
  a.own_ta = own_ta; a.out_spec = (out_ta && own_ta) ? out_spec : nullptr;
  a.inc = rp->ray_increment; a.near_plane = near_plane; a.far_plane = far_plane; a.has_color = has_color;
  { static int em = -1; if (em < 0) em = KF_EXP_ENV("KF_RAYCAST_EXP"); a.exp_mode = em; }
  { static int tb = -1; if (tb < 0) { const char* e = getenv("KF_RAYCAST_BOUNDS"); tb = e ? atoi(e) : 1; } a.tile_bounds = tb; }
  { static int bm = -1; if (bm < 0) { const char* e = getenv("KF_RAYCAST_BOUNDS_MESO"); bm = e ? atoi(e) : 1; } a.bounds_meso = bm; }
  a.shared_grad = rc_shared_grad_for(c->vol);
  a.work = c->count_work ? c->counters : nullptr;
  size_t macro_bytes = (size_t)(c->vol.macro_words + c->vol.super_words) * 4;
  const size_t neg_bytes = kf_negbit_words(c->n_stored_bricks) * 4, meso_bytes = (size_t)c->vol.meso_words * 4;
  if (macro_bytes > RAYCAST_LDS_BYTES) return KF_ERR_STATE;
  static int meso_env = -1;
  if (meso_env < 0) { const char* e = getenv("KF_RAYCAST_MESO"); meso_env = e ? atoi(e) : 1; }
  a.meso_words = (meso_env && macro_bytes + meso_bytes <= RAYCAST_LDS_BYTES) ? c->vol.meso_words : 0;      // (2048^3: 256 KiB, no)
  macro_bytes += (size_t)a.meso_words * 4;                  // from here on: everything in front of the per-brick bits
  a.neg_words = (macro_bytes + neg_bytes <= RAYCAST_LDS_BYTES) ? (int)(neg_bytes / 4) : 0;
  kf_evt_begin(c, KF_STAGE_RAYCAST);
  {
    hipEvent_t ke0 = nullptr, ke1 = nullptr;               // the kernel's own timer rides on its dispatch (kf_evt_attach): the kernel as rocprofv3 sees it
    const dim3 grid(kf_div_up(c->cols, 32), kf_div_up(c->rows, 16));
    const size_t pyr_lds = model_pyr ? (size_t)(32 * 16 + 16 * 8) * 2 * sizeof(float4) : 0;      // the tile's two maps and their level 1
    const size_t lds = macro_bytes + (size_t)a.neg_words * 4 > pyr_lds ? macro_bytes + (size_t)a.neg_words * 4 : pyr_lds;
    const bool timed = kf_evt_attach(c, KF_STAGE_RAYCAST_KERNEL, &ke0, &ke1);
    if (c->fp_pending && c->alt_raw) {
      // kf_prefetch_frame left a note: the next frame's u16 -> f32 + gate + bilateral rides in this launch (k_raycast_prefetch), its vertices /
      // normals follow; the set lands in the alternate buffers and kf_preprocess adopts it when it is asked for exactly that frame
      c->fp_pending = 0;
      const int behind = c->fp_filtered; c->fp_filtered = 0;                 // the filter rode in the tracking launch: the tile tables and the vertices / normals are left
      KfBilateralArgs b; bool fast;
      const bool build_tiles = c->tiles_clear && c->fuse_max_dist > 0.f;      // the fusion pass of this frame has cleared the tables: they can be built for the next depth map
      kf_bilateral_args(c, c->fp_src, nullptr, c->alt_raw, c->alt_trunced, c->alt_filtered, c->fp_params[0], c->fp_params[1], c->fp_params[2], c->fp_params[3],
                        build_tiles, &b, &fast);
      KfFrontTail ft; memset(&ft, 0, sizeof(ft));
      ft.behind = behind;
      if (behind) {
        ft.out_v = c->alt_v0; ft.out_n = c->alt_n0;
        if (c->levels == 3 && c->alt_v12[0]) {
          ft.pyr.v1 = c->alt_v12[0]; ft.pyr.n1 = c->alt_n12[0]; ft.pyr.v2 = c->alt_v12[1]; ft.pyr.n2 = c->alt_n12[1];
          ft.pyr.c1 = c->cols >> 1; ft.pyr.r1 = c->rows >> 1; ft.pyr.c2 = ft.pyr.c1 >> 1; ft.pyr.r2 = ft.pyr.r1 >> 1;
        }
        ft.cam.cols = (int)c->fp_cam.cols; ft.cam.rows = (int)c->fp_cam.rows; ft.cam.cx = c->fp_cam.cx; ft.cam.cy = c->fp_cam.cy; ft.cam.fx = c->fp_cam.fx; ft.cam.fy = c->fp_cam.fy;
      }
      const int bil_gx = kf_div_up(c->cols, BIL_TX), bil_tiles = bil_gx * kf_div_up(c->rows, BIL_TY);
      const int n_rc = (int)(grid.x * grid.y), n_bil = (bil_tiles + 1) / 2;
      const size_t rider_lds = behind ? (size_t)RIDER_LDS_BYTES : 2 * (BIL_TX + 8) * (BIL_TY + 8) * sizeof(float);
      const size_t lds2 = lds > rider_lds ? lds : rider_lds;
      const dim3 g2((unsigned)(n_rc + n_bil));
      if (fast) {
        if (timed) hipExtLaunchKernelGGL(k_raycast_prefetch<true>, g2, dim3(RAYCAST_THREADS), (unsigned)lds2, c->stream, ke0, ke1, 0, a, b, ft, (int)grid.x, n_rc, bil_gx, bil_tiles);
        else hipLaunchKernelGGL(k_raycast_prefetch<true>, g2, dim3(RAYCAST_THREADS), lds2, c->stream, a, b, ft, (int)grid.x, n_rc, bil_gx, bil_tiles);
      } else {
        if (timed) hipExtLaunchKernelGGL(k_raycast_prefetch<false>, g2, dim3(RAYCAST_THREADS), (unsigned)lds2, c->stream, ke0, ke1, 0, a, b, ft, (int)grid.x, n_rc, bil_gx, bil_tiles);
        else hipLaunchKernelGGL(k_raycast_prefetch<false>, g2, dim3(RAYCAST_THREADS), lds2, c->stream, a, b, ft, (int)grid.x, n_rc, bil_gx, bil_tiles);
      }
      if (timed) kf_evt_attached_done(c, KF_STAGE_RAYCAST_KERNEL);
      if (!behind) {
        const int st = kf_launch_vertices_normals(c, c->stream, c->alt_filtered, c->alt_v0, c->alt_n0, &c->fp_cam);
        if (st) return st;
      }
      c->alt_pyr_ok = (behind && ft.pyr.v1) ? 1 : 0;
      c->prefetch_src = c->fp_src; memcpy(c->prefetch_params, c->fp_params, sizeof(c->prefetch_params));
      c->prefetch_cam = c->fp_cam;
      c->prefetch_valid = 1; c->fp_done = 1;
      c->fp_tiles = build_tiles ? 1 : 0; c->fp_tiles_dist = c->fuse_max_dist; c->fp_tiles_min = (build_tiles && b.acc.n) ? 1 : 0;
      if (build_tiles) c->tiles_clear = 0;
    } else if (timed) {
      hipExtLaunchKernelGGL(k_raycast, grid, dim3(RAYCAST_THREADS), (unsigned)lds, c->stream, ke0, ke1, 0, a);
      kf_evt_attached_done(c, KF_STAGE_RAYCAST_KERNEL);
    } else hipLaunchKernelGGL(k_raycast, grid, dim3(RAYCAST_THREADS), lds, c->stream, a);
  }
  kf_evt_end(c, KF_STAGE_RAYCAST);
  if (model_pyr) c->model_pyr_ok = 1;
  return (int)hipGetLastError();
}

extern "C" int kf_raycast_volume(kf_ctx* c, int has_color, const kf_mat44* transform, const kf_raycast_params* rp,
                                 const kf_camera_params* cam, float near_plane, float far_plane) {
  return raycast_launch(c, has_color, transform, rp, cam, near_plane, far_plane, nullptr, nullptr, nullptr);
}

// z-slab variant, MAP FORM (the earlier protocol, kept for per-kernel tests: it evaluates the gradient in the slab that met the crossing and therefore
// drops the rare pixel whose extrapolated vertex leaves that slab's halo -- see kf_raycast_volume_slab_cross below, which SlabPipeline uses):
// this context marches every ray but reports only crossings whose negative sample lies in the voxel layers it owns.  dev_t[pixel] = ray parameter of that crossing (+inf if none), dev_v / dev_n = the vertex / normal it produced
// (zeros when the reference would have given up at that crossing).  The caller reduces over the slabs -- first crossing
// along the ray wins, exactly the reference's sequential march -- and hands the result back with kf_set_model_maps_device.
// The previous sample of the first owned one lies up to x = inc/cell layers outside the owned range and the trilinear +
// gradient taps around a vertex next to it reach ceil(x) + 2 layers: a thinner halo would silently lose crossings at the
// slab faces (those reads fail), so it is refused.  Layers clipped by the volume's own faces do not count.
static int slab_halo_check(const kf_ctx* c, const kf_raycast_params* rp) {
  const int need = (int)ceilf(rp->ray_increment / c->vol.cell) + 2;
  const int lo = c->vol.own_z0 - c->vol.bz0 * KF_BRICK, hi = c->vol.bz1 * KF_BRICK - c->vol.own_z1;
  return ((c->vol.own_z0 > 0 && lo < need) || (c->vol.own_z1 < c->vol.res && hi < need)) ? KF_ERR_ARG : 0;
}
extern "C" int kf_raycast_volume_slab(kf_ctx* c, int has_color, const kf_mat44* transform, const kf_raycast_params* rp,
                                      const kf_camera_params* cam, float near_plane, float far_plane,
                                      float* dev_t, float* dev_v, float* dev_n) {
  if (!c || !rp || !dev_t || !dev_v || !dev_n) return KF_ERR_ARG;
  const int st = slab_halo_check(c, rp);
  if (st) return st;
  return raycast_launch(c, has_color, transform, rp, cam, near_plane, far_plane, dev_t, (float4*)dev_v, (float4*)dev_n);
}

// after the MIN reduction of the crossing parameters over the slabs: keep this context's candidate where it IS the first
// crossing (t == tmin, finite), zero it elsewhere -- the integer SUM reduction that follows then returns the winner's bits
__global__ void __launch_bounds__(256) k_slab_mask(const float* __restrict__ t, const float* __restrict__ tmin, float4* __restrict__ v, float4* __restrict__ n, int npx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npx) return;
  const float ti = t[i];
  if (!(ti == tmin[i] && ti < __builtin_huge_valf())) { const float4 z = make_float4(0.f, 0.f, 0.f, 0.f); v[i] = z; n[i] = z; }
}
extern "C" int kf_slab_mask_candidates(kf_ctx* c, const float* dev_t, const float* dev_tmin, float* dev_v, float* dev_n) {
  if (!c || !dev_t || !dev_tmin || !dev_v || !dev_n) return KF_ERR_ARG;
  const int npx = c->cols * c->rows;
  hipLaunchKernelGGL(k_slab_mask, dim3(kf_div_up(npx, 256)), dim3(256), 0, c->stream, dev_t, dev_tmin, (float4*)dev_v, (float4*)dev_n, npx);
  return (int)hipGetLastError();
}

// ---- the merge SlabPipeline runs: crossings first, normals by the vertex's owner -----------------------------------------------------------------------
// A crossing's vertex is org + dir * alpha with alpha = t - inc * f(t) / (f(t) - f(t - inc)) (raycastingVolume.cu:89-90): an EXTRAPOLATION whenever the two
// interpolated values have the same sign -- an isolated negative voxel at a silhouette does that -- and then the vertex lies anywhere along the ray, far
// outside the halo of the slab that met the crossing (found in round 4: one pixel every few frames on a moving camera; the slab could not read the
// gradient taps and dropped a pixel the single-GPU march keeps).  So the work is split where the data is:
//   1. kf_raycast_volume_slab_cross: every slab marches every ray and reports, per pixel, ONE 64-bit word (crossing parameter << 32 | alpha), what it can
//      decide alone (the two interpolations at the crossing lie within its halo);
//   2. the caller's MIN all-reduce over those words: positive floats order like their bit patterns, so the first crossing along the ray wins and brings its
//      alpha along (alpha 0: the reference gave up at that crossing -- the pixel stays empty, as in the sequential march);
//   3. kf_slab_ray_normals: every rank rebuilds the winners' vertices from the rays (a pure function of pose and camera, which all ranks hold bit for bit);
//      the rank that OWNS the vertex's voxel layer evaluates gradientForPoint (:16-42) -- its taps reach two layers, inside any halo -- and contributes
//      the normal (three words; all-zero bits = none), everybody else zeros; the previous sample's position, whose voxel the function bounds-tests, is found by replaying the chain of additions
//      from t_min (:116), exactly as the march walked it;
//   4. the caller's integer SUM all-reduce (exactly one contributor per pixel: the winner's bits, -0.0 included) and kf_set_model_maps_rays, which writes
//      the model maps and levels 1 and 2 of their pyramids.
// Two collectives and three launches per frame; 8 + 12 bytes per pixel on the wire (round 4: 8 + 16 -- the fourth word only said "valid", which a unit normal says itself).
extern "C" int kf_raycast_volume_slab_cross(kf_ctx* c, const kf_mat44* transform, const kf_raycast_params* rp, const kf_camera_params* cam,
                                            float near_plane, float far_plane, uint64_t* dev_ta) {
  if (!c || !rp || !dev_ta) return KF_ERR_ARG;
  const int st = slab_halo_check(c, rp);
  if (st) return st;
  return raycast_launch(c, 0, transform, rp, cam, near_plane, far_plane, nullptr, nullptr, nullptr, (unsigned long long*)dev_ta);
}
extern "C" int kf_raycast_volume_slab_cross_spec(kf_ctx* c, const kf_mat44* transform, const kf_raycast_params* rp, const kf_camera_params* cam,
                                                 float near_plane, float far_plane, uint64_t* dev_ta, uint64_t* dev_ta_own, float* dev_spec) {
  if (!c || !rp || !dev_ta || !dev_ta_own || !dev_spec) return KF_ERR_ARG;
  const int st = slab_halo_check(c, rp);
  if (st) return st;
  return raycast_launch(c, 0, transform, rp, cam, near_plane, far_plane, nullptr, nullptr, nullptr, (unsigned long long*)dev_ta, (unsigned long long*)dev_ta_own, dev_spec);
}
struct SlabNormalArgs { KfVolume vol; KfCam cam; const float* pose; KfMat pose_val; const unsigned long long* ta; float* cand; float inc, near_plane, far_plane; int shared_grad;
                        const unsigned long long* own_ta; const float* spec; };   // own_ta / spec: kf_raycast_volume_slab_cross_spec's second outputs, or null   // cand: 3 floats per pixel
__global__ void __launch_bounds__(256) k_slab_ray_normals(SlabNormalArgs a) {
  // (a workgroup is a 32x8 pixel tile, a wave an 8x8 patch of it: its 64 vertices stay inside a few bricks -- fewer cache lines per gather instruction)
  const int x = (int)blockIdx.x * 32 + (int)(threadIdx.x >> 6) * 8 + (int)(threadIdx.x & 7), y = (int)blockIdx.y * 8 + (int)((threadIdx.x >> 3) & 7);
  if (x >= a.cam.cols || y >= a.cam.rows) return;
  const KfVolume& v = a.vol;
  const int i = y * a.cam.cols + x;
  const unsigned long long w = a.ta[i];
  const float t_cross = __uint_as_float((unsigned)(w >> 32));
  const unsigned alpha_bits = (unsigned)w;
  float3 out = kf3(0.f, 0.f, 0.f);                                             // all-zero bits: not this rank's vertex, or no gradient (a found gradient is a unit vector)
  if (t_cross < __builtin_huge_valf() && alpha_bits != 0u && a.own_ta && a.own_ta[i] == w) {
    // this context's own crossing won: its march has evaluated the vertex already, if the vertex is this context's (zeros otherwise: the owner's
    // own crossing word differs from the winner's, so the owner takes the branch below)
    out = kf3(a.spec[3 * i], a.spec[3 * i + 1], a.spec[3 * i + 2]);
  } else if (t_cross < __builtin_huge_valf() && alpha_bits != 0u) {
    float3 org, dir, cam_dir;
    rc_pixel_ray(a.cam, a.pose ? a.pose : a.pose_val.m, x, y, org, dir, cam_dir);
    const float3 vtx = kf_add(org, kf_scale(dir, __uint_as_float(alpha_bits)));
    const KfRecip rS = kf_recip(v.size), rcell = kf_recip(v.cell);
    int gz = kf_f2i(kf_div(vtx.z * (float)v.res, rS));                       // the vertex's voxel layer (tsdfVolume.h:50-56), clamped: exactly one owner
    gz = max(0, min(gz, v.res - 1));
    if (gz >= v.own_z0 && gz < v.own_z1) {
      float tmin, tmax;
      rc_ray_interval(v.size, a.near_plane, a.far_plane, org, dir, cam_dir, tmin, tmax);
      float t = tmin, t_prev = tmin;
      if (t < t_cross) kf_ray_advance(t, t_prev, a.inc, t_cross);            // the march's own chain of additions (its closed form, exact: kf_selftest_div mode 12): t ends ON t_cross, t_prev on the sample before it
      const float3 last_pos = kf_add(org, kf_scale(dir, t_prev));
      float3 grad;
      if (gradient_for_point_either<6, RC_SLAB_GRAD_ROUNDS>(a.shared_grad, ((blockIdx.x + blockIdx.y + (threadIdx.x >> 6)) & 1u) != 0u, v, last_pos, vtx, rS, rcell, grad)) out = grad;                  // (the six taps' 48 gathers in one batch: this kernel has the registers)
    }
  }
  a.cand[3 * i] = out.x; a.cand[3 * i + 1] = out.y; a.cand[3 * i + 2] = out.z;
}
static int slab_ray_normals(kf_ctx* c, const kf_mat44* transform, const kf_raycast_params* rp, const kf_camera_params* cam,
                            float near_plane, float far_plane, const uint64_t* dev_ta_min, const uint64_t* dev_ta_own, const float* dev_spec, float* dev_cand) {
  if (!c || !rp || !cam || !dev_ta_min || !dev_cand) return KF_ERR_ARG;
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  SlabNormalArgs a;
  a.vol = c->vol; a.ta = (const unsigned long long*)dev_ta_min; a.cand = dev_cand;
  a.own_ta = (dev_ta_own && dev_spec) ? (const unsigned long long*)dev_ta_own : nullptr; a.spec = dev_spec;
  a.cam.cols = (int)cam->cols; a.cam.rows = (int)cam->rows; a.cam.cx = cam->cx; a.cam.cy = cam->cy; a.cam.fx = cam->fx; a.cam.fy = cam->fy;
  a.inc = rp->ray_increment; a.near_plane = near_plane; a.far_plane = far_plane;
  a.shared_grad = rc_shared_grad_for(c->vol);
  if (transform) { for (int k = 0; k < 16; ++k) a.pose_val.m[k] = transform->m[k]; a.pose = nullptr; }
  else a.pose = c->track->pose;
  hipLaunchKernelGGL(k_slab_ray_normals, dim3(kf_div_up(c->cols, 32), kf_div_up(c->rows, 8)), dim3(256), 0, c->stream, a);
  return (int)hipGetLastError();
}
extern "C" int kf_slab_ray_normals(kf_ctx* c, const kf_mat44* transform, const kf_raycast_params* rp, const kf_camera_params* cam,
                                   float near_plane, float far_plane, const uint64_t* dev_ta_min, float* dev_cand) {
  return slab_ray_normals(c, transform, rp, cam, near_plane, far_plane, dev_ta_min, nullptr, nullptr, dev_cand);
}
extern "C" int kf_slab_ray_normals_spec(kf_ctx* c, const kf_mat44* transform, const kf_raycast_params* rp, const kf_camera_params* cam,
                                        float near_plane, float far_plane, const uint64_t* dev_ta_min, const uint64_t* dev_ta_own, const float* dev_spec, float* dev_cand) {
  if (!dev_ta_own || !dev_spec) return KF_ERR_ARG;
  return slab_ray_normals(c, transform, rp, cam, near_plane, far_plane, dev_ta_min, dev_ta_own, dev_spec, dev_cand);
}
struct SlabUnpackArgs { const unsigned long long* ta; const float* cand; float4* v; float4* n; KfCam cam; const float* pose; KfMat pose_val; KfPyrOut pyr; };   // cand: 3 floats per pixel
// one 32x8 pixel tile per workgroup: the tile's whole 2x2 and 4x4 blocks also give levels 1 and 2 of the model maps' pyramids (kf_tile_pyramid),
// so the tracker that follows finds them done, as after a single-GPU raycast
__global__ void __launch_bounds__(256) k_slab_rays_unpack(SlabUnpackArgs a) {
  __shared__ float4 s_v[32 * 8], s_n[32 * 8], s1_v[16 * 4], s1_n[16 * 4];
  const int x = (int)blockIdx.x * 32 + (int)(threadIdx.x & 31), y = (int)blockIdx.y * 8 + (int)(threadIdx.x >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f), n = v;
  if (x < a.cam.cols && y < a.cam.rows) {
    const int i = y * a.cam.cols + x;
    const float3 cd = kf3(a.cand[3 * i], a.cand[3 * i + 1], a.cand[3 * i + 2]);
    if ((__float_as_uint(cd.x) | __float_as_uint(cd.y) | __float_as_uint(cd.z)) != 0u) {     // the vertex's owner found a gradient (a unit vector: some bit is set): vertex = org + dir * alpha (raycastingVolume.cu:90), w = 1
      float3 org, dir, cam_dir;
      rc_pixel_ray(a.cam, a.pose ? a.pose : a.pose_val.m, x, y, org, dir, cam_dir);
      const float3 vtx = kf_add(org, kf_scale(dir, __uint_as_float((unsigned)a.ta[i])));
      v = make_float4(vtx.x, vtx.y, vtx.z, 1.0f);
      n = make_float4(cd.x, cd.y, cd.z, 0.f);
    }
    a.v[i] = v; a.n[i] = n;
  }
  if (a.pyr.v1) {                                                            // uniform
    s_v[threadIdx.x] = v; s_n[threadIdx.x] = n;
    kf_tile_pyramid<32, 8>(a.pyr, (int)blockIdx.x * 32, (int)blockIdx.y * 8, (int)threadIdx.x, s_v, s_n, s1_v, s1_n, [] { __syncthreads(); });
  }
}
extern "C" int kf_set_model_maps_rays(kf_ctx* c, const kf_mat44* transform, const kf_camera_params* cam, const uint64_t* dev_ta_min, const float* dev_cand) {
  if (!c || !cam || !dev_ta_min || !dev_cand) return KF_ERR_ARG;
  if ((int)cam->cols != c->cols || (int)cam->rows != c->rows) return KF_ERR_ARG;
  SlabUnpackArgs a;
  c->model_pyr_ok = 0;
  memset(&a.pyr, 0, sizeof(a.pyr));
  if (c->levels == 3) {
    a.pyr.v1 = c->model_v[1]; a.pyr.n1 = c->model_n[1]; a.pyr.v2 = c->model_v[2]; a.pyr.n2 = c->model_n[2];
    a.pyr.c1 = c->cols >> 1; a.pyr.r1 = c->rows >> 1; a.pyr.c2 = a.pyr.c1 >> 1; a.pyr.r2 = a.pyr.r1 >> 1;
  }
  a.ta = (const unsigned long long*)dev_ta_min; a.cand = dev_cand; a.v = c->model_v[0]; a.n = c->model_n[0];
  a.cam.cols = (int)cam->cols; a.cam.rows = (int)cam->rows; a.cam.cx = cam->cx; a.cam.cy = cam->cy; a.cam.fx = cam->fx; a.cam.fy = cam->fy;
  if (transform) { for (int k = 0; k < 16; ++k) a.pose_val.m[k] = transform->m[k]; a.pose = nullptr; }
  else a.pose = c->track->pose;                       // the pose the raycast used: nothing moves it between the raycast and this call
  hipLaunchKernelGGL(k_slab_rays_unpack, dim3(kf_div_up(c->cols, 32), kf_div_up(c->rows, 8)), dim3(256), 0, c->stream, a);
  if (a.pyr.v1) c->model_pyr_ok = 1;
  return (int)hipGetLastError();
}

extern "C" int kf_set_model_maps_device(kf_ctx* c, const float* dev_v, const float* dev_n) {
  if (!c || !dev_v || !dev_n) return KF_ERR_ARG;
  const size_t bytes = (size_t)c->cols * c->rows * sizeof(float4);
  c->model_pyr_ok = 0;
  KF_CHECK(hipMemcpyAsync(c->model_v[0], dev_v, bytes, hipMemcpyDeviceToDevice, c->stream));
  KF_CHECK(hipMemcpyAsync(c->model_n[0], dev_n, bytes, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
