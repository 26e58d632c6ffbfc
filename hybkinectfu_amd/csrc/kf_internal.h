// kf_internal.h -- context layout and device-side arithmetic shared by the gfx950 kernels.
//
// All device arithmetic that feeds a discrete decision (pixel rounding, update predicate, cube index) is written one
// fp32 operation per reference source operation and the library is built with -ffp-contract=off, so the result bits
// equal the reference's C++ semantics (see DESIGN.md "Parity").  Citations are relative to /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/hybkf.h"

#define KF_BRICK 8                 // voxels per brick edge
#define KF_BRICK_VOX 512           // voxels per brick: 4 KiB of (tsdf, weight) pairs, contiguous in HBM
#define KF_FLAG_OBSERVED 1u        // some voxel of the brick has weight > 0
#define KF_FLAG_HASNEG 2u          // some voxel of the brick has (or once had) tsdf < 0
// Deferred free-space weights (integrate.hip, "DEFER"): one 16-bit word per QUARTER brick (the 128 voxels of z layers 2q, 2q+1 -- what one
// wave of the fusion pass owns), KfVolume::pend[slot] holding the four of a brick.  0: the voxels are authoritative.  p >= 1: all 128 voxels
// hold tsdf 1.0f and a stored weight w >= 1, and k = p - 1 whole-quarter free-space observations are pending: the voxel's TRUE weight is
// fminf(w + k, max_weight) -- k applications of tsdfVolume.h:65 on small integers, while tsdfVolume.h:66 leaves tsdf at (1 * w + 1) / (w + 1) = 1.
// KF_PEND_SAT: every true weight has reached max_weight (k >= max_weight - 1 and w >= 1): any further free-space observation is the identity.
#define KF_PEND_SAT 0xFFFFu
#define KF_PEND_MAX_WEIGHT 65000.f // deferral needs 1 <= max_weight <= this (w + k stays an exact integer sum; KF_PEND_SAT - 1 pending steps saturate any weight)
#define KF_MACRO 32                // voxels per macro-cell edge (raycast empty-space skipping)
#ifndef KF_SUPER_SHIFT
#define KF_SUPER_SHIFT 2           // a super cell is (1 << KF_SUPER_SHIFT)^3 macro cells: 128^3 voxels, the raycast's coarsest skip level
#endif
#define KF_MAX_LEVELS 3
#define KF_ICP_MAX_WG 2048             // workgroups of one ICP / SDF step launch (1536 pixels each: up to 3.1 M pixels)
#define KF_ICP_LOOP_MAX_WG 1024        // publishers of one step of the persistent ICP loops (k_icp_loop: its workgroups, bounded by the CU count; k_icp_loop_batched: the
                                       // workgroups of the dealing, 800 at 1280x960)
#define KF_ICP_LOOP_STEPS 32           // Gauss-Newton steps of one persistent ICP launch (stock: 4 + 5 + 10)

// Result-changing timing experiments (KF_INTEGRATE_EXP / KF_ICP_EXP / KF_RAYCAST_EXP: skipped stores, partial queues, clock
// stamps written into the outputs) exist only in the variant built with -DKF_EXPERIMENTS (`make experiments` ->
// libhybkf_exp.so, loaded by tools/ via KF_LIB).  In the product library the mode is the constant 0: the branches fold away
// and the environment is never consulted for them.
#ifdef KF_EXPERIMENTS
#include <stdlib.h>
static inline int kf_exp_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#define KF_EXP_MODE(a) ((a).exp_mode)
#define KF_EXP_ENV(name) kf_exp_env(name)
#else
#define KF_EXP_MODE(a) 0
#define KF_EXP_ENV(name) 0
#endif

#define KF_PINNED_STALL_WORD 2048
enum { KF_ERR_ARG = 1001, KF_ERR_STATE = 1002, KF_ERR_ALLOC = 1003 };

// Dense TSDF volume, bricked: voxel (x,y,z) lives in brick (x>>3, y>>3, z>>3) at offset ((z&7)<<6 | (y&7)<<3 | (x&7)).
// Only brick layers [bz0, bz1) are stored (z-slab + halo).  Reference: tsdfVolume.h:14-19,57-60 (12-byte AoS, linear).
struct KfVolume {
  float2* tw;            // (tsdf, weight) per voxel
  uchar4* color;         // (c0, c1, c2, unused) per voxel; null when the context has no colour
  uint8_t* flags;        // per brick KF_FLAG_*
  unsigned* macrobits;   // packed bit tables over the WHOLE volume, [macro_words | super_words | meso_words]: bit = some voxel of the 32^3-voxel macro cell /
                         // of the 128^3-voxel super cell has (had) tsdf < 0; set with atomicOr by whoever finds a brick's first negative voxel
  unsigned* negbits;     // one bit per STORED brick slot: the brick's KF_FLAG_HASNEG, packed so the raycast can keep the table in LDS
  unsigned long long* pend;   // per STORED brick slot: four 16-bit deferred-weight words (quarter q at bits 16q), see KF_PEND_SAT above
  int nm;                // macro cells per axis = ceil(res / 32)
  int ns;                // super cells per axis = ceil(nm / 4)
  int macro_words, super_words;   // words of the two tables (each padded to whole 16-byte vectors)
  int nq, meso_words;             // a third table behind them (round 5): one bit per MESO cell = 2 x 2 x 2 bricks (16^3 voxels), nq cells per axis; the raycast keeps it in LDS
                                  // when it fits (1024^3: 32 KiB, where the per-brick bits -- 256 KiB -- do not) and walks two bricks at a time through empty meso cells
  int res;               // voxels per axis
  int nb;                // bricks per axis
  int bz0, bz1;          // stored brick layers
  int own_z0, own_z1;    // owned voxel layers (subset of the stored ones)
  float size;            // metres
  float cell;            // size / (float)res, the reference's fp32 quotient (tsdfVolume.h:44-46)
  float max_weight;
};

// (integrate.hip) does this context defer whole-quarter free-space weight updates (max_weight in range, KF_INTEGRATE_SAT != 0)?
bool kf_defer_enabled(const struct kf_ctx* c);
// (integrate.hip) apply every pending count to the voxels (the counts drop to "nothing pending"; the states stay valid)
int kf_flush_pending(struct kf_ctx* c);

// the true weight of a voxel whose quarter brick carries the deferred-weight word p (KF_PEND_SAT - 1 pending steps saturate every w >= 1)
__host__ __device__ static inline float kf_pend_weight(float w, unsigned p, float max_weight) {
  return p >= 2u ? fminf(w + (float)(p - 1u), max_weight) : w;
}

// Device-resident tracker state (what CameraPoseFinder keeps in _pose plus the Gauss-Newton scratch).
struct KfTrackState {
  float pose[16];        // CameraPoseFinder::_pose
  float cur[2][16];      // cur_transform of the running estimate, double-buffered across Gauss-Newton steps
  float last_inv[16];    // _pose.getInverse() at the start of the frame
  float reduced[27];     // rigid_align_buf_reduced
  int   status;          // KF_TRACK_*
  int   tracked;         // result of the last findCameraPose
  int   iterations;
  int   converged;       // SDF tracker: |x| < 1e-3 reached
  unsigned arrive;       // arrival counter of the persistent ICP loop's grid barrier (monotonic within a frame)
  unsigned rescue_tag;   // persistent ICP loop: tag_base of the last launch that timed out waiting and was finished by ONE workgroup alone (track.hip)
  int      rescued;      // the committed verdict of the last tracking call comes from that solo finish (kf_track_result::launch_form 3)
  unsigned pad_[1];
  float pose_inv[16];    // pose.getInverse(), written wherever pose is committed: integrate reads it (integrateVolume.cu:84)
  // who ends a persistent tracking launch (track.hip: k_sdf_loop): tag_base | 1 the launch's own workgroups (workgroup 0 commits, each runs its share of the
  // tail), tag_base | 2 ONE workgroup that claimed the launch after a time-out (it commits and runs every share); any other upper bits: an older launch's
  // word, i.e. open.  Set by compare-and-swap, so exactly one of the two parties ends a launch.
  unsigned commit_word;
};

// software grid barrier of the persistent ICP loop: every word on its own 128-byte line
struct KfPaddedCounter { unsigned v; unsigned pad_[31]; };
struct KfGridBarrier { KfPaddedCounter group[8]; KfPaddedCounter top; KfPaddedCounter gen; };

struct KfCounters {
  unsigned long long n_upd;       // (unused)
  unsigned n_active_bricks;       // (unused: see n_active)
  unsigned n_triangles;           // MarchingcubeData::_ptr_num_triangles
  unsigned long long weight_gt0;
  unsigned scan_total;
  unsigned frames_lost;           // integrate calls skipped because tracking failed (device-resident pose path)
  unsigned long long n_upd_total; // running sum of n_upd over integrate calls (reset with kf_reset_volume)
  unsigned frames_fused;
  unsigned pad_;
  // update counts are sharded over 64 cache lines: one address takes ~11 ns per atomic (MI355X_MICROARCH.md 'dequeue'),
  // so thousands of workgroups adding to ONE word would cost more than the fusion itself
  // Per-integrate counters are double-buffered by the call's parity p (kf_ctx::int_parity): the cull adds to n_active[p], the
  // fusion pass reads it and adds to upd_shard[p], and ITS workgroup 0 retires the other parity's set (folds upd_shard[1-p] into
  // upd_total_shard, zeroes n_active[1-p]) -- nobody else touches that set during the launch, so no extra kernel is needed.
  unsigned n_active[2];                      // bricks queued by the cull
  unsigned pad2_[2];
  unsigned long long upd_shard[2][64 * 16];  // this integrate's update count, slot s at [s*16]
  unsigned long long upd_total_shard[64];    // folded sum of the earlier integrates
  // work counters of the measurement passes (kf_stage_timers bit 16 turns them on, kf_read_work_counters reads them): sharded like
  // the update counts, one 128-byte line per shard
  unsigned long long rc_steps[64 * 16];      // raycast: samples the reference's march takes (per ray: up to its first crossing or t_max)
  unsigned long long rc_hits[64 * 16];       // raycast: rays whose crossing was evaluated (trilinear + gradient taps)
  unsigned long long mc_blocks[64 * 16];     // marching cubes: 4-KiB bricks the extraction reads (those with a negative voxel in their 3x3x3 brick neighbourhood)
  // voxels of the OWNED layers whose weight went from 0 to > 0 since the count was last based (kf_ctx::wgt0_base): every fusion kernel adds its own,
  // sharded like the update counts -- kf_get_volume_stats then needs no sweep of the volume (the reference prints the count every frame: integrateVolume.cu:91-94)
  unsigned long long wgt0_shard[64 * 16];
};
// this wave's 0 -> > 0 weight transitions for one pair per lane (w0 / w1: the weights the update starts from); `count` is wave-uniform: owned layers, max_weight > 0
__device__ __forceinline__ unsigned kf_new_voxels(bool count, bool u0, float w0, bool u1, float w1) {
#ifdef KF_NO_OBSERVED_COUNT                      // A/B variant only (tools/build_variant.sh): what the running count costs the fusion kernels
  return 0u;
#endif
  const unsigned long long a = __ballot(u0 && w0 == 0.f), b = __ballot(u1 && w1 == 0.f);
  return count ? (unsigned)__popcll(a) + (unsigned)__popcll(b) : 0u;
}

#define KF_UP_SLOTS 3                  // host-upload slots: the current frame and up to two staged ahead of it
struct kf_ctx {
  kf_config cfg;
  hipStream_t stream;                 // where work is enqueued (private, or adopted through kf_set_stream)
  hipStream_t own_stream;             // the private stream once another one has been adopted
  int cols, rows;
  int levels;
  int registered;                     // counted in the per-device live-context registry
  int num_cus;                        // compute units of the device (co-residency bound of the persistent ICP loop)
  int lvl_cols[KF_MAX_LEVELS], lvl_rows[KF_MAX_LEVELS];
  // frame maps (CudaDeviceDataMan.h:56-67)
  // host uploads (kf_upload_depth_mm) are double-buffered: pinned host staging -> DMA on a copy stream -> device buffer, so
  // frame k+1 crosses PCIe while frame k is computed; events order the copy after the last reader of the buffer it reuses
  uint16_t* up_host[KF_UP_SLOTS]; uint16_t* up_dev[KF_UP_SLOTS]; hipStream_t up_stream; hipEvent_t up_copied[KF_UP_SLOTS], up_consumed[KF_UP_SLOTS];
  int up_next, up_used[KF_UP_SLOTS], pending_slot;   // pending_slot: which up_dev[] pending_mm points at (-1: a caller-owned device frame)
  int up_unwaited[KF_UP_SLOTS];       // the context's stream has not been made to wait for this slot's copy yet (frames staged ahead)
  int staged[2], n_staged;            // kf_upload_depth_mm_next: the slots that hold the frames AFTER the current one, oldest first
  const uint16_t* pending_mm;         // device u16 frame whose conversion is deferred into the fused preprocess kernel
  float* raw_depth; float* trunced_depth; float* filtered_depth;
  uchar4* raw_rgb; uchar4* raycast_rgb;   // stored 4 bytes/pixel on the device
  unsigned char* rgb_staging;             // kf_upload_rgb: the host's 3-byte pixels before they are widened (allocated on first use)
  // next-frame prefetch (kf_prefetch_frame): a second set of the per-frame preprocess outputs, filled on a side stream while the
  // current frame is tracked (the persistent ICP loop leaves ~100 CUs idle); kf_preprocess swaps the sets when it is asked for
  // exactly that frame with exactly those parameters.  Allocated on first use.
  float* alt_raw; float* alt_trunced; float* alt_filtered; float4* alt_v0; float4* alt_n0;
  float4* alt_v12[2]; float4* alt_n12[2];   // levels 1 and 2 of the alternate set's vertex / normal pyramids (written by the raycast launch's riders)
  // levels 1.. of the new / model map pyramids describe their level 0 (whoever writes a level 0 without its pyramid clears the flag; the
  // tracker's pyramid launch builds what is missing).  alt_pyr_ok: the same for the alternate (prefetched) set, adopted with it.
  int new_pyr_ok, model_pyr_ok, alt_pyr_ok;
  hipStream_t side_stream; hipEvent_t ev_preprocessed, ev_prefetched;
  const uint16_t* prefetch_src; float prefetch_params[4]; int prefetch_valid, prefetch_in_use;
  kf_camera_params prefetch_cam;      // the intrinsics the prefetched vertices / normals / pyramids were built with: kf_preprocess adopts the set only for the same camera
  // fused form of the prefetch (default): kf_prefetch_frame only records the request; the next kf_raycast_volume launches the raycast with the
  // next frame's gate + bilateral filter riding along (k_raycast_prefetch) and the vertices / normals launch behind it, on the context's own stream
  int fp_pending;                     // a request waits for the next raycast
  int fp_filtered;                    // ... and its gate + bilateral filter has already run as riders of the tracking launch (track.hip)
  const uint16_t* fp_src; float fp_params[4]; kf_camera_params fp_cam;
  int fp_done;                        // the prefetched set was produced on the context's stream (no event to wait for)
  int fp_tiles; float fp_tiles_dist; int fp_tiles_min;   // the integrate tile tables were built for the prefetched depth map (distance; minima too)
  float4* new_v[KF_MAX_LEVELS]; float4* new_n[KF_MAX_LEVELS];
  float4* model_v[KF_MAX_LEVELS]; float4* model_n[KF_MAX_LEVELS];
  float* icp_partials;                // KF_ICP_MAX_WG x 32 floats
  unsigned long long* icp_loop_slots; // persistent ICP loop: KF_ICP_LOOP_STEPS x KF_ICP_LOOP_MAX_WG x 32 tagged partial sums
  unsigned icp_loop_seq;              // host-side launch counter of that loop (x 64): tags never repeat across launches
  hipEvent_t ev_track; int track_requested;   // kf_request_track_result / kf_wait_track_result
  // the persistent ICP loop after a stall (KF_TRACK_STALLED: some of its workgroups were not resident -- a foreign process on the GPU):
  // `persistent_backoff` frames track with one launch per step, then the loop is tried again; every further stall doubles the wait
  int persistent_backoff, persistent_backoff_len;
  int loop_clean_frames;              // loop launches since the last stall was noted (1024 in a row forget the back-off history)
  int inject_stall;                   // kf_inject_track_stall: loop launches that still get a workgroup playing dead
  int loop_refused;                   // the device cannot hold the loop's workgroups at once (occupancy check / cooperative launch refused): never tried again
  int loop_occupancy;                 // workgroups of k_icp_loop one CU can hold (0: not asked yet)
  int loop_occupancy_batched;         // the same for k_icp_loop_batched
  int sdf_loop_occupancy;             // the same for k_sdf_loop (the addressing variant this volume uses)
  unsigned long long wgt0_base;       // observed voxels (weight > 0, owned layers) when the running count KfCounters::wgt0_shard was last zeroed
  int wgt0_valid;                     // base + shards IS the count (0: something else wrote weights -- upload, resize, a fusion launch that did not count -- : the next kf_get_volume_stats sweeps and re-bases)
  int wgt0_tracking;                  // the fusion launches use their COUNT instantiations: switched on when kf_get_volume_stats is asked twice within 8 fused frames, off after 64 frames without a question
  int wgt0_frames_unasked;            // fusion launches since the last kf_get_volume_stats
  int wgt0_asked_before;              // kf_get_volume_stats has been called at least once
  int last_track_form;                // kf_track_result::launch_form of the last tracking call
  KfTrackState* track;                // device
  KfCounters* counters;               // device
  KfGridBarrier* grid_barrier;        // device
  float* scratch_mats;                // device: 8 x 16 floats for host-supplied transforms
  KfVolume vol;
  size_t n_stored_vox, n_stored_bricks;
  unsigned* active_bricks;            // device: brick ids queued by the cull kernel
  float* tile_max_depth;              // device: per 8x8- and 16x16-pixel tile max of the depth gated by the integration distance
  int n_tile_floats;                  // entries of both tables together
  // The tables are normally built by the fused preprocess kernel (atomic max into a CLEARED table) for the integration distance
  // the last kf_integrate_volume used; integrate falls back to k_integrate_prepare when they do not describe the current trunced
  // depth map (serials differ) or another distance is asked for.  The fusion pass clears them again once the cull has read them.
  unsigned long long tile_min_serial;   // trunc_serial the tile MINIMA were built for (only built while saturation can exist)
  unsigned long long trunc_serial, tile_serial;   // bumped by every writer of trunced_depth / copied when the tables are built
  float tile_built_dist, fuse_max_dist;
  int tiles_clear, int_parity, last_parity;
  // The cull as the tail of the tracking launch (track.hip / cull.h): kf_integrate_volume(transform == NULL) leaves its parameters behind as a hint,
  // the next kf_icp_track that runs the persistent loop arms the tail with them, the next kf_integrate_volume consumes it (same parameters, same depth
  // map, same slab) or undoes it (kf_tail_cull_discard: the queue counter of that parity back to zero) and culls in a launch of its own.
  struct { int valid; float sdf_trunc, max_dist; kf_camera_params dcam; } cull_hint;
  struct { int armed, parity, bz0, bz1; float sdf_trunc, max_dist; kf_camera_params dcam; unsigned long long trunc_serial; unsigned consumed, undone; } tail_cull;
  kf_triangle* triangles; uint32_t max_triangles;
  unsigned* mc_block_counts; size_t mc_blocks_cap;
  unsigned* mc_list; unsigned* mc_nbr_bits; unsigned* mc_partials;   // extraction scratch, allocated by the first kf_marching_cubes
  unsigned short* mc_codes; unsigned char* mc_surv; unsigned* mc_block_bits; uint2* mc_recs; unsigned* mc_d1_list;   // voxel classes, sieve bits, cell records, brick list (mcubes.hip), same scratch
  unsigned long long* layer_work; int layer_work_frames;   // per-brick-layer update counts of the next `layer_work_frames` integrate calls (kf_count_layer_work)
  int defer_override;            // kf_set_defer: -1 follow the environment (default), 0 never defer, 1 defer
  int pend_live;                 // a DEFER fusion pass has run since the volume was last reset / uploaded / flushed: deferred-weight words may be set
  unsigned vol_flags_serial, mc_zero_serial;   // bumped when brick flags may have been CLEARED (reset, upload) / the serial the class tables were last zeroed for
  void* host_pinned;                  // small pinned staging buffer (4 KiB); byte KF_PINNED_STALL_WORD: the ICP loop's stall word
  // per-stage hipEvent timers (KF_STAGE_*): bit s of timers_enabled turns stage s on
  int timers_enabled;
  int count_work;                     // kf_stage_timers bit 16: the raycast / marching-cubes work counters are maintained
  unsigned timers_period;             // time every timers_period-th interval of a stage (>= 1)
  unsigned ev_seen[8]; int ev_open[8];
  hipEvent_t ev[8][2][64];            // [stage][begin/end][ring]
  int ev_n[8];                        // pairs recorded and not yet folded
  double ev_ms[8]; unsigned ev_count[8];
};

enum { KF_STAGE_UPLOAD = 0, KF_STAGE_PREPROCESS = 1, KF_STAGE_TRACK = 2, KF_STAGE_INTEGRATE = 3, KF_STAGE_RAYCAST = 4,
       KF_STAGE_INTEGRATE_KERNEL = 5, KF_STAGE_MCUBES = 6, KF_STAGE_RAYCAST_KERNEL = 7 };
void kf_evt_begin(kf_ctx* c, int stage);
void kf_evt_end(kf_ctx* c, int stage);
bool kf_evt_attach(kf_ctx* c, int stage, hipEvent_t* e0, hipEvent_t* e1);     // events stamped by the kernel's own dispatch (hipExtLaunchKernelGGL)
void kf_evt_attached_done(kf_ctx* c, int stage);

#define KF_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

static inline int kf_div_up(int a, int b) { return (a + b - 1) / b; }
// words of a packed table of n bits, padded to whole 16-byte vectors
static inline int kf_bit_words(size_t n_bits) { return (int)((((n_bits + 31) >> 5) + 3) & ~(size_t)3); }
// words of KfVolume::negbits, padded to whole 16-byte vectors (the raycast copies the table with uint4 loads)
static inline size_t kf_negbit_words(size_t n_bricks) { return ((n_bricks + 127) / 128) * 4; }

// ------------------------------------------------------------------------------------------------------------------
// device arithmetic
// ------------------------------------------------------------------------------------------------------------------
struct KfMat { float m[16]; };
struct KfCam { int cols, rows; float cx, cy, fx, fy; };

__device__ __forceinline__ float3 kf3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 kf_sub(float3 a, float3 b) { return kf3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 kf_add(float3 a, float3 b) { return kf3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 kf_scale(float3 a, float s) { return kf3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float kf_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 kf_cross(float3 a, float3 b) {
  return kf3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float kf_norm(float3 v) { return sqrtf(kf_dot(v, v)); }
// cuda_declar.h:89-94 -- reciprocal taken in double, narrowed, then an fp32 multiply
__device__ __forceinline__ float3 kf_normalize(float3 v) {
  float len = kf_norm(v);
  if ((double)len < 1e-8) return kf3(0.f, 0.f, 0.f);
  float r = (float)(1.0 / (double)len);
  return kf_scale(v, r);
}
__device__ __forceinline__ bool kf_is_zero4(float4 v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f; }

// Mat.h:230-238: row * vector, summed left to right
__device__ __forceinline__ float4 kf_mat_vec(const float* m, float4 v) {
  return make_float4(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w,
                     m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * v.w,
                     m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * v.w,
                     m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w);
}
__device__ __forceinline__ float3 kf_mat_point3(const float* m, float x, float y, float z, float w) {
  return kf3(m[0] * x + m[1] * y + m[2] * z + m[3] * w,
             m[4] * x + m[5] * y + m[6] * z + m[7] * w,
             m[8] * x + m[9] * y + m[10] * z + m[11] * w);
}

// two fp32 values carried as one operand of gfx950's packed instructions (v_pk_mul / add / fma_f32: one issue slot for both)
typedef float kf_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ kf_f2 f2_splat(float a) { kf_f2 r = {a, a}; return r; }
__device__ __forceinline__ kf_f2 f2_fma(kf_f2 a, kf_f2 b, kf_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// identity the compiler cannot see through: the value becomes lane-varying as far as it knows (no instruction is emitted)
__device__ __forceinline__ unsigned kf_opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }

// (int) of a double / float the way the reference's CUDA path converts (cvt.rzi.s32): truncation, saturating, NaN -> 0 -- which is exactly
// what gfx950's v_cvt_i32_f64 / v_cvt_i32_f32 do, so the conversion IS that one instruction.  (C++ leaves the out-of-range cast undefined,
// so it cannot be written as a cast; spelled out with comparisons -- kf_to_int_spelled / kf_f2i_spelled, kept for the self-test -- it costs
// three branches and ten instructions per coordinate, a fifth of a raycast march trip.)  kf_selftest_div modes 10 / 11 compare the two
// forms on the GPU over random bit patterns and every special value (tests/test_gpu_parity.py::test_exact_division_helper).
__device__ __forceinline__ int kf_to_int_spelled(double v) {
  if (v != v) return 0;
  if (v >= 2147483648.0) return 2147483647;
  if (v <= -2147483649.0) return (int)0x80000000;
  return (int)v;
}
__device__ __forceinline__ int kf_f2i_spelled(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.f) return 2147483647;
  if (v <= -2147483648.f) return (int)0x80000000;
  return (int)v;
}
__device__ __forceinline__ int kf_to_int(double v) { int r; asm("v_cvt_i32_f64_e32 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ int kf_f2i(float v) { int r; asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(v)); return r; }

// Correctly rounded fp32 quotient a / b for operands in the normal range: the very FMA sequence hipcc emits for `a / b`
// (v_rcp_f32, two reciprocal refinements, quotient, two residual corrections; LLVM legalizeFDIV32) minus the v_div_scale /
// v_div_fixup steps that only act when an exponent is within 2^32 of the fp32 limits.  Splitting it lets a divisor that is
// shared by several quotients (the voxel's depth pf.z, the truncation distance, the volume size) pay for its reciprocal
// once.  kf_selftest_div compares it with `/` on the GPU (tests/test_gpu_parity.py::test_exact_division_helper).
struct KfRecip { float den, r; };
__device__ __forceinline__ KfRecip kf_recip(float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float e0 = __builtin_fmaf(-b, r0, 1.0f);
  KfRecip k; k.den = b; k.r = __builtin_fmaf(e0, r0, r0);
  return k;
}
__device__ __forceinline__ float kf_div(float a, const KfRecip& k) {
  const float q0 = a * k.r;
  const float e1 = __builtin_fmaf(-k.den, q0, a);
  const float q1 = __builtin_fmaf(e1, k.r, q0);
  const float e2 = __builtin_fmaf(-k.den, q1, a);
  return __builtin_fmaf(e2, k.r, q1);
}
// `(int)(p + 0.5)` of DepthCamera.h:42 (double literal) for callers that only accept results >= 1: for p >= 0.5 the double sum
// is exact and its truncation equals floor(p) + (frac(p) >= 0.5), all exact in fp32; every p < 0.5, NaN or huge value maps to a
// result the caller rejects in both formulations (0 here, <= 0 there; a huge p saturates to INT_MAX in both).
__device__ __forceinline__ int kf_round_px(float p) {
  if (!(p >= 0.5f)) return 0;
  const float t = floorf(p);
  return (int)t + ((p - t) >= 0.5f ? 1 : 0);
}

// DepthCamera.h:19-29
__device__ __forceinline__ float3 kf_depth_to_skeleton(unsigned ux, unsigned uy, float depth, const KfCam& c) {
  float vx = depth * ((float)ux - c.cx) / c.fx;
  float vy = depth * ((float)uy - c.cy) / c.fy;
  return kf3(vx, vy, depth);
}
// DepthCamera.h:30-43: `(int)(p + 0.5)` with a double literal
__device__ __forceinline__ int2 kf_project(float3 v, const KfCam& c) {
  float px = v.x * c.fx / v.z + c.cx;
  float py = v.y * c.fy / v.z + c.cy;
  return make_int2(kf_to_int((double)px + 0.5), kf_to_int((double)py + 0.5));
}

// ---- volume addressing -------------------------------------------------------------------------------------------
// (24-bit multiplies: v_mad_u32_u24 is a full-rate instruction, a 32- or 64-bit integer multiply is not, and the gather-bound kernels form
// dozens of these per pixel.  Exact while bricks per axis <= 1024 -- kf_create refuses more --: every factor is below 2^24 and every
// product below 2^32.  A layer below the stored range gives a meaningless slot, as the 64-bit form did: callers test kf_z_stored first.)
// The ray parameter's walk `do { t_prev = t; t += inc; } while (t < t_exit)` (raycastingVolume.cu:116 repeats `ray_current += fRayIncrement`;
// every sample parameter is the result of that chain of fp32 additions) without running the chain: inside one binade [2^e, 2^(e+1)) every t is
// a multiple of u = ulp(t), so RN(t + inc) = t + s with s = inc rounded to the u-grid -- the SAME s at every step unless inc lies exactly half-way
// between two grid points (a tie, resolved by the parity of t: detected, then the plain chain runs).  k steps therefore land on t + k*s exactly,
// which is representable (a multiple of u below 2^(e+1)), so one fused multiply-add produces it without error.  The jump stops short of
// min(t_exit, 2^(e+1)); the step across either is a plain addition again (RN on the coarser grid above 2^(e+1) is whatever the hardware says).
// Exact (kf_selftest_div mode 12: 2^23 random walks incl. binade crossings and ties) and NOT used by default: a ray of 3 m takes ~85 additions
// at the stock increment, but spread over ~10 walks of three instructions per step -- the chain is cheaper than the 35 instructions of this form
// (raycast 75.9 vs 79.4 us at C2, 110.9 vs 115.3 us at C4, one box).  It pays when the increment is small against the cells (-DKF_RAY_ADVANCE_CLOSED).
__device__ __forceinline__ void kf_ray_advance_plain(float& t, float& t_prev, float inc, float t_exit) {
  do { t_prev = t; t += inc; } while (t < t_exit);
}
__device__ __forceinline__ void kf_ray_advance(float& t, float& t_prev, float inc, float t_exit) {
  t_prev = t; t += inc;
  if (t < t_exit) { t_prev = t; t += inc; }                      // short walks (a brick is two or three steps) never reach the closed form
  while (t < t_exit) {
    const float t1 = t + inc;
    if (!(t1 < t_exit)) { t_prev = t; t = t1; break; }
    const unsigned tb = __float_as_uint(t), eb = tb & 0x7F800000u;
    const float hi = __uint_as_float(eb + 0x00800000u);          // 2^(e+1)
    const float u = __uint_as_float(eb - (23u << 23));           // ulp(t)
    const float s = t1 - t;                                      // exact while t1 stays in the binade (Sterbenz)
    const float r = inc - s;                                     // exact: what the rounding dropped, |r| <= u/2
    const float lim = fminf(t_exit, hi);
    const bool closed = (tb - 0x0C800000u) < 0x72000000u         // t positive, normal, ulp normal, below 2^126
                        && t1 < hi && s > 0.f && fabsf(r) * 2.f != u;
    if (!closed) { t_prev = t; t = t1; continue; }               // one plain step: binade crossing, tie, degenerate increment
    // the largest m >= 1 with t + m*s < lim (t1 = t + s < lim already): estimate, then settle it with exact tests
    float m = fmaxf(floorf((lim - t) * __builtin_amdgcn_rcpf(s)), 1.f);
    while (__builtin_fmaf(m, s, t) >= lim) m -= 1.f;
    while (__builtin_fmaf(m + 1.f, s, t) < lim) m += 1.f;
    t_prev = __builtin_fmaf(m - 1.f, s, t);
    t = __builtin_fmaf(m, s, t);                                 // < lim <= t_exit: the loop goes on with a plain step
  }
}

// brick (bx, by, bz) of the whole volume has found its first negative voxel: its macro cell and super cell are no longer empty
__device__ __forceinline__ void kf_mark_macro(const KfVolume& v, int bx, int by, int bz) {
  const int mx = bx >> 2, my = by >> 2, mz = bz >> 2;                                         // 4 bricks per macro edge
  const unsigned mi = (unsigned)((mz * v.nm + my) * v.nm + mx);
  atomicOr(&v.macrobits[mi >> 5], 1u << (mi & 31u));
  const unsigned si = (unsigned)((((mz >> KF_SUPER_SHIFT) * v.ns) + (my >> KF_SUPER_SHIFT)) * v.ns + (mx >> KF_SUPER_SHIFT));
  atomicOr(&v.macrobits[v.macro_words + (si >> 5)], 1u << (si & 31u));
  const unsigned qi = (unsigned)((((bz >> 1) * v.nq) + (by >> 1)) * v.nq + (bx >> 1));       // the meso cell: 2 bricks per edge
  atomicOr(&v.macrobits[v.macro_words + v.super_words + (qi >> 5)], 1u << (qi & 31u));
}
__host__ __device__ static inline size_t kf_skip_table_words(const KfVolume& v) { return (size_t)v.macro_words + (size_t)v.super_words + (size_t)v.meso_words; }
__device__ __forceinline__ unsigned kf_brick_slot(const KfVolume& v, int bx, int by, int bz) {
  return __umul24(__umul24((unsigned)(bz - v.bz0), (unsigned)v.nb) + (unsigned)by, (unsigned)v.nb) + (unsigned)bx;
}
__device__ __forceinline__ size_t kf_vox_index(const KfVolume& v, int x, int y, int z) {
  return (size_t)kf_brick_slot(v, x >> 3, y >> 3, z >> 3) * KF_BRICK_VOX + (size_t)(((z & 7) << 6) | ((y & 7) << 3) | (x & 7));
}
__device__ __forceinline__ bool kf_z_stored(const KfVolume& v, int z) { return z >= v.bz0 * KF_BRICK && z < v.bz1 * KF_BRICK; }

// tsdfVolume.h:50-56
__device__ __forceinline__ int3 kf_world_to_voxel(const KfVolume& v, float3 p) {
  float r = (float)v.res;
  return make_int3(kf_f2i(p.x * r / v.size), kf_f2i(p.y * r / v.size), kf_f2i(p.z * r / v.size));
}
// tsdfVolume.h:151-172
__device__ __forceinline__ bool kf_interp_params(const KfVolume& v, float3 pos, int3& base, float& a, float& b, float& c) {
  int3 g = kf_world_to_voxel(v, pos);
  const int R = v.res;
  if (g.x <= 0 || g.x >= R - 1) return false;
  if (g.y <= 0 || g.y >= R - 1) return false;
  if (g.z <= 0 || g.z >= R - 1) return false;
  const float cell = v.cell;
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
// tsdfVolume.h:98-122.  Returns false as the reference does when any of the 8 voxels has weight 0.
// Layers outside the stored slab read as unobserved (only reachable in multi-GPU runs, where the halo covers them).
__device__ __forceinline__ bool kf_interpolate_sdf(const KfVolume& v, float3 pos, float& dist) {
  int3 g; float a, b, c;
  if (!kf_interp_params(v, pos, g, a, b, c)) return false;
  if (!kf_z_stored(v, g.z) || !kf_z_stored(v, g.z + 1)) return false;
  float2 q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = v.tw[kf_vox_index(v, g.x + (k >> 2), g.y + ((k >> 1) & 1), g.z + (k & 1))];
#pragma unroll
  for (int k = 0; k < 8; ++k) if (q[k].y == 0.f) return false;
  float ia = 1 - a, ib = 1 - b, ic = 1 - c;
  dist = q[0].x * ia * ib * ic + q[1].x * ia * ib * c + q[2].x * ia * b * ic + q[3].x * ia * b * c +
         q[4].x * a * ib * ic + q[5].x * a * ib * c + q[6].x * a * b * ic + q[7].x * a * b * c;
  return true;
}
// The same interpolation split into prepare / load / finish so that SEVERAL lookups can have their 8 voxel gathers in flight
// together (the reference's early-outs are pure, so evaluating a batch and then testing the results in the reference's
// order gives the same outcome).  Divisions by the volume size and the cell size go through kf_div with hoisted reciprocals.
struct KfInterp { bool ok; int3 g; float a, b, c; };
__device__ __forceinline__ KfInterp kf_interp_prepare(const KfVolume& v, float3 pos, const KfRecip& rS, const KfRecip& rcell) {
  KfInterp it; it.ok = false; it.a = it.b = it.c = 0.f;
  const float r = (float)v.res;
  int3 g = make_int3(kf_f2i(kf_div(pos.x * r, rS)), kf_f2i(kf_div(pos.y * r, rS)), kf_f2i(kf_div(pos.z * r, rS)));   // tsdfVolume.h:50-56
  it.g = g;
  const int R = v.res;
  if (g.x <= 0 || g.x >= R - 1 || g.y <= 0 || g.y >= R - 1 || g.z <= 0 || g.z >= R - 1) return it;
  const float cell = v.cell;
  g.x = (pos.x < ((float)g.x + 0.5f) * cell) ? (g.x - 1) : g.x;
  g.y = (pos.y < ((float)g.y + 0.5f) * cell) ? (g.y - 1) : g.y;
  g.z = (pos.z < ((float)g.z + 0.5f) * cell) ? (g.z - 1) : g.z;
  it.g = g;
  it.a = kf_div(pos.x - ((float)g.x + 0.5f) * cell, rcell);
  it.b = kf_div(pos.y - ((float)g.y + 0.5f) * cell, rcell);
  it.c = kf_div(pos.z - ((float)g.z + 0.5f) * cell, rcell);
  it.ok = kf_z_stored(v, g.z) && kf_z_stored(v, g.z + 1);
  return it;
}
__device__ __forceinline__ void kf_interp_load(const KfVolume& v, const KfInterp& it, float2 q[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = make_float2(0.f, 0.f);
  if (it.ok) {
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = v.tw[kf_vox_index(v, it.g.x + (k >> 2), it.g.y + ((k >> 1) & 1), it.g.z + (k & 1))];
  }
}
__device__ __forceinline__ bool kf_interp_finish(const KfInterp& it, const float2 q[8], float& dist) {
  if (!it.ok) return false;
#pragma unroll
  for (int k = 0; k < 8; ++k) if (q[k].y == 0.f) return false;
  const float a = it.a, b = it.b, c = it.c, ia = 1 - a, ib = 1 - b, ic = 1 - c;
  dist = q[0].x * ia * ib * ic + q[1].x * ia * ib * c + q[2].x * ia * b * ic + q[3].x * ia * b * c +
         q[4].x * a * ib * ic + q[5].x * a * ib * c + q[6].x * a * b * ic + q[7].x * a * b * c;
  return true;
}
// The same without a branch around anything that leads to a load: a lookup that fails its range test still computes (meaningless) fractions and reads voxel 0
// of the stored volume -- valid memory, value unused -- so that the gathers of SEVERAL lookups really go out together (behind `if (ok)` each lookup's eight
// loads sit in a block of their own with its own wait: six "batched" gradient taps were six dependent round trips).  Used by k_slab_ray_normals only: in the raycast's
// own evaluation (pairs, 80-register cap) the branch-free form measured 8 us SLOWER at 1024^3 and equal at 512^3 (profiles/r05_raycast_bounds.txt)
__device__ __forceinline__ KfInterp kf_interp_prepare_nb(const KfVolume& v, float3 pos, const KfRecip& rS, const KfRecip& rcell) {
  KfInterp it;
  const float r = (float)v.res;
  int3 g = make_int3(kf_f2i(kf_div(pos.x * r, rS)), kf_f2i(kf_div(pos.y * r, rS)), kf_f2i(kf_div(pos.z * r, rS)));   // tsdfVolume.h:50-56
  const int R = v.res;
  const bool in = !(g.x <= 0 || g.x >= R - 1 || g.y <= 0 || g.y >= R - 1 || g.z <= 0 || g.z >= R - 1);
  const float cell = v.cell;
  g.x = (pos.x < ((float)g.x + 0.5f) * cell) ? (g.x - 1) : g.x;
  g.y = (pos.y < ((float)g.y + 0.5f) * cell) ? (g.y - 1) : g.y;
  g.z = (pos.z < ((float)g.z + 0.5f) * cell) ? (g.z - 1) : g.z;
  it.g = g;
  it.a = kf_div(pos.x - ((float)g.x + 0.5f) * cell, rcell);
  it.b = kf_div(pos.y - ((float)g.y + 0.5f) * cell, rcell);
  it.c = kf_div(pos.z - ((float)g.z + 0.5f) * cell, rcell);
  it.ok = in && kf_z_stored(v, g.z) && kf_z_stored(v, g.z + 1);
  return it;
}
__device__ __forceinline__ void kf_interp_load_nb(const KfVolume& v, const KfInterp& it, float2 q[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const size_t idx = kf_vox_index(v, it.g.x + (k >> 2), it.g.y + ((k >> 1) & 1), it.g.z + (k & 1));
    q[k] = v.tw[it.ok ? idx : (size_t)0];
  }
}
// two lookups, 16 gathers in flight
__device__ __forceinline__ void kf_interpolate_sdf_pair(const KfVolume& v, float3 p1, float3 p2, const KfRecip& rS, const KfRecip& rcell,
                                                        bool& ok1, float& d1, bool& ok2, float& d2) {
  const KfInterp i1 = kf_interp_prepare(v, p1, rS, rcell), i2 = kf_interp_prepare(v, p2, rS, rcell);
  float2 q1[8], q2[8];
  kf_interp_load(v, i1, q1); kf_interp_load(v, i2, q2);
  d1 = 0.f; d2 = 0.f;
  ok1 = kf_interp_finish(i1, q1, d1); ok2 = kf_interp_finish(i2, q2, d2);
}

// tsdfVolume.h:123-148 (float -> uchar truncation).  The sixteen loads go out together; the reference's early-outs (a voxel with weight 0)
// are pure, so testing them afterwards, in its order, gives the same verdict -- as eight `if (weight == 0) return` in a row each load
// waited for its predecessor's verdict (eight dependent round trips per coloured hit).
__device__ __forceinline__ bool kf_interpolate_color(const KfVolume& v, float3 pos, uchar4& out) {
  int3 g; float a, b, c;
  if (!kf_interp_params(v, pos, g, a, b, c)) return false;
  if (!kf_z_stored(v, g.z) || !kf_z_stored(v, g.z + 1)) return false;
  float ia = 1 - a, ib = 1 - b, ic = 1 - c;
  float wgt[8]; uchar4 col[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const size_t idx = kf_vox_index(v, g.x + (k >> 2), g.y + ((k >> 1) & 1), g.z + (k & 1));
    wgt[k] = v.tw[idx].y; col[k] = v.color[idx];
  }
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (wgt[k] == 0.f) return false;
    float wa = (k >> 2) ? a : ia, wb = ((k >> 1) & 1) ? b : ib, wc = (k & 1) ? c : ic;
    float t0 = (float)col[k].x * wa * wb * wc, t1 = (float)col[k].y * wa * wb * wc, t2 = (float)col[k].z * wa * wb * wc;
    acc[0] = (k == 0) ? t0 : acc[0] + t0; acc[1] = (k == 0) ? t1 : acc[1] + t1; acc[2] = (k == 0) ? t2 : acc[2] + t2;
  }
  out = make_uchar4((unsigned char)acc[0], (unsigned char)acc[1], (unsigned char)acc[2], 0);
  return true;
}

// wave64 sum by DPP-free shuffles (deterministic order)
__device__ __forceinline__ float kf_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Mat44::getInverse (src/cuda/Mat.h:319-440): cofactor expansion; products and sums in the reference's order.
#define T3(a, b, c) (e[a] * e[b] * e[c])
__host__ __device__ static inline void kf_mat44_inverse(const float* e, float* out) {
  float inv[16];
  inv[0] = T3(5, 10, 15) - T3(5, 11, 14) - T3(9, 6, 15) + T3(9, 7, 14) + T3(13, 6, 11) - T3(13, 7, 10);
  inv[4] = -T3(4, 10, 15) + T3(4, 11, 14) + T3(8, 6, 15) - T3(8, 7, 14) - T3(12, 6, 11) + T3(12, 7, 10);
  inv[8] = T3(4, 9, 15) - T3(4, 11, 13) - T3(8, 5, 15) + T3(8, 7, 13) + T3(12, 5, 11) - T3(12, 7, 9);
  inv[12] = -T3(4, 9, 14) + T3(4, 10, 13) + T3(8, 5, 14) - T3(8, 6, 13) - T3(12, 5, 10) + T3(12, 6, 9);
  inv[1] = -T3(1, 10, 15) + T3(1, 11, 14) + T3(9, 2, 15) - T3(9, 3, 14) - T3(13, 2, 11) + T3(13, 3, 10);
  inv[5] = T3(0, 10, 15) - T3(0, 11, 14) - T3(8, 2, 15) + T3(8, 3, 14) + T3(12, 2, 11) - T3(12, 3, 10);
  inv[9] = -T3(0, 9, 15) + T3(0, 11, 13) + T3(8, 1, 15) - T3(8, 3, 13) - T3(12, 1, 11) + T3(12, 3, 9);
  inv[13] = T3(0, 9, 14) - T3(0, 10, 13) - T3(8, 1, 14) + T3(8, 2, 13) + T3(12, 1, 10) - T3(12, 2, 9);
  inv[2] = T3(1, 6, 15) - T3(1, 7, 14) - T3(5, 2, 15) + T3(5, 3, 14) + T3(13, 2, 7) - T3(13, 3, 6);
  inv[6] = -T3(0, 6, 15) + T3(0, 7, 14) + T3(4, 2, 15) - T3(4, 3, 14) - T3(12, 2, 7) + T3(12, 3, 6);
  inv[10] = T3(0, 5, 15) - T3(0, 7, 13) - T3(4, 1, 15) + T3(4, 3, 13) + T3(12, 1, 7) - T3(12, 3, 5);
  inv[14] = -T3(0, 5, 14) + T3(0, 6, 13) + T3(4, 1, 14) - T3(4, 2, 13) - T3(12, 1, 6) + T3(12, 2, 5);
  inv[3] = -T3(1, 6, 11) + T3(1, 7, 10) + T3(5, 2, 11) - T3(5, 3, 10) - T3(9, 2, 7) + T3(9, 3, 6);
  inv[7] = T3(0, 6, 11) - T3(0, 7, 10) - T3(4, 2, 11) + T3(4, 3, 10) + T3(8, 2, 7) - T3(8, 3, 6);
  inv[11] = -T3(0, 5, 11) + T3(0, 7, 9) + T3(4, 1, 11) - T3(4, 3, 9) - T3(8, 1, 7) + T3(8, 3, 5);
  inv[15] = T3(0, 5, 10) - T3(0, 6, 9) - T3(4, 1, 10) + T3(4, 2, 9) + T3(8, 1, 6) - T3(8, 2, 5);
  float det = e[0] * inv[0] + e[1] * inv[4] + e[2] * inv[8] + e[3] * inv[12];
  float detr = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * detr;
}
#undef T3

// Sum over each 16-lane row with four DPP row-shift adds (an inclusive scan: lane 15 of a row ends with the row total).
// One VALU instruction per step and no LDS round trip -- a `__shfl_down` tree costs a ds_bpermute (~110 cycles, serialised
// by its waitcnt) per step, which made 27 wave sums the most expensive part of a Gauss-Newton step (measured 7.6 us).
__device__ __forceinline__ float kf_row_scan_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));   // row_shr:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));   // row_shr:8
  return v;
}

// entry points implemented across the .hip files (internal linkage between translation units)
int kf_launch_pyramids(kf_ctx* ctx, bool model, bool vertices, bool normals);
int kf_launch_pyramids_and_begin(kf_ctx* ctx, int begin_mode);
int kf_live_contexts(int device);
int kf_device_shared(int device);
int kf_materialize_raw_depth(kf_ctx* ctx);
int kf_pending_depth_consumed(kf_ctx* ctx);
int kf_tail_cull_discard(kf_ctx* ctx);   // a cull that ran as the tail of a tracking launch and will not be consumed: its queue counter back to zero
int kf_upload_wait_for(kf_ctx* ctx, const uint16_t* dev_mm);   // dev_mm is about to be read on the context's stream: wait for its staged copy, if it is one
