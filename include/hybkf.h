/*
 * hybkf.h -- C ABI of the MI355X-native KinectFusion core (libhybkf.so).
 *
 * This is the drop-in boundary for the reference's per-frame path: every entry point replaces one of
 * the reference's `extern "C" void cuda*()` launch wrappers (src/cuda/CudaWrappers.h:22-36) or one of the
 * direct accesses its host classes make to the CudaDeviceDataMan singleton (src/cuda/CudaDeviceDataMan.h:54-67).
 * Differences from the reference boundary, all deliberate:
 *   - state lives in an explicit, opaque `kf_ctx` (one per GPU / z-slab) instead of a process singleton;
 *   - every call returns an int status (0 = ok) instead of void, and never throws;
 *   - calls are asynchronous on the context's HIP stream; only the kf_read_ and kf_download_ calls and kf_synchronize block;
 *   - the Gauss-Newton loops of CameraPoseFinderICP/SDF can run entirely on the device (kf_icp_track /
 *     kf_sdf_track) so the 19 host round trips per frame of the reference (src/CameraPoseFinderICP.cpp:117)
 *     disappear; the per-iteration wrappers are still exported for drop-in use and for parity tests.
 * Plain pointers and sizes only; no C++ or torch types.  Paths cited are relative to /root/reference.
 */
#ifndef HYBKF_H_
#define HYBKF_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kf_ctx kf_ctx;

/* src/AppParams.h:36-43 (24 bytes) */
typedef struct kf_camera_params { uint32_t cols, rows; float cx, cy, fx, fy; } kf_camera_params;
/* src/cuda/Mat.h:196-206: row-major 4x4, m[3], m[7], m[11] = translation, camera -> world */
typedef struct kf_mat44 { float m[16]; } kf_mat44;
/* src/AppParams.h:86-90 */
typedef struct kf_integrate_params { float sdf_truncation; float max_integrate_dist; } kf_integrate_params;
/* src/AppParams.h:59-62 */
typedef struct kf_raycast_params { float ray_increment; } kf_raycast_params;
/* src/AppParams.h:70-75 */
typedef struct kf_volume_params { uint32_t resolution; float size_m; float max_weight; } kf_volume_params;
/* src/AppParams.h:22-30 */
typedef struct kf_icp_params { uint32_t pyramid_levels; float norm_sin_thres, dist_thres, dist_shake, angle_shake; } kf_icp_params;
/* src/AppParams.h:31-35 */
typedef struct kf_sdf_tracker_params { uint32_t max_iter_nums; float dist_shake, angle_shake; } kf_sdf_tracker_params;
/* src/cuda/MarchingcubeData.h:15-27 (24 / 72 bytes) */
typedef struct kf_vertex { float pos[3]; float color[3]; } kf_vertex;
typedef struct kf_triangle { kf_vertex v0, v1, v2; } kf_triangle;

/* What CudaDeviceDataMan::init() reads from AppParams (src/cuda/CudaDeviceDataMan.h:24-51), plus the z-slab. */
typedef struct kf_config {
  kf_camera_params depth_camera;
  kf_camera_params rgb_camera;
  kf_volume_params volume;        /* resolution must be a multiple of 8 */
  uint32_t pyramid_levels;        /* 1..3 */
  uint32_t max_triangles;
  int32_t  has_color;             /* allocate the colour plane + rgb maps */
  int32_t  device;                /* HIP device ordinal */
  uint32_t slab_z_begin;          /* z range of voxel layers this context OWNS; 0 / resolution for one GPU */
  uint32_t slab_z_end;
  uint32_t slab_halo;             /* extra layers stored (and integrated) on each side, clipped to the volume */
} kf_config;

/* map ids for kf_download_map / kf_upload_map (members of CudaDeviceDataMan.h:56-67) */
enum {
  KF_MAP_RAW_DEPTH = 0, KF_MAP_TRUNCED_DEPTH = 1, KF_MAP_FILTERED_DEPTH = 2,   /* float,  cols*rows    */
  KF_MAP_NEW_VERTICES = 3, KF_MAP_NEW_NORMALS = 4,                             /* float4, per level    */
  KF_MAP_MODEL_VERTICES = 5, KF_MAP_MODEL_NORMALS = 6,                         /* float4, per level    */
  KF_MAP_RAW_RGB = 7, KF_MAP_RAYCAST_RGB = 8                                   /* uchar3, cols*rows    */
};

/* status of the device-side tracker after kf_icp_track / kf_sdf_track */
enum { KF_TRACK_OK = 0, KF_TRACK_LOST_DET = 1, KF_TRACK_LOST_SHAKE = 2,
       KF_TRACK_STALLED = 3 /* (no longer produced: a device-side wait that times out is finished by one workgroup alone, see kf_track_result::launch_form) */ };

typedef struct kf_track_result {
  kf_mat44 pose;            /* CameraPoseFinder::_pose after the call (unchanged when lost) */
  int32_t  tracked;         /* bool returned by findCameraPose */
  int32_t  status;          /* KF_TRACK_* */
  int32_t  iterations;      /* Gauss-Newton iterations actually applied */
  int32_t  launch_form;     /* how the last kf_icp_track / kf_sdf_track call was launched: 0 none (frame 0), 1 persistent device loop, 2 one launch per
                             * Gauss-Newton step / iteration (GPU shared, a second context, after a stall), 3 persistent loop that timed
                             * out waiting for a workgroup that was not resident and was finished by one workgroup alone (the frame is kept,
                             * milliseconds late); the same pose bits in every form */
} kf_track_result;

typedef struct kf_volume_stats {
  uint64_t updated_last;    /* N_upd: voxels that passed the update predicate in the last integrate */
  uint64_t weight_gt0;      /* voxels with weight > 0 (owned slab only) */
  uint64_t bricks_active;   /* 8^3 bricks queued for the last integrate's fusion pass (after max_weight fused frames, bricks of saturated free
                             * space that is seen as free space again are counted into updated_last without being queued) */
  uint64_t bricks_total;
  uint64_t updated_total;   /* running sum of updated_last since kf_reset_volume */
  uint64_t frames_fused;    /* integrate calls that ran */
  uint64_t frames_lost;     /* integrate calls skipped because the device-side tracker reported lost */
} kf_volume_stats;

const char* kf_error_string(int status);
const char* kf_version(void);

/* CudaDeviceDataMan::init  src/cuda/CudaDeviceDataMan.h:24-51 */
int kf_create(const kf_config* cfg, kf_ctx** out);
int kf_destroy(kf_ctx* ctx);
int kf_synchronize(kf_ctx* ctx);
void* kf_stream(kf_ctx* ctx);                        /* hipStream_t of the context */
/* enqueue on a caller-owned hipStream_t.  NULL means "back to the private stream", NOT the null stream: a framework whose
 * default stream has the null handle (PyTorch) must create a stream of its own and make it current around its own work */
int kf_set_stream(kf_ctx* ctx, void* hip_stream);
int kf_reset_volume(kf_ctx* ctx);                    /* tsdfvolume::init clearData  src/cuda/tsdfVolume.h:29-37 */

/* HybKinectfu::copyFrameToGPU  src/HybKinectfu.cpp:63-96 : u16 mm -> f32 m ((float)((double)mm*0.001)) into raw_depth */
int kf_upload_depth_mm(kf_ctx* ctx, const uint16_t* host_mm, uint32_t cols, uint32_t rows);
/* the same copy AHEAD of its use (src/HybKinectfu.cpp:63-96 run early): the frame goes into a free upload slot and does not replace the current one;
 * up to two frames may be staged (a third: KF_ERR_STATE), kf_take_next_depth makes the oldest of them the current frame (KF_ERR_STATE if none);
 * kf_upload_depth_mm drops whatever is staged.  *dev_mm (may be NULL) = where the frame lies on the device, for kf_prefetch_frame: its front end
 * then rides in its predecessor's launches.  Staging TWO ahead -- upload frame k+2 while frame k is processed and frame k+1 rides along -- takes
 * the PCIe copy off the critical path: a host-fed stream then runs at the rate of one that is already in HBM (bench.py value_pcie_inclusive). */
int kf_upload_depth_mm_next(kf_ctx* ctx, const uint16_t* host_mm, uint32_t cols, uint32_t rows, const uint16_t** dev_mm);
int kf_take_next_depth(kf_ctx* ctx);
int kf_set_depth_mm_device(kf_ctx* ctx, const uint16_t* dev_mm, uint32_t cols, uint32_t rows);   /* frame already in HBM */
int kf_upload_rgb(kf_ctx* ctx, const uint8_t* host_bgr, uint32_t cols, uint32_t rows);

/* cudaTruncDepth  src/cuda/DataPreprocesser.cu:80-88 */
int kf_trunc_depth(kf_ctx* ctx, float trunc_min, float trunc_max);
/* cudaBiliearFilterDepth  src/cuda/DataPreprocesser.cu:89-100 */
int kf_bilateral_filter_depth(kf_ctx* ctx, float sigma_pixel, float sigma_depth);
/* cudaCalculateNewVertices / cudaCalculateNewNormals  src/cuda/VerticesNormalsCalculater.cu:67-85 */
int kf_calculate_new_vertices(kf_ctx* ctx, const kf_camera_params* depth_camera);
int kf_calculate_new_normals(kf_ctx* ctx);
/* the four above fused (what HybKinectfu::processNewFrame runs, src/HybKinectfu.cpp:106-110) */
int kf_preprocess(kf_ctx* ctx, float trunc_min, float trunc_max, float sigma_pixel, float sigma_depth,
                  const kf_camera_params* depth_camera);
/* No reference counterpart (the reference synchronises after every launch): have kf_preprocess's work for the NEXT frame -- a device
 * u16 millimetre image, as for kf_set_depth_mm_device -- done ahead of time into a second buffer set.  Default form: the call only
 * leaves a note.  Called BEFORE kf_icp_track, the next frame's conversion + gate + bilateral filter ride in the tracking launch as extra
 * workgroups on the CUs the persistent loop leaves idle, and its integrate tile tables and vertices / normals as extra workgroups of the
 * next kf_raycast_volume(_slab) launch; when the tracker took another launch form (or the call came after it) the whole filter rides in
 * the raycast launch and the vertices / normals launch follows it -- same stream, no events either way.  KF_PREFETCH_FUSED=0 selects
 * the older form: the two preprocess launches on a side stream, concurrent with whatever is enqueued next.  Either way the next
 * kf_set_depth_mm_device(same pointer) + kf_preprocess(same parameters) adopts the result; any other sequence ignores it.
 * Results are bit-identical to the unprefetched path. */
int kf_prefetch_frame(kf_ctx* ctx, const uint16_t* dev_depth_mm, uint32_t cols, uint32_t rows, float trunc_min, float trunc_max,
                      float sigma_pixel, float sigma_depth, const kf_camera_params* depth_camera);

/* cudaDownSample{New,Model}{Vertices,Normals}  src/cuda/sample.cu:63-112 */
int kf_downsample_new_vertices(kf_ctx* ctx);
int kf_downsample_new_normals(kf_ctx* ctx);
int kf_downsample_model_vertices(kf_ctx* ctx);
int kf_downsample_model_normals(kf_ctx* ctx);

/* cudaCalPointToPlaneErrSolverParams  src/cuda/CalPointToPlaneErrSolverParams.cu:110-129 -> rigid_align_buf_reduced */
int kf_cal_point_to_plane_solver_params(kf_ctx* ctx, uint32_t pyramid_level, const kf_mat44* cur_transform,
                                        const kf_mat44* last_transform_inv, const kf_camera_params* cam,
                                        float dist_thres, float norm_sin_thres);
/* cudaCalSDFSolverParams  src/cuda/CalSDFErrSolverParams.cu:110-138 */
int kf_cal_sdf_solver_params(kf_ctx* ctx, const kf_camera_params* cam, const kf_mat44* cur_transform);
/* rigid_align_buf_reduced.clone(CPU)  src/CameraPoseFinderICP.cpp:117 : blocking 27-float read-back */
int kf_read_solver_params(kf_ctx* ctx, float out27[27]);

/* Device-resident pose + whole Gauss-Newton loops.
 * kf_set_pose: CameraPoseFinder::init / setCameraPose (src/CameraPoseFinder.h:22-36).
 * kf_icp_track: CameraPoseFinderICP::estimateCameraPose (src/CameraPoseFinderICP.cpp:50-94) incl. the four pyramid builds.
 * kf_sdf_track: CameraPoseFinderSDF::estimateCameraPose (src/CameraPoseFinderSDF.cpp:44-106).
 * frame_id == 0 returns "tracked" without touching the pose, as the reference does.  Asynchronous. */
int kf_set_pose(kf_ctx* ctx, const kf_mat44* pose);
int kf_icp_track(kf_ctx* ctx, uint32_t frame_id, const kf_icp_params* icp, const kf_camera_params* depth_camera);
int kf_sdf_track(kf_ctx* ctx, uint32_t frame_id, const kf_sdf_tracker_params* sdf, const kf_camera_params* depth_camera);
int kf_read_track_result(kf_ctx* ctx, kf_track_result* out);          /* blocking */
/* the same read-back in two halves: `request` enqueues the copy where the stream stands, `wait` blocks until that copy has
 * arrived -- work enqueued in between (integrate and raycast with transform == NULL) keeps the GPU busy meanwhile */
int kf_request_track_result(kf_ctx* ctx);
int kf_wait_track_result(kf_ctx* ctx, kf_track_result* out);
/* Fault injection (tests, rehearsals; no reference counterpart): each of the next `launches` launches of the persistent ICP loop gets one
 * workgroup that exits at once, as if a foreign process had kept it off the chip.  The others time out after 20 ms and one of them finishes
 * the frame's Gauss-Newton loop alone: same pose bits, launch_form 3, no frame lost. */
int kf_inject_track_stall(kf_ctx* ctx, int launches);
/* diagnostics: culls that ran as the tail of a tracking launch and were consumed by kf_integrate_volume / were undone (see kf_integrate_volume) */
int kf_cull_tail_counts(kf_ctx* ctx, uint32_t* consumed, uint32_t* undone);

/* cudaIntegrateVolume  src/cuda/integrateVolume.cu:78-96.  transform == NULL: use the device-resident pose and
 * integrate only if the last kf_*_track call tracked (src/HybKinectfu.cpp:123-140).
 * Three launches per streamed frame: after a call with transform == NULL the next kf_icp_track (persistent loop, volumes without deferred weights
 * and of at most ~6400 macro cells of 32^3 voxels) runs this call's brick cull as the tail of its own launch, for the parameters seen here; the
 * kf_integrate_volume that follows consumes it when it asks for exactly that (pose on the device, same parameters, depth map, slab) and
 * otherwise undoes it and culls in a launch of its own -- the result never depends on it.  kf_cull_tail_counts: how often either happened. */
int kf_integrate_volume(kf_ctx* ctx, int has_color, int use_angle_weight_color, const kf_mat44* transform,
                        const kf_integrate_params* integrate_params, const kf_camera_params* depth_camera,
                        const kf_camera_params* rgb_camera);
/* Deferred free-space weights (no reference counterpart: a property of this implementation of integrateKernel / updateVoxel,
 * src/cuda/integrateVolume.cu:15-77, src/cuda/tsdfVolume.h:57-75 -- results are bit-identical either way).  A wave of the fusion pass
 * whose 128 voxels all hold tsdf 1 and are all observed as free space again only counts the observation; the count is applied to the
 * weights (w <- fminf(w + k, max_weight)) when anything else writes into those voxels and by kf_download_volume.  mode 1: on, 0: off (the
 * plain read-modify-write kernel on every frame), -1 (default): on for volumes of 768^3 voxels and finer, where the fusion pass is memory-bound
 * (KF_INTEGRATE_SAT=0 / 2 in the environment: never / always).  Needs 1 <= max_weight <= 65000 and no colour; otherwise the plain kernel runs
 * whatever the mode. */
int kf_set_defer(kf_ctx* ctx, int mode);
/* cudaRaycastingVolume  src/cuda/raycastingVolume.cu:158-176.  transform == NULL: device-resident pose. */
int kf_raycast_volume(kf_ctx* ctx, int has_color, const kf_mat44* transform, const kf_raycast_params* raycast_params,
                      const kf_camera_params* depth_camera, float near_plane, float far_plane);

/* z-slab partitioning (no counterpart in the single-GPU reference: raycastKernel / raySample / gradientForPoint,
 * src/cuda/raycastingVolume.cu:16-156, run on one whole volume).  What pipeline.SlabPipeline runs per frame:
 *   kf_raycast_volume_slab_cross  every slab context marches every ray and writes, per pixel, ONE 64-bit word: (bits of the ray parameter of the first
 *                                 crossing whose negative sample lies in the layers it owns) << 32 | bits of the VERTEX's ray parameter alpha
 *                                 (:89-90) -- +inf / 0 without a crossing, alpha 0 where the reference gives up at the crossing (:87-88);
 *   MIN all-reduce of the words   (caller; positive floats order like their bits: the first crossing along the ray wins and brings its alpha);
 *   kf_slab_ray_normals           every context rebuilds the winners' vertices from the pixels' rays -- a pure function of pose and camera, which all
 *                                 contexts hold bit for bit -- and the one that OWNS a vertex's voxel layer evaluates gradientForPoint (:16-42) for it:
 *                                 dev_cand[px] = 3 floats (the unit normal), all-zero bits elsewhere.  (The vertex is an extrapolation that can land far from the
 *                                 crossing, outside the crossing slab's halo: its taps belong to the vertex's owner.)
 *   integer SUM all-reduce        of dev_cand (caller; one contributor per pixel: the owner's bits);
 *   kf_set_model_maps_rays        vertices from alpha, normals from dev_cand -> model maps and levels 1, 2 of their pyramids.
 * Same transform / camera / increment / planes in all three calls (NULL transform: the device-resident pose). */
int kf_raycast_volume_slab_cross(kf_ctx* ctx, const kf_mat44* transform, const kf_raycast_params* raycast_params,
                                 const kf_camera_params* depth_camera, float near_plane, float far_plane, uint64_t* dev_ta);
int kf_slab_ray_normals(kf_ctx* ctx, const kf_mat44* transform, const kf_raycast_params* raycast_params, const kf_camera_params* depth_camera,
                        float near_plane, float far_plane, const uint64_t* dev_ta_min, float* dev_cand);
int kf_set_model_maps_rays(kf_ctx* ctx, const kf_mat44* transform, const kf_camera_params* depth_camera, const uint64_t* dev_ta_min, const float* dev_cand);
/* The same merge with the normals evaluated SPECULATIVELY by the marching launch (what pipeline.SlabPipeline runs): a context's own crossing is the likely
 * winner of its pixel, and its vertex nearly always lies in the layers the context owns -- so kf_raycast_volume_slab_cross_spec also evaluates
 * gradientForPoint (:16-42) there, in the shadow of the march, and leaves dev_ta_own[px] = a second copy of its word (the caller all-reduces dev_ta in place)
 * and dev_spec[px] = 3 floats: that gradient, or zeros when the vertex is not this context's.  kf_slab_ray_normals_spec then copies dev_spec where the
 * context's own word won (dev_ta_own[px] == dev_ta_min[px]) and evaluates only the rest -- vertices this context owns under a crossing another context
 * met.  Same dev_cand as kf_slab_ray_normals, bit for bit; the caller pairs the buffers of ONE frame (same volume state, pose, camera, increment, planes). */
int kf_raycast_volume_slab_cross_spec(kf_ctx* ctx, const kf_mat44* transform, const kf_raycast_params* raycast_params, const kf_camera_params* depth_camera,
                                      float near_plane, float far_plane, uint64_t* dev_ta, uint64_t* dev_ta_own, float* dev_spec);
int kf_slab_ray_normals_spec(kf_ctx* ctx, const kf_mat44* transform, const kf_raycast_params* raycast_params, const kf_camera_params* depth_camera,
                             float near_plane, float far_plane, const uint64_t* dev_ta_min, const uint64_t* dev_ta_own, const float* dev_spec, float* dev_cand);
/* MAP FORM of the merge (the earlier protocol, kept for per-kernel tests): the slab that meets a crossing evaluates the whole hit itself -- dev_t[px] =
 * the crossing's ray parameter (+inf if none), dev_v / dev_n[px] = float4 vertex / normal (zeros when the march gives up there) -- and the caller keeps,
 * per pixel, the entry with the smallest t (kf_slab_mask_candidates zeroes the losers for an integer SUM).  It drops the rare pixel whose extrapolated
 * vertex leaves the crossing slab's halo; SlabPipeline does not use it. */
int kf_raycast_volume_slab(kf_ctx* ctx, int has_color, const kf_mat44* transform, const kf_raycast_params* raycast_params,
                           const kf_camera_params* depth_camera, float near_plane, float far_plane,
                           float* dev_t, float* dev_v, float* dev_n);
int kf_slab_mask_candidates(kf_ctx* ctx, const float* dev_t, const float* dev_tmin, float* dev_v, float* dev_n);
int kf_set_model_maps_device(kf_ctx* ctx, const float* dev_v, const float* dev_n);   /* model_{vertices,normals}_pyramid[0] <- device buffers */

/* Pixel-partitioned ICP ("partition pixels across GPUs, all-reduce the 27-float system").  `dev_sums` is a
 * caller-owned 32-float device buffer.  kf_icp_partition_begin builds the pyramids and arms the loop; for step = 0 ..
 * kf_icp_partition_steps()-1, kf_icp_partition_step consumes the all-reduced system of the previous step from dev_sums,
 * sums this rank's share (image rows part/parts) of the pixels and leaves its 27 sums in dev_sums for the caller to
 * all-reduce (SUM); kf_icp_partition_finish applies the last system and commits the pose.  Asynchronous. */
int kf_icp_partition_begin(kf_ctx* ctx, uint32_t frame_id);
int kf_icp_partition_steps(kf_ctx* ctx);
int kf_icp_partition_step(kf_ctx* ctx, uint32_t step, const kf_icp_params* icp, const kf_camera_params* depth_camera,
                          uint32_t part, uint32_t parts, float* dev_sums);
int kf_icp_partition_finish(kf_ctx* ctx, const kf_icp_params* icp, const float* dev_sums);
/* The same protocol for CameraPoseFinderSDF on z-slabs (src/CameraPoseFinderSDF.cpp:44-106, src/cuda/CalSDFErrSolverParams.cu:68-138): a pixel is
 * summed by the context that OWNS the voxel of its world point (the 13 lookups around it reach into the halo at most: a thinner halo is
 * refused with an argument error); the caller all-reduces (SUM) the 27-float system in `dev_sums` between the calls.  All max_iter_nums steps
 * are issued on every rank (after convergence they return at once), so the ranks' collective calls line up. */
int kf_sdf_partition_begin(kf_ctx* ctx, uint32_t frame_id);
int kf_sdf_partition_step(kf_ctx* ctx, uint32_t step, const kf_sdf_tracker_params* params, const kf_camera_params* depth_camera, float* dev_sums);
int kf_sdf_partition_finish(kf_ctx* ctx, const kf_sdf_tracker_params* params, const kf_camera_params* depth_camera, const float* dev_sums);

/* cudaMarchingcube  src/cuda/marchingcube.cu:154-164.  Triangles are appended after those already stored
 * (the reference never clears its counter, src/cuda/MarchingcubeData.h:56,99) in the canonical order (z, y, x, k). */
int kf_marching_cubes(kf_ctx* ctx, int has_color, float threshold_marchingcube);
int kf_clear_triangles(kf_ctx* ctx);                                   /* MarchingcubeData::clearData */
int kf_triangle_count(kf_ctx* ctx, uint32_t* count);                   /* MarchingcubeData::triangleNums, blocking */
int kf_read_triangles(kf_ctx* ctx, kf_triangle* dst, uint32_t first, uint32_t count);   /* MarchingcubeData::clone(CPU) */

/* CudaMap2D::clone(CPU) / copyDataFrom on the singleton's maps (debug + parity; blocking) */
int kf_download_map(kf_ctx* ctx, int map_id, uint32_t level, void* dst, size_t dst_bytes);
int kf_upload_map(kf_ctx* ctx, int map_id, uint32_t level, const void* src, size_t src_bytes);
/* volume in the reference's index order (z*R+y)*R+x for stored layers [z_begin, z_end): tsdf, weight (float) and
 * colour (3 bytes/voxel, may be NULL).  Blocking. */
int kf_download_volume(kf_ctx* ctx, uint32_t z_begin, uint32_t z_end, float* tsdf, float* weight, uint8_t* color);
int kf_upload_volume(kf_ctx* ctx, uint32_t z_begin, uint32_t z_end, const float* tsdf, const float* weight, const uint8_t* color);
int kf_get_volume_stats(kf_ctx* ctx, kf_volume_stats* out);                /* blocking.  weight_gt0 -- the count the reference prints per frame,
                                                                              integrateVolume.cu:91-94 -- comes from a sweep of the volume for a host that asks now
                                                                              and then; asked twice within 8 fused frames, the fusion launches keep it as a running
                                                                              count (+3.6 us per frame at 512^3) and the call is a read-back until 64 frames pass
                                                                              without a question (KF_OBSERVED_COUNT=0 / 1: always sweep / always count) */
int kf_get_fusion_counters(kf_ctx* ctx, kf_volume_stats* out);             /* the same read-back WITHOUT the observed-voxel count (weight_gt0 = 0): never sweeps, and does
                                                                              not count as a question for that count -- what a measurement harness brackets its regions with */
int kf_count_observed_voxels(kf_ctx* ctx, uint64_t* out);                  /* the same number by a sweep of the owned layers (blocking): the tests' cross-check
                                                                              of the running count; src/cuda/integrateVolume.cu:78-96 counts the same way */
int kf_stored_z_range(kf_ctx* ctx, uint32_t* z_begin, uint32_t* z_end);

/* z-slab re-balancing (BASELINE.json north_star "xGMI ... exchange of boundary slabs").  No reference counterpart: the reference
 * holds one whole volume (src/cuda/tsdfVolume.h:29-37).
 * kf_download_volume_device / kf_upload_volume_device: kf_download_volume / kf_upload_volume with caller-owned DEVICE buffers, asynchronous on the
 *   context's stream -- the planes two ranks exchange (torch.distributed send / recv: RCCL point-to-point) when a brick layer changes its owner.
 * kf_resize_slab: the context now owns [z_begin, z_end) (+ halo): layers stored before and after keep their voxels, new layers read as never observed
 *   until uploaded, the rest is dropped; pose, tracker state, frame maps and counters stay.  Blocking.
 * kf_count_layer_work / kf_read_layer_work: per brick layer (resolution / 8 entries) the voxels of queued bricks the next `frames` integrate calls update:
 *   the work measure the boundaries are balanced on. */
int kf_download_volume_device(kf_ctx* ctx, uint32_t z_begin, uint32_t z_end, float* dev_tsdf, float* dev_weight, uint8_t* dev_color);
int kf_upload_volume_device(kf_ctx* ctx, uint32_t z_begin, uint32_t z_end, const float* dev_tsdf, const float* dev_weight, const uint8_t* dev_color);
int kf_resize_slab(kf_ctx* ctx, uint32_t z_begin, uint32_t z_end, uint32_t halo);
int kf_count_layer_work(kf_ctx* ctx, int frames);
int kf_read_layer_work(kf_ctx* ctx, uint64_t* out, int reset);                /* blocking */

/* test hook: counts fp32 quotients where the kernels' split exact-division helper differs from the compiler's `/` (must be 0) */
int kf_selftest_div(kf_ctx* ctx, unsigned n, unsigned seed, int mode, unsigned* mismatches);

/* Per-stage device timers (hipEvent pairs on the context's stream).  `stage_mask` bit s enables stage s:
 * 0 depth upload/convert, 1 preprocess, 2 track, 3 integrate (all passes), 4 raycast, 5 integrate fusion kernel only,
 * 6 marching cubes, 7 raycast kernel only.  Bits 8-15 of `stage_mask`, when > 1, are a sampling period N: only every N-th
 * interval of a stage is timed (fewer event records in a benchmark's timed region).  kf_stage_timers resets the accumulators; kf_read_stage_ms blocks and returns
 * accumulated milliseconds and the number of timed intervals per stage. */
int kf_stage_timers(kf_ctx* ctx, int stage_mask);
int kf_read_stage_ms(kf_ctx* ctx, float out_ms[8], uint32_t counts[8]);
/* Work counters for roofline accounting, maintained only while bit 16 of `stage_mask` is set (they cost a
 * few atomics per wave) and reset by kf_stage_timers: out[0] = voxel samples the reference's ray march takes (per ray from t_min
 * to its first crossing or t_max, src/cuda/raycastingVolume.cu:65-119), out[1] = rays whose crossing was evaluated,
 * out[2] = 4-KiB bricks the marching-cubes extractions read (those with a negative voxel in their 3x3x3 brick neighbourhood),
 * out[3] = triangles in the buffer.  Blocking. */
int kf_read_work_counters(kf_ctx* ctx, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* HYBKF_H_ */
