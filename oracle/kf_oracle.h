/*
 * kf_oracle.h -- CPU restatement of the reference's per-frame KinectFusion path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (hybkinectfu_amd/) never links, includes or calls this code.
 *
 * Every function restates one reference function in plain scalar C++ (one rounded
 * fp32 operation per source operation: build with -ffp-contract=off) and cites the
 * reference file:line it follows (paths relative to /root/reference).
 *
 * Pinning status: the reference ships no tests, golden vectors or fixtures for this
 * path and its .cu kernels need nvcc (absent here) -> the kernel bodies are
 * "parity unpinned".  The arithmetic that lives in the reference's
 * __host__ __device__ headers (tsdfVolume.h, Mat.h, DepthCamera.h, cuda_declar.h)
 * IS pinned: oracle/ref_harness.cpp compiles those headers as they lie into
 * oracle/_ref/libkfref.so and tests/test_oracle_vs_ref.py checks this restatement
 * against it bit for bit.
 */
#ifndef KF_ORACLE_H_
#define KF_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/AppParams.h:36-43 */
typedef struct okf_cam { uint32_t cols, rows; float cx, cy, fx, fy; } okf_cam;

/* src/cuda/tsdfVolume.h:14-19 -- 12-byte AoS voxel, index (z*R+y)*R+x (tsdfVolume.h:57-60) */
typedef struct okf_voxel { float tsdf; float weight; uint8_t color[3]; uint8_t pad_; } okf_voxel;

typedef struct okf_volume {
  okf_voxel* data;      /* res^3 voxels -- or, for a z-band (z_layers != 0), only layers [z_base, z_base + z_layers) of the res^3 grid */
  int32_t res;          /* cubic resolution (tsdfVolume.h:32) */
  float size;           /* metres (tsdfVolume.h:31) */
  float max_weight;     /* tsdfVolume.h:33 */
  int32_t z_base;       /* z-band (tests of volumes too large for host memory, e.g. 2048^3 = 103 GB): first stored layer */
  int32_t z_layers;     /* z-band: stored layers; 0 = the whole volume (z_base ignored) */
} okf_volume;
/* a z-band volume answers reads outside its layers with an unobserved voxel and counts them: a test that stays inside its band reads 0 here */
uint64_t okf_band_violations(void);     /* returns the count and resets it */

/* src/cuda/MarchingcubeData.h:15-27 -- 72-byte triangle */
typedef struct okf_vertex { float pos[3]; float color[3]; } okf_vertex;
typedef struct okf_triangle { okf_vertex v[3]; } okf_triangle;

void okf_set_perturbation(int mode);    /* tests only: last-bit perturbations of the tracker's arithmetic (bit 0 reversed sums, 1 exp2f taps, 2 fused accumulation, 3 reciprocal-product solve); 0 = off */
int  okf_set_threads(int n);            /* OpenMP threads used by the loops below; returns the count in effect */

/* a1  src/HybKinectfu.cpp:63-96 */
void okf_depth_mm_to_m(const uint16_t* mm, int n, float* out);
/* a2  src/cuda/DataPreprocesser.cu:17-36 */
void okf_trunc_depth(const float* in, int cols, int rows, float tmin, float tmax, float* out);
/* a3  src/cuda/DataPreprocesser.cu:37-79,89-100 */
void okf_bilateral(const float* in, int cols, int rows, float sigma_pixel, float sigma_depth, float* out);
/* a4  src/cuda/VerticesNormalsCalculater.cu:15-66 */
void okf_depth_to_vertices(const float* depth, const okf_cam* cam, float* v4);
void okf_vertices_to_normals(const float* v4, int cols, int rows, float* n4);
/* a5  src/cuda/sample.cu:16-61 (intended semantics: every output pixel exactly once) */
void okf_pyrdown_vertices(const float* in4, int in_cols, int in_rows, float* out4);
void okf_pyrdown_normals(const float* in4, int in_cols, int in_rows, float* out4);
/* a6  src/cuda/CalPointToPlaneErrSolverParams.cu:7-108.  out27d: double accumulation (ground truth),
 *     out27f: fp32 sequential accumulation; valid: number of pixels with a correspondence.  Any may be NULL. */
void okf_icp_system(const float* new_v, const float* new_n, const float* model_v, const float* model_n,
                    const okf_cam* cam, const float cur[16], const float last_inv[16],
                    float dist_thres, float sin_thres, double* out27d, float* out27f, int* valid);
/* a7  src/CameraPoseFinderICP.cpp:12-145.  maps: per level pointers (level 0 first), levels<=3.
 *     returns 1 tracked / 0 lost; pose updated in place only when tracked. */
int  okf_icp_estimate(const float* const* new_v, const float* const* new_n,
                      const float* const* model_v, const float* const* model_n,
                      int levels, const okf_cam* cam0, float dist_thres, float sin_thres,
                      float dist_shake, float angle_shake, float pose[16]);
/* a7 pieces, exposed for unit tests */
int  okf_solve6(const float in27[27], int check_det, float x[6]);          /* ICP.cpp:117-143 */
int  okf_vector6_to_transform(const float x[6], float dist_shake, float angle_shake, float t[16]); /* ICP.cpp:95-111 */
void okf_mat44_inverse(const float m[16], float out[16]);                   /* src/cuda/Mat.h:319-440 */
void okf_mat44_mul(const float a[16], const float b[16], float out[16]);    /* src/cuda/Mat.h:240-262 */
/* a8  src/cuda/CalSDFErrSolverParams.cu:7-138 */
void okf_sdf_system(const okf_volume* vol, const float* trunced_depth, const okf_cam* cam, const float cur[16],
                    double* out27d, float* out27f, int* valid);
/* a9  src/CameraPoseFinderSDF.cpp:25-106, src/utils/eigen_utils.cpp:60-127 */
int  okf_sdf_estimate(const okf_volume* vol, const float* trunced_depth, const okf_cam* cam, int max_iter,
                      float dist_shake, float angle_shake, float pose[16], int* iters_done);
void okf_exp_map(const double v[6], double rot[9], double trans[3]);        /* eigen_utils.cpp:60-127 */
/* a10 src/cuda/integrateVolume.cu:15-96, src/cuda/tsdfVolume.h:38-75.  z range [z0,z1) lets a slab be integrated.
 *     returns the number of voxels updated (N_upd). */
uint64_t okf_integrate(okf_volume* vol, int z0, int z1, const float* trunced_depth, const float* normals4,
                       const uint8_t* rgb, int has_color, int color_angled, const float pose[16],
                       float sdf_trunc, float max_dist, const okf_cam* depth_cam, const okf_cam* rgb_cam);
uint64_t okf_count_weight_gt0(const okf_volume* vol);
/* a11 src/cuda/raycastingVolume.cu:16-176 ; steps (may be NULL): per-pixel number of nearest-voxel samples taken */
void okf_raycast(const okf_volume* vol, int has_color, const float pose[16], float ray_inc, const okf_cam* cam,
                 float near_plane, float far_plane, float* v4, float* n4, uint8_t* rgb, uint32_t* steps);
/* a12 src/cuda/marchingcube.cu:5-164.  Canonical order (z, y, x, k); at most max_tris are stored; returns stored count. */
uint32_t okf_marching_cubes(const okf_volume* vol, int z0, int z1, int has_color, float threshold,
                            okf_triangle* tris, uint32_t max_tris);
/* tsdfVolume.h:98-122 exposed for pin tests: returns 1 and *dist when valid */
int  okf_interpolate_sdf(const okf_volume* vol, const float pos[3], float* dist);
int  okf_interpolate_color(const okf_volume* vol, const float pos[3], uint8_t out[3]);   /* tsdfVolume.h:123-148 */
void okf_world_to_voxel(const okf_volume* vol, const float pos[3], int out[3]);          /* tsdfVolume.h:50-56 */
float okf_norm(const float v[3]);                                                        /* cuda_declar.h norm() */
int  okf_to_int(double v);                                                               /* the (int) conversion rule the oracle uses: CUDA's (saturating, NaN -> 0) */

#ifdef __cplusplus
}
#endif
#endif
