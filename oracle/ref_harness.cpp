/*
 * ref_harness.cpp -- thin C entry points around the REFERENCE's own __host__ __device__ headers.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libkfref.so, in this container only,
 * from the headers where they lie under /root/reference/src (nothing is copied into the repo):
 *   cuda/tsdfVolume.h  cuda/Mat.h  cuda/DepthCamera.h  cuda/cuda_declar.h  cuda/DataMap.h  cuda/marchingcube_table.h  AppParams.h
 * The CUDA *headers* they include (cuda_runtime_api.h, vector_functions.h) ship in this image inside the
 * triton wheel; no CUDA runtime library exists here, so the GPU branches of DataMap.h (cudaMalloc/cudaFree,
 * never executed for DeviceKind CPU) are left unresolved at link time (-Wl,--unresolved-symbols=ignore-all).
 * The reference's .cu kernels need nvcc and are NOT built (see DESIGN.md "Oracle").
 *
 * Used by tests/test_oracle_vs_ref.py to pin oracle/kf_oracle.cpp's helper arithmetic bit for bit.
 */
#include <cmath>
#include <cstdio>
#include <cstring>
#include "cuda/tsdfVolume.h"
#include "cuda/Mat.h"
#include "cuda/DepthCamera.h"
#include "cuda/marchingcube_table.h"     /* edgeTable :19, triTable :56 -- compiled as they lie; pins csrc/mc_tables.inc and oracle/mc_tables.inc */

namespace {
struct VolAccess : public tsdfvolume {
  VolAccess() : tsdfvolume(CPU) {}
  Voxel& raw(int x, int y, int z) { return _data.at(x, z * _resolution.y + y); }
};
}

extern "C" {

void* ref_vol_create(unsigned res, float size, float max_weight) {
  VolAccess* v = new VolAccess();
  tsdfVolumeParams p; p.nResolution = res; p.fVolumeMeterSize = size; p.fWeightMax = max_weight;
  v->init(p);
  return v;
}
void ref_vol_destroy(void* h) { delete (VolAccess*)h; }
void ref_vol_set(void* h, int x, int y, int z, float tsdf, float weight, const unsigned char c[3]) {
  Voxel& q = ((VolAccess*)h)->raw(x, y, z); q.tsdf = tsdf; q.weight = weight; q.color = make_uchar3(c[0], c[1], c[2]);
}
void ref_vol_get(void* h, int x, int y, int z, float* tsdf, float* weight, unsigned char c[3]) {
  Voxel q = ((VolAccess*)h)->getVoxel(make_int3(x, y, z)); *tsdf = q.tsdf; *weight = q.weight; c[0] = q.color.x; c[1] = q.color.y; c[2] = q.color.z;
}
void ref_vol_update(void* h, int x, int y, int z, float tsdf, float weight, const unsigned char c[3], float wcolor) {
  ((VolAccess*)h)->updateVoxel(x, y, z, tsdf, weight, make_uchar3(c[0], c[1], c[2]), wcolor);
}
void ref_vol_voxel_to_world(void* h, int x, int y, int z, float out[3]) {
  float3 w = ((VolAccess*)h)->voxelPosToWorld(make_int3(x, y, z)); out[0] = w.x; out[1] = w.y; out[2] = w.z;
}
void ref_vol_world_to_voxel(void* h, const float p[3], int out[3]) {
  int3 g = ((VolAccess*)h)->worldPosToVoxel(make_float3(p[0], p[1], p[2])); out[0] = g.x; out[1] = g.y; out[2] = g.z;
}
void ref_vol_nearest(void* h, const float p[3], float* tsdf, float* weight) {
  Voxel q; ((VolAccess*)h)->getVoxel(make_float3(p[0], p[1], p[2]), q); *tsdf = q.tsdf; *weight = q.weight;
}
int ref_vol_interp_sdf(void* h, const float p[3], float* dist) {
  float d = 0; bool ok = ((VolAccess*)h)->interpolateSDF(make_float3(p[0], p[1], p[2]), d); if (ok) *dist = d; return ok ? 1 : 0;
}
int ref_vol_interp_color(void* h, const float p[3], unsigned char c[3]) {
  uchar3 q = make_uchar3(0, 0, 0); bool ok = ((VolAccess*)h)->interpolateColor(make_float3(p[0], p[1], p[2]), q);
  if (ok) { c[0] = q.x; c[1] = q.y; c[2] = q.z; } return ok ? 1 : 0;
}

void ref_mat44_inverse(const float m[16], float out[16]) { Mat44 a(m); Mat44 r = a.getInverse(); memcpy(out, r.entries, 64); }
void ref_mat44_mul(const float a[16], const float b[16], float out[16]) { Mat44 x(a), y(b); Mat44 r = x * y; memcpy(out, r.entries, 64); }
void ref_mat44_vec(const float m[16], const float v[4], float out[4]) {
  Mat44 a(m); float4 r = a * make_float4(v[0], v[1], v[2], v[3]); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void ref_depth_to_skeleton(unsigned ux, unsigned uy, float depth, const CameraParams* cam, float out[3]) {
  float3 r = DepthCamera::depthToSkeleton(ux, uy, depth, *cam); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_project_to_screen(const float v[3], const CameraParams* cam, int out[2]) {
  int2 r = DepthCamera::projectSkeletonToScreen(make_float3(v[0], v[1], v[2]), *cam); out[0] = r.x; out[1] = r.y;
}
void ref_normalize(const float v[3], float out[3]) { float3 r = normalize(make_float3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_cross(const float a[3], const float b[3], float out[3]) {
  float3 r = cross(make_float3(a[0], a[1], a[2]), make_float3(b[0], b[1], b[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_norm(const float v[3]) { return norm(make_float3(v[0], v[1], v[2])); }
/* the reference's marching-cubes tables, element for element (src/cuda/marchingcube_table.h:19,56) */
void ref_mc_tables(int edge_out[256], int tri_out[256 * 16]) {
  for (int c = 0; c < 256; ++c) { edge_out[c] = edgeTable[c]; for (int i = 0; i < 16; ++i) tri_out[c * 16 + i] = triTable[c][i]; }
}
unsigned ref_sizeof_voxel() { return (unsigned)sizeof(Voxel); }
unsigned ref_sizeof_camera_params() { return (unsigned)sizeof(CameraParams); }
}
