// hybkf_host.cpp -- implementation of the host classes in hybkf_host.hpp over the C ABI (libhybkf.so).
// Reference files followed: src/HybKinectfu.cpp, src/CameraPoseFinderICP.cpp, src/CameraPoseFinderSDF.cpp,
// src/utils/eigen_utils.cpp, src/MeshGeneratorMarchingcube.cpp, src/utils/mesh/meshData.{h,cpp}, src/utils/mesh/MeshIO.cpp.
#include "hybkf_host.hpp"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <fstream>
#include <unordered_map>
#include <unordered_set>

// ---- AppParams --------------------------------------------------------------------------------------------------------
void AppParams::setDefaults(unsigned res, float size) {
  _switch_params = {false, false, false, true, true, false, false};
  _depth_camera_params = {640, 480, 319.5f, 239.5f, 525.0f, 525.0f};
  _rgb_camera_params = _depth_camera_params;
  _depth_prepocess_params = {4.0f, 0.3f, 0.03f, 2.0f};                  // max, min, sigma depth, sigma pixel
  _icp_params = {3, 0.1f, 0.1f, 0.3f, 0.3f};
  _sdf_tracker_params = {6, 0.3f, 0.3f};
  _volume_params = {res, size, 128.0f};
  _integrate_params = {0.05f, 2.0f};
  _raycast_params.fRayIncrement = 0.7f * _integrate_params.fSdfTruncation;  // AppParamsProducer.cpp:113-117
  _marchingcube_params.uMaxTriangles = 6500000;
}

// ---- Mat44 (src/cuda/Mat.h) ---------------------------------------------------------------------------------------------
Mat44 Mat44::getIdentity() { Mat44 m; memset(m.entries, 0, sizeof(m.entries)); m.entries[0] = m.entries[5] = m.entries[10] = m.entries[15] = 1.f; return m; }
Mat44 Mat44::operator*(const Mat44& o) const {
  Mat44 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      r.entries[i * 4 + j] = entries[i * 4] * o.entries[j] + entries[i * 4 + 1] * o.entries[4 + j] + entries[i * 4 + 2] * o.entries[8 + j] + entries[i * 4 + 3] * o.entries[12 + j];
  return r;
}
#define T3(a, b, c) (e[a] * e[b] * e[c])
Mat44 Mat44::getInverse() const {                              // Mat.h:319-440
  const float* e = entries; float inv[16];
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
  Mat44 r;
  for (int i = 0; i < 16; ++i) r.entries[i] = inv[i] * detr;
  return r;
}
#undef T3
static kf_mat44 to_kf(const Mat44& m) { kf_mat44 k; memcpy(k.m, m.entries, 64); return k; }

// ---- small fp32 dense algebra (Eigen stand-ins for the host loop) --------------------------------------------------------
static void unpack27(const float* in, float A[36], float b[6]) {   // ICP.cpp:119-136
  int s = 0;
  for (int i = 0; i < 6; ++i) for (int j = i; j < 7; ++j) { float v = in[s++]; if (j == 6) b[i] = v; else { A[i * 6 + j] = v; A[j * 6 + i] = v; } }
}
static float det6(const float A[36]) {                            // partial-pivot LU, Eigen's 6x6 determinant() path
  float m[36]; memcpy(m, A, sizeof(m)); float det = 1.f;
  for (int k = 0; k < 6; ++k) {
    int p = k; float best = fabsf(m[k * 6 + k]);
    for (int r = k + 1; r < 6; ++r) { float v = fabsf(m[r * 6 + k]); if (v > best) { best = v; p = r; } }
    if (best == 0.f) return 0.f;
    if (p != k) { for (int c = 0; c < 6; ++c) std::swap(m[k * 6 + c], m[p * 6 + c]); det = -det; }
    float piv = m[k * 6 + k]; det *= piv;
    for (int r = k + 1; r < 6; ++r) { float f = m[r * 6 + k] / piv; for (int c = k + 1; c < 6; ++c) m[r * 6 + c] -= f * m[k * 6 + c]; }
  }
  return det;
}
static void llt_solve6(const float A[36], const float b[6], float x[6]) {
  float L[36]; memset(L, 0, sizeof(L));
  for (int j = 0; j < 6; ++j) {
    float s = A[j * 6 + j]; for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
    float d = sqrtf(s); L[j * 6 + j] = d;
    for (int i = j + 1; i < 6; ++i) { float t = A[i * 6 + j]; for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k]; L[i * 6 + j] = t / d; }
  }
  float y[6];
  for (int i = 0; i < 6; ++i) { float t = b[i]; for (int k = 0; k < i; ++k) t -= L[i * 6 + k] * y[k]; y[i] = t / L[i * 6 + i]; }
  for (int i = 5; i >= 0; --i) { float t = y[i]; for (int k = i + 1; k < 6; ++k) t -= L[k * 6 + i] * x[k]; x[i] = t / L[i * 6 + i]; }
}
// R = Rx(x0) Ry(x1) Rz(x2); rotation angle of R (== Eigen AngleAxisf(R).angle()) and |t| against the shake limits
static bool euler_increment(const float x[6], float dist_shake, float angle_shake, Mat44& out) {
  float c0 = cosf(x[0]), s0 = sinf(x[0]), c1 = cosf(x[1]), s1 = sinf(x[1]), c2 = cosf(x[2]), s2 = sinf(x[2]);
  float Rx[9] = {1, 0, 0, 0, c0, -s0, 0, s0, c0}, Ry[9] = {c1, 0, s1, 0, 1, 0, -s1, 0, c1}, Rz[9] = {c2, -s2, 0, s2, c2, 0, 0, 0, 1};
  float Rxy[9], R[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rxy[r * 3 + c] = Rx[r * 3] * Ry[c] + Rx[r * 3 + 1] * Ry[3 + c] + Rx[r * 3 + 2] * Ry[6 + c];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = Rxy[r * 3] * Rz[c] + Rxy[r * 3 + 1] * Rz[3 + c] + Rxy[r * 3 + 2] * Rz[6 + c];
  float ca = (R[0] + R[4] + R[8] - 1.f) * 0.5f; ca = ca > 1.f ? 1.f : (ca < -1.f ? -1.f : ca);
  float angle = acosf(ca), d = sqrtf(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
  if (angle > angle_shake || d > dist_shake) return false;
  out = Mat44::getIdentity();
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) out.entries[r * 4 + c] = R[r * 3 + c]; out.entries[r * 4 + 3] = x[3 + r]; }
  return true;
}

// ---- CudaDeviceDataMan ------------------------------------------------------------------------------------------------------
bool CudaDeviceDataMan::init() {
  release();
  const AppParams* p = AppParams::instance();
  kf_config cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.depth_camera = p->_depth_camera_params; cfg.rgb_camera = p->_rgb_camera_params;
  cfg.volume.resolution = p->_volume_params.nResolution; cfg.volume.size_m = p->_volume_params.fVolumeMeterSize; cfg.volume.max_weight = p->_volume_params.fWeightMax;
  cfg.pyramid_levels = p->_icp_params.nPyramidLevels; cfg.max_triangles = p->_marchingcube_params.uMaxTriangles;
  cfg.has_color = p->_switch_params.useRGBData ? 1 : 0; cfg.device = p->device;
  cfg.slab_z_begin = p->slab_z_begin; cfg.slab_z_end = p->slab_z_end; cfg.slab_halo = p->slab_halo;
  _err = kf_create(&cfg, &_ctx);
  return _err == 0;
}
void CudaDeviceDataMan::release() { if (_ctx) { kf_destroy(_ctx); _ctx = nullptr; } }

// ---- CameraPoseFinder (src/CameraPoseFinder.h:15-43) ---------------------------------------------------------------------------
bool CameraPoseFinder::init(const Mat44& reference_transform) {
  if (_inited) return false;
  _pose = reference_transform;
  if (!initPoseFinder()) return false;
  kf_mat44 k = to_kf(_pose);
  if (CudaDeviceDataMan::instance()->check(kf_set_pose(CudaDeviceDataMan::instance()->ctx(), &k))) return false;
  _inited = true;
  return _inited;
}
bool CameraPoseFinder::findCameraPose(const DepthFrameData& d, const ColorFrameData& c) {
  if (!_inited) return false;
  return estimateCameraPose(d, c);
}
void CameraPoseFinder::setCameraPose(const Mat44& t) {
  _pose = t;
  kf_mat44 k = to_kf(_pose);
  CudaDeviceDataMan::instance()->check(kf_set_pose(CudaDeviceDataMan::instance()->ctx(), &k));
}
bool CameraPoseFinder::enqueueCameraPose(const DepthFrameData& d) { return _inited && enqueueEstimate(d); }
bool CameraPoseFinder::enqueueEstimate(const DepthFrameData& d) {   // default for two-virtual plugins (src/CameraPoseFinder.h:38-39)
  ColorFrameData none;
  if (!estimateCameraPose(d, none)) return false;
  setCameraPose(_pose);                                              // publish pose + "tracked" to the device-resident state
  return true;
}
bool CameraPoseFinder::requestPose() {
  return !CudaDeviceDataMan::instance()->check(kf_request_track_result(CudaDeviceDataMan::instance()->ctx()));
}
bool CameraPoseFinder::waitPose() {
  kf_track_result r;
  if (CudaDeviceDataMan::instance()->check(kf_wait_track_result(CudaDeviceDataMan::instance()->ctx(), &r))) return false;
  memcpy(_pose.entries, r.pose.m, 64);
  return r.tracked != 0;
}
bool CameraPoseFinder::syncPose() {
  kf_track_result r;
  if (CudaDeviceDataMan::instance()->check(kf_read_track_result(CudaDeviceDataMan::instance()->ctx(), &r))) return false;
  memcpy(_pose.entries, r.pose.m, 64);
  return r.tracked != 0;
}

// ---- CameraPoseFinderICP (src/CameraPoseFinderICP.cpp) ---------------------------------------------------------------------------
bool CameraPoseFinderICP::initPoseFinder() {                    // :12-49
  unsigned levels = AppParams::instance()->_icp_params.nPyramidLevels;
  _iter_nums.resize(levels);
  if (levels == 1) { _iter_nums[0] = 3; }
  else if (levels == 2) { _iter_nums[0] = 10; _iter_nums[1] = 5; }
  else if (levels == 3) { _iter_nums[0] = 10; _iter_nums[1] = 5; _iter_nums[2] = 4; }
  else return false;
  _camera_params_pyramid.clear();
  _camera_params_pyramid.push_back(AppParams::instance()->_depth_camera_params);
  for (unsigned l = 1; l < levels; l++) {
    CameraParams cur, prev = _camera_params_pyramid[l - 1];
    cur.cols = prev.cols / 2; cur.rows = prev.rows / 2; cur.cx = prev.cx / 2; cur.cy = prev.cy / 2; cur.fx = prev.fx / 2; cur.fy = prev.fy / 2;
    _camera_params_pyramid.push_back(cur);
  }
  return true;
}
bool CameraPoseFinderICP::enqueueEstimate(const DepthFrameData& depth_frame) {
  const IcpParams& ip = AppParams::instance()->_icp_params;
  kf_icp_params k = {ip.nPyramidLevels, ip.fNormSinThres, ip.fDistThres, ip.fDistShake, ip.fAngleShake};
  return 0 == CudaDeviceDataMan::instance()->check(kf_icp_track(CudaDeviceDataMan::instance()->ctx(), depth_frame.frameId(), &k,
                                                                &AppParams::instance()->_depth_camera_params));
}
bool CameraPoseFinderICP::estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData&) {   // :50-94
  if (!_host_loop) return enqueueEstimate(depth_frame) && syncPose();
  if (depth_frame.frameId() == 0) return true;
  kf_ctx* ctx = CudaDeviceDataMan::instance()->ctx();
  kf_downsample_new_vertices(ctx); kf_downsample_new_normals(ctx); kf_downsample_model_vertices(ctx); kf_downsample_model_normals(ctx);
  Mat44 cur_transform = _pose, last_transform_inv = _pose.getInverse();
  float delta_dof[6];
  for (int l = (int)_iter_nums.size() - 1; l >= 0; l--) {
    int it_nums = _iter_nums[l];
    while (it_nums--) {
      if (!minimizePointToPlaneErrFunc(l, delta_dof, cur_transform, last_transform_inv)) return false;
      Mat44 t;
      if (!vector6ToTransformMatrix(delta_dof, t)) return false;   // "camera shaking detected"
      cur_transform = t * cur_transform;
    }
  }
  setCameraPose(cur_transform);
  return true;
}
bool CameraPoseFinderICP::vector6ToTransformMatrix(const float x[6], Mat44& output) {   // :95-111
  return euler_increment(x, AppParams::instance()->_icp_params.fDistShake, AppParams::instance()->_icp_params.fAngleShake, output);
}
bool CameraPoseFinderICP::minimizePointToPlaneErrFunc(unsigned level, float six_dof[6], const Mat44& cur, const Mat44& last_inv) {   // :113-145
  kf_ctx* ctx = CudaDeviceDataMan::instance()->ctx();
  kf_mat44 kc = to_kf(cur), kl = to_kf(last_inv);
  if (kf_cal_point_to_plane_solver_params(ctx, level, &kc, &kl, &_camera_params_pyramid[level], AppParams::instance()->_icp_params.fDistThres,
                                          AppParams::instance()->_icp_params.fNormSinThres)) return false;
  float buf[27], A[36], b[6];
  if (kf_read_solver_params(ctx, buf)) return false;
  unpack27(buf, A, b);
  if (det6(A) < 1E-10) return false;
  llt_solve6(A, b, six_dof);
  return true;
}

// ---- CameraPoseFinderSDF (src/CameraPoseFinderSDF.cpp, src/utils/eigen_utils.cpp) ---------------------------------------------------
bool CameraPoseFinderSDF::initPoseFinder() { return true; }
bool CameraPoseFinderSDF::vector6ToTransformMatrix(const float x[6], Mat44& output) {
  return euler_increment(x, AppParams::instance()->_sdf_tracker_params.fDistShake, AppParams::instance()->_sdf_tracker_params.fAngleShake, output);
}
bool CameraPoseFinderSDF::enqueueEstimate(const DepthFrameData& depth_frame) {
  const SDFTrackerParams& sp = AppParams::instance()->_sdf_tracker_params;
  kf_sdf_tracker_params k = {sp.maxIterNums, sp.fDistShake, sp.fAngleShake};
  return 0 == CudaDeviceDataMan::instance()->check(kf_sdf_track(CudaDeviceDataMan::instance()->ctx(), depth_frame.frameId(), &k,
                                                                &AppParams::instance()->_depth_camera_params));
}
static void exp_map(const double v[6], double R[9], double dt[3]) {   // eigen_utils.cpp:60-127
  double u0 = v[0], u1 = v[1], u2 = v[2];
  double theta = sqrt(u0 * u0 + u1 * u1 + u2 * u2), si = sin(theta), co = cos(theta);
  double sinc = fabs(theta) < 1.0e-8 ? 1.0 : si / theta;
  double mcosc = fabs(theta) < 2.5e-4 ? 0.5 : (1.0 - co) / theta / theta;
  double msinc = fabs(theta) < 2.5e-4 ? (1. / 6.0) : (1.0 - si / theta) / theta / theta;
  R[0] = co + mcosc * u0 * u0; R[1] = -sinc * u2 + mcosc * u0 * u1; R[2] = sinc * u1 + mcosc * u0 * u2;
  R[3] = sinc * u2 + mcosc * u1 * u0; R[4] = co + mcosc * u1 * u1; R[5] = -sinc * u0 + mcosc * u1 * u2;
  R[6] = -sinc * u1 + mcosc * u2 * u0; R[7] = sinc * u0 + mcosc * u2 * u1; R[8] = co + mcosc * u2 * u2;
  dt[0] = v[3] * (sinc + u0 * u0 * msinc) + v[4] * (u0 * u1 * msinc - u2 * mcosc) + v[5] * (u0 * u2 * msinc + u1 * mcosc);
  dt[1] = v[3] * (u0 * u1 * msinc + u2 * mcosc) + v[4] * (sinc + u1 * u1 * msinc) + v[5] * (u1 * u2 * msinc - u0 * mcosc);
  dt[2] = v[3] * (u0 * u2 * msinc - u1 * mcosc) + v[4] * (u1 * u2 * msinc + u0 * mcosc) + v[5] * (sinc + u2 * u2 * msinc);
}
bool CameraPoseFinderSDF::estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData&) {   // :44-106
  if (!_host_loop) return enqueueEstimate(depth_frame) && syncPose();
  if (depth_frame.frameId() == 0) return true;
  kf_ctx* ctx = CudaDeviceDataMan::instance()->ctx();
  unsigned iter = 0; const float e = 0.001f;
  Mat44 cur = _pose;
  while (iter < AppParams::instance()->_sdf_tracker_params.maxIterNums) {
    kf_mat44 kc = to_kf(cur);
    float buf[27], A[36], b[6], x[6];
    if (kf_cal_sdf_solver_params(ctx, &AppParams::instance()->_depth_camera_params, &kc) || kf_read_solver_params(ctx, buf)) return false;
    unpack27(buf, A, b);
    llt_solve6(A, b, x);
    Mat44 t;
    if (!vector6ToTransformMatrix(x, t)) return false;
    float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
    if (nx < e) break;
    double xd[6], R[9], dt[3];
    for (int k = 0; k < 6; ++k) xd[k] = (double)x[k];
    exp_map(xd, R, dt);
    Mat44 n = Mat44::getIdentity();
    for (int r = 0; r < 3; ++r) {
      float rt = 0.f;
      for (int c = 0; c < 3; ++c) n.entries[r * 4 + c] = (float)R[0 * 3 + r] * cur.entries[c] + (float)R[1 * 3 + r] * cur.entries[4 + c] + (float)R[2 * 3 + r] * cur.entries[8 + c];
      rt = (float)R[0 * 3 + r] * (float)dt[0] + (float)R[1 * 3 + r] * (float)dt[1] + (float)R[2 * 3 + r] * (float)dt[2];
      n.entries[r * 4 + 3] = cur.entries[r * 4 + 3] - rt;
    }
    cur = n;
    iter++;
  }
  setCameraPose(cur);
  return true;
}

// ---- HybKinectfu (src/HybKinectfu.cpp) ---------------------------------------------------------------------------------------------
HybKinectfu::HybKinectfu() : _camera_pose_finder(nullptr), _inited(false) {}
HybKinectfu::~HybKinectfu() { delete _camera_pose_recorder; _camera_pose_recorder = nullptr; delete _camera_pose_finder; _camera_pose_finder = nullptr; }

bool HybKinectfu::init() {                                     // :28-61
  if (_inited) return false;
  if (!CudaDeviceDataMan::instance()->ctx() && !CudaDeviceDataMan::instance()->init()) return false;
  if (AppParams::instance()->_switch_params.useTrajFromFile) _camera_pose_finder = new CameraPoseFinderFromFile();
  else if (AppParams::instance()->_switch_params.useSdfTracker) _camera_pose_finder = new CameraPoseFinderSDF();
  else _camera_pose_finder = new CameraPoseFinderICP();
  if (AppParams::instance()->_switch_params.recordTrajectory)
    _camera_pose_recorder = new TrajectoryRecorder(AppParams::instance()->_io_params.trajWriteFilename);
  Mat44 camera_pose0 = Mat44::getIdentity();
  camera_pose0.setTranslation((float)(AppParams::instance()->_volume_params.fVolumeMeterSize / 2.0),
                              (float)(AppParams::instance()->_volume_params.fVolumeMeterSize / 2.0),
                              -AppParams::instance()->_depth_prepocess_params.fMinTrunc);
  if (!_camera_pose_finder->init(camera_pose0)) return false;
  _inited = true;
  return _inited;
}

void HybKinectfu::copyFrameToGPU(const DepthFrameData& d, const ColorFrameData& c) {   // :63-96 (the u16 -> f32 conversion runs on the device)
  CudaDeviceDataMan* dm = CudaDeviceDataMan::instance();
  if (d.on_device) dm->check(kf_set_depth_mm_device(dm->ctx(), d.mm, d.cols, d.rows));
  else dm->check(kf_upload_depth_mm(dm->ctx(), d.mm, d.cols, d.rows));
  if (AppParams::instance()->_switch_params.useRGBData && c.bgr) dm->check(kf_upload_rgb(dm->ctx(), c.bgr, c.cols, c.rows));
}

bool HybKinectfu::enqueueFrame(const DepthFrameData& depth_frame, const ColorFrameData& rgb_frame) {
  if (!_inited) return false;
  const AppParams* p = AppParams::instance();
  CudaDeviceDataMan* dm = CudaDeviceDataMan::instance();
  kf_ctx* ctx = dm->ctx();
  // a finder that computes its pose on the host (file-driven, or a plugin with the reference's two virtuals) has no streaming
  // form: the frame takes the reference's own sequence, which also handles a lost frame (no integrate, raycast with the old pose)
  if (!_camera_pose_finder->deviceResident()) return processNewFrame(depth_frame, rgb_frame);
  copyFrameToGPU(depth_frame, rgb_frame);
  if (dm->check(kf_preprocess(ctx, p->_depth_prepocess_params.fMinTrunc, p->_depth_prepocess_params.fMaxTrunc, p->_depth_prepocess_params.fSigmaPixel,
                              p->_depth_prepocess_params.fSigmaDepth, &p->_depth_camera_params))) return false;      // :106-110
  if (!_camera_pose_finder->enqueueCameraPose(depth_frame)) return false;                                            // :116
  _pending = true;
  kf_integrate_params ip = {p->_integrate_params.fSdfTruncation, p->_integrate_params.fMaxIntegrateDist};
  if (dm->check(kf_integrate_volume(ctx, p->_switch_params.useRGBData, p->_switch_params.colorAngleWeight, nullptr, &ip,
                                    &p->_depth_camera_params, &p->_rgb_camera_params))) return false;                // :125-140, predicated on device
  kf_raycast_params rp = {p->_raycast_params.fRayIncrement};
  if (dm->check(kf_raycast_volume(ctx, p->_switch_params.useRGBData, nullptr, &rp, &p->_depth_camera_params,
                                  p->_depth_prepocess_params.fMinTrunc, p->_depth_prepocess_params.fMaxTrunc))) return false;   // :149-154
  return true;
}

bool HybKinectfu::processNewFrame(const DepthFrameData& depth_frame, const ColorFrameData& rgb_frame) {   // :98-160
  if (!_inited) return false;
  const AppParams* p = AppParams::instance();
  CudaDeviceDataMan* dm = CudaDeviceDataMan::instance();
  kf_ctx* ctx = dm->ctx();
  copyFrameToGPU(depth_frame, rgb_frame);
  if (dm->check(kf_preprocess(ctx, p->_depth_prepocess_params.fMinTrunc, p->_depth_prepocess_params.fMaxTrunc, p->_depth_prepocess_params.fSigmaPixel,
                              p->_depth_prepocess_params.fSigmaDepth, &p->_depth_camera_params))) return false;
  if (_camera_pose_finder->deviceResident()) {
    // Same results as the sequence below, without a bubble: the verdict is requested right behind the tracker, integrate and
    // raycast are enqueued with the device-resident pose (integrate is predicated on the verdict inside the kernel, :123; a lost
    // frame leaves the pose untouched, so the raycast sees what getCameraPose() would have returned), and only then does the
    // host wait -- for the tracker's result, not for the rest of the frame.
    if (!_camera_pose_finder->enqueueCameraPose(depth_frame) || !_camera_pose_finder->requestPose()) return false;
    kf_integrate_params ip = {p->_integrate_params.fSdfTruncation, p->_integrate_params.fMaxIntegrateDist};
    if (dm->check(kf_integrate_volume(ctx, p->_switch_params.useRGBData, p->_switch_params.colorAngleWeight, nullptr, &ip,
                                      &p->_depth_camera_params, &p->_rgb_camera_params))) return false;
    kf_raycast_params rp = {p->_raycast_params.fRayIncrement};
    if (dm->check(kf_raycast_volume(ctx, p->_switch_params.useRGBData, nullptr, &rp, &p->_depth_camera_params,
                                    p->_depth_prepocess_params.fMinTrunc, p->_depth_prepocess_params.fMaxTrunc))) return false;
    _last_tracked = _camera_pose_finder->waitPose(); _pending = false;
    if (_last_tracked && _camera_pose_recorder) _camera_pose_recorder->recordCameraPose(_camera_pose_finder->getCameraPose(), depth_frame.timeStamp());
    return true;
  }
  bool camera_tracking_success = _camera_pose_finder->findCameraPose(depth_frame, rgb_frame);
  _last_tracked = camera_tracking_success; _pending = false;
  Mat44 cur_camera_pose = _camera_pose_finder->getCameraPose();
  kf_mat44 kp = to_kf(cur_camera_pose);
  if (camera_tracking_success) {
    if (_camera_pose_recorder) _camera_pose_recorder->recordCameraPose(cur_camera_pose, depth_frame.timeStamp());     // :129-132
    kf_integrate_params ip = {p->_integrate_params.fSdfTruncation, p->_integrate_params.fMaxIntegrateDist};
    if (dm->check(kf_integrate_volume(ctx, p->_switch_params.useRGBData, p->_switch_params.colorAngleWeight, &kp, &ip,
                                      &p->_depth_camera_params, &p->_rgb_camera_params))) return false;
  }                                                            // else: "camera lost" -- the reference blocks on cv::waitKey(); we never block
  kf_raycast_params rp = {p->_raycast_params.fRayIncrement};
  if (dm->check(kf_raycast_volume(ctx, p->_switch_params.useRGBData, &kp, &rp, &p->_depth_camera_params,
                                  p->_depth_prepocess_params.fMinTrunc, p->_depth_prepocess_params.fMaxTrunc))) return false;
  return true;
}
bool HybKinectfu::lastTracked() {
  if (_pending && _camera_pose_finder) { _last_tracked = _camera_pose_finder->syncPose(); _pending = false; }
  return _last_tracked;
}
Mat44 HybKinectfu::getCameraPose() {
  if (!_camera_pose_finder) return Mat44::getIdentity();
  lastTracked();
  return _camera_pose_finder->getCameraPose();
}

// ---- MeshData: the pieces of src/utils/mesh/meshData.* that saveMesh uses ---------------------------------------------------------------
namespace {
struct I3 { int x, y, z; bool operator==(const I3& o) const { return x == o.x && y == o.y && z == o.z; } };
struct I3Hash { size_t operator()(const I3& v) const { return ((size_t)v.x * 73856093u) ^ ((size_t)v.y * 19349669u) ^ ((size_t)v.z * 83492791u); } };
inline int sgn(float v) { return (0 < v) - (v < 0); }
inline I3 virtual_voxel(const float* v, float voxel) {          // meshData.h:750-753
  float r = (float)(1.0 / (double)voxel);
  return I3{(int)(v[0] * r + (float)(sgn(v[0]) * 0.5)), (int)(v[1] * r + (float)(sgn(v[1]) * 0.5)), (int)(v[2] * r + (float)(sgn(v[2]) * 0.5))};
}
}
unsigned MeshData::mergeCloseVertices(float thresh) {            // meshData.cpp:198-283 (approx): first vertex in a 3x3x3 cell block wins
  const unsigned numV = (unsigned)(vertices.size() / 3);
  std::vector<unsigned> lookup(numV);
  std::vector<float> nv, nc; nv.reserve(vertices.size());
  const bool has_col = colors.size() == (size_t)numV * 4;
  std::unordered_map<I3, unsigned, I3Hash> grid; grid.reserve(numV * 2);
  unsigned cnt = 0;
  for (unsigned v = 0; v < numV; ++v) {
    I3 c = virtual_voxel(&vertices[3 * v], thresh);
    unsigned nn = (unsigned)-1;
    for (int i = -1; i <= 1 && nn == (unsigned)-1; ++i)
      for (int j = -1; j <= 1 && nn == (unsigned)-1; ++j)
        for (int k = -1; k <= 1; ++k) { auto it = grid.find(I3{c.x + i, c.y + j, c.z + k}); if (it != grid.end()) { nn = it->second; break; } }
    if (nn == (unsigned)-1) {
      grid[c] = cnt; lookup[v] = cnt++;
      nv.insert(nv.end(), &vertices[3 * v], &vertices[3 * v] + 3);
      if (has_col) nc.insert(nc.end(), &colors[4 * v], &colors[4 * v] + 4);
    } else lookup[v] = nn;
  }
  for (auto& f : faces) f = lookup[f];
  vertices.swap(nv);
  if (has_col) colors.swap(nc);
  removeDegeneratedFaces();                                      // meshData.cpp:281
  return cnt;
}
unsigned MeshData::removeDegeneratedFaces() {                    // meshData.cpp:289-310: a face that names a vertex twice (an edge the weld collapsed) goes
  size_t w = 0;
  for (size_t i = 0; i + 2 < faces.size(); i += 3) {
    const unsigned a = faces[i], b = faces[i + 1], c = faces[i + 2];
    if (a == b || a == c || b == c) continue;
    faces[w] = a; faces[w + 1] = b; faces[w + 2] = c; w += 3;
  }
  faces.resize(w);
  return (unsigned)(w / 3);
}
unsigned MeshData::removeDuplicateFaces() {                      // meshData.cpp:42-82: same index set in any order = duplicate, first kept
  struct Key { unsigned a, b, c; bool operator==(const Key& o) const { return a == o.a && b == o.b && c == o.c; } };
  struct KeyHash { size_t operator()(const Key& k) const { return ((size_t)k.a * 73856093u) ^ ((size_t)k.b * 19349669u) ^ ((size_t)k.c * 83492791u); } };
  std::unordered_set<Key, KeyHash> seen; seen.reserve(faces.size() / 3 * 2);
  std::vector<unsigned> nf; nf.reserve(faces.size());
  for (size_t i = 0; i + 2 < faces.size(); i += 3) {
    unsigned s[3] = {faces[i], faces[i + 1], faces[i + 2]};
    std::sort(s, s + 3);
    if (seen.insert(Key{s[0], s[1], s[2]}).second) nf.insert(nf.end(), &faces[i], &faces[i] + 3);
  }
  faces.swap(nf);
  return (unsigned)(faces.size() / 3);
}
void MeshData::computeVertexNormals() {                          // meshData.h:713-736: unit face normals accumulated per vertex, renormalised
  normals.assign(vertices.size(), 0.f);
  auto nrm = [](float* n) { float l = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); if (l < 1e-8) { n[0] = n[1] = n[2] = 0.f; } else { float r = (float)(1.0 / (double)l); n[0] *= r; n[1] *= r; n[2] *= r; } };
  for (size_t i = 0; i + 2 < faces.size(); i += 3) {
    const float* a = &vertices[3 * faces[i]]; const float* b = &vertices[3 * faces[i + 1]]; const float* c = &vertices[3 * faces[i + 2]];
    float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    nrm(n);
    for (int k = 0; k < 3; ++k) { float* d = &normals[3 * faces[i + k]]; d[0] += n[0]; d[1] += n[1]; d[2] += n[2]; }
  }
  for (size_t v = 0; v + 2 < normals.size(); v += 3) nrm(&normals[v]);
}
bool MeshData::saveToFile(const std::string& filename) const {  // MeshIO.h:50-76 (dispatch), MeshIO.cpp:490-662 (writers): same bytes as the reference's files
  const size_t nv = vertices.size() / 3, nf = faces.size() / 3;
  if (nv == 0) return false;                                     // MeshIO.h:52 "empty mesh"
  const bool has_col = colors.size() == nv * 4 && nv > 0, has_n = normals.size() == nv * 3 && nv > 0;
  std::string ext = filename.substr(filename.find_last_of(".") + 1);      // MeshIO.h:20-26: text after the last '.', case-insensitive
  for (auto& ch : ext) ch = (char)tolower(ch);
  if (ext == "obj") {                                            // MeshIO.cpp:609-662
    std::ofstream f(filename);
    if (!f.is_open()) return false;
    f << "####\n#\n# OBJ file Generated by MLIB\n#\n####\n# Object " << filename << "\n#\n# Vertices: " << nv << "\n# Faces: " << nf << "\n#\n####\n";
    for (size_t i = 0; i < nv; ++i) {
      f << "v " << vertices[3 * i] << " " << vertices[3 * i + 1] << " " << vertices[3 * i + 2];
      if (has_col) f << " " << colors[4 * i] << " " << colors[4 * i + 1] << " " << colors[4 * i + 2];
      f << "\n";
    }
    if (has_n) for (size_t i = 0; i < nv; ++i) f << "vn " << normals[3 * i] << " " << normals[3 * i + 1] << " " << normals[3 * i + 2] << "\n";
    for (size_t i = 0; i < nf; ++i) f << "f " << faces[3 * i] + 1 << " " << faces[3 * i + 1] + 1 << " " << faces[3 * i + 2] + 1 << " \n";
    return f.good();
  }
  if (ext == "ply") {                                            // MeshIO.cpp:490-571: binary little-endian, interleaved vertex records
    std::ofstream f(filename, std::ios::binary);
    if (!f.is_open()) return false;
    f << "ply\nformat binary_little_endian 1.0\ncomment MLIB generated\nelement vertex " << nv << "\nproperty float x\nproperty float y\nproperty float z\n";
    if (has_n) f << "property float nx\nproperty float ny\nproperty float nz\n";
    if (has_col) f << "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\n";
    f << "element face " << nf << "\nproperty list uchar int vertex_indices\nend_header\n";
    std::vector<unsigned char> rec;
    rec.reserve(nv * 28);
    for (size_t i = 0; i < nv; ++i) {
      const unsigned char* p = (const unsigned char*)&vertices[3 * i];
      rec.insert(rec.end(), p, p + 12);
      if (has_n) { const unsigned char* q = (const unsigned char*)&normals[3 * i]; rec.insert(rec.end(), q, q + 12); }
      if (has_col) {
        // :547-548: uchar3(r*255, g*255, b*255) copied as FOUR bytes -- the reference's 4th (alpha) byte is whatever follows the
        // 3-byte local on its stack; we write 255 (opaque), the only byte of the file that is not defined by the reference
        rec.push_back((unsigned char)(colors[4 * i] * 255)); rec.push_back((unsigned char)(colors[4 * i + 1] * 255)); rec.push_back((unsigned char)(colors[4 * i + 2] * 255));
        rec.push_back(255);
      }
    }
    f.write((const char*)rec.data(), (std::streamsize)rec.size());
    for (size_t i = 0; i < nf; ++i) { const unsigned char three = 3; f.write((const char*)&three, 1); f.write((const char*)&faces[3 * i], 12); }
    return f.good();
  }
  if (ext == "off") {                                            // MeshIO.cpp:574-606: "COFF", integer colours with a trailing blank
    std::ofstream f(filename);
    if (!f.is_open()) return false;
    f << "COFF\n" << nv << " " << nf << " " << 0 << "\n";
    for (size_t i = 0; i < nv; ++i) {
      f << vertices[3 * i] << " " << vertices[3 * i + 1] << " " << vertices[3 * i + 2];
      if (has_col) f << " " << (unsigned)(colors[4 * i] * 255) << " " << (unsigned)(colors[4 * i + 1] * 255) << " " << (unsigned)(colors[4 * i + 2] * 255) << " " << (unsigned)(colors[4 * i + 3] * 255) << " ";
      f << "\n";
    }
    for (size_t i = 0; i < nf; ++i) f << 3 << " " << faces[3 * i] << " " << faces[3 * i + 1] << " " << faces[3 * i + 2] << "\n";
    return f.good();
  }
  return false;                                                  // MeshIO.h:72 "unknown file format"
}

// ---- MeshGeneratorMarchingcube (src/MeshGeneratorMarchingcube.cpp) -------------------------------------------------------------------------
void MeshGeneratorMarchingcube::generateMesh() {                // :23-29
  const AppParams* p = AppParams::instance();
  CudaDeviceDataMan* dm = CudaDeviceDataMan::instance();
  dm->check(kf_marching_cubes(dm->ctx(), p->_switch_params.useRGBData, 300 * p->_volume_params.fVolumeMeterSize / p->_volume_params.nResolution));
}
unsigned MeshGeneratorMarchingcube::triangleCount() {
  uint32_t n = 0; CudaDeviceDataMan* dm = CudaDeviceDataMan::instance();
  dm->check(kf_triangle_count(dm->ctx(), &n));
  return n;
}
bool MeshGeneratorMarchingcube::copyTrianglesToCPU() {          // :30-60
  CudaDeviceDataMan* dm = CudaDeviceDataMan::instance();
  unsigned n = triangleCount();
  if (n == 0) return false;
  std::vector<kf_triangle> tris(n);
  if (dm->check(kf_read_triangles(dm->ctx(), tris.data(), 0, n))) return false;
  setTriangles(tris.data(), n, AppParams::instance()->_switch_params.useRGBData);
  return true;
}
void MeshGeneratorMarchingcube::setTriangles(const kf_triangle* tris, unsigned n, bool col) {   // :39-58, the copy loop
  _meshes = MeshData();
  _meshes.vertices.resize((size_t)n * 9);
  if (col) _meshes.colors.resize((size_t)n * 12);
  for (unsigned i = 0; i < n; ++i) {
    const kf_vertex* v[3] = {&tris[i].v0, &tris[i].v1, &tris[i].v2};
    for (int k = 0; k < 3; ++k) {
      memcpy(&_meshes.vertices[(size_t)(3 * i + k) * 3], v[k]->pos, 12);
      if (col) { float* c = &_meshes.colors[(size_t)(3 * i + k) * 4]; c[0] = v[k]->color[2]; c[1] = v[k]->color[1]; c[2] = v[k]->color[0]; c[3] = 1.0f; }   // :53 x<->z swap
    }
  }
}
void MeshGeneratorMarchingcube::weldMesh() {                    // :69-84
  // The reference sizes its index buffer by the VERTEX count (:70) and fills one face per triangle: the surplus two thirds are
  // (0,0,0) faces, which mergeCloseVertices' closing removeDegeneratedFaces drops again -- same result as one face per triangle.
  const size_t nv = _meshes.vertices.size() / 3;
  _meshes.faces.resize(nv);
  for (size_t i = 0; i < nv; ++i) _meshes.faces[i] = (unsigned)i;   // index buffer of the triangle soup
  _meshes.mergeCloseVertices(0.0001f);
  _meshes.removeDuplicateFaces();
  _meshes.computeVertexNormals();
}
bool MeshGeneratorMarchingcube::saveMesh(const std::string& filename) {   // :61-96
  if (!copyTrianglesToCPU()) return false;
  weldMesh();
  // the reference returns 0 here even on success (:95); we report whether the file was written
  return _meshes.saveToFile(filename);
}
