// host_capi.cpp -- a flat C surface over the C++ host classes so tests and bench.py (ctypes) can drive them exactly the
// way src/MainController.cpp drives the reference: init -> per frame processNewFrame -> generateMesh/saveMesh.
#include "hybkf_host.hpp"
#include <string.h>

static HybKinectfu* g_app = nullptr;
static MeshGeneratorMarchingcube* g_mesh = nullptr;

extern "C" {

// MainController::init (src/MainController.cpp:74-108) with parameters instead of config.ini
int hkf_app_init(unsigned volume_res, float volume_size, unsigned depth_cols, unsigned depth_rows, float cx, float cy, float fx, float fy,
                 int use_sdf_tracker, int host_loop, unsigned max_triangles, float sdf_trunc, float integrate_dist, float trunc_max,
                 int device, unsigned slab_z_begin, unsigned slab_z_end, unsigned slab_halo) {
  AppParams* p = AppParams::instance();
  p->setDefaults(volume_res, volume_size);
  p->_depth_camera_params = {depth_cols, depth_rows, cx, cy, fx, fy};
  p->_rgb_camera_params = p->_depth_camera_params;
  p->_switch_params.useSdfTracker = use_sdf_tracker != 0;
  p->_marchingcube_params.uMaxTriangles = max_triangles;
  if (sdf_trunc > 0) { p->_integrate_params.fSdfTruncation = sdf_trunc; p->_raycast_params.fRayIncrement = 0.7f * sdf_trunc; }
  if (integrate_dist > 0) p->_integrate_params.fMaxIntegrateDist = integrate_dist;
  if (trunc_max > 0) p->_depth_prepocess_params.fMaxTrunc = trunc_max;
  p->device = device; p->slab_z_begin = slab_z_begin; p->slab_z_end = slab_z_end; p->slab_halo = slab_halo;
  delete g_app; g_app = nullptr; delete g_mesh; g_mesh = nullptr;
  if (!CudaDeviceDataMan::instance()->init()) return CudaDeviceDataMan::instance()->lastError();
  g_app = new HybKinectfu();
  if (!g_app->init()) return 1002;
  g_app->poseFinder()->setHostLoop(host_loop != 0);
  g_mesh = new MeshGeneratorMarchingcube();
  return 0;
}
void hkf_app_shutdown() { delete g_app; g_app = nullptr; delete g_mesh; g_mesh = nullptr; CudaDeviceDataMan::instance()->release(); }
void* hkf_app_ctx() { return CudaDeviceDataMan::instance()->ctx(); }

// HybKinectfu::processNewFrame; returns 1 when the frame was tracked, 0 when lost, <0 on error
int hkf_app_process_frame(const uint16_t* mm, int on_device, unsigned frame_id, double stamp) {
  if (!g_app) return -1;
  const CameraParams& c = AppParams::instance()->_depth_camera_params;
  DepthFrameData d; d.mm = mm; d.cols = (int)c.cols; d.rows = (int)c.rows; d.frame_id = frame_id; d.time_stamp = stamp; d.on_device = on_device != 0;
  ColorFrameData col;
  if (!g_app->processNewFrame(d, col)) return -2;
  return g_app->lastTracked() ? 1 : 0;
}
// streaming: no host synchronisation
int hkf_app_enqueue_frame(const uint16_t* mm, int on_device, unsigned frame_id) {
  if (!g_app) return -1;
  const CameraParams& c = AppParams::instance()->_depth_camera_params;
  DepthFrameData d; d.mm = mm; d.cols = (int)c.cols; d.rows = (int)c.rows; d.frame_id = frame_id; d.on_device = on_device != 0;
  ColorFrameData col;
  return g_app->enqueueFrame(d, col) ? 0 : -2;
}
int hkf_app_get_pose(float out16[16]) {
  if (!g_app) return -1;
  Mat44 m = g_app->getCameraPose();
  memcpy(out16, m.entries, 64);
  return g_app->lastTracked() ? 1 : 0;
}
int hkf_app_generate_mesh() { if (!g_mesh) return -1; g_mesh->generateMesh(); return (int)g_mesh->triangleCount(); }
int hkf_app_save_mesh(const char* filename, unsigned* n_vertices, unsigned* n_faces) {
  if (!g_mesh) return -1;
  bool ok = g_mesh->saveMesh(filename);
  if (n_vertices) *n_vertices = (unsigned)(g_mesh->mesh().vertices.size() / 3);
  if (n_faces) *n_faces = (unsigned)(g_mesh->mesh().faces.size() / 3);
  return ok ? 1 : 0;
}
}
