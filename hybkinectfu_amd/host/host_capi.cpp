// host_capi.cpp -- a flat C surface over the C++ host classes so tests and bench.py (ctypes) can drive them exactly the
// way src/MainController.cpp drives the reference: init -> per frame processNewFrame -> generateMesh/saveMesh.
#include "hybkf_host.hpp"
#include "png_reader.hpp"
#include <string.h>

static HybKinectfu* g_app = nullptr;
static MeshGeneratorMarchingcube* g_mesh = nullptr;
static DataSourceProducerRGBDDataset* g_source = nullptr;
// the [IO] / [Switch] entries of src/config.ini that select the dataset reader, the file tracker and the trajectory recorder;
// hkf_app_init applies them after AppParams::setDefaults
static struct { std::string rgbd_dir, traj_read, traj_write; bool use_traj_file = false, record_traj = false, use_rgb = false; } g_io;

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
  p->_io_params.rgbdReadFilename = g_io.rgbd_dir; p->_io_params.trajReadFilename = g_io.traj_read; p->_io_params.trajWriteFilename = g_io.traj_write;
  p->_switch_params.useTrajFromFile = g_io.use_traj_file; p->_switch_params.recordTrajectory = g_io.record_traj;
  p->_switch_params.useRGBData = g_io.use_rgb; p->_switch_params.useDatasetRGBD = !g_io.rgbd_dir.empty();
  delete g_app; g_app = nullptr; delete g_mesh; g_mesh = nullptr; delete g_source; g_source = nullptr;
  if (!CudaDeviceDataMan::instance()->init()) return CudaDeviceDataMan::instance()->lastError();
  g_app = new HybKinectfu();
  if (!g_app->init()) return 1002;
  g_app->poseFinder()->setHostLoop(host_loop != 0);
  g_mesh = new MeshGeneratorMarchingcube();
  return 0;
}
void hkf_app_shutdown() {
  delete g_app; g_app = nullptr; delete g_mesh; g_mesh = nullptr; delete g_source; g_source = nullptr;
  CudaDeviceDataMan::instance()->release();
}
// call BEFORE hkf_app_init; empty strings / zeros switch a feature off
void hkf_app_configure_io(const char* rgbd_dir, const char* traj_read, const char* traj_write, int use_rgb) {
  g_io.rgbd_dir = rgbd_dir ? rgbd_dir : ""; g_io.traj_read = traj_read ? traj_read : ""; g_io.traj_write = traj_write ? traj_write : "";
  g_io.use_traj_file = !g_io.traj_read.empty(); g_io.record_traj = !g_io.traj_write.empty(); g_io.use_rgb = use_rgb != 0;
}
// MainController::mainLoop body (src/MainController.cpp:33-48) on the dataset reader: read the next frame, process it.
// returns 1 tracked, 0 lost, -3 end of data / unreadable frame, <0 other errors; *stamp = the depth frame's time stamp
int hkf_app_process_dataset_frame(unsigned frame_id, double* stamp) {
  if (!g_app) return -1;
  if (!g_source) { g_source = new DataSourceProducerRGBDDataset(); if (!g_source->init()) { delete g_source; g_source = nullptr; return -4; } }
  DepthFrameData d; ColorFrameData c;
  d.frame_id = frame_id; c.frame_id = frame_id;
  if (!g_source->readNewFrame(d, c)) return -3;
  if (stamp) *stamp = d.time_stamp;
  if (!g_app->processNewFrame(d, c)) return -2;
  return g_app->lastTracked() ? 1 : 0;
}
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

// ---- GPU-free mesh post-processing on a caller-supplied triangle soup (kf_triangle layout) ---------------------------------------
static MeshGeneratorMarchingcube* g_soup = nullptr;
int hkf_mesh_from_soup(const void* triangles, unsigned n, int with_color, unsigned* n_vertices, unsigned* n_faces) {
  delete g_soup; g_soup = new MeshGeneratorMarchingcube();
  g_soup->setTriangles((const kf_triangle*)triangles, n, with_color != 0);
  g_soup->weldMesh();
  if (n_vertices) *n_vertices = (unsigned)(g_soup->mesh().vertices.size() / 3);
  if (n_faces) *n_faces = (unsigned)(g_soup->mesh().faces.size() / 3);
  return 0;
}
// which: 0 = the soup mesh (hkf_mesh_from_soup), 1 = the application's mesh (after hkf_app_save_mesh)
static const MeshData* mesh_of(int which) { return which == 0 ? (g_soup ? &g_soup->mesh() : nullptr) : (g_mesh ? &g_mesh->mesh() : nullptr); }
int hkf_mesh_sizes(int which, unsigned* n_vertices, unsigned* n_faces, unsigned* n_colors) {
  const MeshData* m = mesh_of(which); if (!m) return -1;
  *n_vertices = (unsigned)(m->vertices.size() / 3); *n_faces = (unsigned)(m->faces.size() / 3); *n_colors = (unsigned)(m->colors.size() / 4);
  return 0;
}
int hkf_mesh_read(int which, float* vertices, float* normals, float* colors, unsigned* faces) {
  const MeshData* m = mesh_of(which); if (!m) return -1;
  if (vertices) memcpy(vertices, m->vertices.data(), m->vertices.size() * 4);
  if (normals) memcpy(normals, m->normals.data(), m->normals.size() * 4);
  if (colors) memcpy(colors, m->colors.data(), m->colors.size() * 4);
  if (faces) memcpy(faces, m->faces.data(), m->faces.size() * 4);
  return 0;
}
int hkf_mesh_save(int which, const char* filename) { const MeshData* m = mesh_of(which); return m && m->saveToFile(filename) ? 1 : 0; }

// ---- GPU-free entry points of the dataset / trajectory code (CPU tests) ---------------------------------------------------------
// reads frames [0, n) of a TUM directory into caller buffers: depth n x rows x cols u16 mm, bgr n x rows x cols x 3 (may be null)
int hkf_dataset_read(const char* dir, unsigned cols, unsigned rows, int with_color, int n, uint16_t* depth_mm, uint8_t* bgr,
                     double* depth_stamps, double* color_stamps) {
  AppParams* p = AppParams::instance();
  p->_io_params.rgbdReadFilename = dir ? dir : ""; p->_switch_params.useRGBData = with_color != 0;
  p->_depth_camera_params.cols = cols; p->_depth_camera_params.rows = rows;
  DataSourceProducerRGBDDataset src;
  if (!src.init()) return -4;
  int k = 0;
  for (; k < n; ++k) {
    DepthFrameData d; ColorFrameData c;
    if (!src.readNewFrame(d, c)) break;
    memcpy(depth_mm + (size_t)k * cols * rows, d.mm, (size_t)cols * rows * 2);
    if (depth_stamps) depth_stamps[k] = d.time_stamp;
    if (with_color && bgr) memcpy(bgr + (size_t)k * c.cols * c.rows * 3, c.bgr, (size_t)c.cols * c.rows * 3);
    if (with_color && color_stamps) color_stamps[k] = c.time_stamp;
  }
  return k;
}
int hkf_png_read(const char* path, unsigned* width, unsigned* height, unsigned* channels, unsigned* bit_depth, uint8_t* out, size_t out_cap) {
  PngImage img;
  if (!readPng(path, img)) return 0;
  *width = img.width; *height = img.height; *channels = img.channels; *bit_depth = img.bit_depth;
  if (out && img.data.size() <= out_cap) memcpy(out, img.data.data(), img.data.size());
  return 1;
}
void hkf_pyrdown16(const uint16_t* src, int cols, int rows, uint16_t* dst) {
  std::vector<uint16_t> o; DataSourceProducerRGBDDataset::pyrDown16(src, cols, rows, o); memcpy(dst, o.data(), o.size() * 2);
}
void hkf_quat_from_pose(const float pose16[16], float q_xyzw[4]) { Mat44 m; memcpy(m.entries, pose16, 64); TrajectoryRecorder::quaternionFromRotation(m, q_xyzw); }
void hkf_pose_from_quat(const float t[3], const float q_xyzw[4], float pose16[16]) {
  Mat44 m = CameraPoseFinderFromFile::transformFromQuaternion(t, q_xyzw); memcpy(pose16, m.entries, 64);
}
int hkf_trajectory_write(const char* path, const float* poses16, const double* stamps, int n) {
  TrajectoryRecorder rec(path);
  for (int k = 0; k < n; ++k) { Mat44 m; memcpy(m.entries, poses16 + 16 * k, 64); if (!rec.recordCameraPose(m, stamps[k])) return k; }
  return n;
}
// nearest-in-time association over a list file: out_stamps[k] = stamp matched to targets[k] (-1: ran off the end)
int hkf_table_nearest(const char* path, int header_lines, const double* targets, int n, double* out_stamps) {
  TimedTable t;
  if (!t.load(path, header_lines, false)) return -1;
  for (int k = 0; k < n; ++k) { TimedRow r; out_stamps[k] = t.nearest(targets[k], r) ? r.stamp : -1.0; }
  return (int)t.size();
}
}
