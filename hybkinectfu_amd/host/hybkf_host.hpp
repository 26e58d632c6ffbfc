// hybkf_host.hpp -- C++ host classes above the C ABI (include/hybkf.h), mirroring the reference's plugin surface for the
// per-frame path: AppParams, FrameData, CameraPoseFinder{,ICP,SDF}, HybKinectfu, MeshGenerator{,Marchingcube}.
// Same class and method names, argument meaning and bool/void error behaviour as the reference, so a caller written
// against src/HybKinectfu.h, src/CameraPoseFinder.h and src/MeshGenerator.h compiles against this header unchanged apart
// from FrameData, which is a plain view instead of a cv::Mat wrapper (OpenCV is not part of the hot path).
// Paths cited are relative to /root/reference.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/hybkf.h"

// ---- src/AppParams.h:12-95,156-166 -----------------------------------------------------------------------------------
struct IOParams { std::string meshFilename, rgbdReadFilename, rgbdWriteFilename, trajReadFilename, trajWriteFilename; };
struct IcpParams { unsigned nPyramidLevels; float fNormSinThres, fDistThres, fDistShake, fAngleShake; };
struct SDFTrackerParams { unsigned maxIterNums; float fDistShake, fAngleShake; };
typedef kf_camera_params CameraParams;                       // {cols, rows, cx, cy, fx, fy}, 24 bytes
struct RayCasterParams { float fRayIncrement; };
struct DepthPrepocessParams { float fMaxTrunc, fMinTrunc, fSigmaDepth, fSigmaPixel; };
struct tsdfVolumeParams { unsigned nResolution; float fVolumeMeterSize, fWeightMax; };
struct MarchingcubeParams { unsigned uMaxTriangles; };
struct IntegrateParams { float fSdfTruncation, fMaxIntegrateDist; };
struct SwitchParams { bool recordRGBD, recordTrajectory, useRGBData, colorAngleWeight, useDatasetRGBD, useTrajFromFile, useSdfTracker; };

class AppParams {
public:
  static AppParams* instance() { static AppParams v; return &v; }
  // values of src/config.ini with the VGA 525/319.5/239.5 camera; raycast increment = factor * trunc
  // (src/AppParamsProducer.cpp:113-117)
  void setDefaults(unsigned volume_resolution, float volume_size_meter);
  SwitchParams _switch_params;
  CameraParams _rgb_camera_params, _depth_camera_params;
  DepthPrepocessParams _depth_prepocess_params;
  IcpParams _icp_params;
  SDFTrackerParams _sdf_tracker_params;
  tsdfVolumeParams _volume_params;
  IntegrateParams _integrate_params;
  RayCasterParams _raycast_params;
  MarchingcubeParams _marchingcube_params;
  IOParams _io_params;
  // not in the reference: which GPU / z-slab this process owns (one process per GPU)
  int device = 0; unsigned slab_z_begin = 0, slab_z_end = 0, slab_halo = 0;
protected:
  AppParams() { setDefaults(256, 3.0f); }
};

// ---- src/cuda/Mat.h:196-440 ------------------------------------------------------------------------------------------
struct Mat44 {
  float entries[16];                                         // row-major; entries[3], [7], [11] = translation
  static Mat44 getIdentity();
  Mat44 getInverse() const;                                  // cofactor expansion, the reference's expression order
  Mat44 operator*(const Mat44& o) const;
  void setTranslation(float x, float y, float z) { entries[3] = x; entries[7] = y; entries[11] = z; }
};

// ---- src/FrameData.h:57-113: 16-bit millimetre depth (0 = invalid) / BGR 8-bit colour, as plain views ------------------
struct DepthFrameData {
  const uint16_t* mm = nullptr; int cols = 0, rows = 0; double time_stamp = 0; unsigned frame_id = 0;
  bool on_device = false;                                    // mm already lives in HBM (bench / streaming input)
  unsigned frameId() const { return frame_id; }
  double timeStamp() const { return time_stamp; }
};
struct ColorFrameData {
  const uint8_t* bgr = nullptr; int cols = 0, rows = 0; double time_stamp = 0; unsigned frame_id = 0;
};

// ---- src/cuda/CudaDeviceDataMan.h:15-78: owner of all device state, here a kf_ctx --------------------------------------
class CudaDeviceDataMan {
public:
  static CudaDeviceDataMan* instance() { static CudaDeviceDataMan v; return &v; }
  bool init();                                               // from AppParams, CudaDeviceDataMan.h:24-51
  void release();
  kf_ctx* ctx() const { return _ctx; }
  int lastError() const { return _err; }
  int check(int status) { if (status) _err = status; return status; }
private:
  kf_ctx* _ctx = nullptr; int _err = 0;
};

// ---- src/CameraPoseFinder.h:15-43 ---------------------------------------------------------------------------------------
class CameraPoseFinder {
public:
  CameraPoseFinder() : _inited(false) {}
  virtual ~CameraPoseFinder() {}
  virtual bool init(const Mat44& reference_transform);
  virtual bool findCameraPose(const DepthFrameData& depth_frame, const ColorFrameData& color_frame);
  Mat44 getCameraPose() const { return _pose; }
  void setCameraPose(const Mat44& transform);
  // false (default): the Gauss-Newton loop runs on the device (kf_icp_track / kf_sdf_track), one read-back per frame;
  // true: the reference's own host loop -- one kernel wrapper + 27-float read-back + host 6x6 solve per iteration.
  void setHostLoop(bool on) { _host_loop = on; }
  // streaming use: enqueue tracking only; the verdict and pose stay on the device until syncPose()
  bool enqueueCameraPose(const DepthFrameData& depth_frame);
  bool syncPose();                                           // blocking read of (tracked, pose) into _pose
  // the same read-back split in two (kf_request_track_result / kf_wait_track_result): work enqueued between the two calls
  // overlaps with the host's wait.  deviceResident(): the verdict and the pose of this finder live on the device until read.
  bool requestPose();
  bool waitPose();
  // A tracker written against the reference's interface (src/CameraPoseFinder.h:38-39: TWO pure virtuals, initPoseFinder and
  // estimateCameraPose) computes its pose on the host: not device resident.  The built-in ICP / SDF finders override this.
  virtual bool deviceResident() const { return false; }
protected:
  Mat44 _pose;
  bool _host_loop = false;
  virtual bool initPoseFinder() = 0;
  virtual bool estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData& color_frame) = 0;
  // NOT pure: the default estimates on the host and publishes the pose (+ "tracked") to the device, which is all a
  // two-virtual plugin can do; the built-in finders override it with their device-resident loops.
  virtual bool enqueueEstimate(const DepthFrameData& depth_frame);
private:
  bool _inited;
};

// ---- src/CameraPoseFinderICP.h / .cpp --------------------------------------------------------------------------------------
class CameraPoseFinderICP : public CameraPoseFinder {
public:
  bool deviceResident() const override { return !_host_loop; }
protected:
  bool initPoseFinder() override;
  bool estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData& color_frame) override;
  bool enqueueEstimate(const DepthFrameData& depth_frame) override;
private:
  bool vector6ToTransformMatrix(const float x[6], Mat44& output);
  bool minimizePointToPlaneErrFunc(unsigned level, float six_dof[6], const Mat44& cur_transform, const Mat44& last_transform_inv);
  std::vector<int> _iter_nums;
  std::vector<CameraParams> _camera_params_pyramid;
};

// ---- src/CameraPoseFinderSDF.h / .cpp --------------------------------------------------------------------------------------
class CameraPoseFinderSDF : public CameraPoseFinder {
public:
  bool deviceResident() const override { return !_host_loop; }
protected:
  bool initPoseFinder() override;
  bool estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData& color_frame) override;
  bool enqueueEstimate(const DepthFrameData& depth_frame) override;
  bool vector6ToTransformMatrix(const float x[6], Mat44& output);
};

// ---- src/CameraPoseFinderFromFile.{h,cpp}: poses from a TUM trajectory file (timestamp tx ty tz qx qy qz qw), the entry nearest in
// time to the depth frame, re-based so that frame 0 keeps the initial pose ---------------------------------------------------------------------
struct TimedRow { double stamp = 0; std::string text; float v[7] = {0, 0, 0, 0, 0, 0, 0}; };
// The reference walks its list files with getline / tellg / seekg; the same association rule on an in-memory table: rows are
// consumed in order, the first row at or after the target is compared with the row read just before it IN THE SAME QUERY, and
// when the earlier one is nearer the later row is pushed back for the next query (CameraPoseFinderFromFile.cpp:34-65,
// DataSourceProducerRGBDDataset.cpp:67-99).
class TimedTable {
public:
  bool load(const std::string& filename, int header_lines, bool numeric_fields);
  bool nearest(double target, TimedRow& out);                // false: ran off the end before reaching the target
  bool next(TimedRow& out);                                  // plain sequential read (depth list)
  size_t size() const { return _rows.size(); }
private:
  std::vector<TimedRow> _rows; size_t _cursor = 0;
};

class CameraPoseFinderFromFile : public CameraPoseFinder {
public:
  bool deviceResident() const override { return false; }     // poses are computed on the host
  static Mat44 transformFromQuaternion(const float t[3], const float q_xyzw[4]);   // Eigen's Quaternion -> Matrix3f, in fp32
protected:
  bool initPoseFinder() override;
  bool estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData& color_frame) override;
  bool enqueueEstimate(const DepthFrameData& depth_frame) override;
private:
  TimedTable _trajectory;
  Mat44 _refer_transform;
};

// ---- src/TrajectoryRecorder.{h,cpp}: TUM-format trajectory writer ---------------------------------------------------------------------------------
class TrajectoryRecorder {
public:
  explicit TrajectoryRecorder(const std::string& record_filename);
  virtual ~TrajectoryRecorder();
  bool recordCameraPose(const Mat44& mat, double timestamp);
  static void quaternionFromRotation(const Mat44& mat, float q_xyzw[4]);           // Eigen's Matrix3f -> Quaternion, in fp32
protected:
  FILE* _record_file;
};

// ---- src/DataSourceProducer.h, src/DataSourceProducerRGBDDataset.{h,cpp}: TUM RGB-D dataset reader ------------------------------------------------
class DataSourceProducer {
public:
  DataSourceProducer() : _capture_color(false), _inited(false) {}
  virtual ~DataSourceProducer() {}
  bool init();                                               // source directory and colour switch from AppParams
  bool readNewFrame(DepthFrameData& depth_data, ColorFrameData& rgb_data);
protected:
  virtual bool initDataSource() = 0;
  virtual bool readDataFromSource(DepthFrameData& depth_data, ColorFrameData& rgb_data) = 0;
  std::string _sourcefilename;
  bool _capture_color;
private:
  bool _inited;
};

class DataSourceProducerRGBDDataset : public DataSourceProducer {
public:
  DataSourceProducerRGBDDataset() : _depth_factor(5) {}
  // (cols+1)/2 x (rows+1)/2 Gaussian pyramid step of a 16-bit image, cv::pyrDown's integer arithmetic (5-tap 1 4 6 4 1 both
  // ways, reflect-101 border, (sum + 128) >> 8)
  static void pyrDown16(const uint16_t* src, int cols, int rows, std::vector<uint16_t>& dst);
protected:
  bool initDataSource() override;
  bool readDataFromSource(DepthFrameData& depth_data, ColorFrameData& rgb_data) override;
private:
  const float _depth_factor;                                 // TUM depth PNGs hold 1/5000 m; /5 -> millimetres
  TimedTable _depth_list, _rgb_list;
  std::vector<uint16_t> _depth_store;                        // the views handed out point into these until the next read
  std::vector<uint8_t> _bgr_store;
};

// ---- src/HybKinectfu.h / .cpp ------------------------------------------------------------------------------------------------
class HybKinectfu {
public:
  HybKinectfu();
  virtual ~HybKinectfu();
  bool init();
  bool processNewFrame(const DepthFrameData& depth_frame, const ColorFrameData& rgb_frame);
  // Streaming variant: enqueue the whole frame (upload, preprocess, track, integrate-if-tracked, raycast) with no host
  // synchronisation; the tracking verdict is applied on the device.  lastTracked()/getCameraPose() sync.
  bool enqueueFrame(const DepthFrameData& depth_frame, const ColorFrameData& rgb_frame);
  bool lastTracked();
  Mat44 getCameraPose();
  CameraPoseFinder* poseFinder() { return _camera_pose_finder; }
private:
  void copyFrameToGPU(const DepthFrameData& depth_frame, const ColorFrameData& color_frame);
  CameraPoseFinder* _camera_pose_finder;
  TrajectoryRecorder* _camera_pose_recorder = nullptr;       // switch recordTrajectory (HybKinectfu.cpp:47-50,129-132)
  bool _inited;
  bool _last_tracked = true;      // verdict of the last processNewFrame
  bool _pending = false;          // frames enqueued whose verdict still lives on the device
};

// ---- src/MeshGenerator.h, src/MeshGeneratorMarchingcube.{h,cpp}, src/utils/mesh/meshData.* (the parts saveMesh uses) ----------
struct MeshData {
  std::vector<float> vertices;                               // xyz
  std::vector<float> colors;                                 // rgba, empty when the volume has no colour
  std::vector<float> normals;                                // xyz
  std::vector<unsigned> faces;                               // 3 indices per face
  unsigned mergeCloseVertices(float thresh);                 // meshData.cpp:198-283, approx = true path (hash grid, first come wins); ends with removeDegeneratedFaces
  unsigned removeDegeneratedFaces();                         // meshData.cpp:289-310: faces with a repeated index are dropped
  unsigned removeDuplicateFaces();                           // meshData.cpp:42-82
  void computeVertexNormals();                               // meshData.h:713-736
  bool saveToFile(const std::string& filename) const;        // by extension: .obj / .ply / .off (MeshIO.cpp:492-662)
};

class MeshGenerator {
public:
  virtual ~MeshGenerator() {}
  virtual void generateMesh() = 0;
  virtual bool saveMesh(const std::string& filename) = 0;
};

class MeshGeneratorMarchingcube : public MeshGenerator {
public:
  void generateMesh() override;
  bool saveMesh(const std::string& filename) override;
  unsigned triangleCount();
  const MeshData& mesh() const { return _meshes; }
  // the two halves of copyTrianglesToCPU + saveMesh that need no device: usable on any triangle soup (CPU tests pin them
  // against the reference's own ml::MeshData, tests/golden/mesh_*.npz)
  void setTriangles(const kf_triangle* tris, unsigned n, bool with_color);   // :39-58
  void weldMesh();                                                           // :69-84 index buffer, weld, dedupe, normals
protected:
  bool copyTrianglesToCPU();
  MeshData _meshes;
};
