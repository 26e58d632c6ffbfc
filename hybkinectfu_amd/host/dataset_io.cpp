// dataset_io.cpp -- the callers and data formats either side of the per-frame path:
//   DataSourceProducerRGBDDataset  TUM RGB-D directory reader        (reference: src/DataSourceProducerRGBDDataset.cpp)
//   CameraPoseFinderFromFile       ground-truth trajectory as tracker (reference: src/CameraPoseFinderFromFile.cpp)
//   TrajectoryRecorder             TUM trajectory writer              (reference: src/TrajectoryRecorder.cpp)
// The reference leans on OpenCV (imread, Mat /= 5, pyrDown) and Eigen (Quaternion <-> Matrix3f), neither vendored in its tree;
// their published algorithms are restated here in the arithmetic types they use (OpenCV 2.4 imgproc/pyramids.cpp integer
// path for CV_16U; Eigen 3 Geometry/Quaternion.h in float).  Paths cited are relative to /root/reference.
#include "hybkf_host.hpp"
#include "png_reader.hpp"
#include <math.h>
#include <string.h>
#include <fstream>
#include <iomanip>
#include <sstream>

// ---- TimedTable -------------------------------------------------------------------------------------------------------------
bool TimedTable::load(const std::string& filename, int header_lines, bool numeric_fields) {
  _rows.clear(); _cursor = 0;
  std::ifstream f(filename.c_str());
  if (!f.is_open()) return false;
  std::string line;
  for (int i = 0; i < header_lines; ++i) std::getline(f, line);            // "# ..." x3 (RGBDDataset.cpp:41,49; FromFile.cpp:26-29)
  while (std::getline(f, line)) {
    std::stringstream ss(line);
    TimedRow r;
    if (!(ss >> r.stamp)) continue;                                         // blank / trailing line
    if (numeric_fields) { for (int k = 0; k < 7; ++k) ss >> r.v[k]; }
    else ss >> r.text;
    _rows.push_back(r);
  }
  return true;
}
bool TimedTable::next(TimedRow& out) {
  if (_cursor >= _rows.size()) return false;
  out = _rows[_cursor++];
  return true;
}
bool TimedTable::nearest(double target, TimedRow& out) {
  TimedRow last;                                                            // stamp 0: "no earlier row in this query"
  while (_cursor < _rows.size()) {
    const size_t at = _cursor;
    const TimedRow& cur = _rows[_cursor++];
    if (cur.stamp >= target) {
      if (cur.stamp - target > target - last.stamp) { out = last; _cursor = at; }   // earlier row is nearer: re-read `cur` next time
      else out = cur;
      return true;
    }
    last = cur;
  }
  return false;
}

// ---- quaternion <-> rotation (Eigen 3, float) --------------------------------------------------------------------------------
// QuaternionBase::toRotationMatrix
Mat44 CameraPoseFinderFromFile::transformFromQuaternion(const float t[3], const float q[4]) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Mat44 m = Mat44::getIdentity();
  m.entries[0] = 1.f - (tyy + tzz); m.entries[1] = txy - twz;         m.entries[2] = txz + twy;
  m.entries[4] = txy + twz;         m.entries[5] = 1.f - (txx + tzz); m.entries[6] = tyz - twx;
  m.entries[8] = txz - twy;         m.entries[9] = tyz + twx;         m.entries[10] = 1.f - (txx + tyy);
  m.setTranslation(t[0], t[1], t[2]);
  return m;
}
// quaternionbase_assign_impl<Matrix3f>: the trace branch, else the largest diagonal element
void TrajectoryRecorder::quaternionFromRotation(const Mat44& mat, float q[4]) {
  const float* e = mat.entries;
  auto R = [e](int r, int c) { return e[r * 4 + c]; };
  float t = R(0, 0) + R(1, 1) + R(2, 2);
  if (t > 0.f) {
    t = sqrtf(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (R(2, 1) - R(1, 2)) * t;
    q[1] = (R(0, 2) - R(2, 0)) * t;
    q[2] = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(R(i, i) - R(j, j) - R(k, k) + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (R(k, j) - R(j, k)) * t;
    q[j] = (R(j, i) + R(i, j)) * t;
    q[k] = (R(k, i) + R(i, k)) * t;
  }
}

// ---- CameraPoseFinderFromFile (src/CameraPoseFinderFromFile.cpp) -------------------------------------------------------------------
bool CameraPoseFinderFromFile::initPoseFinder() {                                  // :23-32
  return _trajectory.load(AppParams::instance()->_io_params.trajReadFilename, 3, true);
}
bool CameraPoseFinderFromFile::estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData&) {   // :66-89
  TimedRow row;
  if (!_trajectory.nearest(depth_frame.timeStamp(), row)) return false;
  const Mat44 file_transform = transformFromQuaternion(&row.v[0], &row.v[3]);
  if (depth_frame.frameId() == 0) {                                                 // :82-86: frame 0 keeps the initial pose
    _refer_transform = _pose * file_transform.getInverse();
    return true;
  }
  setCameraPose(_refer_transform * file_transform);                                 // :87 (+ the device-resident copy)
  return true;
}
bool CameraPoseFinderFromFile::enqueueEstimate(const DepthFrameData& depth_frame) {
  ColorFrameData none;
  if (!estimateCameraPose(depth_frame, none)) return false;
  if (depth_frame.frameId() == 0) setCameraPose(_pose);                             // publish pose + "tracked" for the device-predicated integrate
  return true;
}

// ---- TrajectoryRecorder (src/TrajectoryRecorder.cpp) -------------------------------------------------------------------------------
TrajectoryRecorder::TrajectoryRecorder(const std::string& record_filename) : _record_file(nullptr) {
  _record_file = fopen(record_filename.c_str(), "w");
  if (!_record_file) return;
  fprintf(_record_file, "# trajectory\n# file: %s\n# timestamp tx ty tz qx qy qz qw\n", record_filename.c_str());   // :12-14
}
TrajectoryRecorder::~TrajectoryRecorder() { if (_record_file) fclose(_record_file); }
bool TrajectoryRecorder::recordCameraPose(const Mat44& mat, double timestamp) {     // :29-41
  if (!_record_file) return false;
  float q[4];
  quaternionFromRotation(mat, q);
  // `<< setprecision(14) << double`, `<< setprecision(6) << float`: iostream's default floatfield is printf's %g
  std::ostringstream os;
  os << std::setprecision(14) << timestamp << " ";
  os << std::setprecision(6) << mat.entries[3] << " " << mat.entries[7] << " " << mat.entries[11] << " ";
  os << std::setprecision(6) << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
  const std::string s = os.str();
  fwrite(s.data(), 1, s.size(), _record_file);
  fflush(_record_file);                                                             // std::endl
  return true;
}

// ---- DataSourceProducer (src/DataSourceProducer.h:15-36) ---------------------------------------------------------------------------
bool DataSourceProducer::init() {
  if (_inited) return false;
  _sourcefilename = AppParams::instance()->_io_params.rgbdReadFilename;
  _capture_color = AppParams::instance()->_switch_params.useRGBData;
  if (!initDataSource()) return false;                                              // (frame recording, :23-26, is OpenNI-only: not built)
  _inited = true;
  return _inited;
}
bool DataSourceProducer::readNewFrame(DepthFrameData& depth_data, ColorFrameData& rgb_data) {
  if (!_inited) return false;
  return readDataFromSource(depth_data, rgb_data);
}

// ---- DataSourceProducerRGBDDataset (src/DataSourceProducerRGBDDataset.cpp) ----------------------------------------------------------
bool DataSourceProducerRGBDDataset::initDataSource() {                              // :35-54
  if (!_depth_list.load(_sourcefilename + "depth.txt", 3, false)) return false;
  if (_capture_color && !_rgb_list.load(_sourcefilename + "rgb.txt", 3, false)) return false;
  return true;
}

static inline int reflect101(int i, int n) {                                        // cv::BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}
void DataSourceProducerRGBDDataset::pyrDown16(const uint16_t* src, int cols, int rows, std::vector<uint16_t>& dst) {
  const int oc = (cols + 1) / 2, orows = (rows + 1) / 2;
  static const int w[5] = {1, 4, 6, 4, 1};
  dst.assign((size_t)oc * orows, 0);
  std::vector<int> hrow((size_t)oc * 5);                                            // horizontal pass of the five source rows, unrounded
  for (int y = 0; y < orows; ++y) {
    for (int k = 0; k < 5; ++k) {
      const uint16_t* s = src + (size_t)reflect101(2 * y + k - 2, rows) * cols;
      for (int x = 0; x < oc; ++x) {
        int acc = 0;
        for (int i = 0; i < 5; ++i) acc += w[i] * (int)s[reflect101(2 * x + i - 2, cols)];
        hrow[(size_t)k * oc + x] = acc;
      }
    }
    for (int x = 0; x < oc; ++x) {
      int acc = 0;
      for (int k = 0; k < 5; ++k) acc += w[k] * hrow[(size_t)k * oc + x];
      dst[(size_t)y * oc + x] = (uint16_t)((acc + 128) >> 8);                       // FixPtCast<ushort, 8>
    }
  }
}

bool DataSourceProducerRGBDDataset::readDataFromSource(DepthFrameData& depth_data, ColorFrameData& rgb_data) {   // :100-138
  TimedRow drow;
  if (!_depth_list.next(drow)) return false;
  const CameraParams& cam = AppParams::instance()->_depth_camera_params;
  {
    PngImage img;
    if (!readPng(_sourcefilename + drow.text, img) || img.channels != 1 || img.bit_depth != 16) return false;   // imread(..., UNCHANGED) -> CV_16UC1
    const size_t n = (size_t)img.width * img.height;
    std::vector<uint16_t> mm(n);
    const uint16_t* raw = img.u16();
    // `cur_depth_mat /= _depth_factor` on CV_16U: scaled in floating point, rounded to nearest (ties to even), saturated (:103)
    const double scale = 1.0 / (double)_depth_factor;
    for (size_t i = 0; i < n; ++i) {
      const double v = nearbyint((double)raw[i] * scale);
      mm[i] = (uint16_t)(v < 0 ? 0 : v > 65535.0 ? 65535 : v);
    }
    int cols = (int)img.width, rows = (int)img.height;
    if ((unsigned)cols != cam.cols || (unsigned)rows != cam.rows) {                 // :105-113: a larger sensor image is halved once
      pyrDown16(mm.data(), cols, rows, _depth_store);
      cols = (cols + 1) / 2; rows = (rows + 1) / 2;
      if ((unsigned)cols != cam.cols || (unsigned)rows != cam.rows) return false;   // (the reference would go on with a mismatched Mat)
    } else _depth_store.swap(mm);
    depth_data.mm = _depth_store.data(); depth_data.cols = cols; depth_data.rows = rows;
    depth_data.time_stamp = drow.stamp; depth_data.on_device = false;
  }
  if (_capture_color) {                                                             // :124-135: the colour image nearest in time
    TimedRow crow;
    if (!_rgb_list.nearest(drow.stamp, crow)) return false;
    PngImage img;
    if (!readPng(_sourcefilename + crow.text, img)) return false;
    const size_t n = (size_t)img.width * img.height;
    _bgr_store.resize(n * 3);
    // cv::imread default flags: 8-bit, 3 channels, B G R order; alpha dropped, grey replicated, 16-bit samples keep their high byte
    for (size_t i = 0; i < n; ++i) {
      uint8_t c[4] = {0, 0, 0, 0};
      for (unsigned k = 0; k < img.channels; ++k)
        c[k] = img.bit_depth == 16 ? (uint8_t)(img.u16()[i * img.channels + k] >> 8) : img.data[i * img.channels + k];
      const bool grey = img.channels < 3;
      _bgr_store[3 * i] = grey ? c[0] : c[2]; _bgr_store[3 * i + 1] = grey ? c[0] : c[1]; _bgr_store[3 * i + 2] = c[0];
    }
    rgb_data.bgr = _bgr_store.data(); rgb_data.cols = (int)img.width; rgb_data.rows = (int)img.height; rgb_data.time_stamp = crow.stamp;
  }
  return true;
}
