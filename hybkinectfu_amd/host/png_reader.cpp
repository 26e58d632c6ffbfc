// png_reader.cpp -- minimal PNG decoder (zlib inflate + the five scanline filters) for the TUM RGB-D input path.
// The reference reads its frames with cv::imread (src/DataSourceProducerRGBDDataset.cpp:101,130); OpenCV is not part of
// this build, and the two formats the dataset uses -- 16-bit grey depth, 8-bit RGB colour -- need nothing more than this.
// Supported: colour types 0/2/3/4/6, bit depth 8 or 16 (palette: 8), non-interlaced.  Everything else returns false.
#include "png_reader.hpp"
#include <stdio.h>
#include <string.h>
#include <zlib.h>

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

bool fail(std::string* err, const char* what) { if (err) *err = what; return false; }

// PNG specification 9.2: Paeth predictor
inline uint8_t paeth(uint8_t a, uint8_t b, uint8_t c) {
  const int p = (int)a + (int)b - (int)c;
  const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

}  // namespace

bool readPng(const std::string& path, PngImage& out, std::string* err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail(err, "cannot open file");
  std::vector<uint8_t> file;
  {
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
  }
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) return fail(err, "not a PNG file");
  uint32_t width = 0, height = 0; int bit_depth = 0, color_type = -1, interlace = 0;
  std::vector<uint8_t> idat, palette;
  bool have_ihdr = false, have_iend = false;
  size_t pos = 8;
  while (pos + 12 <= file.size() && !have_iend) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) return fail(err, "truncated chunk");
    const uint8_t* data = &file[pos + 8];
    const uint32_t crc = be32(data + len);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != crc) return fail(err, "chunk CRC mismatch");
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return fail(err, "bad IHDR");
      width = be32(data); height = be32(data + 4); bit_depth = data[8]; color_type = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0) return fail(err, "unknown compression / filter method");
      have_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) have_iend = true;
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat.empty()) return fail(err, "missing IHDR / IDAT");
  if (interlace != 0) return fail(err, "interlaced PNG not supported");
  if (width == 0 || height == 0 || width > 16384 || height > 16384) return fail(err, "unreasonable image size");
  int samples;
  switch (color_type) {
    case 0: samples = 1; break;
    case 2: samples = 3; break;
    case 3: samples = 1; break;
    case 4: samples = 2; break;
    case 6: samples = 4; break;
    default: return fail(err, "unknown colour type");
  }
  if (!(bit_depth == 8 || (bit_depth == 16 && color_type != 3))) return fail(err, "bit depth not supported");
  const size_t bpp = (size_t)samples * (bit_depth / 8), stride = (size_t)width * bpp;
  std::vector<uint8_t> raw((stride + 1) * height);
  {
    uLongf dst_len = (uLongf)raw.size();
    const int z = uncompress(raw.data(), &dst_len, idat.data(), (uLong)idat.size());
    if (z != Z_OK || dst_len != raw.size()) return fail(err, "inflate failed");
  }
  // PNG specification 9: reconstruct the scanlines in place (filter type byte in front of each)
  std::vector<uint8_t> pix(stride * height);
  std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t* in = &raw[(stride + 1) * y + 1];
    uint8_t* cur = &pix[stride * y];
    const uint8_t* up = y ? &pix[stride * (y - 1)] : zero.data();
    switch (ft) {
      case 0: memcpy(cur, in, stride); break;
      case 1: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(in[i] + (i >= bpp ? cur[i - bpp] : 0)); break;
      case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(in[i] + up[i]); break;
      case 3: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(in[i] + (((i >= bpp ? cur[i - bpp] : 0) + up[i]) >> 1)); break;
      case 4: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(in[i] + paeth(i >= bpp ? cur[i - bpp] : 0, up[i], i >= bpp ? up[i - bpp] : 0)); break;
      default: return fail(err, "unknown filter type");
    }
  }
  out.width = width; out.height = height; out.bit_depth = (unsigned)bit_depth;
  if (color_type == 3) {                                     // palette -> RGB
    out.channels = 3; out.data.resize((size_t)width * height * 3);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
      const size_t e = (size_t)pix[i] * 3;
      if (e + 3 > palette.size()) return fail(err, "palette index out of range");
      out.data[3 * i] = palette[e]; out.data[3 * i + 1] = palette[e + 1]; out.data[3 * i + 2] = palette[e + 2];
    }
    return true;
  }
  out.channels = (unsigned)samples;
  if (bit_depth == 16) {                                     // network byte order -> host uint16
    out.data.resize(pix.size());
    uint16_t* dst = reinterpret_cast<uint16_t*>(out.data.data());
    for (size_t i = 0; i < pix.size() / 2; ++i) dst[i] = (uint16_t)(((uint16_t)pix[2 * i] << 8) | pix[2 * i + 1]);
  } else out.data.swap(pix);
  return true;
}
