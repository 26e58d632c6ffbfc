// png_reader.hpp -- minimal PNG decoder for the dataset input path (see png_reader.cpp).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

struct PngImage {
  unsigned width = 0, height = 0;
  unsigned channels = 0;          // 1 grey, 2 grey+alpha, 3 RGB (also palette images), 4 RGBA
  unsigned bit_depth = 0;         // 8 or 16
  std::vector<uint8_t> data;      // row-major, interleaved; 16-bit samples as host-endian uint16_t
  const uint16_t* u16() const { return reinterpret_cast<const uint16_t*>(data.data()); }
};

// false (and *err, if given) when the file is missing, damaged or uses a feature outside the supported subset
bool readPng(const std::string& path, PngImage& out, std::string* err = nullptr);
