// jpeg_lite.h — baseline JPEG decoding for image textures (see jpeg_lite.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hrthost {

// 8-bit RGB, rows top first (grey files are replicated to three channels, like stbi_load(..., 3)).
bool decodeJPEG(const uint8_t* bytes, size_t n_bytes, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
bool readJPEG(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);

}  // namespace hrthost
