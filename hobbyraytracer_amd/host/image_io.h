// image_io.h — file codecs behind Texture loading and Film output (see image_io.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace hrthost {

bool readHDR(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err);
bool writeHDR(const std::string& path, const float* rgb, int w, int h);
// Portable Float Map ("PF", little-endian, rows bottom first): the fp32 film bit for bit, for exact diffs (RGBE .hdr
// keeps 8 mantissa bits).
bool writePFM(const std::string& path, const float* rgb, int w, int h);
bool readPFM(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err);
bool readPNG(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
bool writePNG(const std::string& path, const uint8_t* rgb, int w, int h, int stride);
bool writeBMP(const std::string& path, const uint8_t* rgb, int w, int h);
bool writeTGA(const std::string& path, const uint8_t* rgb, int w, int h);
// stbi_load(path, .., 3) / stbi_loadf(path, .., 0)
bool loadImageRGB8(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
bool loadImageF32(const std::string& path, std::vector<float>& data, int& w, int& h, int& channels, std::string& err);

}  // namespace hrthost
