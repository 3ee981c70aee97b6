// assets.h — procedural stand-ins for the assets the reference does not ship (assets.cpp).
#pragma once
#include <string>

namespace hrthost {
long writeTeapotObj(const std::string& path, double detail);
long writeBustObj(const std::string& path, double detail);
bool writeHallHdr(const std::string& path, int width, int height);
}  // namespace hrthost
