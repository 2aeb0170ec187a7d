#pragma once
#include <filesystem>
namespace boost { namespace filesystem {
inline bool create_directories(const std::string& p) { return std::filesystem::create_directories(p); }
}}  // namespace boost::filesystem
