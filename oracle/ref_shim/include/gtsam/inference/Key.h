// Stand-in for GTSAM 4.3a0 <gtsam/inference/Key.h> (oracle/_ref only)
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
namespace gtsam {
using Key = std::uint64_t;
using KeyVector = std::vector<Key>;
using KeyFormatter = std::function<std::string(Key)>;
inline std::string _defaultKeyFormatter(Key k) { return std::to_string(k); }
static const KeyFormatter DefaultKeyFormatter = &_defaultKeyFormatter;
}  // namespace gtsam
