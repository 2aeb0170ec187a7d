#pragma once
#include <memory>
namespace gtsam {
template <typename T, typename... Args>
std::shared_ptr<T> make_shared(Args&&... args) { return std::make_shared<T>(std::forward<Args>(args)...); }
}  // namespace gtsam
