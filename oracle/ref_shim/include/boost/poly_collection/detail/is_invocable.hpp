#pragma once
#include <type_traits>
namespace boost { namespace poly_collection { namespace detail {
template <typename F, typename... Args>
using is_invocable = std::is_invocable<F, Args...>;
}}}  // namespace boost::poly_collection::detail
