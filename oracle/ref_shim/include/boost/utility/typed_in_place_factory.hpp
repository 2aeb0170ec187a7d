// Stand-in for <boost/utility/typed_in_place_factory.hpp> (included, not used, by factors/nonlinear_factor_gpu.hpp)
#pragma once
