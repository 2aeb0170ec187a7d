// Stand-in for <gtsam/nonlinear/Values.h>: the class lives next to NonlinearFactor in this stand-in
#pragma once
#include <gtsam/nonlinear/NonlinearFactor.h>
