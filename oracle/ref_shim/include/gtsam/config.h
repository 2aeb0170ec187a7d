#pragma once
#define GTSAM_VERSION_NUMERIC 40300
