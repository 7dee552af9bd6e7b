// reference: src/theia/sfm/feature.h (Feature = Eigen::Vector2d, pixel x, y)
#ifndef THEIA_MI355_SFM_FEATURE_H_
#define THEIA_MI355_SFM_FEATURE_H_
#include "theia/util/eigen_lite.h"
namespace theia {
typedef Eigen::Vector2d Feature;
}
#endif
