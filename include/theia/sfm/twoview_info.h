// reference: src/theia/sfm/twoview_info.h:54-106 -- match and relative-pose data of a view pair (view 1 at
// the origin with identity rotation).  Data members as in the reference; no serialization here.
#ifndef THEIA_MI355_TWOVIEW_INFO_H_
#define THEIA_MI355_TWOVIEW_INFO_H_
#include "theia/util/eigen_lite.h"

namespace theia {
class TwoViewInfo {
 public:
  TwoViewInfo()
      : focal_length_1(0.0), focal_length_2(0.0), position_2(Eigen::Vector3d::Zero()),
        rotation_2(Eigen::Vector3d::Zero()), num_verified_matches(0), num_homography_inliers(0),
        visibility_score(0) {}
  double focal_length_1;
  double focal_length_2;
  Eigen::Vector3d position_2;
  Eigen::Vector3d rotation_2;
  int num_verified_matches;
  int num_homography_inliers;
  int visibility_score;
};
}  // namespace theia
#endif
