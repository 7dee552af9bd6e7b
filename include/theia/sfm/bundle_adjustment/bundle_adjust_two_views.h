// reference: src/theia/sfm/bundle_adjustment/bundle_adjust_two_views.h:46-75 (declarations) and
// bundle_adjust_two_views.cc:113-191 (semantics): bundle adjustment of two views that both observe
// all of the 3D points -- camera 1 held constant, camera 2's pose, (optionally) the focal lengths
// and the points optimised.  Implemented on the C ABI (tmi_ba_adjust_two_views); the batched form
// adjusts many view pairs in ONE device launch, one wavefront per pair, where the reference runs one
// Ceres solve per pair per CPU thread (two_view_match_geometric_verification.cc:285).
#ifndef THEIA_MI355_BUNDLE_ADJUST_TWO_VIEWS_H_
#define THEIA_MI355_BUNDLE_ADJUST_TWO_VIEWS_H_
#include <vector>

#include "theia/matching/feature_correspondence.h"
#include "theia/sfm/bundle_adjustment/bundle_adjustment.h"
#include "theia/sfm/camera/camera.h"
#include "theia/sfm/twoview_info.h"
#include "theia/util/eigen_lite.h"

namespace theia {
struct TwoViewBundleAdjustmentOptions {
  BundleAdjustmentOptions ba_options;  // only point_dof / device are read: the reference hard-wires the
                                       // solver options of this entry point (:58-68)
  bool constant_camera1_intrinsics = true;
  bool constant_camera2_intrinsics = true;
};

BundleAdjustmentSummary BundleAdjustTwoViews(const TwoViewBundleAdjustmentOptions& options,
                                             const std::vector<FeatureCorrespondence>& correspondences,
                                             Camera* camera1, Camera* camera2,
                                             std::vector<Eigen::Vector4d>* points3d);

// reference: bundle_adjust_two_views.h:77-84, bundle_adjust_two_views.cc:193-240 -- the relative pose
// (info->rotation_2, info->position_2 kept at unit norm) from the angular epipolar error of the
// correspondences (features in normalised image coordinates); only options.device is read.
BundleAdjustmentSummary BundleAdjustTwoViewsAngular(const BundleAdjustmentOptions& options,
                                                    const std::vector<FeatureCorrespondence>& correspondences,
                                                    TwoViewInfo* info);
// Extension of the MI355X path: many pairs in one launch (one wavefront per pair).
struct TwoViewAngularProblem {
  const std::vector<FeatureCorrespondence>* correspondences = nullptr;
  TwoViewInfo* info = nullptr;
};
std::vector<BundleAdjustmentSummary> BundleAdjustTwoViewsAngularBatch(const BundleAdjustmentOptions& options,
                                                                      std::vector<TwoViewAngularProblem>* problems);

// One entry per view pair; every pointer must stay valid for the call.
struct TwoViewBundleAdjustmentProblem {
  TwoViewBundleAdjustmentOptions options;
  const std::vector<FeatureCorrespondence>* correspondences = nullptr;
  Camera* camera1 = nullptr;
  Camera* camera2 = nullptr;
  std::vector<Eigen::Vector4d>* points3d = nullptr;
};
// Extension of the MI355X path: all pairs in one launch.  Returns one summary per problem, in order.
std::vector<BundleAdjustmentSummary> BundleAdjustTwoViewsBatch(std::vector<TwoViewBundleAdjustmentProblem>* problems);
}  // namespace theia
#endif
