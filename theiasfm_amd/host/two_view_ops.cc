// Host shim of theia::BundleAdjustTwoViews (reference bundle_adjust_two_views.cc:113-191) on the
// C ABI: the pairs are flattened into one tmi_ba_two_view_batch, adjusted in one device launch
// (one wavefront per pair) and written back for the pairs whose solve is usable.
#include <chrono>
#include <cstdio>
#include <vector>

#include "theia/sfm/bundle_adjustment/bundle_adjust_two_views.h"
#include "theia_mi355_ba.h"

namespace theia {

std::vector<BundleAdjustmentSummary> BundleAdjustTwoViewsBatch(std::vector<TwoViewBundleAdjustmentProblem>* problems) {
  std::vector<BundleAdjustmentSummary> out;
  if (problems == nullptr || problems->empty()) return out;
  const auto t0 = std::chrono::steady_clock::now();
  const size_t P = problems->size();
  out.resize(P);
  std::vector<double> e1(6 * P), e2(6 * P), k1(10 * P, 0.0), k2(10 * P, 0.0), f1, f2, pts;
  std::vector<int32_t> m1(P), m2(P);
  std::vector<uint8_t> c1(P), c2(P);
  std::vector<int64_t> ptr(P + 1, 0);
  std::vector<char> valid(P, 0);
  // point_dof / device of the batch: those of its first VALID problem; a problem that asks for others is reported as
  // failed instead of being solved with options it did not ask for (split such a batch)
  int point_dof = 4, device = -1;
  bool have_options = false;
  for (size_t p = 0; p < P; ++p) {
    const TwoViewBundleAdjustmentProblem& q = (*problems)[p];
    ptr[p + 1] = ptr[p];
    // the reference CHECK-fails on null arguments / a size mismatch; the shim reports failure
    if (!q.camera1 || !q.camera2 || !q.points3d || !q.correspondences || q.points3d->size() != q.correspondences->size())
      continue;
    if (!have_options) {
      point_dof = q.options.ba_options.point_dof;
      device = q.options.ba_options.device;
      have_options = true;
    } else if (q.options.ba_options.point_dof != point_dof || q.options.ba_options.device != device) {
      continue;
    }
    valid[p] = 1;
    for (int a = 0; a < 6; ++a) {
      e1[6 * p + a] = q.camera1->extrinsics()[a];
      e2[6 * p + a] = q.camera2->extrinsics()[a];
    }
    m1[p] = static_cast<int32_t>(q.camera1->GetCameraIntrinsicsModelType());
    m2[p] = static_cast<int32_t>(q.camera2->GetCameraIntrinsicsModelType());
    for (int a = 0; a < q.camera1->CameraIntrinsics()->NumParameters(); ++a) k1[10 * p + a] = q.camera1->intrinsics()[a];
    for (int a = 0; a < q.camera2->CameraIntrinsics()->NumParameters(); ++a) k2[10 * p + a] = q.camera2->intrinsics()[a];
    c1[p] = q.options.constant_camera1_intrinsics ? 1 : 0;
    c2[p] = q.options.constant_camera2_intrinsics ? 1 : 0;
    for (size_t i = 0; i < q.correspondences->size(); ++i) {
      const FeatureCorrespondence& m = (*q.correspondences)[i];
      f1.push_back(m.feature1.x());
      f1.push_back(m.feature1.y());
      f2.push_back(m.feature2.x());
      f2.push_back(m.feature2.y());
      const Eigen::Vector4d& X = (*q.points3d)[i];
      pts.insert(pts.end(), X.data(), X.data() + 4);
    }
    ptr[p + 1] = ptr[p] + static_cast<int64_t>(q.correspondences->size());
  }
  tmi_ba_two_view_batch B;
  B.num_pairs = static_cast<int32_t>(P);
  B.extrinsics1 = e1.data();
  B.extrinsics2 = e2.data();
  B.model1 = m1.data();
  B.model2 = m2.data();
  B.intrinsics1 = k1.data();
  B.intrinsics2 = k2.data();
  B.constant_intrinsics1 = c1.data();
  B.constant_intrinsics2 = c2.data();
  B.correspondence_ptr = ptr.data();
  B.features1 = f1.data();
  B.features2 = f2.data();
  B.points = pts.data();
  std::vector<int8_t> term(P, 2);
  std::vector<double> c0(P, 0.0), cf(P, 0.0);
  tmi_ba_track_batch_summary bs;
  const double setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const int rc = tmi_ba_adjust_two_views(&B, point_dof, /*max_num_iterations=*/200, device, term.data(), nullptr,
                                         c0.data(), cf.data(), &bs);
  if (rc != TMI_BA_OK) {
    std::fprintf(stderr, "[theia::BundleAdjustTwoViews] device batch failed: %s\n", tmi_ba_last_error());
    return out;  // success = false everywhere
  }
  for (size_t p = 0; p < P; ++p) {
    BundleAdjustmentSummary& s = out[p];
    s.setup_time_in_seconds = setup / static_cast<double>(P);
    s.solve_time_in_seconds = bs.seconds / static_cast<double>(P);
    if (!valid[p]) continue;
    s.initial_cost = c0[p];
    s.final_cost = cf[p];
    // termination_type != FAILURE (:187-189); a pair without correspondences is an empty, converged problem
    s.success = term[p] == 0 || term[p] == 1 || term[p] == -1;
    if (term[p] != 0 && term[p] != 1) continue;
    TwoViewBundleAdjustmentProblem& q = (*problems)[p];
    for (int a = 0; a < 6; ++a) q.camera2->mutable_extrinsics()[a] = e2[6 * p + a];
    q.camera1->mutable_intrinsics()[0] = k1[10 * p];
    q.camera2->mutable_intrinsics()[0] = k2[10 * p];
    for (size_t i = 0; i < q.points3d->size(); ++i)
      for (int a = 0; a < 4; ++a) (*q.points3d)[i][a] = pts[4 * (ptr[p] + static_cast<int64_t>(i)) + a];
  }
  return out;
}

BundleAdjustmentSummary BundleAdjustTwoViews(const TwoViewBundleAdjustmentOptions& options,
                                             const std::vector<FeatureCorrespondence>& correspondences,
                                             Camera* camera1, Camera* camera2,
                                             std::vector<Eigen::Vector4d>* points3d) {
  std::vector<TwoViewBundleAdjustmentProblem> one(1);
  one[0].options = options;
  one[0].correspondences = &correspondences;
  one[0].camera1 = camera1;
  one[0].camera2 = camera2;
  one[0].points3d = points3d;
  const std::vector<BundleAdjustmentSummary> r = BundleAdjustTwoViewsBatch(&one);
  return r.empty() ? BundleAdjustmentSummary() : r[0];
}

// ---- BundleAdjustTwoViewsAngular (bundle_adjust_two_views.cc:193-240) ---------------------------
std::vector<BundleAdjustmentSummary> BundleAdjustTwoViewsAngularBatch(const BundleAdjustmentOptions& options,
                                                                      std::vector<TwoViewAngularProblem>* problems) {
  std::vector<BundleAdjustmentSummary> out;
  if (problems == nullptr || problems->empty()) return out;
  const auto t0 = std::chrono::steady_clock::now();
  const size_t P = problems->size();
  out.resize(P);
  std::vector<double> rot(3 * P, 0.0), pos(3 * P, 0.0), f1, f2;
  std::vector<int64_t> ptr(P + 1, 0);
  std::vector<char> valid(P, 0);
  for (size_t p = 0; p < P; ++p) {
    const TwoViewAngularProblem& q = (*problems)[p];
    ptr[p + 1] = ptr[p];
    if (!q.info || !q.correspondences) continue;  // the reference CHECK-fails; the shim reports failure
    valid[p] = 1;
    for (int a = 0; a < 3; ++a) {
      rot[3 * p + a] = q.info->rotation_2[a];
      pos[3 * p + a] = q.info->position_2[a];
    }
    for (const FeatureCorrespondence& m : *q.correspondences) {
      f1.push_back(m.feature1.x());
      f1.push_back(m.feature1.y());
      f2.push_back(m.feature2.x());
      f2.push_back(m.feature2.y());
    }
    ptr[p + 1] = ptr[p] + static_cast<int64_t>(q.correspondences->size());
  }
  tmi_ba_two_view_angular_batch B;
  B.num_pairs = static_cast<int32_t>(P);
  B.rotation2 = rot.data();
  B.position2 = pos.data();
  B.correspondence_ptr = ptr.data();
  B.features1 = f1.data();
  B.features2 = f2.data();
  std::vector<int8_t> term(P, -1);
  std::vector<int32_t> iters(P, 0);
  std::vector<double> c0(P, 0.0), cf(P, 0.0);
  tmi_ba_track_batch_summary bs;
  const double setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const int rc = tmi_ba_adjust_two_views_angular(&B, 200 /* SetSolverOptions :64 */, options.device, term.data(),
                                                 iters.data(), c0.data(), cf.data(), &bs);
  if (rc != TMI_BA_OK) {
    std::fprintf(stderr, "[theia::BundleAdjustTwoViewsAngular] device batch failed: %s\n", tmi_ba_last_error());
    return out;  // success = false everywhere
  }
  for (size_t p = 0; p < P; ++p) {
    BundleAdjustmentSummary& s = out[p];
    s.setup_time_in_seconds = setup / static_cast<double>(P);
    s.solve_time_in_seconds = bs.seconds / static_cast<double>(P);
    if (!valid[p]) continue;
    s.initial_cost = c0[p];
    s.final_cost = cf[p];
    s.success = term[p] == 0 || term[p] == 1 || term[p] == -1;  // termination_type != FAILURE (:236-238)
    if (term[p] != 0 && term[p] != 1) continue;
    TwoViewInfo* info = (*problems)[p].info;
    for (int a = 0; a < 3; ++a) {
      info->rotation_2[a] = rot[3 * p + a];
      info->position_2[a] = pos[3 * p + a];
    }
  }
  return out;
}

BundleAdjustmentSummary BundleAdjustTwoViewsAngular(const BundleAdjustmentOptions& options,
                                                    const std::vector<FeatureCorrespondence>& correspondences,
                                                    TwoViewInfo* info) {
  std::vector<TwoViewAngularProblem> one(1);
  one[0].correspondences = &correspondences;
  one[0].info = info;
  const std::vector<BundleAdjustmentSummary> r = BundleAdjustTwoViewsAngularBatch(options, &one);
  return r.empty() ? BundleAdjustmentSummary() : r[0];
}

}  // namespace theia
