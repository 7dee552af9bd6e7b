// Host shim: TheiaSfM's BundleAdjuster / BundleAdjust* free functions on top of
// the MI355X C ABI.  Follows the reference's control flow line by line where it
// defines problem semantics (which residuals exist, which blocks are constant):
//   bundle_adjuster.cc:82-180   ctor, AddView, AddTrack
//   bundle_adjuster.cc:182-302  Optimize, parameterizations
//   bundle_adjustment.cc:47-107 the four free functions
// and replaces the Ceres problem / solve with flatten -> tmi_ba_solve -> write back.
#include "theia/sfm/bundle_adjustment/bundle_adjuster.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>

#include "theia/sfm/camera/camera.h"
#include "theia/sfm/reconstruction.h"

namespace theia {

tmi_ba_problem FlattenedBundleAdjustmentProblem::AsC() {
  tmi_ba_problem p;
  std::memset(&p, 0, sizeof(p));
  p.num_cameras = static_cast<int32_t>(view_ids.size());
  p.extrinsics = extrinsics.data();
  p.camera_group = camera_group.data();
  p.camera_flags = camera_flags.data();
  p.num_groups = static_cast<int32_t>(group_ids.size());
  p.group_model = group_model.data();
  p.group_offset = group_offset.data();
  p.intrinsics = intrinsics.data();
  p.intrinsics_constant = intrinsics_constant.data();
  p.num_points = static_cast<int32_t>(track_ids.size());
  p.points = points.data();
  p.point_constant = point_constant.data();
  p.num_observations = static_cast<int64_t>(obs_camera.size());
  p.obs_camera = obs_camera.data();
  p.obs_point = obs_point.data();
  p.obs_xy = obs_xy.data();
  return p;
}

BundleAdjuster::BundleAdjuster(const BundleAdjustmentOptions& options, Reconstruction* reconstruction)
    : options_(options), reconstruction_(reconstruction), timer_start_(std::chrono::steady_clock::now()) {
  std::memset(&device_summary_, 0, sizeof(device_summary_));
}

// bundle_adjuster.cc:102-139
void BundleAdjuster::AddView(const ViewId view_id) {
  View* view = reconstruction_ ? reconstruction_->MutableView(view_id) : nullptr;
  if (view == nullptr) return;  // the reference CHECK-fails; the shim never aborts
  if (!view->IsEstimated() || optimized_views_.count(view_id)) return;
  optimized_views_.emplace(view_id);
  SetCameraSchurGroups(view_id);
  optimized_camera_intrinsics_groups_.emplace(reconstruction_->CameraIntrinsicsGroupIdFromViewId(view_id));
  for (const TrackId track_id : view->TrackIds()) {
    const Feature* feature = view->GetFeature(track_id);
    Track* track = reconstruction_->MutableTrack(track_id);
    if (feature == nullptr || track == nullptr || !track->IsEstimated()) continue;
    AddReprojectionErrorResidual(*feature, view_id, track_id);
    SetTrackConstant(track_id);
  }
}

// bundle_adjuster.cc:141-180
void BundleAdjuster::AddTrack(const TrackId track_id) {
  Track* track = reconstruction_ ? reconstruction_->MutableTrack(track_id) : nullptr;
  if (track == nullptr) return;
  if (!track->IsEstimated() || optimized_tracks_.count(track_id)) return;
  optimized_tracks_.emplace(track_id);
  for (const ViewId view_id : track->ViewIds()) {
    View* view = reconstruction_->MutableView(view_id);
    if (view == nullptr) continue;
    if (optimized_views_.count(view_id) || !view->IsEstimated()) continue;
    const Feature* feature = view->GetFeature(track_id);
    if (feature == nullptr) continue;
    AddReprojectionErrorResidual(*feature, view_id, track_id);
    SetCameraExtrinsicsConstant(view_id);
    potentially_constant_camera_intrinsics_groups_.emplace(
        reconstruction_->CameraIntrinsicsGroupIdFromViewId(view_id));
  }
  SetTrackVariable(track_id);
  SetTrackSchurGroup(track_id);
}

// bundle_adjuster.cc:223-240
void BundleAdjuster::SetCameraExtrinsicsParameterization() {
  if (options_.constant_camera_orientation && options_.constant_camera_position) {
    for (const ViewId v : optimized_views_) SetCameraExtrinsicsConstant(v);
  } else if (options_.constant_camera_orientation) {
    for (const ViewId v : optimized_views_) SetCameraOrientationConstant(v);
  } else if (options_.constant_camera_position) {
    for (const ViewId v : optimized_views_) SetCameraPositionConstant(v);
  }
}

// bundle_adjuster.cc:242-287
void BundleAdjuster::SetCameraIntrinsicsParameterization() {
  for (const CameraIntrinsicsGroupId g : optimized_camera_intrinsics_groups_) {
    std::shared_ptr<CameraIntrinsicsModel> intr = GetIntrinsicsForCameraIntrinsicsGroup(g);
    if (!intr) continue;
    std::vector<uint8_t> mask(intr->NumParameters(), 0);
    for (const int idx : intr->GetSubsetFromOptimizeIntrinsicsType(options_.intrinsics_to_optimize)) mask[idx] = 1;
    intrinsics_constant_[g] = mask;
  }
  for (const CameraIntrinsicsGroupId g : potentially_constant_camera_intrinsics_groups_) {
    if (optimized_camera_intrinsics_groups_.count(g)) continue;
    std::shared_ptr<CameraIntrinsicsModel> intr = GetIntrinsicsForCameraIntrinsicsGroup(g);
    if (!intr) continue;
    intrinsics_constant_[g] = std::vector<uint8_t>(intr->NumParameters(), 1);
  }
}

// bundle_adjuster.cc:289-302
std::shared_ptr<CameraIntrinsicsModel> BundleAdjuster::GetIntrinsicsForCameraIntrinsicsGroup(
    const CameraIntrinsicsGroupId camera_intrinsics_group) {
  const auto views = reconstruction_->GetViewsInCameraIntrinsicGroup(camera_intrinsics_group);
  if (views.empty()) return nullptr;
  return reconstruction_->MutableView(*views.begin())->MutableCamera()->MutableCameraIntrinsics();
}

// bundle_adjuster.cc:304-344: constancy is recorded as flags instead of Ceres calls
void BundleAdjuster::SetCameraExtrinsicsConstant(const ViewId v) {
  camera_flags_[v] |= TMI_BA_CAMERA_POSITION_CONSTANT | TMI_BA_CAMERA_ORIENTATION_CONSTANT;
}
void BundleAdjuster::SetCameraPositionConstant(const ViewId v) { camera_flags_[v] |= TMI_BA_CAMERA_POSITION_CONSTANT; }
void BundleAdjuster::SetCameraOrientationConstant(const ViewId v) {
  camera_flags_[v] |= TMI_BA_CAMERA_ORIENTATION_CONSTANT;
}
void BundleAdjuster::SetTrackConstant(const TrackId t) { track_constant_[t] = true; }
void BundleAdjuster::SetTrackVariable(const TrackId t) { track_constant_[t] = false; }
// bundle_adjuster.cc:346-371: the elimination order (points, then intrinsics, then
// extrinsics) is structural in the device path; nothing to record.
void BundleAdjuster::SetCameraSchurGroups(const ViewId) {}
void BundleAdjuster::SetTrackSchurGroup(const TrackId) {}

// bundle_adjuster.cc:373-386
void BundleAdjuster::AddReprojectionErrorResidual(const Feature& feature, const ViewId view_id,
                                                  const TrackId track_id) {
  residuals_.push_back(Residual{view_id, track_id, feature.x(), feature.y()});
  camera_flags_.emplace(view_id, 0);
  track_constant_.emplace(track_id, true);
}

bool BundleAdjuster::Flatten(FlattenedBundleAdjustmentProblem* f) {
  if (f == nullptr || reconstruction_ == nullptr) return false;
  *f = FlattenedBundleAdjustmentProblem();
  SetCameraExtrinsicsParameterization();
  SetCameraIntrinsicsParameterization();
  std::map<ViewId, int> cam_index;
  std::map<TrackId, int> pt_index;
  std::map<CameraIntrinsicsGroupId, int> grp_index;
  for (const Residual& r : residuals_) {
    cam_index[r.view] = 0;
    pt_index[r.track] = 0;
    grp_index[reconstruction_->CameraIntrinsicsGroupIdFromViewId(r.view)] = 0;
  }
  int n = 0;
  for (auto& g : grp_index) {
    g.second = n++;
    f->group_ids.push_back(g.first);
  }
  f->group_offset.push_back(0);
  for (const CameraIntrinsicsGroupId g : f->group_ids) {
    std::shared_ptr<CameraIntrinsicsModel> intr = GetIntrinsicsForCameraIntrinsicsGroup(g);
    f->group_model.push_back(static_cast<int32_t>(intr->Type()));
    const int np = intr->NumParameters();
    f->intrinsics.insert(f->intrinsics.end(), intr->parameters(), intr->parameters() + np);
    auto it = intrinsics_constant_.find(g);
    if (it != intrinsics_constant_.end())
      f->intrinsics_constant.insert(f->intrinsics_constant.end(), it->second.begin(), it->second.end());
    else
      f->intrinsics_constant.insert(f->intrinsics_constant.end(), np, 1);
    f->group_offset.push_back(f->group_offset.back() + np);
  }
  n = 0;
  for (auto& c : cam_index) {
    c.second = n++;
    f->view_ids.push_back(c.first);
    const Camera& cam = reconstruction_->View(c.first)->Camera();
    f->extrinsics.insert(f->extrinsics.end(), cam.extrinsics(), cam.extrinsics() + 6);
    f->camera_group.push_back(grp_index[reconstruction_->CameraIntrinsicsGroupIdFromViewId(c.first)]);
    f->camera_flags.push_back(camera_flags_[c.first]);
  }
  n = 0;
  for (auto& p : pt_index) {
    p.second = n++;
    f->track_ids.push_back(p.first);
    const Eigen::Vector4d& X = reconstruction_->Track(p.first)->Point();
    f->points.insert(f->points.end(), X.data(), X.data() + 4);
    f->point_constant.push_back(track_constant_[p.first] ? 1 : 0);
  }
  // deterministic observation order: by (track, view)
  std::vector<Residual> sorted = residuals_;
  std::sort(sorted.begin(), sorted.end(), [](const Residual& a, const Residual& b) {
    return a.track != b.track ? a.track < b.track : a.view < b.view;
  });
  for (const Residual& r : sorted) {
    f->obs_camera.push_back(cam_index[r.view]);
    f->obs_point.push_back(pt_index[r.track]);
    f->obs_xy.push_back(r.x);
    f->obs_xy.push_back(r.y);
  }
  return true;
}

static int ToAbiSolver(ceres::LinearSolverType t) {
  switch (t) {
    case ceres::DENSE_QR: return TMI_BA_DENSE_QR;
    case ceres::DENSE_SCHUR: return TMI_BA_DENSE_SCHUR;
    case ceres::SPARSE_SCHUR: return TMI_BA_SPARSE_SCHUR;
    case ceres::ITERATIVE_SCHUR: return TMI_BA_ITERATIVE_SCHUR;
    case ceres::CGNR: return TMI_BA_CGNR;
    default: return TMI_BA_SPARSE_SCHUR;  // *_NORMAL_CHOLESKY: same normal equations, solved exactly
  }
}

// BundleAdjustmentOptions -> tmi_ba_options, field for field (SetSolverOptions,
// bundle_adjuster.cc:57-79)
void ToDeviceOptions(const BundleAdjustmentOptions& options, tmi_ba_options* o) {
  tmi_ba_options_init(o);
  o->loss_function_type = static_cast<int32_t>(options.loss_function_type);
  o->robust_loss_width = options.robust_loss_width;
  o->linear_solver_type = ToAbiSolver(options.linear_solver_type);
  o->preconditioner_type = static_cast<int32_t>(options.preconditioner_type);
  o->verbose = options.verbose ? 1 : 0;
  o->num_threads = options.num_threads;
  o->max_num_iterations = options.max_num_iterations;
  o->max_solver_time_in_seconds = options.max_solver_time_in_seconds;
  o->use_inner_iterations = options.use_inner_iterations ? 1 : 0;
  o->function_tolerance = options.function_tolerance;
  o->gradient_tolerance = options.gradient_tolerance;
  o->parameter_tolerance = options.parameter_tolerance;
  o->max_trust_region_radius = options.max_trust_region_radius;
  o->point_dof = options.point_dof;
  o->device = options.device;
}

// bundle_adjuster.cc:182-221
BundleAdjustmentSummary BundleAdjuster::Optimize() {
  BundleAdjustmentSummary summary;
  FlattenedBundleAdjustmentProblem flat;
  if (!Flatten(&flat)) return summary;
  const double internal_setup_time =
      std::chrono::duration<double>(std::chrono::steady_clock::now() - timer_start_).count();
  if (flat.obs_camera.empty()) {
    // nothing to optimise: Ceres reports a usable, zero-cost solve
    summary.success = true;
    summary.setup_time_in_seconds = internal_setup_time;
    return summary;
  }
  tmi_ba_options o;
  ToDeviceOptions(options_, &o);
  tmi_ba_problem p = flat.AsC();
  tmi_ba_solve(&p, &o, &device_summary_);
  summary.success = device_summary_.success != 0;
  summary.initial_cost = device_summary_.initial_cost;
  summary.final_cost = device_summary_.final_cost;
  summary.setup_time_in_seconds = internal_setup_time + device_summary_.setup_time_in_seconds;
  summary.solve_time_in_seconds = device_summary_.solve_time_in_seconds;
  if (options_.verbose)
    std::fprintf(stderr, "[theia::BundleAdjuster] %s: cost %.9e -> %.9e, %d iterations, rmse %.6f px\n",
                 device_summary_.message, summary.initial_cost, summary.final_cost,
                 device_summary_.num_iterations, device_summary_.final_rmse);
  if (!summary.success) return summary;
  // write back in place (Ceres updates the caller's arrays through raw pointers)
  for (size_t c = 0; c < flat.view_ids.size(); ++c) {
    double* e = reconstruction_->MutableView(flat.view_ids[c])->MutableCamera()->mutable_extrinsics();
    std::copy(flat.extrinsics.begin() + 6 * c, flat.extrinsics.begin() + 6 * c + 6, e);
  }
  for (size_t g = 0; g < flat.group_ids.size(); ++g) {
    std::shared_ptr<CameraIntrinsicsModel> intr = GetIntrinsicsForCameraIntrinsicsGroup(flat.group_ids[g]);
    std::copy(flat.intrinsics.begin() + flat.group_offset[g], flat.intrinsics.begin() + flat.group_offset[g + 1],
              intr->mutable_parameters());
  }
  for (size_t t = 0; t < flat.track_ids.size(); ++t) {
    double* X = reconstruction_->MutableTrack(flat.track_ids[t])->MutablePoint()->data();
    std::copy(flat.points.begin() + 4 * t, flat.points.begin() + 4 * t + 4, X);
  }
  return summary;
}

// ---- bundle_adjustment.cc:47-107 ----------------------------------------------------
BundleAdjustmentSummary BundleAdjustPartialReconstruction(const BundleAdjustmentOptions& options,
                                                          const std::unordered_set<ViewId>& view_ids,
                                                          const std::unordered_set<TrackId>& track_ids,
                                                          Reconstruction* reconstruction) {
  BundleAdjuster bundle_adjuster(options, reconstruction);
  for (const ViewId v : view_ids) bundle_adjuster.AddView(v);
  for (const TrackId t : track_ids) bundle_adjuster.AddTrack(t);
  return bundle_adjuster.Optimize();
}

BundleAdjustmentSummary BundleAdjustReconstruction(const BundleAdjustmentOptions& options,
                                                   Reconstruction* reconstruction) {
  BundleAdjuster bundle_adjuster(options, reconstruction);
  if (reconstruction == nullptr) return BundleAdjustmentSummary();
  for (const ViewId v : reconstruction->ViewIds()) bundle_adjuster.AddView(v);
  for (const TrackId t : reconstruction->TrackIds()) bundle_adjuster.AddTrack(t);
  return bundle_adjuster.Optimize();
}

BundleAdjustmentSummary BundleAdjustView(const BundleAdjustmentOptions& options, const ViewId view_id,
                                         Reconstruction* reconstruction) {
  BundleAdjustmentOptions ba_options = options;
  ba_options.linear_solver_type = ceres::DENSE_QR;
  ba_options.use_inner_iterations = false;
  BundleAdjuster bundle_adjuster(ba_options, reconstruction);
  bundle_adjuster.AddView(view_id);
  return bundle_adjuster.Optimize();
}

BundleAdjustmentSummary BundleAdjustTrack(const BundleAdjustmentOptions& options, const TrackId track_id,
                                          Reconstruction* reconstruction) {
  BundleAdjustmentOptions ba_options = options;
  ba_options.linear_solver_type = ceres::DENSE_QR;
  ba_options.use_inner_iterations = false;
  BundleAdjuster bundle_adjuster(ba_options, reconstruction);
  bundle_adjuster.AddTrack(track_id);
  return bundle_adjuster.Optimize();
}

}  // namespace theia
