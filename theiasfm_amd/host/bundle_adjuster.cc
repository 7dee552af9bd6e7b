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

namespace {
// id -> dense index over a sorted list of distinct ids: direct table when the ids are
// compact (Reconstruction hands them out consecutively), binary search otherwise
template <class Id>
struct IdIndex {
  std::vector<Id> ids;       // ascending, distinct
  std::vector<int> table;    // [max id + 1] or empty
  void Build(std::vector<Id>* raw) {
    std::sort(raw->begin(), raw->end());
    raw->erase(std::unique(raw->begin(), raw->end()), raw->end());
    ids.swap(*raw);
    table.clear();
    if (!ids.empty() && static_cast<uint64_t>(ids.back()) < 8ull * ids.size() + (1u << 20)) {
      table.assign(static_cast<size_t>(ids.back()) + 1, -1);
      for (size_t i = 0; i < ids.size(); ++i) table[ids[i]] = static_cast<int>(i);
    }
  }
  int operator()(Id id) const {
    if (!table.empty()) return table[id];
    return static_cast<int>(std::lower_bound(ids.begin(), ids.end(), id) - ids.begin());
  }
};
}  // namespace

bool BundleAdjuster::Flatten(FlattenedBundleAdjustmentProblem* f) {
  if (f == nullptr || reconstruction_ == nullptr) return false;
  *f = FlattenedBundleAdjustmentProblem();
  SetCameraExtrinsicsParameterization();
  SetCameraIntrinsicsParameterization();
  // sorted ids => deterministic block order (Ceres follows unordered_map iteration)
  IdIndex<ViewId> cam_index;
  IdIndex<TrackId> pt_index;
  IdIndex<CameraIntrinsicsGroupId> grp_index;
  {
    std::vector<ViewId> v;
    std::vector<TrackId> t;
    v.reserve(residuals_.size());
    t.reserve(residuals_.size());
    for (const Residual& r : residuals_) {
      v.push_back(r.view);
      t.push_back(r.track);
    }
    cam_index.Build(&v);
    pt_index.Build(&t);
    std::vector<CameraIntrinsicsGroupId> g;
    g.reserve(cam_index.ids.size());
    for (const ViewId id : cam_index.ids) g.push_back(reconstruction_->CameraIntrinsicsGroupIdFromViewId(id));
    grp_index.Build(&g);
  }
  f->group_ids = grp_index.ids;
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
  f->view_ids = cam_index.ids;
  f->extrinsics.reserve(6 * f->view_ids.size());
  for (const ViewId id : f->view_ids) {
    const Camera& cam = reconstruction_->View(id)->Camera();
    f->extrinsics.insert(f->extrinsics.end(), cam.extrinsics(), cam.extrinsics() + 6);
    f->camera_group.push_back(grp_index(reconstruction_->CameraIntrinsicsGroupIdFromViewId(id)));
    f->camera_flags.push_back(camera_flags_[id]);
  }
  f->track_ids = pt_index.ids;
  f->points.reserve(4 * f->track_ids.size());
  for (const TrackId id : f->track_ids) {
    const Eigen::Vector4d& X = reconstruction_->Track(id)->Point();
    f->points.insert(f->points.end(), X.data(), X.data() + 4);
    f->point_constant.push_back(track_constant_[id] ? 1 : 0);
  }
  // deterministic observation order: by (track, view) -- a counting sort by track, then each
  // track's few observations by view
  const size_t n = residuals_.size(), np = f->track_ids.size();
  std::vector<int> pt_of(n);
  std::vector<size_t> first(np + 1, 0);
  for (size_t i = 0; i < n; ++i) {
    pt_of[i] = pt_index(residuals_[i].track);
    first[pt_of[i] + 1]++;
  }
  for (size_t p = 0; p < np; ++p) first[p + 1] += first[p];
  std::vector<uint32_t> order(n);
  {
    std::vector<size_t> fill(first.begin(), first.end() - 1);
    for (size_t i = 0; i < n; ++i) order[fill[pt_of[i]]++] = static_cast<uint32_t>(i);
  }
  for (size_t p = 0; p < np; ++p)
    std::sort(order.begin() + first[p], order.begin() + first[p + 1],
              [&](uint32_t a, uint32_t b) { return residuals_[a].view < residuals_[b].view; });
  // A (view, track) pair can have been pushed twice when AddTrack(t) pulled in a view that a
  // later AddView(v) adds again -- a call order outside the class contract ("AddView before
  // AddTrack", bundle_adjuster.h:58-59; the free functions follow it).  The reference would
  // then hold the residual block twice; the device layout holds an observation once, so the
  // duplicate is dropped here (the view's extrinsics stay constant, as in the reference).
  f->obs_camera.reserve(n);
  f->obs_point.reserve(n);
  f->obs_xy.reserve(2 * n);
  for (size_t q = 0; q < n; ++q) {
    const Residual& r = residuals_[order[q]];
    if (q > 0) {
      const Residual& prev = residuals_[order[q - 1]];
      if (prev.view == r.view && prev.track == r.track) continue;
    }
    f->obs_camera.push_back(cam_index(r.view));
    f->obs_point.push_back(pt_of[order[q]]);
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
