// Host shim for the steps either side of the full adjustment:
//   SetOutlierTracksToUnestimated   set_outlier_tracks_to_unestimated.cc:49-133
//   BundleAdjustTracks              batched form of bundle_adjustment.cc:96-107
//   SelectGoodTracksForBundleAdjustment   select_good_tracks_for_bundle_adjustment.cc:251-327
// Both flatten "the estimated views observing these estimated tracks" -- exactly the
// residual set BundleAdjuster::AddTrack builds (bundle_adjuster.cc:141-180) -- and hand it
// to the C ABI.
#include <cstdio>
#include <vector>

#include "theia/sfm/bundle_adjustment/bundle_adjuster.h"
#include "theia/sfm/bundle_adjustment/bundle_adjustment.h"
#include "theia/sfm/reconstruction.h"
#include "theia/sfm/select_good_tracks_for_bundle_adjustment.h"
#include "theia/sfm/set_outlier_tracks_to_unestimated.h"

namespace theia {
namespace {
bool FlattenTracks(const BundleAdjustmentOptions& options, const std::unordered_set<TrackId>& track_ids,
                   Reconstruction* reconstruction, FlattenedBundleAdjustmentProblem* flat) {
  BundleAdjuster adjuster(options, reconstruction);
  for (const TrackId t : track_ids) adjuster.AddTrack(t);
  return adjuster.Flatten(flat);
}
}  // namespace

int SetOutlierTracksToUnestimated(const double max_inlier_reprojection_error,
                                  const double min_triangulation_angle_degrees,
                                  Reconstruction* reconstruction) {
  if (reconstruction == nullptr) return 0;
  const auto ids = reconstruction->TrackIds();
  const std::unordered_set<TrackId> all_tracks(ids.begin(), ids.end());
  return SetOutlierTracksToUnestimated(all_tracks, max_inlier_reprojection_error,
                                       min_triangulation_angle_degrees, reconstruction);
}

int SetOutlierTracksToUnestimated(const std::unordered_set<TrackId>& track_ids,
                                  const double max_inlier_reprojection_error,
                                  const double min_triangulation_angle_degrees,
                                  Reconstruction* reconstruction) {
  if (reconstruction == nullptr) return 0;
  FlattenedBundleAdjustmentProblem flat;
  if (!FlattenTracks(BundleAdjustmentOptions(), track_ids, reconstruction, &flat)) return -1;
  int num_bad_reprojections = 0, num_insufficient_viewing_angles = 0;
  std::unordered_set<TrackId> observed(flat.track_ids.begin(), flat.track_ids.end());
  if (!flat.track_ids.empty()) {
    std::vector<uint8_t> flag(flat.track_ids.size(), 0);
    tmi_ba_filter_summary fs;
    tmi_ba_problem p = flat.AsC();
    const int rc = tmi_ba_filter_outlier_tracks(&p, -1, max_inlier_reprojection_error,
                                                min_triangulation_angle_degrees, flag.data(), nullptr, &fs);
    if (rc != TMI_BA_OK) {
      std::fprintf(stderr, "[theia::SetOutlierTracksToUnestimated] device filter failed: %s\n",
                   tmi_ba_last_error());
      return -1;
    }
    for (size_t t = 0; t < flat.track_ids.size(); ++t) {
      if (flag[t] == 0) continue;
      reconstruction->MutableTrack(flat.track_ids[t])->SetEstimated(false);
    }
    num_bad_reprojections = static_cast<int>(fs.num_bad_reprojections);
    num_insufficient_viewing_angles = static_cast<int>(fs.num_insufficient_viewing_angles);
  }
  // an estimated track none of whose views is estimated has no ray pair: the reference
  // removes it for an insufficient viewing angle (:120-125 with an empty ray list)
  for (const TrackId t : track_ids) {
    Track* track = reconstruction->MutableTrack(t);
    if (track == nullptr || !track->IsEstimated() || observed.count(t)) continue;
    track->SetEstimated(false);
    ++num_insufficient_viewing_angles;
  }
  return num_bad_reprojections + num_insufficient_viewing_angles;
}

std::unordered_map<TrackId, BundleAdjustmentSummary> BundleAdjustTracks(
    const BundleAdjustmentOptions& options, const std::unordered_set<TrackId>& track_ids,
    Reconstruction* reconstruction) {
  std::unordered_map<TrackId, BundleAdjustmentSummary> result;
  if (reconstruction == nullptr) return result;
  BundleAdjustmentOptions ba_options = options;
  ba_options.linear_solver_type = ceres::DENSE_QR;  // bundle_adjustment.cc:100-101
  ba_options.use_inner_iterations = false;
  FlattenedBundleAdjustmentProblem flat;
  if (!FlattenTracks(ba_options, track_ids, reconstruction, &flat)) return result;
  // estimated tracks without a residual: Optimize() of an empty problem succeeds at zero cost
  for (const TrackId t : track_ids) {
    const Track* track = reconstruction->Track(t);
    if (track == nullptr || !track->IsEstimated()) continue;
    BundleAdjustmentSummary s;
    s.success = true;
    result.emplace(t, s);
  }
  if (flat.track_ids.empty()) return result;
  tmi_ba_options o;
  ToDeviceOptions(ba_options, &o);
  const size_t n = flat.track_ids.size();
  std::vector<int8_t> termination(n, -1);
  std::vector<double> initial_cost(n, 0.0), final_cost(n, 0.0);
  tmi_ba_track_batch_summary ts;
  tmi_ba_problem p = flat.AsC();
  const int rc = tmi_ba_adjust_tracks(&p, &o, termination.data(), nullptr, initial_cost.data(),
                                      final_cost.data(), &ts);
  for (size_t t = 0; t < n; ++t) {
    BundleAdjustmentSummary& s = result[flat.track_ids[t]];
    s.success = rc == TMI_BA_OK && (termination[t] == 0 || termination[t] == 1);
    s.initial_cost = initial_cost[t];
    s.final_cost = final_cost[t];
    s.solve_time_in_seconds = ts.kernel_seconds / static_cast<double>(n);
    s.setup_time_in_seconds = (ts.seconds - ts.kernel_seconds) / static_cast<double>(n);
    if (!s.success) continue;
    double* X = reconstruction->MutableTrack(flat.track_ids[t])->MutablePoint()->data();
    std::copy(flat.points.begin() + 4 * t, flat.points.begin() + 4 * t + 4, X);
  }
  return result;
}

// select_good_tracks_for_bundle_adjustment.cc:264-277
bool SelectGoodTracksForBundleAdjustment(const Reconstruction& reconstruction,
                                         const int long_track_length_threshold,
                                         const int image_grid_cell_size_pixels,
                                         const int min_num_optimized_tracks_per_view,
                                         std::unordered_set<TrackId>* tracks_to_optimize) {
  std::unordered_set<ViewId> view_ids;  // GetEstimatedViewsFromReconstruction
  for (const ViewId v : reconstruction.ViewIds())
    if (reconstruction.View(v)->IsEstimated()) view_ids.insert(v);
  return SelectGoodTracksForBundleAdjustment(reconstruction, view_ids, long_track_length_threshold,
                                             image_grid_cell_size_pixels,
                                             min_num_optimized_tracks_per_view, tracks_to_optimize);
}

// select_good_tracks_for_bundle_adjustment.cc:280-327
bool SelectGoodTracksForBundleAdjustment(const Reconstruction& reconstruction,
                                         const std::unordered_set<ViewId>& view_ids,
                                         const int long_track_length_threshold,
                                         const int image_grid_cell_size_pixels,
                                         const int min_num_optimized_tracks_per_view,
                                         std::unordered_set<TrackId>* tracks_to_optimize) {
  if (tracks_to_optimize == nullptr) return false;
  // the estimated tracks seen by these views, with ALL their estimated views (the track
  // statistics run over every view of a track, :92-104)
  std::unordered_set<TrackId> tracks;
  for (const ViewId v : view_ids) {
    const View* view = reconstruction.View(v);
    if (view == nullptr) continue;
    for (const TrackId t : view->TrackIds()) {
      const Track* track = reconstruction.Track(t);
      if (track != nullptr && track->IsEstimated()) tracks.insert(t);
    }
  }
  FlattenedBundleAdjustmentProblem flat;
  // flattening reads the reconstruction only
  if (!FlattenTracks(BundleAdjustmentOptions(), tracks, const_cast<Reconstruction*>(&reconstruction), &flat))
    return false;
  if (flat.track_ids.empty()) return true;
  std::vector<uint8_t> view_mask(flat.view_ids.size(), 0), selected(flat.track_ids.size(), 0);
  for (size_t c = 0; c < flat.view_ids.size(); ++c) view_mask[c] = view_ids.count(flat.view_ids[c]) ? 1 : 0;
  tmi_ba_select_summary ss;
  tmi_ba_problem p = flat.AsC();
  const int rc = tmi_ba_select_good_tracks(&p, -1, long_track_length_threshold, image_grid_cell_size_pixels,
                                           min_num_optimized_tracks_per_view, view_mask.data(),
                                           selected.data(), nullptr, nullptr, &ss);
  if (rc != TMI_BA_OK) {
    std::fprintf(stderr, "[theia::SelectGoodTracksForBundleAdjustment] device path failed: %s\n",
                 tmi_ba_last_error());
    return false;
  }
  for (size_t t = 0; t < flat.track_ids.size(); ++t)
    if (selected[t]) tracks_to_optimize->insert(flat.track_ids[t]);
  return true;
}

}  // namespace theia
