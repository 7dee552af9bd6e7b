// Source-compatible mirror of
//   /root/reference/src/theia/sfm/select_good_tracks_for_bundle_adjustment.h:50-82
// The track statistics (5 M projections on a Venice-sized scene) run on the MI355X through
// tmi_ba_select_good_tracks; grid binning and ranking follow on the host inside the engine.
#ifndef THEIA_MI355_SFM_SELECT_GOOD_TRACKS_FOR_BUNDLE_ADJUSTMENT_H_
#define THEIA_MI355_SFM_SELECT_GOOD_TRACKS_FOR_BUNDLE_ADJUSTMENT_H_
#include <unordered_set>
#include "theia/sfm/types.h"

namespace theia {
class Reconstruction;

// Chooses a subset of the estimated tracks for bundle adjustment: per image, the best
// ranked track of every grid cell (spatial coverage), then the top ranked remaining tracks
// until every view is constrained by min_num_optimized_tracks_per_view tracks.
// Returns false only if the device path failed (the reference always returns true).
bool SelectGoodTracksForBundleAdjustment(const Reconstruction& reconstruction,
                                         const int long_track_length_threshold,
                                         const int image_grid_cell_size_pixels,
                                         const int min_num_optimized_tracks_per_view,
                                         std::unordered_set<TrackId>* tracks_to_optimize);
// Same, considering only the given views.
bool SelectGoodTracksForBundleAdjustment(const Reconstruction& reconstruction,
                                         const std::unordered_set<ViewId>& view_ids,
                                         const int long_track_length_threshold,
                                         const int image_grid_cell_size_pixels,
                                         const int min_num_optimized_tracks_per_view,
                                         std::unordered_set<TrackId>* tracks_to_optimize);
}  // namespace theia
#endif
