// Source-compatible mirror of
//   /root/reference/src/theia/sfm/set_outlier_tracks_to_unestimated.h:45-63
// The per-track tests (reprojection error, cheirality, viewing angle) run on the MI355X
// through tmi_ba_filter_outlier_tracks; this function flattens the estimated tracks and
// views, and applies the flags.
#ifndef THEIA_MI355_SFM_SET_OUTLIER_TRACKS_TO_UNESTIMATED_H_
#define THEIA_MI355_SFM_SET_OUTLIER_TRACKS_TO_UNESTIMATED_H_
#include <unordered_set>
#include "theia/sfm/types.h"

namespace theia {
class Reconstruction;

// Removes features that have a reprojection error larger than the reprojection error
// threshold. Additionally, any features that are poorly constrained because of a small
// viewing angle are removed. Returns the number of features removed (-1 if the device path
// failed; the reference cannot fail). Only the input tracks are checked.
int SetOutlierTracksToUnestimated(const std::unordered_set<TrackId>& tracks,
                                  const double max_inlier_reprojection_error,
                                  const double min_triangulation_angle_degrees,
                                  Reconstruction* reconstruction);
// Same as above, but checks all tracks.
int SetOutlierTracksToUnestimated(const double max_inlier_reprojection_error,
                                  const double min_triangulation_angle_degrees,
                                  Reconstruction* reconstruction);
}  // namespace theia
#endif
