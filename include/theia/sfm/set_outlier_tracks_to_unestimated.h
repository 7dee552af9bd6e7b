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

// Post-BA clean-up.  A track of `tracks` loses its estimated flag when (a) one of its views sees
// the point behind the camera, (b) its mean squared reprojection error over the estimated views
// exceeds max_inlier_reprojection_error^2, or (c) no two viewing rays are at least
// min_triangulation_angle_degrees apart.  The return value counts the tracks dropped by (a)-(c)
// (-1 if the device path failed; the reference cannot fail).
int SetOutlierTracksToUnestimated(const std::unordered_set<TrackId>& tracks,
                                  const double max_inlier_reprojection_error,
                                  const double min_triangulation_angle_degrees,
                                  Reconstruction* reconstruction);
// The same over every track of the reconstruction.
int SetOutlierTracksToUnestimated(const double max_inlier_reprojection_error,
                                  const double min_triangulation_angle_degrees,
                                  Reconstruction* reconstruction);
}  // namespace theia
#endif
