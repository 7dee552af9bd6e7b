// reference: src/theia/sfm/reconstruction.h:66-181, reconstruction.cc:93-223,352-367
// The container subset the BA path reads and writes, with the reference's
// intrinsics-group sharing: views added to an existing group point at the same
// CameraIntrinsicsModel object (reconstruction.cc:113-124).
#ifndef THEIA_MI355_SFM_RECONSTRUCTION_H_
#define THEIA_MI355_SFM_RECONSTRUCTION_H_
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>
#include "theia/sfm/track.h"
#include "theia/sfm/types.h"
#include "theia/sfm/view.h"
namespace theia {
class Reconstruction {
 public:
  Reconstruction() : next_track_id_(0), next_view_id_(0), next_camera_intrinsics_group_id_(0) {}

  ViewId ViewIdFromName(const std::string& view_name) const {
    auto it = view_name_to_id_.find(view_name);
    return it == view_name_to_id_.end() ? kInvalidViewId : it->second;
  }
  ViewId AddView(const std::string& view_name) {
    const ViewId id = AddView(view_name, next_camera_intrinsics_group_id_);
    ++next_camera_intrinsics_group_id_;
    return id;
  }
  ViewId AddView(const std::string& view_name, const CameraIntrinsicsGroupId group_id) {
    if (view_name.empty() || view_name_to_id_.count(view_name)) return kInvalidViewId;
    internal::BumpDataModelEpoch();
    class View new_view(view_name);
    auto& group = camera_intrinsics_groups_[group_id];
    if (!group.empty()) {
      const ViewId other = *group.begin();
      new_view.MutableCamera()->MutableCameraIntrinsics() = views_.at(other).Camera().CameraIntrinsics();
    }
    views_.emplace(next_view_id_, new_view);
    view_name_to_id_.emplace(view_name, next_view_id_);
    view_id_to_camera_intrinsics_group_id_.emplace(next_view_id_, group_id);
    group.emplace(next_view_id_);
    if (group_id >= next_camera_intrinsics_group_id_) next_camera_intrinsics_group_id_ = group_id + 1;
    ++next_view_id_;
    return next_view_id_ - 1;
  }
  int NumViews() const { return static_cast<int>(views_.size()); }
  const class View* View(const ViewId id) const {
    auto it = views_.find(id);
    return it == views_.end() ? nullptr : &it->second;
  }
  class View* MutableView(const ViewId id) {
    auto it = views_.find(id);
    return it == views_.end() ? nullptr : &it->second;
  }
  std::vector<ViewId> ViewIds() const {
    std::vector<ViewId> ids;
    ids.reserve(views_.size());
    for (const auto& v : views_) ids.push_back(v.first);
    return ids;
  }
  CameraIntrinsicsGroupId CameraIntrinsicsGroupIdFromViewId(const ViewId id) const {
    auto it = view_id_to_camera_intrinsics_group_id_.find(id);
    return it == view_id_to_camera_intrinsics_group_id_.end() ? kInvalidCameraIntrinsicsGroupId : it->second;
  }
  std::unordered_set<ViewId> GetViewsInCameraIntrinsicGroup(const CameraIntrinsicsGroupId g) const {
    auto it = camera_intrinsics_groups_.find(g);
    return it == camera_intrinsics_groups_.end() ? std::unordered_set<ViewId>() : it->second;
  }
  int NumCameraIntrinsicGroups() const { return static_cast<int>(camera_intrinsics_groups_.size()); }

  TrackId AddTrack() {
    internal::BumpDataModelEpoch();
    class Track new_track;
    tracks_.emplace(next_track_id_, new_track);
    return next_track_id_++;
  }
  bool AddObservation(const ViewId view_id, const TrackId track_id, const Feature& feature) {
    class View* view = MutableView(view_id);
    class Track* track = MutableTrack(track_id);
    if (view == nullptr || track == nullptr) return false;
    if (view->GetFeature(track_id) != nullptr) return false;
    view->AddFeature(track_id, feature);
    track->AddView(view_id);
    return true;
  }
  TrackId AddTrack(const std::vector<std::pair<ViewId, Feature> >& track) {
    if (track.size() < 2) return kInvalidTrackId;
    std::unordered_set<ViewId> seen;
    for (const auto& o : track)
      if (!seen.insert(o.first).second || View(o.first) == nullptr) return kInvalidTrackId;
    const TrackId id = AddTrack();
    for (const auto& o : track) AddObservation(o.first, id, o.second);
    return id;
  }
  int NumTracks() const { return static_cast<int>(tracks_.size()); }
  const class Track* Track(const TrackId id) const {
    auto it = tracks_.find(id);
    return it == tracks_.end() ? nullptr : &it->second;
  }
  class Track* MutableTrack(const TrackId id) {
    auto it = tracks_.find(id);
    return it == tracks_.end() ? nullptr : &it->second;
  }
  std::vector<TrackId> TrackIds() const {
    std::vector<TrackId> ids;
    ids.reserve(tracks_.size());
    for (const auto& t : tracks_) ids.push_back(t.first);
    return ids;
  }

  // Extension: identity of this object for the shim's resident-session cache (types.h)
  std::uint64_t Uid() const { return uid_.value; }

 private:
  internal::ObjectUid uid_;
  TrackId next_track_id_;
  ViewId next_view_id_;
  CameraIntrinsicsGroupId next_camera_intrinsics_group_id_;
  std::unordered_map<std::string, ViewId> view_name_to_id_;
  std::unordered_map<ViewId, class View> views_;
  std::unordered_map<TrackId, class Track> tracks_;
  std::unordered_map<ViewId, CameraIntrinsicsGroupId> view_id_to_camera_intrinsics_group_id_;
  std::unordered_map<CameraIntrinsicsGroupId, std::unordered_set<ViewId> > camera_intrinsics_groups_;
};
}  // namespace theia
#endif
