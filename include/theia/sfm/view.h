// reference: src/theia/sfm/view.h:57-101
#ifndef THEIA_MI355_SFM_VIEW_H_
#define THEIA_MI355_SFM_VIEW_H_
#include <string>
#include <unordered_map>
#include <vector>
#include "theia/sfm/camera/camera.h"
#include "theia/sfm/feature.h"
#include "theia/sfm/types.h"
namespace theia {
class View {
 public:
  View() : name_(""), is_estimated_(false) {}
  explicit View(const std::string& name) : name_(name), is_estimated_(false) {}
  const std::string& Name() const { return name_; }
  void SetEstimated(bool is_estimated) {
    if (is_estimated != is_estimated_) internal::BumpDataModelEpoch();
    is_estimated_ = is_estimated;
  }
  bool IsEstimated() const { return is_estimated_; }
  const class Camera& Camera() const { return camera_; }
  class Camera* MutableCamera() { return &camera_; }
  int NumFeatures() const { return static_cast<int>(features_.size()); }
  std::vector<TrackId> TrackIds() const {
    std::vector<TrackId> ids;
    ids.reserve(features_.size());
    for (const auto& f : features_) ids.push_back(f.first);
    return ids;
  }
  const Feature* GetFeature(const TrackId track_id) const {
    auto it = features_.find(track_id);
    return it == features_.end() ? nullptr : &it->second;
  }
  // Extension: the (track id -> feature) table itself, for callers that walk every feature
  // (the reference's TrackIds() + GetFeature() pair costs a copy and a look-up per feature).
  const std::unordered_map<TrackId, Feature>& Features() const { return features_; }
  void AddFeature(const TrackId track_id, const Feature& feature) {
    internal::BumpDataModelEpoch();
    features_[track_id] = feature;
  }
  bool RemoveFeature(const TrackId track_id) {
    internal::BumpDataModelEpoch();
    return features_.erase(track_id) > 0;
  }

 private:
  std::string name_;
  bool is_estimated_;
  class Camera camera_;
  std::unordered_map<TrackId, Feature> features_;
};
}  // namespace theia
#endif
