// reference: src/theia/sfm/track.h:50-88 (same accessors; Eigen-free storage)
#ifndef THEIA_MI355_SFM_TRACK_H_
#define THEIA_MI355_SFM_TRACK_H_
#include <unordered_set>
#include "theia/sfm/types.h"
#include "theia/util/eigen_lite.h"
namespace theia {
class Track {
 public:
  Track() : is_estimated_(false) { point_[3] = 1.0; }
  int NumViews() const { return static_cast<int>(view_ids_.size()); }
  void SetEstimated(const bool is_estimated) {
    if (is_estimated != is_estimated_) internal::BumpDataModelEpoch();
    is_estimated_ = is_estimated;
  }
  bool IsEstimated() const { return is_estimated_; }
  const Eigen::Vector4d& Point() const { return point_; }
  Eigen::Vector4d* MutablePoint() { return &point_; }
  void AddView(const ViewId view_id) {
    internal::BumpDataModelEpoch();
    view_ids_.insert(view_id);
  }
  bool RemoveView(const ViewId view_id) {
    internal::BumpDataModelEpoch();
    return view_ids_.erase(view_id) > 0;
  }
  const std::unordered_set<ViewId>& ViewIds() const { return view_ids_; }

 private:
  bool is_estimated_;
  std::unordered_set<ViewId> view_ids_;
  Eigen::Vector4d point_;
};
}  // namespace theia
#endif
