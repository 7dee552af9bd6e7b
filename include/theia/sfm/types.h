// Id types of the data model, as in the reference (src/theia/sfm/types.h:47-53): 32-bit
// unsigned ids, the all-ones value marks "no such id".
#ifndef THEIA_MI355_SFM_TYPES_H_
#define THEIA_MI355_SFM_TYPES_H_
#include <atomic>
#include <cstdint>
namespace theia {
using ViewId = std::uint32_t;
using TrackId = std::uint32_t;
using CameraIntrinsicsGroupId = std::uint32_t;
constexpr ViewId kInvalidViewId = UINT32_MAX;
constexpr TrackId kInvalidTrackId = UINT32_MAX;
constexpr CameraIntrinsicsGroupId kInvalidCameraIntrinsicsGroupId = UINT32_MAX;

// Extension (not in the reference): a process-wide mutation stamp of the data model.  Every call that can change
// which residuals a bundle adjustment holds -- adding / removing views, tracks, observations, flipping an
// is_estimated flag -- bumps it; parameter VALUES (camera poses, points, intrinsics) do not.  The host shim keeps the
// flattened problem and the device-resident solver of the last BundleAdjustReconstruction alive and re-uses them when
// the same Reconstruction comes back with the stamp unchanged (theiasfm_amd/host/bundle_adjuster.cc); a TheiaSfM
// maintainer adds the same one-line bump to the corresponding mutators of the real classes.
namespace internal {
inline std::atomic<std::uint64_t>& DataModelEpoch() {
  static std::atomic<std::uint64_t> epoch(1);
  return epoch;
}
inline void BumpDataModelEpoch() { DataModelEpoch().fetch_add(1, std::memory_order_relaxed); }
// distinguishes a Reconstruction from another one that later lives at the same address (copies get a new id)
struct ObjectUid {
  std::uint64_t value;
  ObjectUid() : value(Next()) {}
  ObjectUid(const ObjectUid&) : value(Next()) {}
  ObjectUid& operator=(const ObjectUid&) {
    value = Next();
    return *this;
  }
  static std::uint64_t Next() {
    static std::atomic<std::uint64_t> next(1);
    return next.fetch_add(1, std::memory_order_relaxed);
  }
};
}  // namespace internal
}  // namespace theia
#endif
