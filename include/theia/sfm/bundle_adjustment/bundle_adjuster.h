// Source-compatible mirror of
//   /root/reference/src/theia/sfm/bundle_adjustment/bundle_adjuster.h:60-132.
// Same public API and the same AddView / AddTrack residual-set semantics
// (bundle_adjuster.cc:102-180); Optimize() flattens the recorded problem into
// tmi_ba_problem and calls tmi_ba_solve() where the reference calls ceres::Solve
// (bundle_adjuster.cc:205), then writes the parameters back in place.
//
// NOTE (reference contract, bundle_adjuster.h:58-59): AddView must be called
// before AddTrack for any view that is to be optimised.
#ifndef THEIA_MI355_BUNDLE_ADJUSTER_H_
#define THEIA_MI355_BUNDLE_ADJUSTER_H_
#include <sys/mman.h>

#include <chrono>
#include <cstdlib>
#include <new>
#include <cstdint>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "theia/sfm/bundle_adjustment/bundle_adjustment.h"
#include "theia/sfm/feature.h"
#include "theia/sfm/types.h"
#include "theia_mi355_ba.h"

namespace theia {
class Camera;
class CameraIntrinsicsModel;
class Reconstruction;
class Track;

// Allocator whose value-less construct() default-initialises: vector::resize(n) of a trivial type then allocates
// without zero-filling.  The per-observation arrays below (and the residual list) are 40-120 MB each at Venice size and
// are filled by threads right after they are sized: a single-threaded memset of memory that is about to be
// overwritten cost 15-20 ms per array of the 0.21 s set-up of BundleAdjustReconstruction (tools/e2e_setup_probe.py).
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = DefaultInitAllocator<U>;
  };
  DefaultInitAllocator() = default;
  template <class U>
  DefaultInitAllocator(const DefaultInitAllocator<U>&) {}
  // big blocks: 2 MB aligned and advised as transparent huge pages -- the arrays are written once, front to back, right
  // after they are sized, and with 4 KB pages that first touch is ~30 000 page faults per 120 MB array
  T* allocate(std::size_t n) {
    // (n * sizeof(T) and the rounding below must not wrap: what std::allocator::allocate checks -- ADVICE r5)
    if (n > (static_cast<std::size_t>(-1) - kHugeAlign) / sizeof(T)) throw std::bad_array_new_length();
    const std::size_t bytes = n * sizeof(T);
    if (bytes >= kHugeFrom) {
      const std::size_t rounded = (bytes + kHugeAlign - 1) / kHugeAlign * kHugeAlign;
      void* p = nullptr;
      if (posix_memalign(&p, kHugeAlign, rounded) == 0 && p != nullptr) {
#ifdef MADV_HUGEPAGE
        (void)madvise(p, rounded, MADV_HUGEPAGE);
#endif
        return static_cast<T*>(p);
      }
    }
    void* p = std::malloc(bytes ? bytes : 1);
    if (p == nullptr) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, std::size_t) noexcept { std::free(p); }
  static constexpr std::size_t kHugeAlign = std::size_t(2) << 20, kHugeFrom = std::size_t(8) << 20;
  template <class U>
  void construct(U* p) {
    ::new (static_cast<void*>(p)) U;
  }
  template <class U, class... Args>
  void construct(U* p, Args&&... args) {
    ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
};
template <class T>
using BulkVector = std::vector<T, DefaultInitAllocator<T>>;

// The flattened problem Optimize() hands to the C ABI (exposed for tests and
// for callers that want to keep a problem resident with tmi_ba_solver_*).
struct FlattenedBundleAdjustmentProblem {
  std::vector<ViewId> view_ids;                   // camera index -> ViewId (ascending)
  std::vector<TrackId> track_ids;                 // point index -> TrackId (ascending)
  std::vector<CameraIntrinsicsGroupId> group_ids; // group index -> group id (ascending)
  std::vector<double> extrinsics, intrinsics, points;
  std::vector<int32_t> camera_group, group_model, group_offset;
  BulkVector<double> obs_xy;                      // [2 x #observations]   (sized without a fill, see BulkVector)
  BulkVector<int32_t> obs_camera, obs_point;      // [#observations], ordered by (track, view)
  std::vector<uint8_t> camera_flags, intrinsics_constant, point_constant;
  tmi_ba_problem AsC();
};

// BundleAdjustmentOptions -> the C ABI's option block.
void ToDeviceOptions(const BundleAdjustmentOptions& options, tmi_ba_options* device_options);

class BundleAdjuster {
 public:
  BundleAdjuster(const BundleAdjustmentOptions& options, Reconstruction* reconstruction);
  virtual ~BundleAdjuster() {}

  // A residual is created for each estimated track that the view observes.
  void AddView(const ViewId view_id);
  // A residual is created for each estimated view (not already optimised) that
  // observes the track; those views are held constant.
  void AddTrack(const TrackId track_id);
  // Optimise the provided views and tracks.
  BundleAdjustmentSummary Optimize();

  // Extensions: AddView / AddTrack for many ids at once.  Same residual set and constancy rules as
  // calling AddView / AddTrack one id at a time in the given order; the walk over the views'
  // feature tables (5 M hash nodes at Venice size, the dominant host cost of a one-shot
  // BundleAdjustReconstruction) runs on several host threads.  A subclass that overrides the
  // protected hooks is detected and served by the one-at-a-time path.
  void AddViews(const std::vector<ViewId>& view_ids);
  void AddTracks(const std::vector<TrackId>& track_ids);

  // Extension: the flattened problem (what Optimize() sends to the device).
  bool Flatten(FlattenedBundleAdjustmentProblem* flat);
  // Extension: Optimize() that leaves the problem resident -- the flattened arrays and the tmi_ba_solver handle
  // (created, solved, downloaded; the caller owns both and calls tmi_ba_solver_destroy).  *solver stays nullptr when
  // there was nothing to optimise or the device refused the problem.  BundleAdjustReconstruction keeps them for the
  // next call on the same Reconstruction (bundle_adjustment.h, "resident session").
  BundleAdjustmentSummary OptimizeResident(FlattenedBundleAdjustmentProblem* flat, tmi_ba_solver** solver);
  // Extension: the full device summary of the last Optimize().
  const tmi_ba_summary& DeviceSummary() const { return device_summary_; }

 protected:
  void SetCameraExtrinsicsParameterization();
  void SetCameraIntrinsicsParameterization();
  std::shared_ptr<CameraIntrinsicsModel> GetIntrinsicsForCameraIntrinsicsGroup(
      const CameraIntrinsicsGroupId camera_intrinsics_group);

  virtual void SetCameraExtrinsicsConstant(const ViewId view_id);
  virtual void SetCameraPositionConstant(const ViewId view_id);
  virtual void SetCameraOrientationConstant(const ViewId view_id);
  virtual void SetTrackConstant(const TrackId track_id);
  virtual void SetTrackVariable(const TrackId track_id);
  virtual void SetCameraSchurGroups(const ViewId view_id);
  virtual void SetTrackSchurGroup(const TrackId track_id);
  // The reference's hook, signature unchanged (bundle_adjuster.h:100-102).  AddView / AddTrack call it with the camera
  // of the view and the track they are adding; the ids those pointers belong to are in hook_view_id_ /
  // hook_track_id_ for the duration of the call (a subclass that forwards OTHER pointers to the base implementation
  // has them looked up in pointer -> id maps built at the first miss; a pointer that is not of this reconstruction
  // drops the residual with one warning per BundleAdjuster).  The bulk AddViews / AddTracks take the one-at-a-time
  // path -- and so call this hook per residual -- whenever the dynamic type is not BundleAdjuster itself.
  virtual void AddReprojectionErrorResidual(const Feature& feature, Camera* camera, Track* track);

  const BundleAdjustmentOptions options_;
  Reconstruction* reconstruction_;
  std::chrono::steady_clock::time_point timer_start_;
  ViewId hook_view_id_ = kInvalidViewId;     // ids behind the pointers handed to AddReprojectionErrorResidual
  TrackId hook_track_id_ = kInvalidTrackId;
  std::unordered_map<const Camera*, ViewId> camera_to_view_;  // built at the first pointer the hook ids do not explain
  std::unordered_map<const Track*, TrackId> track_to_id_;
  bool warned_foreign_residual_ = false;

  std::unordered_set<ViewId> optimized_views_;
  std::unordered_set<CameraIntrinsicsGroupId> optimized_camera_intrinsics_groups_;
  std::unordered_set<CameraIntrinsicsGroupId> potentially_constant_camera_intrinsics_groups_;

  // what the reference keeps inside ceres::Problem
  struct Residual { ViewId view; TrackId track; double x, y; };
  BulkVector<Residual> residuals_;
  // id -> small state, a flat table when the ids are compact (Reconstruction hands them out
  // consecutively) and a hash map otherwise: the reference pays hash look-ups into ceres::Problem
  // per residual (5 M at Venice size); the per-residual path here is one array access.
  class IdState {
   public:
    // -1 = absent
    int Get(uint32_t id) const {
      if (id < flat_.size() && flat_[id] >= 0) return flat_[id];
      if (sparse_.empty()) return -1;  // (an id hashed earlier can lie below a table that grew since)
      auto it = sparse_.find(id);
      return it == sparse_.end() ? -1 : it->second;
    }
    void Set(uint32_t id, int value) {
      // flat while the table stays dense enough (at most ~8 slots per id on top of a 1 M floor), hashed beyond
      if (id < kFlatLimit && !sparse_.count(id) &&
          (id < flat_.size() || static_cast<uint64_t>(id) <= 8ull * (count_ + 1) + (1u << 20))) {
        if (id >= flat_.size()) flat_.resize(static_cast<size_t>(id) + 1 + flat_.size() / 2, -1);
        if (flat_[id] < 0) ++count_;
        flat_[id] = static_cast<int16_t>(value);
      } else {
        if (sparse_.emplace(id, static_cast<int16_t>(value)).second) ++count_;
        else sparse_[id] = static_cast<int16_t>(value);
      }
    }
    void SetIfAbsent(uint32_t id, int value) {
      if (Get(id) < 0) Set(id, value);
    }
    // For threads that write DISJOINT ids of a table Reserve() has pre-sized: touches flat_[id] and nothing else (no
    // resize, no count_, no sparse_ -- Set() updates all three and must not run concurrently).  True when the id was
    // absent; the caller sums those and reports them with NoteAdded() after the join.  sparse_ must not be mutated
    // while such a pass runs (Get() reads it).
    bool SetPresized(uint32_t id, int value) {
      const bool fresh = flat_[id] < 0;
      flat_[id] = static_cast<int16_t>(value);
      return fresh;
    }
    void NoteAdded(size_t n) { count_ += n; }
    // ids that were hashed (Set() beyond the flat range at the time): SetPresized() must not be used while there are
    // any -- it would add a second, flat entry for such an id
    bool HasSparse() const { return !sparse_.empty(); }
    size_t FlatSize() const { return flat_.size(); }
    void Clear() {
      count_ = 0;
      std::fill(flat_.begin(), flat_.end(), static_cast<int16_t>(-1));
      sparse_.clear();
    }
    // make ids 0..max_id flat-addressable up front (so that concurrent readers never see a resize);
    // false if max_id is beyond the flat range
    // ... or so far above the number of ids that a dense table would be mostly holes (sub-reconstructions, merged
    // id spaces): the caller then takes the hash-map path, O(#ids) memory
    bool Reserve(uint32_t max_id, size_t num_ids = SIZE_MAX) {
      if (max_id >= kFlatLimit) return false;
      if (num_ids != SIZE_MAX && static_cast<uint64_t>(max_id) > 8ull * num_ids + (1u << 20)) return false;
      if (max_id >= flat_.size()) flat_.resize(static_cast<size_t>(max_id) + 1, -1);
      return true;
    }
    // ids present, ascending
    std::vector<uint32_t> Ids() const;

   private:
    static constexpr uint32_t kFlatLimit = 1u << 27;
    size_t count_ = 0;  // ids present
    std::vector<int16_t> flat_;
    std::unordered_map<uint32_t, int16_t> sparse_;
  };
  IdState camera_flags_;     // TMI_BA_CAMERA_* bits of the views that take part
  IdState track_constant_;   // 1 constant / 0 variable for the tracks that take part
  IdState track_estimated_;  // memo of Track::IsEstimated for the tracks AddView met
  IdState view_optimized_;   // flat mirror of optimized_views_ (AddTrack tests it per observation)
  IdState track_optimized_;  // the reference's optimized_tracks_ set (1 = AddTrack took the track)
  std::unordered_map<CameraIntrinsicsGroupId, std::vector<uint8_t> > intrinsics_constant_;
  tmi_ba_summary device_summary_;
};
}  // namespace theia
#endif
