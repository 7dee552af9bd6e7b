// Host shim: TheiaSfM's BundleAdjuster / BundleAdjust* free functions on top of
// the MI355X C ABI.  Follows the reference's control flow line by line where it
// defines problem semantics (which residuals exist, which blocks are constant):
//   bundle_adjuster.cc:82-180   ctor, AddView, AddTrack
//   bundle_adjuster.cc:182-302  Optimize, parameterizations
//   bundle_adjustment.cc:47-107 the four free functions
// and replaces the Ceres problem / solve with flatten -> tmi_ba_solve -> write back.
#include "theia/sfm/bundle_adjustment/bundle_adjuster.h"

#include <algorithm>
#include <mutex>
#include <thread>
#include <typeinfo>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>

#include "theia/sfm/camera/camera.h"
#include "theia/sfm/reconstruction.h"

namespace theia {

namespace {
int HostThreads(size_t work) {
  if (const char* e = std::getenv("TMI_BA_HOST_THREADS")) return std::max(1, std::atoi(e));  // tests
  if (work < 200000) return 1;
  // one thread per ~100 k items of hash-container walking, at most half the hardware threads and at
  // most 64 (Venice size on a 256-thread host: AddViews 0.092 s with 16 threads, 0.052 s with 50)
  const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
  const unsigned want = static_cast<unsigned>(std::min<size_t>(work / 100000 + 1, 64));
  return static_cast<int>(std::max(1u, std::min(want, hw / 2)));
}
template <class Body>
void RunThreads(int n_threads, Body&& body) {
  if (n_threads <= 1) {
    body(0);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t) pool.emplace_back([&, t] { body(t); });
  for (auto& th : pool) th.join();
}
}  // namespace

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
  view_optimized_.Set(view_id, 1);
  SetCameraSchurGroups(view_id);
  optimized_camera_intrinsics_groups_.emplace(reconstruction_->CameraIntrinsicsGroupIdFromViewId(view_id));
  camera_flags_.SetIfAbsent(view_id, 0);
  for (const auto& kv : view->Features()) {
    const TrackId track_id = kv.first;
    // Track::IsEstimated, remembered per track: one hash look-up per track instead of one per
    // observation (the flag cannot change while this adjuster collects residuals)
    int est = track_estimated_.Get(track_id);
    if (est < 0) {
      const Track* track = reconstruction_->Track(track_id);
      est = (track != nullptr && track->IsEstimated()) ? 1 : 0;
      track_estimated_.Set(track_id, est);
    }
    if (!est) continue;
    hook_view_id_ = view_id;
    hook_track_id_ = track_id;
    AddReprojectionErrorResidual(kv.second, view->MutableCamera(), reconstruction_->MutableTrack(track_id));
    SetTrackConstant(track_id);
  }
}

// bundle_adjuster.cc:141-180
void BundleAdjuster::AddTrack(const TrackId track_id) {
  Track* track = reconstruction_ ? reconstruction_->MutableTrack(track_id) : nullptr;
  if (track == nullptr) return;
  if (!track->IsEstimated() || track_optimized_.Get(track_id) == 1) return;
  track_optimized_.Set(track_id, 1);
  for (const ViewId view_id : track->ViewIds()) {
    View* view = reconstruction_->MutableView(view_id);
    if (view == nullptr) continue;
    if (view_optimized_.Get(view_id) == 1 || !view->IsEstimated()) continue;
    const Feature* feature = view->GetFeature(track_id);
    if (feature == nullptr) continue;
    hook_view_id_ = view_id;
    hook_track_id_ = track_id;
    AddReprojectionErrorResidual(*feature, view->MutableCamera(), track);
    SetCameraExtrinsicsConstant(view_id);
    potentially_constant_camera_intrinsics_groups_.emplace(
        reconstruction_->CameraIntrinsicsGroupIdFromViewId(view_id));
  }
  SetTrackVariable(track_id);
  SetTrackSchurGroup(track_id);
}

void BundleAdjuster::AddViews(const std::vector<ViewId>& view_ids) {
  if (reconstruction_ == nullptr) return;
  // the one-at-a-time path for subclasses (their hooks must see every residual) and for ids
  // outside the flat tables
  uint32_t max_track = 0;
  bool flat_ok = typeid(*this) == typeid(BundleAdjuster);
  std::vector<TrackId> all_tracks;
  if (flat_ok) {
    all_tracks = reconstruction_->TrackIds();
    for (const TrackId t : all_tracks) max_track = std::max<uint32_t>(max_track, t);
    flat_ok = track_estimated_.Reserve(max_track, all_tracks.size()) && track_constant_.Reserve(max_track, all_tracks.size());
    // SetPresized() below writes flat slots only: an id that an earlier AddTrack hashed (above the flat range of the
    // time) would end up in both stores and be flattened twice -- such tables take the one-at-a-time path
    flat_ok = flat_ok && !track_constant_.HasSparse() && !track_estimated_.HasSparse();
  }
  if (!flat_ok) {
    for (const ViewId v : view_ids) AddView(v);
    return;
  }
  // sequential bookkeeping per view (cheap), exactly what AddView does before its feature loop
  std::vector<const View*> todo;
  std::vector<ViewId> todo_id;
  size_t work = 0;
  for (const ViewId view_id : view_ids) {
    View* view = reconstruction_->MutableView(view_id);
    if (view == nullptr || !view->IsEstimated() || optimized_views_.count(view_id)) continue;
    optimized_views_.emplace(view_id);
    view_optimized_.Set(view_id, 1);
    SetCameraSchurGroups(view_id);
    optimized_camera_intrinsics_groups_.emplace(reconstruction_->CameraIntrinsicsGroupIdFromViewId(view_id));
    camera_flags_.SetIfAbsent(view_id, 0);
    todo.push_back(view);
    todo_id.push_back(view_id);
    work += view->Features().size();
  }
  const int n_threads = HostThreads(work);
  // The IsEstimated memo is rebuilt from the CURRENT tracks: an entry left by an earlier call may belong to a track
  // that has been removed since (its id could lie above max_track, which sizes `touched` below), and the
  // reference looks the track up at every feature (bundle_adjuster.cc:121-125).
  track_estimated_.Clear();
  // Track::IsEstimated for every track, threads own disjoint id ranges (read-only look-ups); the table was
  // pre-sized by Reserve above, so a thread writes its own flat slots and nothing else (IdState::SetPresized)
  std::vector<size_t> fresh(n_threads, 0);
  RunThreads(n_threads, [&](int t) {
    const size_t i0 = all_tracks.size() * t / n_threads, i1 = all_tracks.size() * (t + 1) / n_threads;
    size_t mine = 0;
    for (size_t i = i0; i < i1; ++i) {
      const TrackId id = all_tracks[i];
      if (track_estimated_.Get(id) >= 0) continue;
      const Track* track = reconstruction_->Track(id);
      if (track_estimated_.SetPresized(id, (track != nullptr && track->IsEstimated()) ? 1 : 0)) ++mine;
    }
    fresh[t] = mine;
  });
  for (const size_t f : fresh) track_estimated_.NoteAdded(f);
  // the feature tables, one view at a time per thread, into thread-local residual lists
  std::vector<std::vector<Residual> > local(n_threads);
  std::atomic<size_t> next(0);
  RunThreads(n_threads, [&](int t) {
    std::vector<Residual>& out = local[t];
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= todo.size()) break;
      const ViewId view_id = todo_id[i];
      for (const auto& kv : todo[i]->Features())
        if (track_estimated_.Get(kv.first) == 1) out.push_back(Residual{view_id, kv.first, kv.second.x(), kv.second.y()});
    }
  });
  if (std::getenv("TMI_BA_SETUP_TIMING") != nullptr)
    std::fprintf(stderr, "[tmi_ba shim] %-28s %.3f s\n", "AddViews: feature tables",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - timer_start_).count());
  // AddReprojectionErrorResidual + SetTrackConstant of every residual (:127-137).  The per-thread lists are copied to
  // their final positions in parallel; the tracks of the residuals are exactly the estimated tracks that some added
  // view sees, marked by threads that own disjoint ID ranges of the (pre-sized) table.
  {
    const size_t base = residuals_.size();
    std::vector<size_t> first(local.size() + 1, base);
    for (size_t l = 0; l < local.size(); ++l) first[l + 1] = first[l] + local[l].size();
    residuals_.resize(first.back());
    RunThreads(n_threads, [&](int t) { std::copy(local[t].begin(), local[t].end(), residuals_.begin() + first[t]); });
    std::vector<uint8_t> touched(static_cast<size_t>(max_track) + 1, 0);
    // (one byte per track id; equal values written by several threads: relaxed atomic stores)
    RunThreads(n_threads, [&](int t) {
      for (const Residual& r : local[t])
        if (r.track <= max_track) __atomic_store_n(&touched[r.track], static_cast<uint8_t>(1), __ATOMIC_RELAXED);
    });
    std::vector<size_t> fresh_c(n_threads, 0);
    RunThreads(n_threads, [&](int t) {
      const size_t i0 = touched.size() * t / n_threads, i1 = touched.size() * (t + 1) / n_threads;
      size_t mine = 0;
      for (size_t id = i0; id < i1; ++id)
        if (touched[id] && track_constant_.SetPresized(static_cast<uint32_t>(id), 1)) ++mine;
      fresh_c[t] = mine;
    });
    for (const size_t c : fresh_c) track_constant_.NoteAdded(c);
  }
  if (std::getenv("TMI_BA_SETUP_TIMING") != nullptr)
    std::fprintf(stderr, "[tmi_ba shim] %-28s %.3f s\n", "AddViews: total",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - timer_start_).count());
}

void BundleAdjuster::AddTracks(const std::vector<TrackId>& track_ids) {
  const auto t0 = std::chrono::steady_clock::now();
  if (reconstruction_ == nullptr) return;
  uint32_t max_track = 0;
  for (const TrackId t : track_ids) max_track = std::max<uint32_t>(max_track, t);
  // one-at-a-time for subclasses (their hooks must see every call) and for ids beyond the flat tables
  if (typeid(*this) != typeid(BundleAdjuster) || !track_optimized_.Reserve(max_track, track_ids.size()) ||
      !track_constant_.Reserve(max_track, track_ids.size())) {
    for (const TrackId t : track_ids) AddTrack(t);
    return;
  }
  // read-only pass, threads own disjoint index ranges: 0 = AddTrack returns at once, 1 = every view
  // of the track is already optimised (AddTrack only marks the track variable), 2 = AddTrack has
  // residuals to add (views outside the optimised set)
  std::vector<uint8_t> kind(track_ids.size(), 0);
  const int n_threads = HostThreads(4 * track_ids.size());
  RunThreads(n_threads, [&](int t) {
    const size_t i0 = track_ids.size() * t / n_threads, i1 = track_ids.size() * (t + 1) / n_threads;
    for (size_t i = i0; i < i1; ++i) {
      const Track* track = reconstruction_->Track(track_ids[i]);
      if (track == nullptr || !track->IsEstimated()) continue;
      uint8_t k = 1;
      for (const ViewId view_id : track->ViewIds())
        if (view_optimized_.Get(view_id) != 1) { k = 2; break; }
      kind[i] = k;
    }
  });
  for (size_t i = 0; i < track_ids.size(); ++i) {
    const TrackId id = track_ids[i];
    if (kind[i] == 0) continue;
    if (kind[i] == 2) { AddTrack(id); continue; }
    if (track_optimized_.Get(id) == 1) continue;
    track_optimized_.Set(id, 1);
    track_constant_.Set(id, 0);  // SetTrackVariable
  }
  if (std::getenv("TMI_BA_SETUP_TIMING") != nullptr)
    std::fprintf(stderr, "[tmi_ba shim] %-28s %.3f s (since construction %.3f s)\n", "AddTracks",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - timer_start_).count());
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
  camera_flags_.Set(v, std::max(camera_flags_.Get(v), 0) | TMI_BA_CAMERA_POSITION_CONSTANT | TMI_BA_CAMERA_ORIENTATION_CONSTANT);
}
void BundleAdjuster::SetCameraPositionConstant(const ViewId v) {
  camera_flags_.Set(v, std::max(camera_flags_.Get(v), 0) | TMI_BA_CAMERA_POSITION_CONSTANT);
}
void BundleAdjuster::SetCameraOrientationConstant(const ViewId v) {
  camera_flags_.Set(v, std::max(camera_flags_.Get(v), 0) | TMI_BA_CAMERA_ORIENTATION_CONSTANT);
}
void BundleAdjuster::SetTrackConstant(const TrackId t) { track_constant_.Set(t, 1); }
void BundleAdjuster::SetTrackVariable(const TrackId t) { track_constant_.Set(t, 0); }

std::vector<uint32_t> BundleAdjuster::IdState::Ids() const {
  std::vector<uint32_t> ids;
  for (size_t i = 0; i < flat_.size(); ++i)
    if (flat_[i] >= 0) ids.push_back(static_cast<uint32_t>(i));
  if (!sparse_.empty()) {
    const size_t first = ids.size();
    for (const auto& kv : sparse_) ids.push_back(kv.first);
    std::sort(ids.begin() + first, ids.end());
    std::inplace_merge(ids.begin(), ids.begin() + first, ids.end());  // a hashed id can lie below a table that grew since
  }
  return ids;
}
// bundle_adjuster.cc:346-371: the elimination order (points, then intrinsics, then
// extrinsics) is structural in the device path; nothing to record.
void BundleAdjuster::SetCameraSchurGroups(const ViewId) {}
void BundleAdjuster::SetTrackSchurGroup(const TrackId) {}

// bundle_adjuster.cc:373-386
void BundleAdjuster::AddReprojectionErrorResidual(const Feature& feature, Camera* camera, Track* track) {
  if (reconstruction_ == nullptr || camera == nullptr || track == nullptr) return;
  // the ids behind the pointers: AddView / AddTrack leave them in hook_*_id_; pointers that are not theirs (a subclass
  // forwarding something else) are looked up
  ViewId view_id = hook_view_id_;
  TrackId track_id = hook_track_id_;
  {
    // (a miss builds the pointer -> id maps ONCE: a subclass that forwards its own pointers for every residual would
    // otherwise pay a scan of all views and tracks per residual -- ADVICE r5)
    View* hv = view_id == kInvalidViewId ? nullptr : reconstruction_->MutableView(view_id);
    if (hv == nullptr || hv->MutableCamera() != camera) {
      if (camera_to_view_.empty())
        for (const ViewId v : reconstruction_->ViewIds()) camera_to_view_.emplace(reconstruction_->MutableView(v)->MutableCamera(), v);
      const auto it = camera_to_view_.find(camera);
      view_id = it == camera_to_view_.end() ? kInvalidViewId : it->second;
    }
    if (track_id == kInvalidTrackId || reconstruction_->MutableTrack(track_id) != track) {
      if (track_to_id_.empty())
        for (const TrackId t : reconstruction_->TrackIds()) track_to_id_.emplace(reconstruction_->MutableTrack(t), t);
      const auto it = track_to_id_.find(track);
      track_id = it == track_to_id_.end() ? kInvalidTrackId : it->second;
    }
  }
  if (view_id == kInvalidViewId || track_id == kInvalidTrackId) {
    // not of this reconstruction: the residual cannot be flattened (the reference would hand Ceres the raw pointers)
    if (!warned_foreign_residual_) {
      warned_foreign_residual_ = true;
      std::fprintf(stderr, "[tmi_ba shim] AddReprojectionErrorResidual: camera / track pointer does not belong to the "
                           "reconstruction being adjusted; residual dropped (reported once per BundleAdjuster)\n");
    }
    return;
  }
  residuals_.push_back(Residual{view_id, track_id, feature.x(), feature.y()});
  camera_flags_.SetIfAbsent(view_id, 0);
  track_constant_.SetIfAbsent(track_id, 1);
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
  const bool timing = std::getenv("TMI_BA_SETUP_TIMING") != nullptr;
  auto t_phase = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[tmi_ba shim] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_phase).count());
    t_phase = now;
  };
  *f = FlattenedBundleAdjustmentProblem();
  SetCameraExtrinsicsParameterization();
  SetCameraIntrinsicsParameterization();
  // sorted ids => deterministic block order (Ceres follows unordered_map iteration)
  IdIndex<ViewId> cam_index;
  IdIndex<TrackId> pt_index;
  IdIndex<CameraIntrinsicsGroupId> grp_index;
  {
    // every view / track of a residual is registered in the state tables: their ids, ascending.  AddView registers
    // its view up front, so an estimated view none of whose tracks is estimated sits in the table without a
    // residual; in the reference its parameter blocks never enter the ceres::Problem (bundle_adjuster.cc:373-386
    // is the only place that adds them) -- it is dropped here
    std::vector<ViewId> v = camera_flags_.Ids();
    std::vector<TrackId> t = track_constant_.Ids();
    if (!v.empty()) {
      std::vector<uint8_t> seen;
      std::unordered_set<ViewId> seen_sparse;
      const bool compact = static_cast<uint64_t>(v.back()) < 8ull * v.size() + (1u << 20);
      if (compact) seen.assign(static_cast<size_t>(v.back()) + 1, 0);
      for (const Residual& r : residuals_) {
        if (compact) seen[r.view] = 1;
        else seen_sparse.insert(r.view);
      }
      v.erase(std::remove_if(v.begin(), v.end(), [&](ViewId id) { return compact ? !seen[id] : !seen_sparse.count(id); }),
              v.end());
    }
    cam_index.Build(&v);
    pt_index.Build(&t);
    std::vector<CameraIntrinsicsGroupId> g;
    g.reserve(cam_index.ids.size());
    for (const ViewId id : cam_index.ids) g.push_back(reconstruction_->CameraIntrinsicsGroupIdFromViewId(id));
    grp_index.Build(&g);
  }
  lap("id tables");
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
    f->camera_flags.push_back(static_cast<uint8_t>(std::max(camera_flags_.Get(id), 0)));
  }
  f->track_ids = pt_index.ids;
  {
    // read-only look-ups of distinct tracks into pre-sized arrays: split over threads
    const size_t nt = f->track_ids.size();
    f->points.resize(4 * nt);
    f->point_constant.resize(nt);
    const int nth = HostThreads(4 * nt);
    RunThreads(nth, [&](int th) {
      for (size_t t = nt * th / nth; t < nt * (th + 1) / nth; ++t) {
        const TrackId id = f->track_ids[t];
        const Eigen::Vector4d& X = reconstruction_->Track(id)->Point();
        std::copy(X.data(), X.data() + 4, f->points.begin() + 4 * t);
        f->point_constant[t] = track_constant_.Get(id) == 1 ? 1 : 0;
      }
    });
  }
  lap("parameters");
  // deterministic observation order: by (track, view).  Two stable counting sorts over compact
  // records (LSD radix: first by camera index, then by track index) -- linear passes instead of a
  // comparison sort per track through an index indirection (1.5 s -> 0.4 s at Venice size on the
  // development box).
  struct Rec { int32_t cam, pt; double x, y; };
  const size_t n = residuals_.size(), np = f->track_ids.size();
  // Order by (track, view) in two cache-friendly steps instead of a 1 M-bucket scatter (three
  // cache misses per observation: 1.6 s at Venice size on the development box):
  //   A. scatter the records into coarse buckets of 1024 consecutive tracks (a few thousand
  //      write streams, each sequential), threads own contiguous slices of the input;
  //   B. per bucket (a few thousand records, cache resident): counting sort by track, then
  //      the few observations of a track by view; output written sequentially.
  const int kShift = 10;
  const size_t nbuckets = (np >> kShift) + 1;
  const int n_threads = HostThreads(n);
  auto run_threads = [&](auto&& body) {
    if (n_threads == 1) {
      body(0);
      return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back([&, t] { body(t); });
    for (auto& th : pool) th.join();
  };
  std::unique_ptr<Rec[]> coarse(new Rec[n ? n : 1]);  // (not value-initialised: every slot is written by the scatter below)
  std::vector<size_t> bucket_first(nbuckets + 1, 0);
  {
    std::vector<std::vector<size_t> > cnt(n_threads, std::vector<size_t>(nbuckets, 0));
    run_threads([&](int t) {
      const size_t i0 = n * t / n_threads, i1 = n * (t + 1) / n_threads;
      for (size_t i = i0; i < i1; ++i) cnt[t][static_cast<size_t>(pt_index(residuals_[i].track)) >> kShift]++;
    });
    size_t run = 0;
    for (size_t bkt = 0; bkt < nbuckets; ++bkt) {
      bucket_first[bkt] = run;
      for (int t = 0; t < n_threads; ++t) {
        const size_t c = cnt[t][bkt];
        cnt[t][bkt] = run;  // this thread's first slot inside the bucket
        run += c;
      }
    }
    bucket_first[nbuckets] = run;
    lap("  order: histogram");
    run_threads([&](int t) {
      const size_t i0 = n * t / n_threads, i1 = n * (t + 1) / n_threads;
      for (size_t i = i0; i < i1; ++i) {
        const Residual& r = residuals_[i];
        const int32_t p = pt_index(r.track);
        coarse[cnt[t][static_cast<size_t>(p) >> kShift]++] = Rec{cam_index(r.view), p, r.x, r.y};
      }
    });
  }
  lap("  order: coarse scatter");
  f->obs_camera.resize(n);
  f->obs_point.resize(n);
  f->obs_xy.resize(2 * n);
  lap("  order: output alloc");
  {
    std::atomic<size_t> next(0);
    run_threads([&](int) {
      std::vector<size_t> first((size_t(1) << kShift) + 1);
      std::vector<Rec> local;
      for (;;) {
        const size_t bkt = next.fetch_add(1);
        if (bkt >= nbuckets) break;
        const size_t b0 = bucket_first[bkt], b1 = bucket_first[bkt + 1];
        if (b0 == b1) continue;
        const int32_t base = static_cast<int32_t>(bkt << kShift);
        std::fill(first.begin(), first.end(), 0);
        for (size_t i = b0; i < b1; ++i) first[coarse[i].pt - base + 1]++;
        for (size_t p = 0; p + 1 < first.size(); ++p) first[p + 1] += first[p];
        local.resize(b1 - b0);
        {
          std::vector<size_t> fill(first.begin(), first.end() - 1);
          for (size_t i = b0; i < b1; ++i) local[fill[coarse[i].pt - base]++] = coarse[i];
        }
        for (size_t p = 0; p + 1 < first.size(); ++p)
          if (first[p + 1] - first[p] > 1)
            std::sort(local.begin() + first[p], local.begin() + first[p + 1],
                      [](const Rec& x, const Rec& y) { return x.cam < y.cam; });
        for (size_t i = 0; i < local.size(); ++i) {
          const size_t q = b0 + i;
          f->obs_camera[q] = local[i].cam;
          f->obs_point[q] = local[i].pt;
          f->obs_xy[2 * q] = local[i].x;
          f->obs_xy[2 * q + 1] = local[i].y;
        }
      }
    });
  }
  lap("  order: buckets");
  // A (view, track) pair can have been pushed twice when AddTrack(t) pulled in a view that a
  // later AddView(v) adds again -- a call order outside the class contract ("AddView before
  // AddTrack", bundle_adjuster.h:58-59; the free functions follow it).  The reference would
  // then hold the residual block twice; the device layout holds an observation once, so the
  // duplicate is dropped here (the view's extrinsics stay constant, as in the reference).
  size_t m = n ? 1 : 0;
  for (size_t q = 1; q < n; ++q) {
    if (f->obs_camera[q] == f->obs_camera[q - 1] && f->obs_point[q] == f->obs_point[q - 1]) continue;
    if (m != q) {
      f->obs_camera[m] = f->obs_camera[q];
      f->obs_point[m] = f->obs_point[q];
      f->obs_xy[2 * m] = f->obs_xy[2 * q];
      f->obs_xy[2 * m + 1] = f->obs_xy[2 * q + 1];
    }
    ++m;
  }
  f->obs_camera.resize(m);
  f->obs_point.resize(m);
  f->obs_xy.resize(2 * m);
  lap("observation order");
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
  // ceres::SCHUR_JACOBI (and JACOBI, which the device path maps onto it) -> Ceres' own block shape, one block per
  // parameter block; the merged per-view block only on request (bundle_adjustment.h, extensions).  CLUSTER_JACOBI /
  // CLUSTER_TRIDIAGONAL go through (theia_mi355_ba.h: clusters = the shared intrinsics blocks with their views, or the
  // visibility clusters; the tridiagonal variant adds the blocks between neighbours of the degree-2 spanning forest).
  o->preconditioner_type = static_cast<int32_t>(options.preconditioner_type);
  if ((options.preconditioner_type == ceres::SCHUR_JACOBI || options.preconditioner_type == ceres::JACOBI) &&
      !options.merged_view_blocks_in_preconditioner)
    o->preconditioner_type = TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS;
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
  o->visibility_clustering_type = static_cast<int32_t>(options.visibility_clustering_type);  // bundle_adjuster.cc:61
}

// which preconditioner ran (tmi_ba_summary::effective_preconditioner_type) in ceres' enumerators, against the request
void FillPreconditionerReport(const BundleAdjustmentOptions& options, const tmi_ba_summary& device, BundleAdjustmentSummary* summary) {
  const bool iterative = options.linear_solver_type == ceres::ITERATIVE_SCHUR || options.linear_solver_type == ceres::CGNR;
  if (!iterative || device.num_linear_solver_iterations == 0) {
    summary->effective_preconditioner_type = iterative ? options.preconditioner_type : ceres::IDENTITY;
    summary->preconditioner_substituted = false;
    return;
  }
  switch (device.effective_preconditioner_type) {
    case TMI_BA_PRECOND_IDENTITY: summary->effective_preconditioner_type = ceres::IDENTITY; break;
    case TMI_BA_PRECOND_CLUSTER_JACOBI: summary->effective_preconditioner_type = ceres::CLUSTER_JACOBI; break;
    case TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL: summary->effective_preconditioner_type = ceres::CLUSTER_TRIDIAGONAL; break;
    default: summary->effective_preconditioner_type = ceres::SCHUR_JACOBI; break;  // either block shape
  }
  summary->preconditioner_substituted = summary->effective_preconditioner_type != options.preconditioner_type;
}

// bundle_adjuster.cc:182-221
BundleAdjustmentSummary BundleAdjuster::Optimize() {
  FlattenedBundleAdjustmentProblem flat;
  return OptimizeResident(&flat, nullptr);
}

// write the optimised parameters back in place (Ceres updates the caller's arrays through raw pointers)
static void WriteBack(const FlattenedBundleAdjustmentProblem& flat, Reconstruction* reconstruction) {
  for (size_t c = 0; c < flat.view_ids.size(); ++c) {
    double* e = reconstruction->MutableView(flat.view_ids[c])->MutableCamera()->mutable_extrinsics();
    std::copy(flat.extrinsics.begin() + 6 * c, flat.extrinsics.begin() + 6 * c + 6, e);
  }
  // distinct tracks, read-only look-ups in the reconstruction's maps: safe to split over threads
  const int n_threads = HostThreads(4 * flat.track_ids.size());
  RunThreads(n_threads, [&](int th) {
    const size_t t0 = flat.track_ids.size() * th / n_threads, t1 = flat.track_ids.size() * (th + 1) / n_threads;
    for (size_t t = t0; t < t1; ++t) {
      double* X = reconstruction->MutableTrack(flat.track_ids[t])->MutablePoint()->data();
      std::copy(flat.points.begin() + 4 * t, flat.points.begin() + 4 * t + 4, X);
    }
  });
}

BundleAdjustmentSummary BundleAdjuster::OptimizeResident(FlattenedBundleAdjustmentProblem* flat_out,
                                                         tmi_ba_solver** solver_out) {
  BundleAdjustmentSummary summary;
  FlattenedBundleAdjustmentProblem& flat = *flat_out;
  if (solver_out) *solver_out = nullptr;
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
  if (solver_out == nullptr) {
    tmi_ba_solve(&p, &o, &device_summary_);
  } else {
    // the same three steps tmi_ba_solve runs, with the handle kept
    tmi_ba_solver* solver = nullptr;
    std::memset(&device_summary_, 0, sizeof(device_summary_));
    int st = tmi_ba_solver_create(&p, &o, 0, 1, &solver);
    if (st == TMI_BA_OK) st = tmi_ba_solver_solve(solver, &o, &device_summary_);  // its setup time is the create call's
    if (st == TMI_BA_OK && device_summary_.success) st = tmi_ba_solver_download(solver, &p);
    if (st != TMI_BA_OK) {
      device_summary_.success = 0;
      device_summary_.status = st;
      if (solver) tmi_ba_solver_destroy(solver);
      solver = nullptr;
    }
    *solver_out = solver;
  }
  summary.success = device_summary_.success != 0;
  summary.initial_cost = device_summary_.initial_cost;
  summary.final_cost = device_summary_.final_cost;
  summary.setup_time_in_seconds = internal_setup_time + device_summary_.setup_time_in_seconds;
  summary.solve_time_in_seconds = device_summary_.solve_time_in_seconds;
  FillPreconditionerReport(options_, device_summary_, &summary);
  if (options_.verbose)
    std::fprintf(stderr, "[theia::BundleAdjuster] %s: cost %.9e -> %.9e, %d iterations, rmse %.6f px\n",
                 device_summary_.message, summary.initial_cost, summary.final_cost,
                 device_summary_.num_iterations, device_summary_.final_rmse);
  if (!summary.success) return summary;
  for (size_t g = 0; g < flat.group_ids.size(); ++g) {
    std::shared_ptr<CameraIntrinsicsModel> intr = GetIntrinsicsForCameraIntrinsicsGroup(flat.group_ids[g]);
    std::copy(flat.intrinsics.begin() + flat.group_offset[g], flat.intrinsics.begin() + flat.group_offset[g + 1],
              intr->mutable_parameters());
  }
  WriteBack(flat, reconstruction_);
  return summary;
}

// ---- bundle_adjustment.cc:47-107 ----------------------------------------------------
BundleAdjustmentSummary BundleAdjustPartialReconstruction(const BundleAdjustmentOptions& options,
                                                          const std::unordered_set<ViewId>& view_ids,
                                                          const std::unordered_set<TrackId>& track_ids,
                                                          Reconstruction* reconstruction) {
  BundleAdjuster bundle_adjuster(options, reconstruction);
  bundle_adjuster.AddViews(std::vector<ViewId>(view_ids.begin(), view_ids.end()));
  bundle_adjuster.AddTracks(std::vector<TrackId>(track_ids.begin(), track_ids.end()));
  return bundle_adjuster.Optimize();
}

// ---- resident session ----------------------------------------------------------------------------------------
// The reference's pipelines call the full BA again and again on one Reconstruction
// (global_reconstruction_estimator.cc:487,522; incremental_reconstruction_estimator.cc:516-607).  Re-flattening 5 M
// observations and re-building the device structure costs 0.25 s of a 0.33 s call at Venice size, so the flattened
// problem and the tmi_ba_solver handle of the last BundleAdjustReconstruction stay alive; a call on the SAME
// Reconstruction whose residual set cannot have changed (data-model mutation stamp, types.h) with the same
// problem-shaping options re-reads the parameter values through cached pointers, uploads them
// (tmi_ba_solver_set_parameters), solves and writes back.  One session per process (it holds the problem's HBM);
// ReleaseBundleAdjustmentSession() frees it.
namespace {
struct ResidentSession {
  const Reconstruction* reconstruction = nullptr;
  std::uint64_t uid = 0, epoch = 0;
  BundleAdjustmentOptions options;
  FlattenedBundleAdjustmentProblem flat;
  tmi_ba_solver* solver = nullptr;
  std::vector<double*> extrinsics_ptr, point_ptr, intrinsics_ptr;
  std::vector<const CameraIntrinsicsModel*> view_model;  // per camera: the intrinsics object its view pointed at
  // structural fingerprint taken when the session was built (StructureFingerprint): catches residual-set changes
  // that did not go through a stamped mutator (copy-assignment through MutableView() / MutableTrack(), data-model
  // classes without the hooks)
  std::uint64_t fingerprint[6] = {0, 0, 0, 0, 0, 0};
  ~ResidentSession() {
    if (solver) tmi_ba_solver_destroy(solver);
  }
};
std::mutex g_session_mutex;
// Deliberately leaked holder: a static destructor would call into the HIP runtime after the runtime's own teardown at
// process exit.  The session is released by an atexit handler registered when the first session is stored -- i.e. after
// the first HIP call, so it runs before the runtime's exit handlers -- or explicitly (ReleaseBundleAdjustmentSession).
std::unique_ptr<ResidentSession>& g_session = *new std::unique_ptr<ResidentSession>;

// options that shape the handle (tmi_ba_solver_create bakes them into the structure: the preconditioner type decides
// which blocks of S exist and which operator applies it, engine.hip create_impl)
bool SameShape(const BundleAdjustmentOptions& a, const BundleAdjustmentOptions& b) {
  return a.constant_camera_orientation == b.constant_camera_orientation &&
         a.constant_camera_position == b.constant_camera_position && a.intrinsics_to_optimize == b.intrinsics_to_optimize &&
         a.linear_solver_type == b.linear_solver_type && a.point_dof == b.point_dof && a.device == b.device &&
         a.preconditioner_type == b.preconditioner_type &&
         a.visibility_clustering_type == b.visibility_clustering_type &&
         a.merged_view_blocks_in_preconditioner == b.merged_view_blocks_in_preconditioner;
}

// What the residual set of a full BA depends on, through public accessors only: container sizes, which of the
// session's views / tracks are estimated, how many features / views they hold, and how many tracks are estimated
// at all (a track that became estimated since belongs to the residual set and is in no cached list).
// O(#views + #tracks) look-ups on the host's threads, no per-observation work: ~2 ms at Venice size.
void StructureFingerprint(const Reconstruction& rec, const FlattenedBundleAdjustmentProblem& flat, std::uint64_t out[6]) {
  out[0] = static_cast<std::uint64_t>(rec.NumViews());
  out[1] = static_cast<std::uint64_t>(rec.NumTracks());
  std::uint64_t est_views = 0, features = 0;
  for (const ViewId id : flat.view_ids) {
    const View* v = rec.View(id);
    if (v == nullptr || !v->IsEstimated()) continue;
    ++est_views;
    features += static_cast<std::uint64_t>(v->NumFeatures());
  }
  out[2] = est_views;
  out[3] = features;
  const size_t nt = flat.track_ids.size();
  const int n_threads = HostThreads(8 * nt);
  std::vector<std::uint64_t> est(n_threads, 0), obs(n_threads, 0);
  RunThreads(n_threads, [&](int th) {
    std::uint64_t e = 0, o = 0;
    for (size_t t = nt * th / n_threads; t < nt * (th + 1) / n_threads; ++t) {
      const Track* tr = rec.Track(flat.track_ids[t]);
      if (tr == nullptr || !tr->IsEstimated()) continue;
      ++e;
      o += static_cast<std::uint64_t>(tr->NumViews());
    }
    est[th] = e;
    obs[th] = o;
  });
  std::uint64_t est_tracks = 0, observations = 0;
  for (int th = 0; th < n_threads; ++th) {
    est_tracks += est[th];
    observations += obs[th];
  }
  if (static_cast<std::uint64_t>(rec.NumTracks()) != nt) {
    // tracks outside the cached list exist (unestimated when the session was built): count the estimated ones
    est_tracks = 0;
    for (const TrackId id : rec.TrackIds()) {
      const Track* tr = rec.Track(id);
      if (tr != nullptr && tr->IsEstimated()) ++est_tracks;
    }
  }
  out[4] = est_tracks;
  out[5] = observations;
}

bool SessionMatches(const ResidentSession& s, const BundleAdjustmentOptions& options, Reconstruction* rec) {
  if (s.solver == nullptr || s.reconstruction != rec || s.uid != rec->Uid() ||
      s.epoch != internal::DataModelEpoch().load(std::memory_order_relaxed) || !SameShape(s.options, options))
    return false;
  // the intrinsics objects can be swapped without touching the containers (Camera::MutableCameraIntrinsics)
  for (size_t c = 0; c < s.flat.view_ids.size(); ++c) {
    const View* v = rec->View(s.flat.view_ids[c]);
    if (v == nullptr || v->Camera().CameraIntrinsics().get() != s.view_model[c] ||
        static_cast<int32_t>(v->Camera().GetCameraIntrinsicsModelType()) != s.flat.group_model[s.flat.camera_group[c]])
      return false;
  }
  std::uint64_t now[6];
  StructureFingerprint(*rec, s.flat, now);
  return std::equal(now, now + 6, s.fingerprint);
}

BundleAdjustmentSummary RunSession(ResidentSession* s, const BundleAdjustmentOptions& options) {
  BundleAdjustmentSummary summary;
  const auto t0 = std::chrono::steady_clock::now();
  FlattenedBundleAdjustmentProblem& flat = s->flat;
  for (size_t c = 0; c < s->extrinsics_ptr.size(); ++c) std::copy(s->extrinsics_ptr[c], s->extrinsics_ptr[c] + 6, flat.extrinsics.begin() + 6 * c);
  for (size_t g = 0; g < s->intrinsics_ptr.size(); ++g)
    std::copy(s->intrinsics_ptr[g], s->intrinsics_ptr[g] + (flat.group_offset[g + 1] - flat.group_offset[g]),
              flat.intrinsics.begin() + flat.group_offset[g]);
  const size_t np = s->point_ptr.size();
  const int n_threads = HostThreads(4 * np);
  RunThreads(n_threads, [&](int th) {
    for (size_t t = np * th / n_threads; t < np * (th + 1) / n_threads; ++t)
      std::copy(s->point_ptr[t], s->point_ptr[t] + 4, flat.points.begin() + 4 * t);
  });
  tmi_ba_options o;
  ToDeviceOptions(options, &o);
  tmi_ba_problem p = flat.AsC();
  tmi_ba_summary ds;
  std::memset(&ds, 0, sizeof(ds));
  int st = tmi_ba_solver_set_parameters(s->solver, &p);
  const double setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (st == TMI_BA_OK) st = tmi_ba_solver_solve(s->solver, &o, &ds);
  if (st == TMI_BA_OK && ds.success) st = tmi_ba_solver_download(s->solver, &p);
  summary.success = st == TMI_BA_OK && ds.success != 0;
  summary.initial_cost = ds.initial_cost;
  summary.final_cost = ds.final_cost;
  summary.setup_time_in_seconds = setup;  // gather + upload (the handle's own figure is its create call, long past)
  summary.solve_time_in_seconds = ds.solve_time_in_seconds;
  FillPreconditionerReport(options, ds, &summary);
  if (options.verbose)
    std::fprintf(stderr, "[theia::BundleAdjustReconstruction, resident session] %s: cost %.9e -> %.9e, %d iterations\n",
                 ds.message, summary.initial_cost, summary.final_cost, ds.num_iterations);
  if (!summary.success) return summary;
  for (size_t c = 0; c < s->extrinsics_ptr.size(); ++c) std::copy(flat.extrinsics.begin() + 6 * c, flat.extrinsics.begin() + 6 * c + 6, s->extrinsics_ptr[c]);
  for (size_t g = 0; g < s->intrinsics_ptr.size(); ++g)
    std::copy(flat.intrinsics.begin() + flat.group_offset[g], flat.intrinsics.begin() + flat.group_offset[g + 1], s->intrinsics_ptr[g]);
  RunThreads(n_threads, [&](int th) {
    for (size_t t = np * th / n_threads; t < np * (th + 1) / n_threads; ++t)
      std::copy(flat.points.begin() + 4 * t, flat.points.begin() + 4 * t + 4, s->point_ptr[t]);
  });
  return summary;
}
}  // namespace

void ReleaseBundleAdjustmentSession() {
  std::lock_guard<std::mutex> lock(g_session_mutex);
  g_session.reset();
}

bool BundleAdjustmentSessionIsResident(const Reconstruction* reconstruction) {
  std::lock_guard<std::mutex> lock(g_session_mutex);
  return g_session && g_session->reconstruction == reconstruction && g_session->solver != nullptr &&
         g_session->uid == reconstruction->Uid() &&
         g_session->epoch == internal::DataModelEpoch().load(std::memory_order_relaxed);
}

BundleAdjustmentSummary BundleAdjustReconstruction(const BundleAdjustmentOptions& options,
                                                   Reconstruction* reconstruction) {
  if (reconstruction == nullptr) return BundleAdjustmentSummary();
  if (!options.keep_problem_resident) {
    BundleAdjuster bundle_adjuster(options, reconstruction);
    bundle_adjuster.AddViews(reconstruction->ViewIds());
    bundle_adjuster.AddTracks(reconstruction->TrackIds());
    return bundle_adjuster.Optimize();
  }
  // The mutex guards the HOLDER only: the session is taken out under the lock, used (flatten / solve) without it and
  // put back, so full BAs of different Reconstructions on different threads do not serialise on each other -- a
  // call that finds the holder empty (another thread is using the session) simply builds its own.
  std::unique_ptr<ResidentSession> mine;
  {
    std::lock_guard<std::mutex> lock(g_session_mutex);
    mine = std::move(g_session);
  }
  auto put_back = [](std::unique_ptr<ResidentSession> keep) {
    std::unique_ptr<ResidentSession> old;  // destroyed (HBM freed) outside the lock
    {
      std::lock_guard<std::mutex> lock(g_session_mutex);
      old = std::move(g_session);
      g_session = std::move(keep);
    }
  };
  if (mine && SessionMatches(*mine, options, reconstruction)) {
    const BundleAdjustmentSummary summary = RunSession(mine.get(), options);
    put_back(std::move(mine));
    return summary;
  }
  mine.reset();  // its HBM goes before the new problem is built
  std::unique_ptr<ResidentSession> s(new ResidentSession);
  BundleAdjuster bundle_adjuster(options, reconstruction);
  bundle_adjuster.AddViews(reconstruction->ViewIds());
  bundle_adjuster.AddTracks(reconstruction->TrackIds());
  const BundleAdjustmentSummary summary = bundle_adjuster.OptimizeResident(&s->flat, &s->solver);
  if (s->solver != nullptr) {
    s->reconstruction = reconstruction;
    s->uid = reconstruction->Uid();
    s->epoch = internal::DataModelEpoch().load(std::memory_order_relaxed);
    s->options = options;
    const FlattenedBundleAdjustmentProblem& f = s->flat;
    s->extrinsics_ptr.resize(f.view_ids.size());
    s->view_model.resize(f.view_ids.size());
    s->intrinsics_ptr.assign(f.group_ids.size(), nullptr);
    for (size_t c = 0; c < f.view_ids.size(); ++c) {
      Camera* cam = reconstruction->MutableView(f.view_ids[c])->MutableCamera();
      s->extrinsics_ptr[c] = cam->mutable_extrinsics();
      s->view_model[c] = cam->CameraIntrinsics().get();
      s->intrinsics_ptr[f.camera_group[c]] = cam->mutable_intrinsics();
    }
    bool complete = true;
    for (double* ptr : s->intrinsics_ptr) complete = complete && ptr != nullptr;
    s->point_ptr.resize(f.track_ids.size());
    for (size_t t = 0; t < f.track_ids.size(); ++t) s->point_ptr[t] = reconstruction->MutableTrack(f.track_ids[t])->MutablePoint()->data();
    if (complete) {
      StructureFingerprint(*reconstruction, s->flat, s->fingerprint);
      static const bool registered = (std::atexit([] {
        // a BA still running in another thread holds its session itself; what sits in the holder is released
        if (g_session_mutex.try_lock()) {
          g_session.reset();
          g_session_mutex.unlock();
        }
      }), true);
      (void)registered;
      put_back(std::move(s));
    }
  }
  return summary;
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
