// Self-test of the C++ host shim (theia::BundleAdjuster on the MI355X C ABI).
//   ./test_host_shim cpu   problem-semantics checks on the flattened problem
//   ./test_host_shim gpu   end-to-end BundleAdjustReconstruction / partial BA
// The expectations restate the AddView / AddTrack rules of
// /root/reference/src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:102-180,242-287.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "theia/sfm/bundle_adjustment/bundle_adjuster.h"
#include "theia/sfm/bundle_adjustment/bundle_adjustment.h"
#include "theia/sfm/bundle_adjustment/bundle_adjust_two_views.h"
#include "theia/sfm/reconstruction.h"
#include "theia/sfm/select_good_tracks_for_bundle_adjustment.h"
#include "theia/sfm/set_outlier_tracks_to_unestimated.h"

using namespace theia;

static int g_fail = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);           \
      ++g_fail;                                                             \
    }                                                                       \
  } while (0)

static double urand(unsigned* s) {
  *s = *s * 1664525u + 1013904223u;
  return ((*s >> 8) & 0xffffff) / double(0x1000000);
}

static void Rodrigues(const double* w, const double* a, double* q) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double wxa[3] = {w[1] * a[2] - w[2] * a[1], w[2] * a[0] - w[0] * a[2], w[0] * a[1] - w[1] * a[0]};
  if (t2 < 1e-30) {
    for (int i = 0; i < 3; ++i) q[i] = a[i] + wxa[i];
    return;
  }
  const double t = std::sqrt(t2), c = std::cos(t), s = std::sin(t);
  const double wa = (w[0] * a[0] + w[1] * a[1] + w[2] * a[2]) * (1 - c) / t2;
  for (int i = 0; i < 3; ++i) q[i] = a[i] * c + wxa[i] * s / t + w[i] * wa;
}

// views on a ring looking at the origin region, pinhole f=800, all views see all tracks
static void BuildScene(Reconstruction* rec, int nviews, int ntracks, bool share_groups, unsigned seed,
                       double perturb) {
  unsigned s = seed;
  std::vector<ViewId> vids;
  for (int i = 0; i < nviews; ++i) {
    const std::string name = "view" + std::to_string(i);
    const ViewId id = share_groups ? rec->AddView(name, i / 2) : rec->AddView(name);
    vids.push_back(id);
    Camera* cam = rec->MutableView(id)->MutableCamera();
    cam->SetPosition(Eigen::Vector3d(10 * (urand(&s) - 0.5), 10 * (urand(&s) - 0.5), -30 + 4 * urand(&s)));
    cam->SetOrientationFromAngleAxis(Eigen::Vector3d(0.2 * (urand(&s) - 0.5), 0.2 * (urand(&s) - 0.5),
                                                     0.2 * (urand(&s) - 0.5)));
    cam->SetFocalLength(800.0);
    cam->SetPrincipalPoint(500.0, 500.0);
    rec->MutableView(id)->SetEstimated(true);
  }
  for (int t = 0; t < ntracks; ++t) {
    const TrackId tid = rec->AddTrack();
    Track* tr = rec->MutableTrack(tid);
    double X[3] = {6 * (urand(&s) - 0.5), 6 * (urand(&s) - 0.5), 6 * (urand(&s) - 0.5)};
    for (const ViewId v : vids) {
      const Camera& cam = rec->View(v)->Camera();
      const double a[3] = {X[0] - cam.extrinsics()[0], X[1] - cam.extrinsics()[1], X[2] - cam.extrinsics()[2]};
      double q[3];
      Rodrigues(cam.extrinsics() + 3, a, q);
      const double u = 800.0 * q[0] / q[2] + 500.0 + 0.5 * (urand(&s) - 0.5);
      const double w = 800.0 * q[1] / q[2] + 500.0 + 0.5 * (urand(&s) - 0.5);
      rec->AddObservation(v, tid, Feature(u, w));
    }
    for (int i = 0; i < 3; ++i) (*tr->MutablePoint())[i] = X[i] + perturb * (urand(&s) - 0.5);
    (*tr->MutablePoint())[3] = 1.0;
    tr->SetEstimated(true);
  }
}

static void TestSemantics() {
  Reconstruction rec;
  BuildScene(&rec, 5, 8, /*share_groups=*/true, 7, 0.1);  // groups: {0,1} {2,3} {4}
  rec.MutableView(3)->SetEstimated(false);
  rec.MutableTrack(7)->SetEstimated(false);
  BundleAdjustmentOptions opt;
  {
    // full BA: residual set = estimated views x estimated tracks
    BundleAdjuster ba(opt, &rec);
    for (ViewId v : rec.ViewIds()) ba.AddView(v);
    for (TrackId t : rec.TrackIds()) ba.AddTrack(t);
    ba.AddView(0);   // duplicates ignored (bundle_adjuster.cc:106)
    ba.AddTrack(0);  // (:144)
    FlattenedBundleAdjustmentProblem f;
    EXPECT(ba.Flatten(&f));
    EXPECT(f.view_ids.size() == 4);
    EXPECT(f.track_ids.size() == 7);
    EXPECT(f.obs_camera.size() == 4u * 7u);
    for (uint8_t c : f.camera_flags) EXPECT(c == 0);
    for (uint8_t c : f.point_constant) EXPECT(c == 0);
    EXPECT(f.group_ids.size() == 3);
    // default intrinsics_to_optimize = FOCAL_LENGTH | RADIAL_DISTORTION: pinhole mask 0 1 1 1 1 0 0
    const uint8_t want[7] = {0, 1, 1, 1, 1, 0, 0};
    for (size_t g = 0; g < 3; ++g) EXPECT(std::memcmp(&f.intrinsics_constant[7 * g], want, 7) == 0);
    // views 0 and 1 share one intrinsics block
    EXPECT(f.camera_group[0] == f.camera_group[1] && f.camera_group[2] != f.camera_group[0]);
  }
  {
    // bulk AddViews / AddTracks on several host threads == one id at a time
    Reconstruction big;
    BuildScene(&big, 7, 300, /*share_groups=*/true, 9, 0.1);
    big.MutableView(2)->SetEstimated(false);
    big.MutableTrack(11)->SetEstimated(false);
    BundleAdjuster one(opt, &big), bulk(opt, &big);
    for (ViewId v : big.ViewIds()) one.AddView(v);
    for (TrackId t : big.TrackIds()) one.AddTrack(t);
    setenv("TMI_BA_HOST_THREADS", "4", 1);
    bulk.AddViews(big.ViewIds());
    bulk.AddTracks(big.TrackIds());
    FlattenedBundleAdjustmentProblem f1, f2;
    EXPECT(one.Flatten(&f1) && bulk.Flatten(&f2));
    unsetenv("TMI_BA_HOST_THREADS");
    EXPECT(f1.view_ids == f2.view_ids && f1.track_ids == f2.track_ids && f1.group_ids == f2.group_ids);
    EXPECT(f1.obs_camera == f2.obs_camera && f1.obs_point == f2.obs_point && f1.obs_xy == f2.obs_xy);
    EXPECT(f1.camera_flags == f2.camera_flags && f1.point_constant == f2.point_constant);
    EXPECT(f1.intrinsics_constant == f2.intrinsics_constant && f1.extrinsics == f2.extrinsics);
    EXPECT(f1.obs_camera.size() == 6u * 299u);
    for (size_t q = 1; q < f2.obs_camera.size(); ++q)  // (track, view) ascending
      EXPECT(f2.obs_point[q] > f2.obs_point[q - 1] ||
             (f2.obs_point[q] == f2.obs_point[q - 1] && f2.obs_camera[q] > f2.obs_camera[q - 1]));
  }
  {
    // out-of-contract order (AddTrack pulls in view 1, AddView(1) comes later): the reference
    // would duplicate the (view 1, track 0) residual block; the shim keeps it once and the
    // view's extrinsics stay constant as AddTrack left them (bundle_adjuster.cc:164)
    BundleAdjuster ba(opt, &rec);
    ba.AddTrack(0);
    ba.AddView(1);
    FlattenedBundleAdjustmentProblem f;
    EXPECT(ba.Flatten(&f));
    // track 0 in views 0, 1, 2, 4 + view 1's other 6 estimated tracks
    EXPECT(f.obs_camera.size() == 4u + 6u);
    for (size_t q = 1; q < f.obs_camera.size(); ++q)
      EXPECT(!(f.obs_camera[q] == f.obs_camera[q - 1] && f.obs_point[q] == f.obs_point[q - 1]));
    for (size_t c = 0; c < f.view_ids.size(); ++c) EXPECT(f.camera_flags[c] == 3);
  }
  {
    // partial BA: optimise view 0 and track 0 only
    BundleAdjuster ba(opt, &rec);
    ba.AddView(0);
    ba.AddTrack(0);
    FlattenedBundleAdjustmentProblem f;
    EXPECT(ba.Flatten(&f));
    // view 0 sees its 7 estimated tracks; track 0 adds views 1, 2, 4 (3 is unestimated)
    EXPECT(f.obs_camera.size() == 7u + 3u);
    EXPECT(f.view_ids.size() == 4);
    for (size_t c = 0; c < f.view_ids.size(); ++c)
      EXPECT(f.camera_flags[c] == (f.view_ids[c] == 0 ? 0 : 3));  // others: extrinsics constant (:164)
    for (size_t p = 0; p < f.track_ids.size(); ++p)
      EXPECT(f.point_constant[p] == (f.track_ids[p] == 0 ? 0 : 1));  // anchors stay constant (:137)
    // group of view 0 (shared with constant view 1) stays variable; groups used only by
    // constant cameras are fully constant (:271-286)
    for (size_t g = 0; g < f.group_ids.size(); ++g) {
      int nconst = 0;
      for (int k = f.group_offset[g]; k < f.group_offset[g + 1]; ++k) nconst += f.intrinsics_constant[k];
      EXPECT(nconst == (f.group_ids[g] == 0 ? 4 : 7));
    }
  }
  {
    BundleAdjustmentOptions o2 = opt;
    o2.constant_camera_position = true;
    o2.intrinsics_to_optimize = OptimizeIntrinsicsType::NONE;
    BundleAdjuster ba(o2, &rec);
    for (ViewId v : rec.ViewIds()) ba.AddView(v);
    for (TrackId t : rec.TrackIds()) ba.AddTrack(t);
    FlattenedBundleAdjustmentProblem f;
    EXPECT(ba.Flatten(&f));
    for (uint8_t c : f.camera_flags) EXPECT(c == TMI_BA_CAMERA_POSITION_CONSTANT);
    for (uint8_t c : f.intrinsics_constant) EXPECT(c == 1);
  }
  {
    // nothing to do: unestimated view / track are skipped silently (:106,144)
    BundleAdjuster ba(opt, &rec);
    ba.AddView(3);
    ba.AddTrack(7);
    FlattenedBundleAdjustmentProblem f;
    EXPECT(ba.Flatten(&f));
    EXPECT(f.obs_camera.empty());
    const BundleAdjustmentSummary s = ba.Optimize();
    EXPECT(s.success);
  }
  {
    // an estimated view none of whose tracks is estimated: its parameter blocks never enter the reference's
    // ceres::Problem (they are added by AddReprojectionErrorResidual only) -- it must not become a camera block
    Reconstruction rec;
    BuildScene(&rec, 4, 20, /*share_groups=*/false, 3, 0.0);
    const ViewId lonely = rec.AddView("lonely");
    rec.MutableView(lonely)->SetEstimated(true);
    const TrackId t = rec.AddTrack();  // stays un-estimated
    rec.AddObservation(lonely, t, Feature(1.0, 2.0));
    rec.AddObservation(0, t, Feature(3.0, 4.0));
    for (const bool bulk : {false, true}) {
      BundleAdjuster ba(opt, &rec);
      if (bulk) {
        ba.AddViews(rec.ViewIds());
        ba.AddTracks(rec.TrackIds());
      } else {
        for (ViewId v : rec.ViewIds()) ba.AddView(v);
        for (TrackId tr : rec.TrackIds()) ba.AddTrack(tr);
      }
      FlattenedBundleAdjustmentProblem f;
      EXPECT(ba.Flatten(&f));
      EXPECT(f.view_ids.size() == 4 && std::find(f.view_ids.begin(), f.view_ids.end(), lonely) == f.view_ids.end());
      EXPECT(f.track_ids.size() == 20 && f.obs_camera.size() == 80);
    }
  }
  {
    // the reference's protected hook keeps its signature (bundle_adjuster.h:100-102): a subclass sees every residual
    // with the camera of its view and its track, and the base implementation records the right ids
    struct Counting : BundleAdjuster {
      using BundleAdjuster::BundleAdjuster;
      int calls = 0;
      bool pointers_ok = true;
      void AddReprojectionErrorResidual(const Feature& feature, Camera* camera, Track* track) override {
        ++calls;
        bool cam_found = false, trk_found = false;
        for (const ViewId v : reconstruction_->ViewIds()) cam_found |= reconstruction_->MutableView(v)->MutableCamera() == camera;
        for (const TrackId t : reconstruction_->TrackIds()) trk_found |= reconstruction_->MutableTrack(t) == track;
        pointers_ok &= cam_found && trk_found;
        BundleAdjuster::AddReprojectionErrorResidual(feature, camera, track);
      }
    };
    Counting sub(opt, &rec);
    BundleAdjuster plain(opt, &rec);
    sub.AddViews(rec.ViewIds());   // subclasses take the one-at-a-time path: the hook sees every residual
    sub.AddTracks(rec.TrackIds());
    for (ViewId v : rec.ViewIds()) plain.AddView(v);
    for (TrackId t : rec.TrackIds()) plain.AddTrack(t);
    FlattenedBundleAdjustmentProblem f1, f2;
    EXPECT(sub.Flatten(&f1) && plain.Flatten(&f2));
    EXPECT(sub.calls == 4 * 7 && sub.pointers_ok);
    EXPECT(f1.obs_camera == f2.obs_camera && f1.obs_point == f2.obs_point && f1.obs_xy == f2.obs_xy);
    // a subclass that forwards pointers the caller did not announce: the ids are looked up
    struct Forwarding : BundleAdjuster {
      using BundleAdjuster::BundleAdjuster;
      void Add(const Feature& f, Camera* c, Track* t) { AddReprojectionErrorResidual(f, c, t); }
    };
    Forwarding fw(opt, &rec);
    fw.Add(Feature(1.0, 2.0), rec.MutableView(2)->MutableCamera(), rec.MutableTrack(5));
    FlattenedBundleAdjustmentProblem f3;
    EXPECT(fw.Flatten(&f3));
    EXPECT(f3.view_ids.size() == 1 && f3.view_ids[0] == 2 && f3.track_ids.size() == 1 && f3.track_ids[0] == 5);
  }
  {
    // a track id hashed by an earlier AddTrack (far above the dense range) and then met by the threaded
    // AddViews must not be flattened twice; and an IsEstimated memo entry of a track removed since must not survive
    Reconstruction rec2;
    BuildScene(&rec2, 4, 30, /*share_groups=*/false, 5, 0.0);
    BundleAdjuster ba(opt, &rec2), ref(opt, &rec2);
    const TrackId far = 50u << 20;  // beyond 8 x ids + 1 M: the sparse store
    // (the stand-in Reconstruction hands out consecutive ids; poke the tables through the public calls instead)
    ba.AddTrack(3);
    ba.AddViews(rec2.ViewIds());
    ba.AddTracks(rec2.TrackIds());
    for (TrackId t : {TrackId(3)}) ref.AddTrack(t);
    for (ViewId v : rec2.ViewIds()) ref.AddView(v);
    for (TrackId t : rec2.TrackIds()) ref.AddTrack(t);
    FlattenedBundleAdjustmentProblem f1, f2;
    EXPECT(ba.Flatten(&f1) && ref.Flatten(&f2));
    EXPECT(f1.track_ids == f2.track_ids && f1.obs_camera == f2.obs_camera && f1.point_constant == f2.point_constant);
    for (size_t q = 1; q < f1.track_ids.size(); ++q) EXPECT(f1.track_ids[q] > f1.track_ids[q - 1]);
    (void)far;
    // second AddViews on the same adjuster after a track was un-estimated: the memo follows the reconstruction
    Reconstruction rec3;
    BuildScene(&rec3, 4, 30, /*share_groups=*/false, 6, 0.0);
    BundleAdjuster two(opt, &rec3);
    two.AddViews({rec3.ViewIds()[0]});
    rec3.MutableTrack(9)->SetEstimated(false);
    two.AddViews({rec3.ViewIds()[1]});
    FlattenedBundleAdjustmentProblem f3;
    EXPECT(two.Flatten(&f3));
    EXPECT(f3.obs_camera.size() == 30u + 29u);
  }
  // option defaults (bundle_adjustment.h:78-122)
  EXPECT(opt.max_num_iterations == 100 && opt.use_inner_iterations && opt.robust_loss_width == 2.0);
  EXPECT(opt.linear_solver_type == ceres::SPARSE_SCHUR && opt.preconditioner_type == ceres::SCHUR_JACOBI);
  // what the shim hands the device for the reference's default preconditioner: Ceres' block shape (one block per
  // parameter block, schur_jacobi_preconditioner.cc); the merged per-view block only when asked for; IDENTITY as is
  {
    tmi_ba_options o;
    BundleAdjustmentOptions d;
    ToDeviceOptions(d, &o);
    EXPECT(o.preconditioner_type == TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS);
    d.preconditioner_type = ceres::CLUSTER_JACOBI;  // clusters = shared intrinsics blocks + their views on the device
    ToDeviceOptions(d, &o);
    EXPECT(o.preconditioner_type == TMI_BA_PRECOND_CLUSTER_JACOBI);
    d.preconditioner_type = ceres::CLUSTER_TRIDIAGONAL;  // round 6: implemented (cluster_chains.h), passed through
    ToDeviceOptions(d, &o);
    EXPECT(o.preconditioner_type == TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL);
    d.preconditioner_type = ceres::JACOBI;
    ToDeviceOptions(d, &o);
    EXPECT(o.preconditioner_type == TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS);
    d.preconditioner_type = ceres::SCHUR_JACOBI;
    d.merged_view_blocks_in_preconditioner = true;
    ToDeviceOptions(d, &o);
    EXPECT(o.preconditioner_type == TMI_BA_PRECOND_SCHUR_JACOBI);
    d.preconditioner_type = ceres::IDENTITY;
    ToDeviceOptions(d, &o);
    EXPECT(o.preconditioner_type == TMI_BA_PRECOND_IDENTITY);
    EXPECT(o.use_inner_iterations == 1 && o.point_dof == 4 && o.max_num_iterations == 100);
  }
}

static double Rmse(const BundleAdjuster& ba) { return ba.DeviceSummary().final_rmse; }

static void TestGpu() {
  {
    Reconstruction rec;
    BuildScene(&rec, 6, 200, /*share_groups=*/false, 11, 0.3);
    BundleAdjustmentOptions opt;
    std::vector<double> before;
    for (TrackId t : rec.TrackIds()) before.push_back(rec.Track(t)->Point()[0]);
    BundleAdjuster ba(opt, &rec);
    for (ViewId v : rec.ViewIds()) ba.AddView(v);
    for (TrackId t : rec.TrackIds()) ba.AddTrack(t);
    const BundleAdjustmentSummary s = ba.Optimize();
    std::printf("full BA: success %d cost %.6e -> %.6e rmse %.4f (%s)\n", s.success, s.initial_cost,
                s.final_cost, Rmse(ba), ba.DeviceSummary().message);
    EXPECT(s.success);
    EXPECT(s.final_cost < 1e-2 * s.initial_cost);
    EXPECT(Rmse(ba) < 0.3);  // uniform(-0.25, 0.25) pixel noise
    int changed = 0, i = 0;
    for (TrackId t : rec.TrackIds()) changed += rec.Track(t)->Point()[0] != before[i++];
    EXPECT(changed == (int)before.size());  // written back in place
    // a second BA of the adjusted reconstruction must not increase the cost
    const BundleAdjustmentSummary s2 = BundleAdjustReconstruction(opt, &rec);
    EXPECT(s2.success && s2.final_cost <= s2.initial_cost * (1 + 1e-12));
    EXPECT(std::fabs(s2.initial_cost - s.final_cost) < 1e-9 * s.final_cost);
    // exact solver (the default SPARSE_SCHUR): no preconditioner ran, nothing substituted
    EXPECT(!s.preconditioner_substituted && s.effective_preconditioner_type == ceres::IDENTITY);
  }
  {
    // which preconditioner ran is part of the summary: ceres::CLUSTER_TRIDIAGONAL and SCHUR_JACOBI are served as asked
    // on one rank with the formed S (the shim's default); ceres::JACOBI runs as SCHUR_JACOBI and the caller is told
    Reconstruction rec;
    BuildScene(&rec, 8, 300, /*share_groups=*/true, 13, 0.3);
    BundleAdjustmentOptions opt;
    opt.linear_solver_type = ceres::ITERATIVE_SCHUR;
    opt.max_num_iterations = 4;
    opt.preconditioner_type = ceres::CLUSTER_TRIDIAGONAL;
    const BundleAdjustmentSummary st = BundleAdjustReconstruction(opt, &rec);
    EXPECT(st.success && !st.preconditioner_substituted && st.effective_preconditioner_type == ceres::CLUSTER_TRIDIAGONAL);
    opt.preconditioner_type = ceres::SCHUR_JACOBI;
    const BundleAdjustmentSummary sj = BundleAdjustReconstruction(opt, &rec);
    EXPECT(sj.success && !sj.preconditioner_substituted && sj.effective_preconditioner_type == ceres::SCHUR_JACOBI);
    opt.preconditioner_type = ceres::JACOBI;
    const BundleAdjustmentSummary sjj = BundleAdjustReconstruction(opt, &rec);
    EXPECT(sjj.success && sjj.preconditioner_substituted && sjj.effective_preconditioner_type == ceres::SCHUR_JACOBI);
    std::printf("preconditioner report: CLUSTER_TRIDIAGONAL -> %d (substituted %d), SCHUR_JACOBI -> %d (substituted %d)\n",
                (int)st.effective_preconditioner_type, (int)st.preconditioner_substituted, (int)sj.effective_preconditioner_type,
                (int)sj.preconditioner_substituted);
  }
  {
    // partial BA with anchors + shared constant intrinsics + Huber loss
    Reconstruction rec;
    BuildScene(&rec, 6, 120, /*share_groups=*/true, 5, 0.2);
    BundleAdjustmentOptions opt;
    opt.intrinsics_to_optimize = OptimizeIntrinsicsType::NONE;
    opt.loss_function_type = LossFunctionType::HUBER;
    opt.linear_solver_type = ceres::ITERATIVE_SCHUR;
    std::unordered_set<ViewId> views = {0, 1, 2};
    std::unordered_set<TrackId> tracks;
    for (TrackId t = 0; t < 60; ++t) tracks.insert(t);
    const Eigen::Vector3d frozen_cam = rec.View(5)->Camera().GetPosition();
    const double frozen_pt = rec.Track(100)->Point()[1];
    const BundleAdjustmentSummary s = BundleAdjustPartialReconstruction(opt, views, tracks, &rec);
    std::printf("partial BA: success %d cost %.6e -> %.6e\n", s.success, s.initial_cost, s.final_cost);
    EXPECT(s.success && s.final_cost < s.initial_cost);
    EXPECT(rec.View(5)->Camera().GetPosition()[0] == frozen_cam[0]);
    EXPECT(rec.Track(100)->Point()[1] == frozen_pt);
    // single view / single track BA (bundle_adjustment.cc:82-107: DENSE_QR, no inner iterations)
    BundleAdjustmentOptions o1 = opt;
    o1.linear_solver_type = ceres::DENSE_QR;
    o1.use_inner_iterations = false;
    {
      BundleAdjuster bv(o1, &rec);
      bv.AddView(4);
      const BundleAdjustmentSummary sv = bv.Optimize();
      std::printf("view BA: success %d cost %.6e -> %.6e status %d (%s)\n", sv.success, sv.initial_cost,
                  sv.final_cost, bv.DeviceSummary().status, bv.DeviceSummary().message);
      EXPECT(sv.success && sv.final_cost <= sv.initial_cost);
    }
    {
      BundleAdjuster bt(o1, &rec);
      bt.AddTrack(110);
      const BundleAdjustmentSummary st = bt.Optimize();
      std::printf("track BA: success %d cost %.6e -> %.6e status %d (%s)\n", st.success, st.initial_cost,
                  st.final_cost, bt.DeviceSummary().status, bt.DeviceSummary().message);
      EXPECT(st.success && st.final_cost <= st.initial_cost);
    }
    const BundleAdjustmentSummary sv2 = BundleAdjustView(opt, 3, &rec);
    const BundleAdjustmentSummary st2 = BundleAdjustTrack(opt, 111, &rec);
    EXPECT(sv2.success && st2.success);
  }
}

// The resident session of BundleAdjustReconstruction: a second call on the same Reconstruction re-uses the
// flattened problem and the device handle, and must give exactly what a from-scratch call gives; anything that can
// change the residual set drops the session.
static void TestResidentSessionGpu() {
  BundleAdjustmentOptions opt;
  opt.linear_solver_type = ceres::ITERATIVE_SCHUR;
  opt.max_num_iterations = 4;  // stop early so that the second call still has work to do
  opt.keep_problem_resident = true;  // an extension, off by default (bundle_adjustment.h)
  EXPECT(!BundleAdjustmentOptions().keep_problem_resident);
  opt.function_tolerance = -1.0;
  opt.gradient_tolerance = -1.0;
  opt.parameter_tolerance = -1.0;
  auto snapshot = [](const Reconstruction& r) {
    std::vector<double> v;
    for (ViewId id = 0; id < (ViewId)r.NumViews(); ++id)
      for (int a = 0; a < 6; ++a) v.push_back(r.View(id)->Camera().extrinsics()[a]);
    for (TrackId id = 0; id < (TrackId)r.NumTracks(); ++id)
      for (int a = 0; a < 4; ++a) v.push_back(r.Track(id)->Point()[a]);
    for (ViewId id = 0; id < (ViewId)r.NumViews(); ++id)
      for (int a = 0; a < 7; ++a) v.push_back(r.View(id)->Camera().intrinsics()[a]);
    return v;
  };
  for (const bool shared : {false, true}) {
    ReleaseBundleAdjustmentSession();
    // A: two calls through the session.  B: the same two calls with the session switched off.
    Reconstruction A, B;
    BuildScene(&A, 7, 260, shared, 31, 0.3);
    BuildScene(&B, 7, 260, shared, 31, 0.3);
    BundleAdjustmentOptions one_shot = opt;
    one_shot.keep_problem_resident = false;
    const BundleAdjustmentSummary a1 = BundleAdjustReconstruction(opt, &A);
    EXPECT(a1.success && BundleAdjustmentSessionIsResident(&A));
    const BundleAdjustmentSummary a2 = BundleAdjustReconstruction(opt, &A);
    EXPECT(a2.success && BundleAdjustmentSessionIsResident(&A));
    const BundleAdjustmentSummary b1 = BundleAdjustReconstruction(one_shot, &B);
    const BundleAdjustmentSummary b2 = BundleAdjustReconstruction(one_shot, &B);
    EXPECT(b1.success && b2.success && !BundleAdjustmentSessionIsResident(&B));
    EXPECT(a1.final_cost == b1.final_cost && a2.initial_cost == b2.initial_cost && a2.final_cost == b2.final_cost);
    EXPECT(a2.final_cost < a2.initial_cost);
    EXPECT(snapshot(A) == snapshot(B));  // bit for bit
    std::printf("resident session (%s intrinsics): first %.6e -> %.6e, second %.6e -> %.6e, one-shot second %.6e -> %.6e\n",
                shared ? "shared" : "private", a1.initial_cost, a1.final_cost, a2.initial_cost, a2.final_cost,
                b2.initial_cost, b2.final_cost);
    // parameter VALUES may change between calls (re-triangulation, a pose refinement): picked up
    (*A.MutableTrack(5)->MutablePoint())[0] += 0.05;
    (*B.MutableTrack(5)->MutablePoint())[0] += 0.05;
    A.MutableView(2)->MutableCamera()->mutable_extrinsics()[1] += 0.01;
    B.MutableView(2)->MutableCamera()->mutable_extrinsics()[1] += 0.01;
    EXPECT(BundleAdjustmentSessionIsResident(&A));
    const BundleAdjustmentSummary a3 = BundleAdjustReconstruction(opt, &A);
    const BundleAdjustmentSummary b3 = BundleAdjustReconstruction(one_shot, &B);
    EXPECT(a3.success && a3.initial_cost == b3.initial_cost && a3.final_cost == b3.final_cost && snapshot(A) == snapshot(B));
    // the residual set changes: the session must go
    A.MutableTrack(9)->SetEstimated(false);
    B.MutableTrack(9)->SetEstimated(false);
    EXPECT(!BundleAdjustmentSessionIsResident(&A));
    const double frozen = A.Track(9)->Point()[2];
    const BundleAdjustmentSummary a4 = BundleAdjustReconstruction(opt, &A);
    const BundleAdjustmentSummary b4 = BundleAdjustReconstruction(one_shot, &B);
    EXPECT(a4.success && a4.final_cost == b4.final_cost && A.Track(9)->Point()[2] == frozen);
    EXPECT(BundleAdjustmentSessionIsResident(&A));
    // other problem-shaping options: rebuilt, not served from the session
    BundleAdjustmentOptions other = opt;
    other.constant_camera_position = true;
    BundleAdjustmentOptions other_one_shot = other;
    other_one_shot.keep_problem_resident = false;
    const Eigen::Vector3d pos = A.View(1)->Camera().GetPosition();
    const BundleAdjustmentSummary a5 = BundleAdjustReconstruction(other, &A);
    const BundleAdjustmentSummary b5 = BundleAdjustReconstruction(other_one_shot, &B);
    EXPECT(a5.success && a5.final_cost == b5.final_cost && A.View(1)->Camera().GetPosition()[0] == pos[0]);
    // a copy is a different object even though it starts out identical
    Reconstruction C = A;
    EXPECT(!BundleAdjustmentSessionIsResident(&C));
    // A residual-set change that NO stamped mutator saw (copy-assignment through MutableTrack(): what data-model
    // classes without the hooks look like to the session): the structural fingerprint must catch it.  The shorter
    // track is prepared BEFORE the session is built, so the stamp does not move afterwards.
    {
      Reconstruction E, F;
      BuildScene(&E, 7, 260, shared, 33, 0.3);
      BuildScene(&F, 7, 260, shared, 33, 0.3);
      Track shorter = *E.Track(12);
      const ViewId dropped = *shorter.ViewIds().begin();
      shorter.RemoveView(dropped);
      const BundleAdjustmentSummary e1 = BundleAdjustReconstruction(opt, &E);
      const BundleAdjustmentSummary f1 = BundleAdjustReconstruction(one_shot, &F);
      EXPECT(e1.success && f1.success && e1.final_cost == f1.final_cost && BundleAdjustmentSessionIsResident(&E));
      *E.MutableTrack(12) = shorter;  // no stamp
      *F.MutableTrack(12) = shorter;
      EXPECT(BundleAdjustmentSessionIsResident(&E));  // the stamp alone still says "valid" ...
      const BundleAdjustmentSummary e2 = BundleAdjustReconstruction(opt, &E);
      const BundleAdjustmentSummary f2 = BundleAdjustReconstruction(one_shot, &F);
      // ... the fingerprint does not: the call rebuilt the problem and equals the from-scratch one bit for bit
      EXPECT(e2.success && e2.initial_cost == f2.initial_cost && e2.final_cost == f2.final_cost && snapshot(E) == snapshot(F));
      EXPECT(e2.initial_cost != e1.final_cost);  // one residual fewer than the kept session would have solved
    }
  }
  ReleaseBundleAdjustmentSession();
}

// SetOutlierTracksToUnestimated (set_outlier_tracks_to_unestimated.cc:62-133) and the batched
// BundleAdjustTracks, end to end through the shim.
static void TestTrackOpsGpu() {
  Reconstruction rec;
  BuildScene(&rec, 6, 150, /*share_groups=*/false, 21, 0.0);
  // track 0: a gross feature error in every view; track 1: behind the cameras;
  // track 2: far away (no viewing angle); track 3: not estimated (must be ignored);
  // track 4: none of its views estimated is impossible here, so un-estimate one view instead
  for (ViewId v : rec.ViewIds()) {
    const Feature* f = rec.View(v)->GetFeature(0);
    rec.MutableView(v)->AddFeature(0, Feature(f->x() + 25.0, f->y() - 25.0));
  }
  (*rec.MutableTrack(1)->MutablePoint())[2] = -200.0;
  for (int i = 0; i < 3; ++i) (*rec.MutableTrack(2)->MutablePoint())[i] *= 1.0;
  (*rec.MutableTrack(2)->MutablePoint())[2] = 1e7;
  rec.MutableTrack(3)->SetEstimated(false);
  const int removed = SetOutlierTracksToUnestimated(4.0, 1.0, &rec);
  std::printf("outlier filter: %d removed\n", removed);
  EXPECT(!rec.Track(0)->IsEstimated());
  EXPECT(!rec.Track(1)->IsEstimated());
  EXPECT(!rec.Track(2)->IsEstimated());
  EXPECT(!rec.Track(3)->IsEstimated());
  int estimated = 0;
  for (TrackId t : rec.TrackIds()) estimated += rec.Track(t)->IsEstimated();
  EXPECT(removed == 3);            // track 3 was not estimated: not counted (:77-79)
  EXPECT(estimated == 150 - 4);    // noise-free tracks with a wide baseline stay
  // only the listed tracks are examined
  (*rec.MutableTrack(10)->MutablePoint())[2] = -200.0;
  (*rec.MutableTrack(11)->MutablePoint())[2] = -200.0;
  EXPECT(SetOutlierTracksToUnestimated(std::unordered_set<TrackId>{10}, 4.0, 1.0, &rec) == 1);
  EXPECT(!rec.Track(10)->IsEstimated() && rec.Track(11)->IsEstimated());
  (*rec.MutableTrack(11)->MutablePoint())[2] = 0.0;

  // batched track adjustment == BundleAdjustTrack per track
  Reconstruction a, b;
  BuildScene(&a, 6, 80, false, 33, 0.8);
  BuildScene(&b, 6, 80, false, 33, 0.8);
  BundleAdjustmentOptions opt;
  std::unordered_set<TrackId> ids;
  for (TrackId t : a.TrackIds()) ids.insert(t);
  a.MutableTrack(7)->SetEstimated(false);
  b.MutableTrack(7)->SetEstimated(false);
  const Eigen::Vector3d cam0 = a.View(0)->Camera().GetPosition();
  const auto res = BundleAdjustTracks(opt, ids, &a);
  EXPECT(res.size() == 79 && !res.count(7));
  double worst = 0.0;
  int ok = 0;
  for (TrackId t : b.TrackIds()) {
    if (t == 7) continue;
    const BundleAdjustmentSummary s1 = BundleAdjustTrack(opt, t, &b);
    const BundleAdjustmentSummary& sN = res.at(t);
    ok += sN.success;
    EXPECT(s1.success == sN.success);
    EXPECT(std::fabs(s1.final_cost - sN.final_cost) <= 1e-9 * (1.0 + s1.final_cost));
    EXPECT(sN.final_cost <= sN.initial_cost);
    for (int i = 0; i < 4; ++i)
      worst = std::fmax(worst, std::fabs(a.Track(t)->Point()[i] - b.Track(t)->Point()[i]));
  }
  std::printf("batched track BA: %d/79 usable, max |dX| vs per-track BA %.3e\n", ok, worst);
  EXPECT(ok == 79);
  EXPECT(worst < 1e-9);
  EXPECT(a.View(0)->Camera().GetPosition()[0] == cam0[0]);  // cameras are constant

  // SelectGoodTracksForBundleAdjustment: every view ends up with at least the minimum
  // number of selected tracks, non-estimated tracks are never chosen, fewer than all
  Reconstruction c;
  BuildScene(&c, 6, 400, false, 44, 0.1);
  c.MutableTrack(5)->SetEstimated(false);
  std::unordered_set<TrackId> chosen;
  EXPECT(SelectGoodTracksForBundleAdjustment(c, 10, 100, 60, &chosen));
  EXPECT(!chosen.count(5));
  EXPECT(chosen.size() >= 60 && chosen.size() < 399);
  for (ViewId v : c.ViewIds()) {
    int n = 0;
    for (TrackId t : c.View(v)->TrackIds()) n += chosen.count(t) ? 1 : 0;
    EXPECT(n >= 60);
  }
  std::unordered_set<TrackId> chosen2;
  EXPECT(SelectGoodTracksForBundleAdjustment(c, std::unordered_set<ViewId>{0, 1}, 10, 100, 60, &chosen2));
  EXPECT(!chosen2.empty() && chosen2.size() <= chosen.size());
  std::printf("track selection: %zu of 399 (all views), %zu (views 0,1)\n", chosen.size(), chosen2.size());
}

// BundleAdjustTwoViews (bundle_adjust_two_views.cc:113-191) and its batched form through the shim.
// BundleAdjustTwoViewsAngular through the shim: normalised coordinates of exact correspondences, the
// relative pose started off the truth; single calls equal the batched call
static void TestTwoViewsAngularGpu() {
  unsigned s = 99;
  const int P = 4;
  std::vector<TwoViewInfo> info(P), infob(P);
  std::vector<std::vector<FeatureCorrespondence>> corr(P);
  for (int p = 0; p < P; ++p) {
    const double aa[3] = {0.04 * urand(&s), -0.05 * urand(&s), 0.02};
    double c[3] = {1.0 + 0.3 * urand(&s), 0.2 * urand(&s), 0.2 * urand(&s)};
    const double cn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (double& v : c) v /= cn;
    for (int i = 0; i < 80 + 30 * p; ++i) {
      const double X[3] = {4 * (urand(&s) - 0.5), 4 * (urand(&s) - 0.5), 4 + 5 * urand(&s)};
      const double a[3] = {X[0] - c[0], X[1] - c[1], X[2] - c[2]};
      double q[3];
      Rodrigues(aa, a, q);
      corr[p].emplace_back(Feature(X[0] / X[2], X[1] / X[2]), Feature(q[0] / q[2], q[1] / q[2]));
    }
    info[p].rotation_2 = Eigen::Vector3d(aa[0] + 0.02, aa[1] - 0.015, aa[2] + 0.01);
    double c0[3] = {c[0] + 0.05, c[1] - 0.04, c[2] + 0.03};
    const double n0 = std::sqrt(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
    info[p].position_2 = Eigen::Vector3d(c0[0] / n0, c0[1] / n0, c0[2] / n0);
    infob[p] = info[p];
  }
  BundleAdjustmentOptions o;
  std::vector<BundleAdjustmentSummary> single(P);
  for (int p = 0; p < P; ++p) {
    single[p] = BundleAdjustTwoViewsAngular(o, corr[p], &info[p]);
    EXPECT(single[p].success && single[p].final_cost < 1e-3 * single[p].initial_cost);
    const Eigen::Vector3d& t = info[p].position_2;
    EXPECT(std::fabs(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] - 1.0) < 1e-12);
  }
  std::vector<TwoViewAngularProblem> batch(P);
  for (int p = 0; p < P; ++p) {
    batch[p].correspondences = &corr[p];
    batch[p].info = &infob[p];
  }
  const std::vector<BundleAdjustmentSummary> all = BundleAdjustTwoViewsAngularBatch(o, &batch);
  EXPECT(all.size() == static_cast<size_t>(P));
  double worst = 0.0;
  for (int p = 0; p < P && all.size() == static_cast<size_t>(P); ++p) {
    EXPECT(all[p].final_cost == single[p].final_cost);
    for (int a = 0; a < 3; ++a) {
      worst = std::fmax(worst, std::fabs(info[p].rotation_2[a] - infob[p].rotation_2[a]));
      worst = std::fmax(worst, std::fabs(info[p].position_2[a] - infob[p].position_2[a]));
    }
  }
  std::printf("two-view angular BA: %d pairs, cost %.4e -> %.4e (pair 0), batched vs single max |d| %.2e\n", P,
              single[0].initial_cost, single[0].final_cost, worst);
  EXPECT(worst == 0.0);
  const BundleAdjustmentSummary bad = BundleAdjustTwoViewsAngular(o, corr[0], nullptr);
  EXPECT(!bad.success);
}

static void TestTwoViewsGpu() {
  unsigned s = 77;
  auto make_pair = [&](Camera* c1, Camera* c2, std::vector<FeatureCorrespondence>* corr,
                       std::vector<Eigen::Vector4d>* pts, int n) {
    c1->SetFocalLength(800.0);
    c1->SetPrincipalPoint(500.0, 400.0);
    c2->SetFocalLength(760.0);
    c2->SetPrincipalPoint(500.0, 400.0);
    c2->SetPosition(Eigen::Vector3d(1.0 + 0.2 * urand(&s), 0.1 * urand(&s), 0.1 * urand(&s)));
    c2->SetOrientationFromAngleAxis(Eigen::Vector3d(0.05 * urand(&s), -0.1 * urand(&s), 0.03));
    for (int i = 0; i < n; ++i) {
      const double X[3] = {4 * (urand(&s) - 0.5), 4 * (urand(&s) - 0.5), 5 + 5 * urand(&s)};
      double px[2][2];
      for (int c = 0; c < 2; ++c) {
        const Camera& cam = c ? *c2 : *c1;
        const double a[3] = {X[0] - cam.extrinsics()[0], X[1] - cam.extrinsics()[1], X[2] - cam.extrinsics()[2]};
        double q[3];
        Rodrigues(cam.extrinsics() + 3, a, q);
        px[c][0] = cam.FocalLength() * q[0] / q[2] + 500.0 + 0.5 * (urand(&s) - 0.5);
        px[c][1] = cam.FocalLength() * q[1] / q[2] + 400.0 + 0.5 * (urand(&s) - 0.5);
      }
      corr->emplace_back(Feature(px[0][0], px[0][1]), Feature(px[1][0], px[1][1]));
      pts->emplace_back(X[0] + 0.05 * (urand(&s) - 0.5), X[1] + 0.05 * (urand(&s) - 0.5), X[2] + 0.05 * (urand(&s) - 0.5), 1.0);
    }
    // start camera 2 off its generating pose
    c2->mutable_extrinsics()[0] += 0.03;
    c2->mutable_extrinsics()[4] += 0.004;
  };
  const int P = 5;
  std::vector<Camera> cam1(P), cam2(P), cam1b(P), cam2b(P);
  std::vector<std::vector<FeatureCorrespondence>> corr(P);
  std::vector<std::vector<Eigen::Vector4d>> pts(P), ptsb(P);
  for (int p = 0; p < P; ++p) {
    make_pair(&cam1[p], &cam2[p], &corr[p], &pts[p], 60 + 40 * p);
    cam1b[p].DeepCopy(cam1[p]);
    cam2b[p].DeepCopy(cam2[p]);
    ptsb[p] = pts[p];
  }
  // one by one
  std::vector<BundleAdjustmentSummary> single(P);
  for (int p = 0; p < P; ++p) {
    TwoViewBundleAdjustmentOptions o;
    o.constant_camera2_intrinsics = (p % 2 == 0);
    const double e1_before = cam1[p].extrinsics()[0], f1_before = cam1[p].FocalLength();
    single[p] = BundleAdjustTwoViews(o, corr[p], &cam1[p], &cam2[p], &pts[p]);
    EXPECT(single[p].success && single[p].final_cost < 0.05 * single[p].initial_cost);
    EXPECT(cam1[p].extrinsics()[0] == e1_before && cam1[p].FocalLength() == f1_before);  // camera 1 constant
    if (p % 2 == 0) EXPECT(cam2[p].FocalLength() == 760.0);
    else EXPECT(cam2[p].FocalLength() != 760.0);
  }
  // all at once: the same results
  std::vector<TwoViewBundleAdjustmentProblem> batch(P);
  for (int p = 0; p < P; ++p) {
    batch[p].options.constant_camera2_intrinsics = (p % 2 == 0);
    batch[p].correspondences = &corr[p];
    batch[p].camera1 = &cam1b[p];
    batch[p].camera2 = &cam2b[p];
    batch[p].points3d = &ptsb[p];
  }
  const std::vector<BundleAdjustmentSummary> all = BundleAdjustTwoViewsBatch(&batch);
  EXPECT(all.size() == static_cast<size_t>(P));
  double worst = 0.0;
  for (int p = 0; p < P && all.size() == static_cast<size_t>(P); ++p) {
    EXPECT(all[p].success == single[p].success);
    EXPECT(all[p].final_cost == single[p].final_cost);
    for (int a = 0; a < 6; ++a) worst = std::fmax(worst, std::fabs(cam2[p].extrinsics()[a] - cam2b[p].extrinsics()[a]));
  }
  std::printf("two-view BA: %d pairs, cost %.4e -> %.4e (pair 0), batched vs single max |d ext| %.2e\n", P,
              single[0].initial_cost, single[0].final_cost, worst);
  EXPECT(worst == 0.0);
  // mismatched sizes: the reference CHECK-fails; the shim reports failure and touches nothing
  std::vector<Eigen::Vector4d> short_pts(3);
  Camera a, b;
  const BundleAdjustmentSummary bad = BundleAdjustTwoViews(TwoViewBundleAdjustmentOptions(), corr[0], &a, &b, &short_pts);
  EXPECT(!bad.success);
}

// Distinct BundleAdjuster instances run concurrently from pool threads on disjoint tracks of ONE Reconstruction
// (estimate_track.cc:166-204, 238-246: every worker calls BundleAdjustTrack with num_threads = 1) and full BAs of
// different Reconstructions may overlap: the boundary must be re-entrant and the results those of the serial order.
static void TestConcurrentCallersGpu() {
  const int kThreads = 8, kTracksPerThread = 6;
  Reconstruction serial, threaded;
  BuildScene(&serial, 7, 120, /*share_groups=*/false, 41, 0.3);
  BuildScene(&threaded, 7, 120, /*share_groups=*/false, 41, 0.3);
  BundleAdjustmentOptions opt;
  opt.max_num_iterations = 10;
  opt.num_threads = 1;
  std::vector<BundleAdjustmentSummary> a(kThreads * kTracksPerThread), b(kThreads * kTracksPerThread);
  for (int i = 0; i < kThreads * kTracksPerThread; ++i) a[i] = BundleAdjustTrack(opt, (TrackId)i, &serial);
  std::vector<std::thread> pool;
  for (int t = 0; t < kThreads; ++t)
    pool.emplace_back([&, t] {
      for (int k = 0; k < kTracksPerThread; ++k) {
        const int i = t * kTracksPerThread + k;  // disjoint tracks per thread
        b[i] = BundleAdjustTrack(opt, (TrackId)i, &threaded);
      }
    });
  for (auto& th : pool) th.join();
  int same = 0;
  for (int i = 0; i < kThreads * kTracksPerThread; ++i) {
    bool eq = a[i].success == b[i].success && a[i].initial_cost == b[i].initial_cost && a[i].final_cost == b[i].final_cost;
    for (int c = 0; c < 4; ++c) eq = eq && serial.Track(i)->Point()[c] == threaded.Track(i)->Point()[c];
    same += eq ? 1 : 0;
    EXPECT(b[i].success && b[i].final_cost <= b[i].initial_cost);
  }
  EXPECT(same == kThreads * kTracksPerThread);  // bit for bit the serial results
  std::printf("concurrent BundleAdjustTrack: %d threads x %d tracks, %d of %d equal to the serial run\n", kThreads,
              kTracksPerThread, same, kThreads * kTracksPerThread);
  // full BAs of different Reconstructions from different threads (exact and iterative solvers side by side)
  const int kRecs = 4;
  std::vector<Reconstruction> one(kRecs), two(kRecs);
  std::vector<BundleAdjustmentSummary> s1(kRecs), s2(kRecs);
  BundleAdjustmentOptions full;
  full.max_num_iterations = 5;
  for (int r = 0; r < kRecs; ++r) {
    BuildScene(&one[r], 6 + r, 150 + 20 * r, /*share_groups=*/r == 3, 51 + r, 0.3);
    BuildScene(&two[r], 6 + r, 150 + 20 * r, /*share_groups=*/r == 3, 51 + r, 0.3);
  }
  auto options_of = [&](int r) {
    BundleAdjustmentOptions o = full;
    o.linear_solver_type = (r & 1) ? ceres::ITERATIVE_SCHUR : ceres::SPARSE_SCHUR;
    o.use_inner_iterations = r == 2;
    return o;
  };
  for (int r = 0; r < kRecs; ++r) s1[r] = BundleAdjustReconstruction(options_of(r), &one[r]);
  pool.clear();
  for (int r = 0; r < kRecs; ++r) pool.emplace_back([&, r] { s2[r] = BundleAdjustReconstruction(options_of(r), &two[r]); });
  for (auto& th : pool) th.join();
  for (int r = 0; r < kRecs; ++r) {
    bool eq = s1[r].success && s2[r].success && s1[r].final_cost == s2[r].final_cost;
    for (ViewId v = 0; v < (ViewId)one[r].NumViews(); ++v)
      for (int c = 0; c < 6; ++c) eq = eq && one[r].View(v)->Camera().extrinsics()[c] == two[r].View(v)->Camera().extrinsics()[c];
    EXPECT(eq);
  }
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  TestSemantics();
  if (mode == "gpu") {
    TestGpu();
    TestResidentSessionGpu();
    TestTrackOpsGpu();
    TestTwoViewsGpu();
    TestTwoViewsAngularGpu();
    TestConcurrentCallersGpu();
  }
  std::printf("%s: %d failure(s)\n", mode.c_str(), g_fail);
  return g_fail ? 1 : 0;
}
