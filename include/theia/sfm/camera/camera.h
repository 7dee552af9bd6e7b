// reference: src/theia/sfm/camera/camera.h:60-252 -- the subset the BA boundary
// touches: 6 extrinsics [position(3), angle_axis(3)] (:195-200), a shared_ptr to
// the intrinsics model (shared between the views of an intrinsics group, :247),
// raw parameter accessors (:181-190) and the copy semantics (:69-74: copies share
// intrinsics).
#ifndef THEIA_MI355_CAMERA_H_
#define THEIA_MI355_CAMERA_H_
#include <memory>
#include "theia/sfm/camera/camera_intrinsics_model.h"
#include "theia/util/eigen_lite.h"

namespace theia {
class Camera {
 public:
  Camera() : Camera(CameraIntrinsicsModelType::PINHOLE) {}
  explicit Camera(const CameraIntrinsicsModelType& camera_type)
      : camera_intrinsics_(CameraIntrinsicsModel::Create(camera_type)) {
    for (double& p : camera_parameters_) p = 0.0;
    image_size_[0] = image_size_[1] = 0;
  }
  // shallow copies share the intrinsics, as in the reference
  Camera(const Camera&) = default;
  Camera& operator=(const Camera&) = default;
  void DeepCopy(const Camera& camera) {
    *this = camera;
    camera_intrinsics_ = std::make_shared<CameraIntrinsicsModel>(*camera.camera_intrinsics_);
  }

  CameraIntrinsicsModelType GetCameraIntrinsicsModelType() const { return camera_intrinsics_->Type(); }
  void SetCameraIntrinsicsModelType(const CameraIntrinsicsModelType& t) {
    if (t != camera_intrinsics_->Type()) camera_intrinsics_ = CameraIntrinsicsModel::Create(t);
  }

  void SetPosition(const Eigen::Vector3d& p) { for (int i = 0; i < 3; ++i) camera_parameters_[POSITION + i] = p[i]; }
  Eigen::Vector3d GetPosition() const {
    return Eigen::Vector3d(camera_parameters_[0], camera_parameters_[1], camera_parameters_[2]);
  }
  void SetOrientationFromAngleAxis(const Eigen::Vector3d& aa) {
    for (int i = 0; i < 3; ++i) camera_parameters_[ORIENTATION + i] = aa[i];
  }
  Eigen::Vector3d GetOrientationAsAngleAxis() const {
    return Eigen::Vector3d(camera_parameters_[3], camera_parameters_[4], camera_parameters_[5]);
  }
  void SetFocalLength(const double f) { camera_intrinsics_->SetFocalLength(f); }
  double FocalLength() const { return camera_intrinsics_->FocalLength(); }
  void SetPrincipalPoint(const double px, const double py) { camera_intrinsics_->SetPrincipalPoint(px, py); }
  void SetImageSize(const int w, const int h) { image_size_[0] = w; image_size_[1] = h; }
  int ImageWidth() const { return image_size_[0]; }
  int ImageHeight() const { return image_size_[1]; }

  const std::shared_ptr<CameraIntrinsicsModel>& CameraIntrinsics() const { return camera_intrinsics_; }
  std::shared_ptr<CameraIntrinsicsModel>& MutableCameraIntrinsics() { return camera_intrinsics_; }

  const double* parameters() const { return camera_parameters_; }
  double* mutable_parameters() { return camera_parameters_; }
  const double* extrinsics() const { return camera_parameters_; }
  double* mutable_extrinsics() { return camera_parameters_; }
  const double* intrinsics() const { return camera_intrinsics_->parameters(); }
  double* mutable_intrinsics() { return camera_intrinsics_->mutable_parameters(); }

  enum ExternalParametersIndex { POSITION = 0, ORIENTATION = 3 };
  static const int kExtrinsicsSize = 6;

 private:
  double camera_parameters_[kExtrinsicsSize];
  std::shared_ptr<CameraIntrinsicsModel> camera_intrinsics_;
  int image_size_[2];
};
}  // namespace theia
#endif
