// Storage-and-metadata mirror of the reference's camera intrinsics models:
//   camera_intrinsics_model.h (base), pinhole_camera_model.h:84-94,
//   pinhole_radial_tangential_camera_model.h:89-102, fisheye_camera_model.h:65-77,
//   fov_camera_model.h:67-75, division_undistortion_camera_model.h:74-82.
// Only what the BA boundary needs: the parameter vector (one Ceres block per
// shared model object), its size/type, and GetSubsetFromOptimizeIntrinsicsType.
// The projection math of the models lives on the device (camera_models.h).
#ifndef THEIA_MI355_CAMERA_INTRINSICS_MODEL_H_
#define THEIA_MI355_CAMERA_INTRINSICS_MODEL_H_
#include <memory>
#include <vector>
#include "theia/sfm/bundle_adjustment/bundle_adjustment.h"
#include "theia/sfm/camera/camera_intrinsics_model_type.h"
#include "theia_mi355_ba.h"

namespace theia {
class CameraIntrinsicsModel {
 public:
  explicit CameraIntrinsicsModel(CameraIntrinsicsModelType type)
      : type_(type), parameters_(tmi_ba_intrinsics_size(static_cast<int>(type)), 0.0) {
    // defaults as in the reference constructors: f = 1, aspect ratio = 1, rest 0
    parameters_[0] = 1.0;
    parameters_[1] = 1.0;
  }
  static std::shared_ptr<CameraIntrinsicsModel> Create(const CameraIntrinsicsModelType& type) {
    return std::make_shared<CameraIntrinsicsModel>(type);
  }
  CameraIntrinsicsModelType Type() const { return type_; }
  int NumParameters() const { return static_cast<int>(parameters_.size()); }
  const double* parameters() const { return parameters_.data(); }
  double* mutable_parameters() { return parameters_.data(); }
  double GetParameter(int i) const { return parameters_[i]; }
  void SetParameter(int i, double v) { parameters_[i] = v; }
  void SetFocalLength(double f) { parameters_[0] = f; }
  double FocalLength() const { return parameters_[0]; }
  void SetPrincipalPoint(double px, double py) {
    const int o = HasSkew() ? 3 : 2;
    parameters_[o] = px;
    parameters_[o + 1] = py;
  }
  // Indices of the parameters held CONSTANT (the reference's contract,
  // pinhole_camera_model.cc:132-162 and the four siblings).
  std::vector<int> GetSubsetFromOptimizeIntrinsicsType(
      const OptimizeIntrinsicsType& intrinsics_to_optimize) const {
    std::vector<uint8_t> mask(parameters_.size());
    tmi_ba_intrinsics_constant_mask(static_cast<int>(type_), static_cast<int>(intrinsics_to_optimize),
                                    mask.data());
    std::vector<int> constant;
    for (size_t i = 0; i < mask.size(); ++i)
      if (mask[i]) constant.push_back(static_cast<int>(i));
    return constant;
  }

 private:
  bool HasSkew() const {
    return type_ == CameraIntrinsicsModelType::PINHOLE ||
           type_ == CameraIntrinsicsModelType::PINHOLE_RADIAL_TANGENTIAL ||
           type_ == CameraIntrinsicsModelType::FISHEYE;
  }
  CameraIntrinsicsModelType type_;
  std::vector<double> parameters_;
};

// Parameter index enums with the reference's names.
struct PinholeCameraModel {
  static const int kIntrinsicsSize = 7;
  enum InternalParametersIndex { FOCAL_LENGTH = 0, ASPECT_RATIO = 1, SKEW = 2, PRINCIPAL_POINT_X = 3,
                                 PRINCIPAL_POINT_Y = 4, RADIAL_DISTORTION_1 = 5, RADIAL_DISTORTION_2 = 6 };
};
struct PinholeRadialTangentialCameraModel {
  static const int kIntrinsicsSize = 10;
  enum InternalParametersIndex { FOCAL_LENGTH = 0, ASPECT_RATIO = 1, SKEW = 2, PRINCIPAL_POINT_X = 3,
                                 PRINCIPAL_POINT_Y = 4, RADIAL_DISTORTION_1 = 5, RADIAL_DISTORTION_2 = 6,
                                 RADIAL_DISTORTION_3 = 7, TANGENTIAL_DISTORTION_1 = 8,
                                 TANGENTIAL_DISTORTION_2 = 9 };
};
struct FisheyeCameraModel {
  static const int kIntrinsicsSize = 9;
  enum InternalParametersIndex { FOCAL_LENGTH = 0, ASPECT_RATIO = 1, SKEW = 2, PRINCIPAL_POINT_X = 3,
                                 PRINCIPAL_POINT_Y = 4, RADIAL_DISTORTION_1 = 5, RADIAL_DISTORTION_2 = 6,
                                 RADIAL_DISTORTION_3 = 7, RADIAL_DISTORTION_4 = 8 };
};
struct FOVCameraModel {
  static const int kIntrinsicsSize = 5;
  enum InternalParametersIndex { FOCAL_LENGTH = 0, ASPECT_RATIO = 1, PRINCIPAL_POINT_X = 2,
                                 PRINCIPAL_POINT_Y = 3, RADIAL_DISTORTION_1 = 4 };
};
struct DivisionUndistortionCameraModel {
  static const int kIntrinsicsSize = 5;
  enum InternalParametersIndex { FOCAL_LENGTH = 0, ASPECT_RATIO = 1, PRINCIPAL_POINT_X = 2,
                                 PRINCIPAL_POINT_Y = 3, RADIAL_DISTORTION_1 = 4 };
};
}  // namespace theia
#endif
