// reference: src/theia/sfm/camera/camera_intrinsics_model_type.h:45-52
#ifndef THEIA_MI355_CAMERA_INTRINSICS_MODEL_TYPE_H_
#define THEIA_MI355_CAMERA_INTRINSICS_MODEL_TYPE_H_
namespace theia {
enum class CameraIntrinsicsModelType {
  INVALID = -1,
  PINHOLE = 0,
  PINHOLE_RADIAL_TANGENTIAL = 1,
  FISHEYE = 2,
  FOV = 3,
  DIVISION_UNDISTORTION = 4,
};
}
#endif
