// The five camera models and their numeric codes (reference:
// src/theia/sfm/camera/camera_intrinsics_model_type.h:45-52); the codes are the values
// tmi_ba_problem::group_model carries across the C ABI.
#ifndef THEIA_MI355_CAMERA_INTRINSICS_MODEL_TYPE_H_
#define THEIA_MI355_CAMERA_INTRINSICS_MODEL_TYPE_H_
namespace theia {
enum class CameraIntrinsicsModelType : int {
  INVALID = -1, PINHOLE, PINHOLE_RADIAL_TANGENTIAL, FISHEYE, FOV, DIVISION_UNDISTORTION
};
static_assert(static_cast<int>(CameraIntrinsicsModelType::DIVISION_UNDISTORTION) == 4, "ABI codes");
}  // namespace theia
#endif
