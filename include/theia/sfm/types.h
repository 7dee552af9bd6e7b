// Id types of the data model, as in the reference (src/theia/sfm/types.h:47-53): 32-bit
// unsigned ids, the all-ones value marks "no such id".
#ifndef THEIA_MI355_SFM_TYPES_H_
#define THEIA_MI355_SFM_TYPES_H_
#include <cstdint>
namespace theia {
using ViewId = std::uint32_t;
using TrackId = std::uint32_t;
using CameraIntrinsicsGroupId = std::uint32_t;
constexpr ViewId kInvalidViewId = UINT32_MAX;
constexpr TrackId kInvalidTrackId = UINT32_MAX;
constexpr CameraIntrinsicsGroupId kInvalidCameraIntrinsicsGroupId = UINT32_MAX;
}  // namespace theia
#endif
