// reference: src/theia/sfm/types.h:47-53
#ifndef THEIA_MI355_SFM_TYPES_H_
#define THEIA_MI355_SFM_TYPES_H_
#include <cstdint>
#include <limits>
namespace theia {
typedef uint32_t ViewId;
typedef uint32_t TrackId;
typedef uint32_t CameraIntrinsicsGroupId;
static const ViewId kInvalidViewId = std::numeric_limits<ViewId>::max();
static const TrackId kInvalidTrackId = std::numeric_limits<TrackId>::max();
static const CameraIntrinsicsGroupId kInvalidCameraIntrinsicsGroupId =
    std::numeric_limits<CameraIntrinsicsGroupId>::max();
}  // namespace theia
#endif
