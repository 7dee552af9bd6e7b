// Robust loss selector (reference: src/theia/sfm/bundle_adjustment/create_loss_function.h:51-58);
// the numeric codes are tmi_ba_options::loss_function_type.  The losses themselves are evaluated on
// the device (camera_models.h loss_eval), there is no ceres::LossFunction to create.
#ifndef THEIA_MI355_CREATE_LOSS_FUNCTION_H_
#define THEIA_MI355_CREATE_LOSS_FUNCTION_H_
namespace theia {
enum class LossFunctionType : int { TRIVIAL, HUBER, SOFTLONE, CAUCHY, ARCTAN, TUKEY };
static_assert(static_cast<int>(LossFunctionType::TUKEY) == 5, "ABI codes");
}  // namespace theia
#endif
