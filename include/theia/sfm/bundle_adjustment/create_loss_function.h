// reference: src/theia/sfm/bundle_adjustment/create_loss_function.h:51-58
#ifndef THEIA_MI355_CREATE_LOSS_FUNCTION_H_
#define THEIA_MI355_CREATE_LOSS_FUNCTION_H_
namespace theia {
enum class LossFunctionType {
  TRIVIAL = 0,
  HUBER = 1,
  SOFTLONE = 2,
  CAUCHY = 3,
  ARCTAN = 4,
  TUKEY = 5
};
}
#endif
