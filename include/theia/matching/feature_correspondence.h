// reference: src/theia/matching/feature_correspondence.h:46-61 -- the pixel (or normalised) location
// of one feature in the two images of a view pair.
#ifndef THEIA_MI355_MATCHING_FEATURE_CORRESPONDENCE_H_
#define THEIA_MI355_MATCHING_FEATURE_CORRESPONDENCE_H_
#include "theia/sfm/feature.h"
namespace theia {
struct FeatureCorrespondence {
  Feature feature1;
  Feature feature2;
  FeatureCorrespondence() {}
  FeatureCorrespondence(const Feature& f1, const Feature& f2) : feature1(f1), feature2(f2) {}
  bool operator==(const FeatureCorrespondence& o) const {
    return feature1.x() == o.feature1.x() && feature1.y() == o.feature1.y() && feature2.x() == o.feature2.x() &&
           feature2.y() == o.feature2.y();
  }
};
}  // namespace theia
#endif
