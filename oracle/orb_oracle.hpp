// orb_oracle.hpp -- CPU ORACLE (test infrastructure): types shared by the oracle's translation units.
#pragma once
#include "oracle_common.hpp"
namespace orc {
struct KP {
    float x, y, size, angle, response;
    int octave;
};
struct OrbResult {
    std::vector<Image> pyramid;                 // mvImagePyramid (ROI part; the 19 px border is never read, App. A.1)
    std::vector<Image> blurred;                 // per-level GaussianBlur(7x7, 2) working images
    std::vector<std::vector<KP>> candidates;    // vToDistributeKeys per level (coords relative to minBorder)
    std::vector<std::vector<KP>> level_kps;     // after octree + border + orientation (level coords)
    std::vector<olf_keypoint> kps;
    std::vector<uint8_t> desc;
};

void orb_extract(const Image& img, const olf_orb_params& p, OrbResult& R);
void orb_scale_tables(const olf_orb_params& p, std::vector<float>& sf, std::vector<float>& inv_sf);
}  // namespace orc
