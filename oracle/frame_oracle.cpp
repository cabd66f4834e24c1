// frame_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
// Restates the feature part of Frame::Frame (stereo + lines), reference src/Frame.cc:136-221:
// ExtractORB x2 (:351-357), ComputeStereoMatches (:702-876).  Line stages are added by
// line_oracle.cpp.  PARITY UNPINNED (see oracle_common.hpp).
#include "orb_oracle.hpp"

namespace orc {
void compute_stereo_matches(const std::vector<olf_keypoint>& keysL, const uint8_t* descL,
                            const std::vector<olf_keypoint>& keysR, const uint8_t* descR,
                            const std::vector<Image>& pyrL, const std::vector<Image>& pyrR,
                            const std::vector<float>& sf, const std::vector<float>& inv_sf, float mbf, float fx,
                            std::vector<float>& uRight, std::vector<float>& depth, std::vector<int>* sad_out);
}
using namespace orc;

extern "C" {

// ORB on both images + ComputeStereoMatches.  Outputs: kpsL/descL/nL, kpsR/descR/nR, uRight[cap], depth[cap]
int orc_stereo_points(const uint8_t* imgL, const uint8_t* imgR, int w, int h, const olf_params* p,
                      olf_keypoint* kpsL, uint8_t* descL, int* nL, olf_keypoint* kpsR, uint8_t* descR, int* nR, int cap,
                      float* uRight, float* depth, int* sad)
{
    Image L(w, h), R(w, h);
    std::memcpy(L.d.data(), imgL, (size_t)w * h);
    std::memcpy(R.d.data(), imgR, (size_t)w * h);
    OrbResult rl, rr;
    orb_extract(L, p->orb, rl);
    orb_extract(R, p->orb, rr);
    *nL = (int)rl.kps.size(); *nR = (int)rr.kps.size();
    if (*nL > cap || *nR > cap) return OLF_ERR_CAPACITY;
    std::memcpy(kpsL, rl.kps.data(), rl.kps.size() * sizeof(olf_keypoint));
    std::memcpy(descL, rl.desc.data(), rl.desc.size());
    std::memcpy(kpsR, rr.kps.data(), rr.kps.size() * sizeof(olf_keypoint));
    std::memcpy(descR, rr.desc.data(), rr.desc.size());
    std::vector<float> sf, inv_sf, u, d;
    std::vector<int> s;
    orb_scale_tables(p->orb, sf, inv_sf);
    if (rl.kps.empty()) return OLF_OK;   // Frame ctor returns early (src/Frame.cc:176-177)
    compute_stereo_matches(rl.kps, rl.desc.data(), rr.kps, rr.desc.data(), rl.pyramid, rr.pyramid, sf, inv_sf, p->stereo.bf, p->stereo.fx, u, d, &s);
    for (int i = 0; i < *nL; ++i) { uRight[i] = u[i]; depth[i] = d[i]; if (sad) sad[i] = s[i]; }
    return OLF_OK;
}

}  // extern "C"
