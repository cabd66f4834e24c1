// maprecord_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
// Restates the write sequence of Map::SaveKeyFrame (src/Map.cc:283-373, built with HasLine) into a byte buffer: one f.write per field,
// native (little-endian x86-64) representation, no padding.  mnFrameId / mnId / MapPoint / MapLine ids are `long unsigned int` (8 bytes).
#include "oracle_common.hpp"

extern "C" size_t orc_keyframe_record(unsigned long mnFrameId, unsigned long mnId, double mTimeStamp, const float* t3, const float* quat4, int N,
                                      const olf_keypoint* keys, const float* uRight, const float* depth, const uint8_t* desc,
                                      const unsigned long* mp_ids, int N_l, const olf_keyline* kls, const float* disp2, const double* le3,
                                      const uint8_t* desc_l, const unsigned long* ml_ids, uint8_t* out)
{
    uint8_t* p = out;
    auto put = [&](const void* src, size_t n) { std::memcpy(p, src, n); p += n; };
    put(&mnFrameId, sizeof(mnFrameId)); put(&mnId, sizeof(mnId)); put(&mTimeStamp, sizeof(mTimeStamp));
    for (int i = 0; i < 3; ++i) put(&t3[i], sizeof(float));
    for (int i = 0; i < 4; i++) put(&quat4[i], sizeof(float));
    put(&N, sizeof(N));
    for (int i = 0; i < N; i++) {
        const olf_keypoint& kp = keys[i];
        put(&kp.x, 4); put(&kp.y, 4); put(&kp.size, 4); put(&kp.angle, 4); put(&kp.response, 4); put(&kp.octave, 4);
        put(&uRight[i], 4); put(&depth[i], 4);
        for (int j = 0; j < 32; ++j) put(&desc[32 * (size_t)i + j], 1);
        put(&mp_ids[i], sizeof(unsigned long));
    }
    put(&N_l, sizeof(N_l));
    for (int i = 0; i < N_l; i++) {
        const olf_keyline& kl = kls[i];
        put(&kl.angle, 4); put(&kl.class_id, 4); put(&kl.octave, 4); put(&kl.pt_x, 4); put(&kl.pt_y, 4); put(&kl.response, 4); put(&kl.size, 4);
        put(&kl.startPointX, 4); put(&kl.startPointY, 4); put(&kl.endPointX, 4); put(&kl.endPointY, 4);
        put(&kl.sPointInOctaveX, 4); put(&kl.sPointInOctaveY, 4); put(&kl.ePointInOctaveX, 4); put(&kl.ePointInOctaveY, 4);
        put(&kl.lineLength, 4); put(&kl.numOfPixels, 4);
        put(&disp2[2 * i], 4); put(&disp2[2 * i + 1], 4);
        put(&le3[3 * i], 8); put(&le3[3 * i + 1], 8); put(&le3[3 * i + 2], 8);
        for (int j = 0; j < 32; ++j) put(&desc_l[32 * (size_t)i + j], 1);
        put(&ml_ids[i], sizeof(unsigned long));
    }
    return (size_t)(p - out);
}
