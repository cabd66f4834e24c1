// maprecord.cpp -- the per-key-frame feature record of the reference's binary map file (host code, like the reference's):
// void Map::SaveKeyFrame(ofstream &f, KeyFrame* kf), src/Map.cc:283-373, and KeyFrame* Map::LoadKeyFrame(ifstream &f, ...), :376-531,
// restricted to what the feature path produces: header (mnFrameId, mnId, mTimeStamp, translation, quaternion), N x {cv::KeyPoint's six
// stored members, mvuRight, mvDepth, the 32-byte ORB descriptor row, the MapPoint id or ULONG_MAX}, NL x {the 17 KeyLine members,
// mvDisparity_l, mvle_l, the 32-byte LBD descriptor row, the MapLine id or ULONG_MAX}.  Little-endian, no padding: the same bytes
// f.write((char*)&member, sizeof(member)) produces member by member.
#include "olf_internal.hpp"
#include <cstring>

namespace {
constexpr size_t kHeader = 8 + 8 + 8 + 3 * 4 + 4 * 4, kKp = 6 * 4 + 4 + 4 + 32 + 8, kKl = sizeof(olf_keyline) + 2 * 4 + 3 * 8 + 32 + 8;
static_assert(kHeader == 52 && kKp == 72 && sizeof(olf_keyline) == 68 && kKl == 140, "record layout");
template <class T> inline void put(uint8_t*& p, const T& v) { std::memcpy(p, &v, sizeof(T)); p += sizeof(T); }
template <class T> inline void get(const uint8_t*& p, T& v) { std::memcpy(&v, p, sizeof(T)); p += sizeof(T); }
}  // namespace

extern "C" {

size_t olf_kf_record_bytes(int n_keys, int n_lines)
{
    if (n_keys < 0 || n_lines < 0) return 0;
    return kHeader + 4 + (size_t)n_keys * kKp + 4 + (size_t)n_lines * kKl;
}

int olf_kf_record_pack(uint64_t frame_id, uint64_t kf_id, double timestamp, const float* t3, const float* quat4, int n_keys,
                       const olf_keypoint* keys, const float* uright, const float* depth, const uint8_t* desc, const uint64_t* mappoint_ids,
                       int n_lines, const olf_keyline* lines, const float* disparity2, const double* le3, const uint8_t* ldesc,
                       const uint64_t* mapline_ids, uint8_t* out, size_t capacity, size_t* written)
{
    using olf::set_error;
    if (!t3 || !quat4 || !out || !written || n_keys < 0 || n_lines < 0 || (n_keys && (!keys || !uright || !depth || !desc)) ||
        (n_lines && (!lines || !disparity2 || !le3 || !ldesc))) { set_error("olf_kf_record_pack: bad argument"); return OLF_ERR_INVALID; }
    const size_t need = olf_kf_record_bytes(n_keys, n_lines);
    if (capacity < need) { set_error("olf_kf_record_pack: output buffer too small"); return OLF_ERR_CAPACITY; }
    uint8_t* p = out;
    put(p, frame_id); put(p, kf_id); put(p, timestamp);
    for (int i = 0; i < 3; ++i) put(p, t3[i]);
    for (int i = 0; i < 4; ++i) put(p, quat4[i]);
    put(p, (int32_t)n_keys);
    const uint64_t none = ~(uint64_t)0;                                   // ULONG_MAX: "no MapPoint / MapLine", src/Map.cc:318-322
    for (int i = 0; i < n_keys; ++i) {
        const olf_keypoint& k = keys[i];
        put(p, k.x); put(p, k.y); put(p, k.size); put(p, k.angle); put(p, k.response); put(p, k.octave);
        put(p, uright[i]); put(p, depth[i]);
        std::memcpy(p, desc + 32 * (size_t)i, 32); p += 32;
        put(p, mappoint_ids ? mappoint_ids[i] : none);
    }
    put(p, (int32_t)n_lines);
    for (int i = 0; i < n_lines; ++i) {
        put(p, lines[i]);                                                  // the 17 members in declaration order, no padding (68 bytes)
        put(p, disparity2[2 * i]); put(p, disparity2[2 * i + 1]);
        for (int k = 0; k < 3; ++k) put(p, le3[3 * (size_t)i + k]);
        std::memcpy(p, ldesc + 32 * (size_t)i, 32); p += 32;
        put(p, mapline_ids ? mapline_ids[i] : none);
    }
    *written = (size_t)(p - out);
    return OLF_OK;
}

int olf_kf_record_counts(const uint8_t* buf, size_t len, int32_t* n_keys, int32_t* n_lines, size_t* record_bytes)
{
    using olf::set_error;
    if (!buf || !n_keys || !n_lines || !record_bytes) { set_error("olf_kf_record_counts: bad argument"); return OLF_ERR_INVALID; }
    if (len < kHeader + 4) { set_error("olf_kf_record_counts: truncated record"); return OLF_ERR_INVALID; }
    int32_t n = 0, nl = 0;
    std::memcpy(&n, buf + kHeader, 4);
    if (n < 0 || len < kHeader + 4 + (size_t)n * kKp + 4) { set_error("olf_kf_record_counts: truncated record"); return OLF_ERR_INVALID; }
    std::memcpy(&nl, buf + kHeader + 4 + (size_t)n * kKp, 4);
    if (nl < 0 || len < olf_kf_record_bytes(n, nl)) { set_error("olf_kf_record_counts: truncated record"); return OLF_ERR_INVALID; }
    *n_keys = n; *n_lines = nl; *record_bytes = olf_kf_record_bytes(n, nl);
    return OLF_OK;
}

int olf_kf_record_unpack(const uint8_t* buf, size_t len, uint64_t* frame_id, uint64_t* kf_id, double* timestamp, float* t3, float* quat4,
                         olf_keypoint* keys, float* uright, float* depth, uint8_t* desc, uint64_t* mappoint_ids, olf_keyline* lines,
                         float* disparity2, double* le3, uint8_t* ldesc, uint64_t* mapline_ids)
{
    using olf::set_error;
    int32_t n = 0, nl = 0; size_t total = 0;
    const int rc = olf_kf_record_counts(buf, len, &n, &nl, &total);
    if (rc != OLF_OK) return rc;
    if (!frame_id || !kf_id || !timestamp || !t3 || !quat4 || (n && (!keys || !uright || !depth || !desc || !mappoint_ids)) ||
        (nl && (!lines || !disparity2 || !le3 || !ldesc || !mapline_ids))) { set_error("olf_kf_record_unpack: bad argument"); return OLF_ERR_INVALID; }
    const uint8_t* p = buf;
    get(p, *frame_id); get(p, *kf_id); get(p, *timestamp);
    for (int i = 0; i < 3; ++i) get(p, t3[i]);
    for (int i = 0; i < 4; ++i) get(p, quat4[i]);
    p += 4;
    for (int i = 0; i < n; ++i) {
        olf_keypoint& k = keys[i];
        get(p, k.x); get(p, k.y); get(p, k.size); get(p, k.angle); get(p, k.response); get(p, k.octave);
        k.class_id = -1;                                                   // not stored; cv::KeyPoint's default
        get(p, uright[i]); get(p, depth[i]);
        std::memcpy(desc + 32 * (size_t)i, p, 32); p += 32;
        get(p, mappoint_ids[i]);
    }
    p += 4;
    for (int i = 0; i < nl; ++i) {
        get(p, lines[i]);
        get(p, disparity2[2 * i]); get(p, disparity2[2 * i + 1]);
        for (int k = 0; k < 3; ++k) get(p, le3[3 * (size_t)i + k]);
        std::memcpy(ldesc + 32 * (size_t)i, p, 32); p += 32;
        get(p, mapline_ids[i]);
    }
    return OLF_OK;
}

}  // extern "C"
