// bow_api.cpp -- vocabulary object + BoW transform entry points of the C ABI.
//   ORBVocabulary / LineVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>     include/ORBVocabulary.h:30-34
//   loadFromTextFile  Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425     transform  :1127-1195, :1217-1261
// The tree lives in HBM (children stored contiguously); the per-descriptor descent is the HIP kernel in bow.hip; the BowVector /
// FeatureVector assembly (two small ordered maps per image, consumed by host code) is host work here, as in the reference.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include "../../include/orbline.h"
#include "olf_internal.hpp"

using namespace olf;

struct olf_voc {
    int k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0, n_slots = 0;
    uint8_t* d_slotDesc = nullptr; int* d_childOff = nullptr; int* d_slotNode = nullptr; int* d_nodeWord = nullptr; double* d_nodeWeight = nullptr;
};

namespace olf {
int launch_bow_descend(const uint8_t* slotDesc, const int* childOff, const int* slotNode, const int* nodeWord, const double* nodeWeight,
                       const uint8_t* desc, int n, int nid_level, int* word, double* weight, int* nodeOut, hipStream_t s);
hipStream_t ctx_stream(olf_ctx* c);
int ctx_scratch(olf_ctx* c, int slot, size_t bytes, void** out);
int launch_search_by_bow_batch(const uint8_t* slotDesc, const int* childOff, const int* slotNode, const double* nodeWeight, int nid_level, int n_frames,
                               int img_stride, int cap, const olf_keypoint* d_kps, const uint8_t* d_desc, const int* d_counts, const uint8_t* d_mp_valid,
                               const uint8_t* d_mp_bad, float nnratio, int check_ori, int* d_nodes, unsigned long long* d_sorted, int* d_m, int* d_matches,
                               int* d_nmatches, hipStream_t s);
}

extern "C" {

void olf_voc_destroy(olf_voc* v)
{
    if (!v) return;
    (void)hipFree(v->d_slotDesc); (void)hipFree(v->d_childOff); (void)hipFree(v->d_slotNode); (void)hipFree(v->d_nodeWord); (void)hipFree(v->d_nodeWeight);
    delete v;
}

int olf_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc,
                   const double* weight, olf_voc** out)
{
    if (!out || !parent || !is_leaf || !desc || !weight || n_nodes < 1 || k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 ||
        weighting < 0 || weighting > 3) { set_error("olf_voc_create: bad argument"); return OLF_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("olf_voc_create: no HIP device visible (this library has no CPU path)"); return OLF_ERR_NODEVICE; }
    // node ids are creation order (TemplatedVocabulary.h:1384-1391): a parent always precedes its children
    std::vector<int> cnt(n_nodes + 1, 0);
    for (int i = 1; i < n_nodes; ++i) {
        if (parent[i] < 0 || parent[i] >= i) { set_error("olf_voc_create: parent[i] must be in [0, i)"); return OLF_ERR_INVALID; }
        ++cnt[parent[i]];
    }
    std::vector<int> off(n_nodes + 1, 0);
    for (int i = 0; i < n_nodes; ++i) off[i + 1] = off[i] + cnt[i];
    const int n_slots = off[n_nodes];
    std::vector<int> fill(off.begin(), off.end() - 1), slotNode(std::max(n_slots, 1));
    std::vector<uint8_t> slotDesc((size_t)std::max(n_slots, 1) * 32);
    std::vector<int> word(n_nodes, 0);               // Node::word_id is 0 unless the file marks the node as a word (:316, :1411-1418)
    int n_words = 0;
    for (int i = 1; i < n_nodes; ++i) {
        const int s = fill[parent[i]]++;            // children in node-id order = push_back order
        slotNode[s] = i;
        std::memcpy(&slotDesc[(size_t)s * 32], desc + (size_t)i * 32, 32);
        if (is_leaf[i]) word[i] = n_words++;
    }
    olf_voc* v = new olf_voc;
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->n_nodes = n_nodes; v->n_words = n_words; v->n_slots = n_slots;
    auto up = [&](void** d, const void* h, size_t bytes) -> bool {
        return hipMalloc(d, std::max<size_t>(bytes, 16)) == hipSuccess && (bytes == 0 || hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess);
    };
    std::vector<double> w(weight, weight + n_nodes);
    w[0] = 0;
    if (!up((void**)&v->d_slotDesc, slotDesc.data(), (size_t)n_slots * 32) || !up((void**)&v->d_childOff, off.data(), (size_t)(n_nodes + 1) * 4) ||
        !up((void**)&v->d_slotNode, slotNode.data(), (size_t)n_slots * 4) || !up((void**)&v->d_nodeWord, word.data(), (size_t)n_nodes * 4) ||
        !up((void**)&v->d_nodeWeight, w.data(), (size_t)n_nodes * 8)) {
        set_error("olf_voc_create: device allocation / upload failed");
        olf_voc_destroy(v);
        return OLF_ERR_HIP;
    }
    *out = v;
    return OLF_OK;
}

int olf_voc_load_text(const char* path, olf_voc** out)
{
    if (!path || !out) { set_error("olf_voc_load_text: null argument"); return OLF_ERR_INVALID; }
    std::ifstream f(path);
    if (!f.is_open()) { set_error(std::string("olf_voc_load_text: cannot open ") + path); return OLF_ERR_INVALID; }
    std::string line;
    std::getline(f, line);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    { std::istringstream h(line); h >> k >> L >> n1 >> n2; }
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        set_error("olf_voc_load_text: not a DBoW2 text vocabulary (header must be 'k L scoring weighting')");
        return OLF_ERR_INVALID;
    }
    std::vector<int32_t> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    // one node per line, INCLUDING the empty string the reference's while(!f.eof()) loop reads after the final newline: that
    // phantom node becomes one more (childless, weight 0) child of the root.  Its descriptor is uninitialised memory in the reference;
    // here it is all zeros (convention C.8 in DESIGN.md).
    while (!f.eof()) {
        std::getline(f, line);
        const char* p = line.c_str();
        char* q = nullptr;
        auto next_int = [&](long& v) -> bool { v = std::strtol(p, &q, 10); const bool ok = q != p; p = q; return ok; };
        long pid = 0, isleaf = 0;
        if (!next_int(pid)) pid = 0;
        if (!next_int(isleaf)) isleaf = 0;
        const size_t id = parent.size();
        if (pid < 0 || (size_t)pid >= id) { set_error("olf_voc_load_text: node refers to a parent that does not precede it"); return OLF_ERR_INVALID; }
        parent.push_back((int32_t)pid); leaf.push_back(isleaf > 0);
        desc.resize((id + 1) * 32, 0);
        for (int i = 0; i < 32; ++i) { long b = 0; if (next_int(b)) desc[id * 32 + i] = (uint8_t)b; }
        double w = std::strtod(p, &q);
        if (q == p) w = 0;
        weight.push_back(w);
    }
    return olf_voc_create(k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

int olf_voc_info(const olf_voc* v, int* k, int* L, int* scoring, int* weighting, int* n_nodes, int* n_words)
{
    if (!v) { set_error("olf_voc_info: null vocabulary"); return OLF_ERR_INVALID; }
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    return OLF_OK;
}

int olf_bow_words_dev(olf_ctx* c, const olf_voc* v, const uint8_t* d_desc, int n, int levelsup, int32_t* d_word, double* d_weight, int32_t* d_node,
                      void* stream)
{
    if (!c || !v || !d_desc || !d_word || !d_weight || !d_node || n < 0) { set_error("olf_bow_words_dev: bad argument"); return OLF_ERR_INVALID; }
    if (v->n_words == 0) { set_error("olf_bow_words_dev: empty vocabulary"); return OLF_ERR_INVALID; }
    return launch_bow_descend(v->d_slotDesc, v->d_childOff, v->d_slotNode, v->d_nodeWord, v->d_nodeWeight, d_desc, n, v->L - levelsup, d_word, d_weight,
                              d_node, stream ? (hipStream_t)stream : ctx_stream(c));
}

int olf_bow_assemble(const olf_voc* v, const int32_t* word, const double* weight, const int32_t* node, int n, int32_t* bow_ids, double* bow_vals,
                     int* n_bow, int32_t* fv_nodes, int32_t* fv_offs, int32_t* fv_idx, int* n_fv)
{
    if (!v || !n_bow || !n_fv || !fv_offs || n < 0 || (n > 0 && (!word || !weight || !node || !bow_ids || !bow_vals || !fv_nodes || !fv_idx))) {
        set_error("olf_bow_assemble: bad argument"); return OLF_ERR_INVALID;
    }
    // features whose word is stopped (weight <= 0) are dropped from both vectors (TemplatedVocabulary.h:1160-1164)
    std::vector<int> keep; keep.reserve(n);
    for (int i = 0; i < n; ++i) if (weight[i] > 0) keep.push_back(i);
    // BowVector: an ordered map word -> value.  TF_IDF / TF add the word's weight once per occurrence (repeated addition, in
    // feature order); IDF / BINARY keep the first.
    std::vector<int> byWord(keep);
    std::stable_sort(byWord.begin(), byWord.end(), [&](int a, int b) { return (uint32_t)word[a] < (uint32_t)word[b]; });
    const bool tf = v->weighting == 0 || v->weighting == 1;
    int nb = 0;
    for (size_t a = 0; a < byWord.size();) {
        size_t b = a;
        double val = weight[byWord[a]];
        for (++b; b < byWord.size() && word[byWord[b]] == word[byWord[a]]; ++b) if (tf) val += weight[byWord[b]];
        bow_ids[nb] = word[byWord[a]]; bow_vals[nb] = val; ++nb;
        a = b;
    }
    const bool must = v->scoring != 5;               // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
    if (tf && nb > 0 && !must) { const double nd = (double)nb; for (int i = 0; i < nb; ++i) bow_vals[i] /= nd; }
    if (must) {
        double norm = 0.0;
        if (v->scoring == 1) { for (int i = 0; i < nb; ++i) norm += bow_vals[i] * bow_vals[i]; norm = std::sqrt(norm); }
        else for (int i = 0; i < nb; ++i) norm += std::fabs(bow_vals[i]);
        if (norm > 0.0) for (int i = 0; i < nb; ++i) bow_vals[i] /= norm;
    }
    *n_bow = nb;
    // FeatureVector: ordered map node -> feature indices in ascending feature order
    std::vector<int> byNode(keep);
    std::stable_sort(byNode.begin(), byNode.end(), [&](int a, int b) { return (uint32_t)node[a] < (uint32_t)node[b]; });
    int nf = 0;
    for (size_t a = 0; a < byNode.size(); ++a) {
        if (a == 0 || node[byNode[a]] != node[byNode[a - 1]]) { fv_nodes[nf] = node[byNode[a]]; fv_offs[nf] = (int)a; ++nf; }
        fv_idx[a] = byNode[a];
    }
    fv_offs[nf] = (int)byNode.size();
    *n_fv = nf;
    return OLF_OK;
}

int olf_search_by_bow_batch_dev(olf_ctx* c, const olf_voc* v, int n_frames, int img_stride, const olf_keypoint* d_kps, const uint8_t* d_desc,
                                const int32_t* d_counts, const uint8_t* d_mp_valid, const uint8_t* d_mp_bad, float nnratio, int check_orientation, int levelsup,
                                int32_t* d_matches, int32_t* d_nmatches, void* stream)
{
    if (!c || !v || !d_kps || !d_desc || !d_counts || !d_matches || !d_nmatches || n_frames < 0 || img_stride < 1 || levelsup < 0) {
        set_error("olf_search_by_bow_batch_dev: bad argument"); return OLF_ERR_INVALID;
    }
    if (v->n_words == 0) { set_error("olf_search_by_bow_batch_dev: empty vocabulary"); return OLF_ERR_INVALID; }
    if (n_frames < 2) return OLF_OK;
    const int cap = olf_orb_capacity(c);
    if (cap > 4096) { set_error("olf_search_by_bow_batch_dev: more than 4096 features per frame (the per-frame node sort runs in 32 KB of LDS)"); return OLF_ERR_CAPACITY; }
    void* st = nullptr;
    const size_t bn = (((size_t)n_frames * cap * 4) + 63) & ~(size_t)63, bs = (size_t)n_frames * cap * 8;
    const int rc = ctx_scratch(c, 2, bn + bs + (size_t)n_frames * 4 + 64, &st);
    if (rc != OLF_OK) return rc;
    int* d_nodes = (int*)st;
    unsigned long long* d_sorted = (unsigned long long*)((uint8_t*)st + bn);
    int* d_m = (int*)((uint8_t*)st + bn + bs);
    return launch_search_by_bow_batch(v->d_slotDesc, v->d_childOff, v->d_slotNode, v->d_nodeWeight, v->L - levelsup, n_frames, img_stride, cap, d_kps, d_desc,
                                      d_counts, d_mp_valid, d_mp_bad, nnratio, check_orientation ? 1 : 0, d_nodes, d_sorted, d_m, d_matches, d_nmatches,
                                      stream ? (hipStream_t)stream : ctx_stream(c));
}

int olf_bow_transform(olf_ctx* c, const olf_voc* v, const uint8_t* desc, int n, int levelsup, int32_t* bow_ids, double* bow_vals, int* n_bow,
                      int32_t* fv_nodes, int32_t* fv_offs, int32_t* fv_idx, int* n_fv)
{
    if (!c || !v || !n_bow || !n_fv || !fv_offs || n < 0 || (n > 0 && !desc)) { set_error("olf_bow_transform: bad argument"); return OLF_ERR_INVALID; }
    if (v->n_words == 0 || n == 0) { *n_bow = 0; *n_fv = 0; fv_offs[0] = 0; return OLF_OK; }      // empty vocabulary -> empty vectors (:1137-1140)
    void* st = nullptr;
    const size_t bd = ((size_t)n * 32 + 15) & ~(size_t)15, bw = (size_t)n * 8;
    int rc = ctx_scratch(c, 2, bd + bw + (size_t)n * 8 + 64, &st);
    if (rc != OLF_OK) return rc;
    uint8_t* d_desc = (uint8_t*)st; double* d_weight = (double*)(d_desc + bd); int* d_word = (int*)(d_weight + n); int* d_node = d_word + n;
    hipStream_t s = ctx_stream(c);
    OLF_HIP_CHECK(hipMemcpyAsync(d_desc, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
    rc = olf_bow_words_dev(c, v, d_desc, n, levelsup, d_word, d_weight, d_node, s);
    if (rc != OLF_OK) return rc;
    std::vector<int32_t> word(n), node(n);
    std::vector<double> weight(n);
    OLF_HIP_CHECK(hipMemcpyAsync(word.data(), d_word, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(node.data(), d_node, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(weight.data(), d_weight, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipStreamSynchronize(s));
    return olf_bow_assemble(v, word.data(), weight.data(), node.data(), n, bow_ids, bow_vals, n_bow, fv_nodes, fv_offs, fv_idx, n_fv);
}

}  // extern "C"
