// bow_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Restates the vendored DBoW2 (paths relative to /root/reference/Thirdparty/DBoW2/DBoW2):
//   TemplatedVocabulary::loadFromTextFile                      TemplatedVocabulary.h:1338-1425  (+ FORB::fromString, FORB.cpp:120-135)
//   TemplatedVocabulary::transform(feature, id, w, nid, lup)   TemplatedVocabulary.h:1217-1261
//   TemplatedVocabulary::transform(features, v, fv, levelsup)  TemplatedVocabulary.h:1127-1195
//   BowVector::addWeight / addIfNotExist / normalize           BowVector.cpp:34-84
//   FeatureVector::addFeature                                  FeatureVector.cpp:31-45
//   FORB::distance (256-bit Hamming)                           FORB.cpp:74-96
// called by Frame::ComputeBoW (src/Frame.cc:585-597) and KeyFrame::ComputeBoW (src/KeyFrame.cc:96-112) for the ORB and the LBD
// vocabulary (both are TemplatedVocabulary<FORB::TDescriptor, FORB>, include/ORBVocabulary.h:30-34).
// The two container classes are STL-only and are also compiled from the reference checkout into oracle/_ref/libref_bow.so;
// tests/test_oracle_cpu.py checks this restatement against them.  TemplatedVocabulary.h itself includes OpenCV -> restated only.
#include "oracle_common.hpp"
#include <fstream>
#include <map>
#include <sstream>
#include <string>

namespace orc {
struct VocNode { int parent = 0; std::vector<int> children; uint8_t desc[32] = {0}; double weight = 0; int word_id = 0; bool leaf = false; };
struct Voc {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<VocNode> nodes;
    int n_words = 0;
};

static void node_transform(const Voc& V, const uint8_t* f, int levelsup, int& word_id, double& weight, int& nid)
{
    const int nid_level = V.L - levelsup;
    if (nid_level <= 0) nid = 0;
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const std::vector<int>& nodes = V.nodes[final_id].children;
        final_id = nodes[0];
        double best_d = hamming256(f, V.nodes[final_id].desc);
        for (size_t i = 1; i < nodes.size(); ++i) {
            const double d = hamming256(f, V.nodes[nodes[i]].desc);
            if (d < best_d) { best_d = d; final_id = nodes[i]; }
        }
        if (current_level == nid_level) nid = final_id;
    } while (!V.nodes[final_id].children.empty());      // Node::isLeaf() = children.empty()
    word_id = V.nodes[final_id].word_id;
    weight = V.nodes[final_id].weight;
}
}  // namespace orc

using namespace orc;
extern "C" {

void* orc_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int* parent, const uint8_t* is_leaf, const uint8_t* desc,
                     const double* weight)
{
    Voc* V = new Voc; V->k = k; V->L = L; V->scoring = scoring; V->weighting = weighting;
    V->nodes.resize(n_nodes);
    for (int i = 1; i < n_nodes; ++i) {
        VocNode& n = V->nodes[i];
        n.parent = parent[i]; std::memcpy(n.desc, desc + 32 * (size_t)i, 32); n.weight = weight[i]; n.leaf = is_leaf[i];
        V->nodes[parent[i]].children.push_back(i);
        if (is_leaf[i]) n.word_id = V->n_words++;
    }
    return V;
}

// loadFromTextFile: header "k L scoring weighting", then one node per line "parent isLeaf d0 .. d31 weight"; node ids in file order from 1.
// (The reference's while(!f.eof()) loop also turns the empty line after the last newline into one more node with parent 0 and
// uninitialised fields; it is never reached by transform() because it is appended after the root's k real children only when k+1 <= ... --
// it IS a child of the root.  We reproduce it: an all-zero descriptor, weight 0, not a leaf -> see the test for its effect.)
void* orc_voc_load_text(const char* path)
{
    std::ifstream f(path);
    if (!f.is_open()) return nullptr;
    std::string s;
    std::getline(f, s);
    std::stringstream ss; ss << s;
    Voc* V = new Voc; int n1 = -1, n2 = -1;
    ss >> V->k >> V->L >> n1 >> n2;
    if (V->k < 0 || V->k > 20 || V->L < 1 || V->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { delete V; return nullptr; }
    V->scoring = n1; V->weighting = n2;
    V->nodes.resize(1);
    while (!f.eof()) {
        std::string snode;
        std::getline(f, snode);
        std::stringstream ssnode; ssnode << snode;
        const int nid = (int)V->nodes.size();
        V->nodes.resize(nid + 1);
        int pid = 0; ssnode >> pid;
        if (ssnode.fail()) pid = 0;
        V->nodes[nid].parent = pid;
        V->nodes[pid].children.push_back(nid);
        int nIsLeaf = 0; ssnode >> nIsLeaf;
        for (int i = 0; i < 32; ++i) { int n = 0; ssnode >> n; if (!ssnode.fail()) V->nodes[nid].desc[i] = (uint8_t)n; }
        ssnode >> V->nodes[nid].weight;
        if (ssnode.fail()) V->nodes[nid].weight = 0;
        if (nIsLeaf > 0) { V->nodes[nid].leaf = true; V->nodes[nid].word_id = V->n_words++; }
    }
    return V;
}

void orc_voc_destroy(void* v) { delete (Voc*)v; }
int orc_voc_info(void* v, int* k, int* L, int* scoring, int* weighting, int* n_words)
{
    Voc* V = (Voc*)v; *k = V->k; *L = V->L; *scoring = V->scoring; *weighting = V->weighting; *n_words = V->n_words;
    return (int)V->nodes.size();
}
void orc_voc_export(void* v, int* parent, uint8_t* is_leaf, uint8_t* desc, double* weight)
{
    Voc* V = (Voc*)v;
    for (size_t i = 0; i < V->nodes.size(); ++i) {
        parent[i] = V->nodes[i].parent; is_leaf[i] = V->nodes[i].leaf; std::memcpy(desc + 32 * i, V->nodes[i].desc, 32); weight[i] = V->nodes[i].weight;
    }
}

// per-feature (word id, weight, node id at level L - levelsup)
void orc_bow_words(void* v, const uint8_t* desc, int n, int levelsup, int* word, double* weight, int* node)
{
    Voc* V = (Voc*)v;
    for (int i = 0; i < n; ++i) { node[i] = 0; node_transform(*V, desc + 32 * (size_t)i, levelsup, word[i], weight[i], node[i]); }
}

// full transform -> BowVector (ascending word id: ids, values) and FeatureVector (CSR: ascending node id, offsets, feature indices)
int orc_bow_transform(void* v, const uint8_t* desc, int n, int levelsup, int* bow_ids, double* bow_vals, int* n_bow, int* fv_nodes, int* fv_offs,
                      int* fv_idx, int* n_fv)
{
    Voc* V = (Voc*)v;
    std::map<unsigned, double> bow;
    std::map<unsigned, std::vector<unsigned>> fv;
    if (V->n_words == 0) { *n_bow = 0; *n_fv = 0; fv_offs[0] = 0; return 0; }           // empty(): m_words.empty()
    // scoring: L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA normalise (L2_NORM with L2, the others L1); DOT_PRODUCT does not
    const bool must = V->scoring != 5;
    const bool l2 = V->scoring == 1;
    const bool tf = V->weighting == 0 || V->weighting == 1;     // TF_IDF or TF
    for (int i = 0; i < n; ++i) {
        int id, nid = 0; double w;
        node_transform(*V, desc + 32 * (size_t)i, levelsup, id, w, nid);
        if (w > 0) {
            auto it = bow.lower_bound((unsigned)id);
            if (it != bow.end() && !(bow.key_comp()((unsigned)id, it->first))) { if (tf) it->second += w; }
            else bow.insert(it, std::make_pair((unsigned)id, w));
            auto ft = fv.lower_bound((unsigned)nid);
            if (ft != fv.end() && ft->first == (unsigned)nid) ft->second.push_back(i);
            else { ft = fv.insert(ft, std::make_pair((unsigned)nid, std::vector<unsigned>())); ft->second.push_back(i); }
        }
    }
    if (tf && !bow.empty() && !must) {
        const double nd = (double)bow.size();
        for (auto& p : bow) p.second /= nd;
    }
    if (must) {
        double norm = 0.0;
        if (!l2) for (auto& p : bow) norm += std::fabs(p.second);
        else { for (auto& p : bow) norm += p.second * p.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto& p : bow) p.second /= norm;
    }
    int k = 0;
    for (auto& p : bow) { bow_ids[k] = (int)p.first; bow_vals[k] = p.second; ++k; }
    *n_bow = k;
    int a = 0, o = 0;
    for (auto& p : fv) { fv_nodes[a] = (int)p.first; fv_offs[a] = o; for (unsigned i : p.second) fv_idx[o++] = (int)i; ++a; }
    fv_offs[a] = o;
    *n_fv = a;
    return k;
}

}  // extern "C"
