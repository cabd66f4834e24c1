// ref_bow_shim.cpp -- C entry points over the REAL DBoW2 container classes BowVector / FeatureVector
// (Thirdparty/DBoW2/DBoW2/BowVector.cpp, FeatureVector.cpp: STL-only, compiled from /root/reference where they lie).
// Test infrastructure: validates oracle/bow_oracle.cpp's restatement of addWeight / addIfNotExist / normalize / addFeature.
#include "BowVector.h"
#include "FeatureVector.h"
extern "C" {
// feed n (word id, weight) pairs in order; mode 0 = addWeight, 1 = addIfNotExist; divide_by_size as in transform()'s "!must" branch;
// norm: -1 none, 0 L1, 1 L2.  Returns the map in iteration order.
int ref_bow_build(const int* ids, const double* w, int n, int mode, int divide_by_size, int norm, int* out_ids, double* out_vals)
{
    DBoW2::BowVector v;
    for (int i = 0; i < n; ++i) {
        if (!(w[i] > 0)) continue;
        if (mode == 0) v.addWeight((DBoW2::WordId)ids[i], w[i]); else v.addIfNotExist((DBoW2::WordId)ids[i], w[i]);
    }
    if (divide_by_size && !v.empty()) { const double nd = v.size(); for (DBoW2::BowVector::iterator it = v.begin(); it != v.end(); it++) it->second /= nd; }
    if (norm >= 0) v.normalize(norm == 0 ? DBoW2::L1 : DBoW2::L2);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { out_ids[k] = (int)it->first; out_vals[k] = it->second; }
    return k;
}
int ref_fv_build(const int* nodes, const double* w, int n, int* out_nodes, int* out_offs, int* out_idx)
{
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; ++i) if (w[i] > 0) fv.addFeature((DBoW2::NodeId)nodes[i], (unsigned)i);
    int a = 0, o = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++a) {
        out_nodes[a] = (int)it->first; out_offs[a] = o;
        for (size_t j = 0; j < it->second.size(); ++j) out_idx[o++] = (int)it->second[j];
    }
    out_offs[a] = o;
    return a;
}
}
