// C entry point in front of the REFERENCE's own DBoW2 map classes -- TEST INFRASTRUCTURE, build container only.
//
// oracle/_ref/libdbow2_fold.so = this file + /root/reference/Thirdparty/DBoW2/DBoW2/BowVector.cpp + FeatureVector.cpp, the two
// reference sources compiled UNMODIFIED where they lie (oracle/Makefile, target ref), with the reference's own BowVector.h /
// FeatureVector.h on the include path and no stand-in header of any kind.  The rest of DBoW2 (TemplatedVocabulary.h, FORB,
// ScoringObject.cpp) includes OpenCV and is unbuildable here, so the ~20 lines of TemplatedVocabulary::transform(features, v, fv,
// levelsup) that DRIVE the maps (TemplatedVocabulary.h:1161-1212) are restated below on a given (word, weight, node) stream;
// every map operation -- addWeight, addIfNotExist, normalize, addFeature, the iteration order -- is the reference's compiled code.
// Same signature as oracle/orb_oracle.c::orc_bow_fold.  weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; norm: 0 none, 1 L1, 2 L2.
#include <cstdint>

#include "BowVector.h"
#include "FeatureVector.h"

extern "C" int ref_bow_fold(const uint32_t* word, const double* weight, const uint32_t* node, int n, int weighting, int norm,
                            uint32_t* bow_words, double* bow_values, uint32_t* fv_node_ids, int32_t* fv_start, uint32_t* fv_items,
                            int* n_fv_nodes)
{
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    const bool must = norm != 0;   // m_scoring_object->mustNormalize(norm), TemplatedVocabulary.h:1155-1156
    if (weighting == 0 || weighting == 1) {   // TF_IDF || TF, :1160-1184
        for (int i = 0; i < n; i++) {
            if (weight[i] > 0) {   // not stopped
                v.addWeight(word[i], weight[i]);
                fv.addFeature(node[i], (unsigned int)i);
            }
        }
        if (!v.empty() && !must) {
            const double nd = v.size();
            for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
        }
    } else {   // IDF || BINARY, :1186-1203
        for (int i = 0; i < n; i++) {
            if (weight[i] > 0) {
                v.addIfNotExist(word[i], weight[i]);
                fv.addFeature(node[i], (unsigned int)i);
            }
        }
    }
    if (must) v.normalize(norm == 1 ? DBoW2::L1 : DBoW2::L2);
    int nw = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++nw) {
        bow_words[nw] = it->first;
        bow_values[nw] = it->second;
    }
    int ns = 0, pos = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++ns) {
        fv_node_ids[ns] = it->first;
        fv_start[ns] = pos;
        for (size_t k = 0; k < it->second.size(); k++) fv_items[pos++] = it->second[k];
    }
    fv_start[ns] = pos;
    *n_fv_nodes = ns;
    return nw;
}

extern "C" const char* ref_bow_fold_sources(void)
{
    return "Thirdparty/DBoW2/DBoW2/BowVector.cpp, Thirdparty/DBoW2/DBoW2/FeatureVector.cpp (unmodified, compiled in place)";
}
