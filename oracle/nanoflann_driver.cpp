// nanoflann_driver.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin extern "C" driver around the reference's OWN vendored nanoflann
// (/root/reference/include/nanoflann.hpp + KDTreeVectorOfVectorsAdaptor.h),
// compiled in place with -I/root/reference/include; no reference source is
// copied.  It calls the tree exactly as descriptor.h does:
//   construction  D.h:1699  (dim = R, vector<vector<float>>, max leaf 10)
//   query         D.h:1710-1716 (KNNResultSet<float>(k), SearchParams(10))
// Output goes to oracle/_ref/ (git-ignored).  Used to produce/verify the
// golden kNN lists in tests/golden/.
#include <cstddef>
#include <memory>
#include <vector>

#include "KDTreeVectorOfVectorsAdaptor.h"

using KeyMat = std::vector<std::vector<float>>;
using InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat, float>;

extern "C" int ref_nanoflann_knn(const float *keys, int N, int R, const float *query, int k,
                                 long long *out_idx, float *out_d2)
{
    KeyMat mat(N, std::vector<float>(R));
    for (int i = 0; i < N; i++)
        for (int d = 0; d < R; d++) mat[i][d] = keys[(size_t)i * R + d];

    std::unique_ptr<InvKeyTree> tree = std::make_unique<InvKeyTree>(R, mat, 10);

    std::vector<size_t> candidate_indexes(k);
    std::vector<float> out_dists_sqr(k);
    nanoflann::KNNResultSet<float> knnsearch_result(k);
    knnsearch_result.init(&candidate_indexes[0], &out_dists_sqr[0]);
    tree->index->findNeighbors(knnsearch_result, query, nanoflann::SearchParams(10));

    int found = (int)knnsearch_result.size();
    for (int i = 0; i < k; i++) {
        out_idx[i] = i < found ? (long long)candidate_indexes[i] : -1;
        out_d2[i] = out_dists_sqr[i];
    }
    return found;
}
