// ============================================================================
// TEST INFRASTRUCTURE ONLY.  Real-code oracle for the kNN stages of
// GMM::searchCorrespondence / GMM::queryPoint (gaussian_mixture.cpp:484-576):
// this wrapper instantiates the reference's OWN vendored nanoflann
// (gmmloc/include/gmmloc/utils/nanoflann.hpp, v0x132), included from where it
// lies under /root/reference -- nothing is copied into this repository.
// The dataset adaptors below restate FLANNPoints2d / FLANNPoints3d
// (gaussian_mixture.h:15-92), which cannot be included directly because that
// header pulls in Eigen/OpenCV.  Tree parameters as in the reference:
// L2_Simple_Adaptor<double>, leaf_max_size 5, exact search (eps 0, sorted).
// Built by oracle/Makefile into oracle/_ref/libnanoflann_ref.so (git-ignored).
// ============================================================================
#include <cstdint>
#include <vector>

#include "gmmloc/utils/nanoflann.hpp"

namespace {
template <int DIM>
struct Cloud {
  const double* pts;
  size_t n;
  inline size_t kdtree_get_point_count() const { return n; }
  inline double kdtree_distance(const double* p1, const size_t idx_p2, size_t) const {
    double s = 0.0;
    for (int d = 0; d < DIM; ++d) {
      const double df = p1[d] - pts[idx_p2 * DIM + d];
      s += df * df;
    }
    return s;
  }
  inline double kdtree_get_pt(const size_t idx, const size_t dim) const { return pts[idx * DIM + dim]; }
  template <class BBOX>
  bool kdtree_get_bbox(BBOX&) const {
    return false;
  }
};

template <int DIM>
void knn(const double* pts, int n, const double* q, int nq, int k, int32_t* idx, double* dist, int32_t* cnt) {
  using Tree = nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<double, Cloud<DIM>>, Cloud<DIM>, DIM>;
  Cloud<DIM> cloud{pts, (size_t)n};
  Tree tree(DIM, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(5));
  tree.buildIndex();
  std::vector<size_t> ri(k);
  std::vector<double> rd(k);
  for (int i = 0; i < nq; ++i) {
    const size_t m = tree.knnSearch(q + (size_t)i * DIM, (size_t)k, ri.data(), rd.data());
    for (int j = 0; j < k; ++j) {
      idx[(size_t)i * k + j] = j < (int)m ? (int32_t)ri[j] : -1;
      dist[(size_t)i * k + j] = j < (int)m ? rd[j] : 0.0;
    }
    cnt[i] = (int32_t)m;
  }
}
}  // namespace

extern "C" {
void nfref_knn2d(const double* pts, int n, const double* q, int nq, int k, int32_t* idx, double* dist, int32_t* cnt) {
  knn<2>(pts, n, q, nq, k, idx, dist, cnt);
}
void nfref_knn3d(const double* pts, int n, const double* q, int nq, int k, int32_t* idx, double* dist, int32_t* cnt) {
  knn<3>(pts, n, q, nq, k, idx, dist, cnt);
}
// Persistent 3-D tree, as the reference keeps it (built once in the GMM constructor, gaussian_mixture.cpp:43-59;
// queried per point by GMM::queryPoint, :545-576): lets the CPU baseline time the queries without the build.
struct NfTree3 {
  Cloud<3> cloud;
  nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<double, Cloud<3>>, Cloud<3>, 3> tree;
  std::vector<double> copy;
  NfTree3(const double* pts, int n) : cloud{nullptr, (size_t)n}, tree(3, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(5)), copy(pts, pts + (size_t)n * 3) {
    cloud.pts = copy.data();
    tree.buildIndex();
  }
};
void* nfref_tree3d_create(const double* pts, int n) { return new NfTree3(pts, n); }
void nfref_tree3d_destroy(void* h) { delete (NfTree3*)h; }
void nfref_tree3d_knn(void* h, const double* q, int nq, int k, int32_t* idx, double* dist) {
  NfTree3* t = (NfTree3*)h;
  std::vector<size_t> ri(k);
  std::vector<double> rd(k);
  for (int i = 0; i < nq; ++i) {
    const size_t m = t->tree.knnSearch(q + (size_t)i * 3, (size_t)k, ri.data(), rd.data());
    for (int j = 0; j < k; ++j) {
      idx[(size_t)i * k + j] = j < (int)m ? (int32_t)ri[j] : -1;
      dist[(size_t)i * k + j] = j < (int)m ? rd[j] : 0.0;
    }
  }
}
}
