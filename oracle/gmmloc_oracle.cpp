// ============================================================================
// TEST INFRASTRUCTURE ONLY -- fp64 CPU restatement ("oracle") of the gmmloc
// hot path.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build / load this; the product never does.
//
// Every function cites the reference lines it follows (paths relative to
// /root/reference).  Eigen and g2o are un-vendored, unpinned third-party
// dependencies of the reference, none of whose tests pin results on this
// path (the reference has no tests at all):
//   ==> PARITY UNPINNED against the real reference binary.
// What pins this oracle instead: (1) the independent numpy/scipy restatement
// in oracle/numpy_ref.py + golden vectors under tests/golden/, (2) the real
// vendored nanoflann (oracle/_ref) for the kNN stages, (3) analytic KATs.
//
// Build: g++ -O2 -ffp-contract=off -mfma (oracle/Makefile).  Source order ==
// evaluation order; the only fused operations are the explicit fma() calls in
// chi2() which define the canonical Mahalanobis evaluation shared with the
// HIP kernels (bit-exact index parity).
// ============================================================================
#include <climits>
#include <cmath>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "og_graph.hpp"
#include "og_math.hpp"

using namespace og;

namespace {

struct Camera {  // PinholeCamera (gmmloc/src/cv/pinhole_camera.cpp) + camera::bf
  double fx, fy, cx, cy, bf;
  int width, height;
};

struct Params {  // gmmloc/include/gmmloc/config.h:31-89 (float scalars stay float)
  double neighbor_dist_thresh;
  float tri_lambda2, tri_str_thresh, ba_lambda2;
  int tri_check_str_chi2;
  int ba_first_as_prior;
  float sigma2_inv[8];
};

struct Comp {  // GaussianComponent (gmmloc/include/gmmloc/gmm/gaussian.h:14-93)
  double mean[3], cov[9], cov_inv[9];
  double det, det_sqrt;
  double scale[3];
  double axis[9];  // row-major, column c = eigenvector c (ascending eigenvalue)
  double sqrt_info[9];
  bool is_degenerated = false, is_salient = false;
  std::vector<int> nbs;
  std::vector<double> nbs_dist;
};

struct Comp2d {  // GaussianComponent2d (gaussian.h:95-162)
  int id;        // parent 3-D component index (id_ / parent_)
  double mean[2], cov[4], cov_inv[4], det, scale[2], theta, proj_d;
};

struct OGmm {
  std::vector<Comp> comps;
  std::vector<Comp2d> comps2d;  // GMM::components2d_ (last renderView)
  bool nbs_built = false;
};

// GaussianComponent ctor + decompose: gaussian.h:30-39, gaussian.cpp:36-63
void build_comp(const double* mean, const double* cov, Comp& c) {
  for (int i = 0; i < 3; ++i) c.mean[i] = mean[i];
  for (int i = 0; i < 9; ++i) c.cov[i] = cov[i];
  inv3(c.cov, c.cov_inv);
  c.det = det3(c.cov);
  c.det_sqrt = std::sqrt(c.det);
  eig_sym(c.cov, 3, c.scale, c.axis);
  c.is_degenerated = c.scale[0] < 1e-4;
  if (!chol3_lower(c.cov_inv, c.sqrt_info))
    for (int i = 0; i < 9; ++i) c.sqrt_info[i] = std::numeric_limits<double>::quiet_NaN();
  const double scale_thresh = 0.2;
  c.is_salient = (c.scale[1] > scale_thresh && c.scale[2] > scale_thresh);
}

// GaussianComponent::chi2 / MDist2: gaussian.cpp:65-70, gaussian.h:53-56
//   delta.transpose() * cov_inv_ * delta   == (delta^T A) . delta
// Canonical evaluation (shared bit-for-bit with the HIP kernel): Eigen's
// coefficient order with fused multiply-adds, as GCC emits for the reference
// under -O3 -march=native (gmmloc/CMakeLists.txt:7) on an FMA host.
inline double chi2(const double* mean, const double* A, const double* x) {
  const double d0 = x[0] - mean[0], d1 = x[1] - mean[1], d2 = x[2] - mean[2];
  const double r0 = std::fma(d2, A[6], std::fma(d1, A[3], d0 * A[0]));
  const double r1 = std::fma(d2, A[7], std::fma(d1, A[4], d0 * A[1]));
  const double r2 = std::fma(d2, A[8], std::fma(d1, A[5], d0 * A[2]));
  return std::fma(r2, d2, std::fma(r1, d1, r0 * d0));
}
// 2-D MDist2: gaussian.h:123-126
inline double mdist2_2d(const double* mean, const double* A, const double* x) {
  const double d0 = x[0] - mean[0], d1 = x[1] - mean[1];
  const double r0 = std::fma(d1, A[2], d0 * A[0]);
  const double r1 = std::fma(d1, A[3], d0 * A[1]);
  return std::fma(r1, d1, r0 * d0);
}

// GMMUtility::BHCoefficient<T>: gmm_utils.h:30-52
double bh3(const Comp& g0, const Comp& g1) {
  double cov[9], inv[9];
  for (int i = 0; i < 9; ++i) cov[i] = (g0.cov[i] + g1.cov[i]) / 2.0;
  const double d[3] = {g1.mean[0] - g0.mean[0], g1.mean[1] - g0.mean[1], g1.mean[2] - g0.mean[2]};
  inv3(cov, inv);
  double r[3];
  for (int j = 0; j < 3; ++j) r[j] = (d[0] * inv[0 * 3 + j] + d[1] * inv[1 * 3 + j]) + d[2] * inv[2 * 3 + j];
  double d0 = (r[0] * d[0] + r[1] * d[1]) + r[2] * d[2];
  d0 /= 8.0;
  const double d1 = std::log(det3(cov) / std::sqrt(g0.det * g1.det)) / 2.0;
  return d0 + d1;
}
double bh2(const Comp2d& g0, const Comp2d& g1) {
  double cov[4], inv[4];
  for (int i = 0; i < 4; ++i) cov[i] = (g0.cov[i] + g1.cov[i]) / 2.0;
  const double d[2] = {g1.mean[0] - g0.mean[0], g1.mean[1] - g0.mean[1]};
  inv2(cov, inv);
  const double r0 = d[0] * inv[0] + d[1] * inv[2];
  const double r1 = d[0] * inv[1] + d[1] * inv[3];
  double d0 = r0 * d[0] + r1 * d[1];
  d0 /= 8.0;
  const double d1 = std::log(det2(cov) / std::sqrt(g0.det * g1.det)) / 2.0;
  return d0 + d1;
}

// GMM::GMM neighbour graph: gaussian_mixture.cpp:61-78
void build_neighbours(OGmm& g, double thresh) {
  const int K = (int)g.comps.size();
  for (int i = 0; i < K; ++i) {
    g.comps[i].nbs.clear();
    g.comps[i].nbs_dist.clear();
    for (int j = 0; j < K; ++j) {
      if (i == j) continue;
      const double dist = bh3(g.comps[i], g.comps[j]);
      if (dist < thresh) {
        g.comps[i].nbs.push_back(j);
        g.comps[i].nbs_dist.push_back(dist);
      }
    }
  }
  g.nbs_built = true;
}

// GMMUtility::projectGaussian: gmm_utils.cpp:121-146 with
// PinholeCamera::project3 (+Jacobian): pinhole_camera.cpp:68-125, visibility :127-150,
// GaussianComponent2d ctor / decompose: gaussian.h:107-117, gaussian.cpp:17-32
bool project_gaussian(const Comp& g, const Camera& cam, const Quat& rot_c_w, const double* t_c_w, Comp2d& out) {
  double mc[3], r[3];
  qrot(rot_c_w, g.mean, r);
  for (int i = 0; i < 3; ++i) mc[i] = r[i] + t_c_w[i];
  const double x = mc[0], y = mc[1], z = mc[2];
  const double rz = 1.0 / z;
  double kx = x * rz, ky = y * rz;
  const double rz2 = rz * rz;
  const double J[6] = {cam.fx * rz, 0.0, -cam.fx * x * rz2, 0.0, cam.fy * rz, -cam.fy * y * rz2};
  kx = cam.fx * kx + cam.cx;
  ky = cam.fy * ky + cam.cy;
  const bool visible = kx >= 0.0 && ky >= 0.0 && kx < (double)cam.width && ky < (double)cam.height;
  if (!(visible && z > 0.0)) return false;
  double R[9], JR[6], JRS[6];
  qtoR(rot_c_w, R);
  // jacob_proj * rot * cov3d * rot.transpose() * jacob_proj.transpose(), left to right
  matmul(J, R, JR, 2, 3, 3);
  matmul(JR, g.cov, JRS, 2, 3, 3);
  double Rt[9], M[6];
  transpose(R, Rt, 3, 3);
  matmul(JRS, Rt, M, 2, 3, 3);
  double Jt[6];
  transpose(J, Jt, 2, 3);
  matmul(M, Jt, out.cov, 2, 3, 2);
  out.mean[0] = kx;
  out.mean[1] = ky;
  inv2(out.cov, out.cov_inv);
  out.det = det2(out.cov);
  double V[4];
  eig_sym(out.cov, 2, out.scale, V);
  out.theta = std::atan(V[2] / V[0]);
  return true;
}

// GMM::renderView(rot_c_w, t_c_w): gaussian_mixture.cpp:271-371
void render_view(OGmm& g, const Camera& cam, const Quat& rot_c_w, const double* t_c_w) {
  auto& out = g.comps2d;
  out.clear();
  const double view_cos_thresh = std::cos(78.0 * M_PI / 180.0);
  for (size_t idx = 0; idx != g.comps.size(); idx++) {
    const Comp& c = g.comps[idx];
    const double* mu = c.mean;
    if (c.is_degenerated) {  // STEP.0 check view cos (:283-302)
      double rt[3];
      qrot(qinverse(rot_c_w), t_c_w, rt);
      const double t_w_c[3] = {-rt[0], -rt[1], -rt[2]};
      double po[3] = {mu[0] - t_w_c[0], mu[1] - t_w_c[1], mu[2] - t_w_c[2]};
      const double n = std::sqrt(po[0] * po[0] + po[1] * po[1] + po[2] * po[2]);
      for (int i = 0; i < 3; ++i) po[i] /= n;
      const double view_cos = std::fabs(po[0] * c.axis[0] + po[1] * c.axis[3] + po[2] * c.axis[6]);
      if (view_cos < view_cos_thresh) continue;
    }
    Comp2d g2d;
    if (!project_gaussian(c, cam, rot_c_w, t_c_w, g2d)) continue;
    const double cov_2d_thresh = 4.0;  // :311-317
    if (g2d.scale[0] < cov_2d_thresh && g2d.scale[1] < cov_2d_thresh) continue;
    g2d.id = (int)idx;
    double r[3];
    qrot(rot_c_w, mu, r);
    g2d.proj_d = r[2] + t_c_w[2];
    const double depth_thresh = 0.8;  // :328-355
    if (!out.empty()) {
      size_t min_idx = 0;  // (uninitialised in the reference if every BH is NaN)
      double min_dist = std::numeric_limits<double>::max();
      for (size_t j = 0; j < out.size(); j++) {
        const double dist = bh2(out[j], g2d);
        if (dist < min_dist) {
          min_dist = dist;
          min_idx = j;
        }
      }
      if (min_dist < depth_thresh) {
        if (g2d.proj_d < out[min_idx].proj_d) out[min_idx] = g2d;
      } else {
        out.push_back(g2d);
      }
    } else {
      out.push_back(g2d);
    }
  }
  // sort by depth descending (:362-364).  std::sort is not stable; entries with
  // bit-equal depth are unordered in the reference -- stable here.
  std::stable_sort(out.begin(), out.end(), [](const Comp2d& a, const Comp2d& b) { return a.proj_d > b.proj_d; });
}

// exact k-NN in ascending squared-L2, dimension DIM, ties by lower index.
// The reference uses nanoflann (exact, eps=0, sorted ascending): the result
// set is identical except for bit-equal distances (traversal order there).
template <int DIM>
int knn_brute(const double* pts, int n, int stride, const double* q, int k, int* idx, double* dist) {
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    double d = 0.0;
    for (int a = 0; a < DIM; ++a) {
      const double df = q[a] - pts[(size_t)i * stride + a];
      d += df * df;  // d0*d0 + d1*d1 (+ d2*d2): gaussian_mixture.h:33-39,71-76
    }
    int pos = cnt;
    while (pos > 0 && dist[pos - 1] > d) --pos;  // KNNResultSet::addPoint (nanoflann.hpp:160-196)
    if (pos >= k) continue;
    const int last = cnt < k ? cnt : k - 1;
    for (int m = last; m > pos; --m) {
      dist[m] = dist[m - 1];
      idx[m] = idx[m - 1];
    }
    dist[pos] = d;
    idx[pos] = i;
    if (cnt < k) ++cnt;
  }
  return cnt;
}

// g2o::RobustKernelHuber delta as the reference sets it: float sqrt(5.991)
inline double fdelta(double chi2_thresh) { return (double)(float)std::sqrt(chi2_thresh); }

struct StrOptStat {  // gmmloc/include/gmmloc/types/map.h:30-35
  bool res = false;
  double chi2_proj = 0, chi2_str = 0;
  double pt_est[3] = {0, 0, 0};
};

// GMMLoc::optimizePoint: gmmloc_opt.cpp:260-342
StrOptStat optimize_point(const double* pt3d, const double* uvr, int octave, const SE3& Tcw, const Comp& comp,
                          double proj_z2, const Camera& cam, const Params& prm) {
  const float sigma2_inv = prm.sigma2_inv[octave];
  Optimizer opt;
  opt.algorithm = ALG_GN;
  opt.linearSolver = LS_EIGEN;
  Vertex* v = opt.addVertex(V_XYZ);
  for (int i = 0; i < 3; ++i) v->p[i] = pt3d[i];
  Edge* factor_proj = opt.addEdge(E_XYZ_STEREO, v);
  {
    Edge* e = factor_proj;
    for (int i = 0; i < 3; ++i) e->meas[i] = uvr[i];
    e->setInformationIdentity((double)sigma2_inv);
    e->rot = Tcw.r;
    for (int i = 0; i < 3; ++i) e->t[i] = Tcw.t[i];
    e->fx = cam.fx;
    e->fy = cam.fy;
    e->cx = cam.cx;
    e->cy = cam.cy;
    e->bf = cam.bf;
  }
  Edge* factor_str = opt.addEdge(E_PT2GAUSS_DEG, v);
  {
    Edge* e = factor_str;
    e->setInformationIdentity(1.0 * prm.tri_lambda2 * proj_z2);
    for (int i = 0; i < 3; ++i) {
      e->normal[i] = comp.axis[i * 3 + 0];  // axis_.col(0)
      e->mean[i] = comp.mean[i];
    }
  }
  StrOptStat r;
  opt.initializeOptimization();
  opt.optimize(5);
  r.res = true;
  r.chi2_proj = factor_proj->chi2();  // NB: error of the last computeActiveErrors()
  r.chi2_str = factor_str->chi2();
  for (int i = 0; i < 3; ++i) r.pt_est[i] = v->p[i];
  const double thresh = 7.815;
  if (r.chi2_proj > thresh) r.res = false;
  if (prm.tri_check_str_chi2 && r.chi2_str > prm.tri_str_thresh * prm.tri_lambda2) r.res = false;
  return r;
}

// GMM::queryPoint: gaussian_mixture.cpp:545-576 (returns the nearest mean,
// the Mahalanobis minimum it computes is discarded)
int query_point(const OGmm& g, const double* pt) {
  int idx[5];
  double dist[5];
  const int n = knn_brute<3>(g.comps[0].mean, (int)g.comps.size(), (int)(sizeof(Comp) / sizeof(double)), pt, 5, idx, dist);
  return n ? idx[0] : -1;
}

// GMMLoc::checkMapAssociation: gmmloc_opt.cpp:156-258.  comps = the feature's
// kf->comps_[idx] as parent 3-D indices.  Returns component index or -1;
// pt3d is updated in place exactly where the reference writes it.
int check_map_association(OGmm& g, double* pt3d, const double* uvr, int octave, const SE3& Tcw, const int* comps,
                          int ncomps, const Camera& cam, const Params& prm) {
  if (ncomps == 0) return -1;
  const double pt_init[3] = {pt3d[0], pt3d[1], pt3d[2]};
  double ptc[3];
  se3_map(Tcw, pt3d, ptc);
  double proj_z = ptc[2];
  proj_z = proj_z > 1.0 ? 1.0 : proj_z;
  const double proj_z2 = proj_z * proj_z;
  int min_idx = -1;
  double min_value = std::numeric_limits<double>::max();
  double min_res[3] = {0, 0, 0};
  for (int i = 0; i < ncomps; i++) {
    if (comps[i] < 0) continue;
    StrOptStat r = optimize_point(pt_init, uvr, octave, Tcw, g.comps[comps[i]], proj_z2, cam, prm);
    if (r.res && r.chi2_proj < min_value) {
      min_idx = i;
      min_value = r.chi2_proj;
      for (int k = 0; k < 3; ++k) min_res[k] = r.pt_est[k];
    }
  }
  if (min_idx != -1) {
    const int g3d = comps[min_idx];
    double ll = chi2(g.comps[g3d].mean, g.comps[g3d].cov_inv, min_res);
    int str = g3d;
    for (size_t n = 0; n < g.comps[g3d].nbs.size(); ++n) {
      const int np = g.comps[g3d].nbs[n];
      const double ln = chi2(g.comps[np].mean, g.comps[np].cov_inv, min_res);
      if (ln < ll) {
        ll = ln;
        str = np;
      }
    }
    if (str != g3d) {
      StrOptStat r = optimize_point(pt_init, uvr, octave, Tcw, g.comps[str], proj_z2, cam, prm);
      if (r.res) {
        for (int k = 0; k < 3; ++k) min_res[k] = r.pt_est[k];
      } else {
        str = g3d;
        ll = chi2(g.comps[g3d].mean, g.comps[g3d].cov_inv, min_res);
      }
    }
    if (ll > 9.0) return -1;
    for (int k = 0; k < 3; ++k) pt3d[k] = min_res[k];
    return str;
  } else {
    const int gi = query_point(g, pt_init);
    if (!g.comps[gi].is_degenerated) return -1;
    StrOptStat r = optimize_point(pt_init, uvr, octave, Tcw, g.comps[gi], proj_z2, cam, prm);
    if (r.res) {
      for (int k = 0; k < 3; ++k) pt3d[k] = r.pt_est[k];  // moves the point, still returns nullptr
    } else {
      return -1;
    }
  }
  return -1;
}

// Localization::optimizeTriangulationVec: localization_opt.cpp:27-204.
// uvr*[2] < 0 <=> mono feature (kp.depth <= 0).  Candidate order: comps1 then
// comps2, de-duplicated (the reference iterates an unordered_set of pointers).
int optimize_triangulation(OGmm& g, double* x3d, const SE3& T1, const double* uvr1, int oct1, bool stereo1,
                           const SE3& T2, const double* uvr2, int oct2, bool stereo2, const int* comps1, int n1,
                           const int* comps2, int n2, const Camera& cam, const Params& prm) {
  (void)oct2;
  Optimizer opt;
  opt.algorithm = ALG_GN;
  opt.linearSolver = LS_EIGEN;
  Vertex* vp = opt.addVertex(V_XYZ);
  const double pt_init[3] = {x3d[0], x3d[1], x3d[2]};
  for (int i = 0; i < 3; ++i) vp->p[i] = pt_init[i];
  auto addEdge = [&](const SE3& Tcw, const double* uvr, bool stereo, double s2i) {
    Edge* e = opt.addEdge(stereo ? E_XYZ_STEREO : E_XYZ_MONO, vp);
    for (int i = 0; i < 3; ++i) e->meas[i] = uvr[i];
    e->setInformationIdentity(s2i);
    e->rot = Tcw.r;
    for (int i = 0; i < 3; ++i) e->t[i] = Tcw.t[i];
    e->fx = cam.fx;
    e->fy = cam.fy;
    e->cx = cam.cx;
    e->cy = cam.cy;
    e->bf = cam.bf;
    return e;
  };
  double th_kf1 = 5.991, th_kf2 = 5.991;
  const float sigma2_inv1 = prm.sigma2_inv[oct1];
  Edge* edge_kf1 = addEdge(T1, uvr1, stereo1, sigma2_inv1);
  if (stereo1) th_kf1 = 7.8;
  Edge* edge_kf2 = addEdge(T2, uvr2, stereo2, sigma2_inv1);  // kp2 weighted with kp1's sigma (:132,135)
  if (stereo2) th_kf2 = 7.8;

  std::vector<int> cand;
  auto push = [&](int c) {
    for (int x : cand)
      if (x == c) return;
    cand.push_back(c);
  };
  for (int i = 0; i < n1; ++i)
    if (comps1[i] >= 0) push(comps1[i]);
  for (int i = 0; i < n2; ++i)
    if (comps2[i] >= 0) push(comps2[i]);
  switch (og::switches().tri_order) {  // the reference walks an unordered_set<GaussianComponent*>: any order is "the" order
    case 1: std::reverse(cand.begin(), cand.end()); break;
    case 2: std::sort(cand.begin(), cand.end()); break;
    case 3: std::sort(cand.begin(), cand.end(), std::greater<int>()); break;
    default: break;
  }

  Edge* edge_str = nullptr;
  int min_comp = -1;
  double min_value = std::numeric_limits<double>::max();
  double min_res[3] = {0, 0, 0};
  for (int ci : cand) {
    const Comp& c = g.comps[ci];
    if (!c.is_degenerated) continue;
    for (int i = 0; i < 3; ++i) vp->p[i] = pt_init[i];
    if (edge_str) opt.removeEdge(edge_str);
    edge_str = opt.addEdge(E_PT2GAUSS_DEG, vp);
    edge_str->setInformationIdentity(1.0 * prm.tri_lambda2);
    for (int i = 0; i < 3; ++i) {
      edge_str->normal[i] = c.axis[i * 3 + 0];
      edge_str->mean[i] = c.mean[i];
    }
    opt.initializeOptimization(0);
    opt.optimize(20);
    bool ok = true;
    if (prm.tri_check_str_chi2 && edge_str->chi2() > prm.tri_str_thresh * prm.tri_lambda2) ok = false;
    const double err1 = edge_kf1->chi2(), err2 = edge_kf2->chi2();
    const double err_sum = err1 + err2;
    if (err1 > th_kf1 || err2 > th_kf2) ok = false;
    if (ok && err_sum < min_value) {
      for (int i = 0; i < 3; ++i) min_res[i] = vp->p[i];
      min_comp = ci;
      min_value = err_sum;
    }
  }
  if (min_comp >= 0)
    for (int i = 0; i < 3; ++i) x3d[i] = min_res[i];
  return min_comp;
}

// Tracking::optimizeCurrentPose: tracking_opt.cpp:21-217
// obs[i] = (u, v, u_right); u_right < 0 => monocular edge; has_pt[i] == 0 => no map point.
int optimize_current_pose(SE3& Tcw, int N, const double* Xw, const double* obs, const int* octave,
                          const uint8_t* has_pt, uint8_t* is_outlier, const Camera& cam, const Params& prm) {
  Optimizer opt;
  opt.algorithm = ALG_LM;
  opt.linearSolver = LS_DENSE;
  int num_init = 0;
  Vertex* vse3 = opt.addVertex(V_SE3);
  vse3->T = Tcw;
  std::vector<Edge*> edges_mono, edges_stereo;
  std::vector<int> idx_mono, idx_stereo;
  const double delta_mono = fdelta(5.991), delta_stereo = fdelta(7.815);
  for (int i = 0; i < N; i++) {
    if (!has_pt[i]) continue;
    const bool mono = obs[i * 3 + 2] < 0;
    num_init++;
    is_outlier[i] = 0;
    Edge* e = opt.addEdge(mono ? E_POSE_MONO : E_POSE_STEREO, vse3);
    for (int k = 0; k < 3; ++k) e->meas[k] = obs[i * 3 + k];
    const float inv_sigma2 = prm.sigma2_inv[octave[i]];
    e->setInformationIdentity((double)inv_sigma2);
    e->robust = true;
    e->delta = mono ? delta_mono : delta_stereo;
    e->fx = cam.fx;
    e->fy = cam.fy;
    e->cx = cam.cx;
    e->cy = cam.cy;
    e->bf = cam.bf;
    for (int k = 0; k < 3; ++k) e->Xw[k] = Xw[i * 3 + k];
    if (mono) {
      edges_mono.push_back(e);
      idx_mono.push_back(i);
    } else {
      edges_stereo.push_back(e);
      idx_stereo.push_back(i);
    }
  }
  if (num_init < 3) return 0;
  const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
  const float chi2Stereo[4] = {7.815, 7.815, 7.815, 7.815};
  const int its[4] = {10, 10, 10, 10};
  const SE3 T0 = Tcw;
  int num_bad = 0;
  for (size_t it = 0; it < 4; it++) {
    vse3->T = T0;
    opt.initializeOptimization(0);
    opt.optimize(its[it]);
    num_bad = 0;
    for (size_t i = 0; i < edges_mono.size(); i++) {
      Edge* e = edges_mono[i];
      const int idx = idx_mono[i];
      if (is_outlier[idx]) e->computeError();
      const float chi2 = (float)e->chi2();
      if (chi2 > chi2Mono[it]) {
        is_outlier[idx] = 1;
        e->level = 1;
        num_bad++;
      } else {
        is_outlier[idx] = 0;
        e->level = 0;
      }
      if (it == 2) e->robust = false;
    }
    for (size_t i = 0; i < edges_stereo.size(); i++) {
      Edge* e = edges_stereo[i];
      const int idx = idx_stereo[i];
      if (is_outlier[idx]) e->computeError();
      const float chi2 = (float)e->chi2();
      if (chi2 > chi2Stereo[it]) {
        is_outlier[idx] = 1;
        e->level = 1;
        num_bad++;
      } else {
        e->level = 0;
        is_outlier[idx] = 0;
      }
      if (it == 2) e->robust = false;
    }
    if (opt.edges.size() < 10) break;
  }
  Tcw = vse3->T;
  return num_init - num_bad;
}

// Localization::jointOptimization: localization_opt.cpp:456-925, on a flat
// problem: poses [0,P) free (local key-frames), [P,P+F) fixed; L points, each
// with at most one GMM association; observations (point, pose, uvr, octave)
// grouped by point in the order the reference would visit them.
struct BAProblem {
  int P, F, L, nobs;
  double* poses;             // (P+F) x 7  (qx qy qz qw tx ty tz), in/out for [0,P)
  const uint8_t* has_prior;  // P
  double* points;            // L x 3 in/out
  const int32_t* assoc;      // L: component index or -1
  const int32_t* obs_ptr;    // L+1 CSR
  const int32_t* obs_pose;   // nobs
  const double* obs_uvr;     // nobs x 3 (u_right < 0 => mono)
  const int32_t* obs_octave; // nobs
  uint8_t* assoc_dropped;    // L out: association cleared (:837-853)
  uint8_t* obs_erase;        // nobs out: observation to erase (:855-879)
};

int joint_optimization(OGmm& g, BAProblem& pb, const Camera& cam, const Params& prm, const bool* stop_flag, int stop_budget = -1) {
  Optimizer opt;
  opt.algorithm = ALG_LM;
  opt.linearSolver = LS_EIGEN;
  opt.forceStop = stop_flag;
  opt.stopBudget = stop_budget;
  auto poseOf = [&](int i) {
    const double* p = pb.poses + (size_t)i * 7;
    return se3_make(Quat{p[0], p[1], p[2], p[3]}, p + 4);
  };
  std::vector<Vertex*> vpose(pb.P + pb.F);
  for (int i = 0; i < pb.P; ++i) {
    Vertex* v = opt.addVertex(V_SE3);
    v->T = poseOf(i);
    vpose[i] = v;
    if (pb.has_prior[i]) {  // :560-577
      if (prm.ba_first_as_prior) {
        Edge* e = opt.addEdge(E_SE3_PRIOR, v);
        e->inv_meas = se3_inverse(v->T);
        const double sigma_rot = 2.0 * M_PI / 180.0;
        const double sigma_rot2_inv = 1.0 / (sigma_rot * sigma_rot);
        const double sigma_trans2_inv = 1.0 / (0.01 * 0.01);
        for (int k = 0; k < 36; ++k) e->info[k] = 0.0;
        for (int k = 0; k < 3; ++k) {
          e->info[k * 6 + k] = sigma_rot2_inv;
          e->info[(k + 3) * 6 + (k + 3)] = sigma_trans2_inv;
        }
      } else {
        v->fixed = true;
      }
    }
  }
  for (int i = pb.P; i < pb.P + pb.F; ++i) {
    Vertex* v = opt.addVertex(V_SE3);
    v->T = poseOf(i);
    v->fixed = true;
    vpose[i] = v;
  }
  std::vector<Edge*> edges_gmm_deg(pb.L, nullptr);
  std::vector<Edge*> eobs(pb.nobs, nullptr);
  std::vector<Vertex*> vpt(pb.L);
  const double thHuberMono = fdelta(5.991), thHuberStereo = fdelta(7.815);
  for (int l = 0; l < pb.L; ++l) {
    Vertex* vP = opt.addVertex(V_XYZ);
    for (int k = 0; k < 3; ++k) vP->p[k] = pb.points[l * 3 + k];
    vP->marginalized = true;
    vpt[l] = vP;
    if (pb.assoc[l] >= 0) {
      const Comp& c = g.comps[pb.assoc[l]];
      if (c.is_degenerated) {  // :657-669
        Edge* e = opt.addEdge(E_PT2GAUSS_DEG, vP);
        e->setInformationIdentity(1.0 * prm.ba_lambda2);
        for (int k = 0; k < 3; ++k) {
          e->normal[k] = c.axis[k * 3 + 0];
          e->mean[k] = c.mean[k];
        }
        edges_gmm_deg[l] = e;
      } else {  // :670-681
        Edge* e = opt.addEdge(E_PT2GAUSS, vP);
        e->setInformationIdentity(1.0);
        for (int k = 0; k < 9; ++k) e->sqrt_info[k] = c.sqrt_info[k];
        for (int k = 0; k < 3; ++k) e->mean[k] = c.mean[k];
      }
    }
    for (int o = pb.obs_ptr[l]; o < pb.obs_ptr[l + 1]; ++o) {
      const bool mono = pb.obs_uvr[o * 3 + 2] < 0;
      Edge* e = opt.addEdge(mono ? E_BA_MONO : E_BA_STEREO, vP, vpose[pb.obs_pose[o]]);
      for (int k = 0; k < 3; ++k) e->meas[k] = pb.obs_uvr[o * 3 + k];
      const float invSigma2 = prm.sigma2_inv[pb.obs_octave[o]];
      e->setInformationIdentity((double)invSigma2);
      e->robust = true;
      e->delta = mono ? thHuberMono : thHuberStereo;
      e->fx = cam.fx;
      e->fy = cam.fy;
      e->cx = cam.cx;
      e->cy = cam.cy;
      e->bf = cam.bf;
      eobs[o] = e;
    }
  }
  if (stop_flag && *stop_flag) return 0;  // :765-767 (the pointer is tested there, not the optimizer's terminate())
  opt.initializeOptimization();
  opt.optimize(5);  // :770-771
  const double str_thresh = prm.tri_str_thresh * prm.ba_lambda2;
  for (int l = 0; l < pb.L; ++l) {  // :773-786
    Edge* e = edges_gmm_deg[l];
    if (!e) continue;
    e->computeError();
    if (e->chi2() > str_thresh) e->level = 1;
    e->robust = false;
  }
  opt.initializeOptimization(0);
  opt.optimize(5);  // :788-789
  bool doMore = true;
  if (opt.terminate()) doMore = false;  // :791-796 (*pbStopFlag; + the test hook's iteration budget)
  int actual_iter = 0;
  if (doMore) {  // :799-828
    for (int o = 0; o < pb.nobs; ++o) {
      Edge* e = eobs[o];
      const double th = (e->kind == E_BA_MONO) ? 5.991 : 7.815;
      if (e->chi2() > th || !e->isDepthPositive()) e->level = 1;
      e->robust = false;
    }
    opt.initializeOptimization(0);
    actual_iter = opt.optimize(40);
  }
  for (int l = 0; l < pb.L; ++l) {  // :837-853
    pb.assoc_dropped[l] = 0;
    Edge* e = edges_gmm_deg[l];
    if (!e) continue;
    e->computeError();
    if (e->chi2() > str_thresh) pb.assoc_dropped[l] = 1;
  }
  for (int o = 0; o < pb.nobs; ++o) {  // :855-879
    Edge* e = eobs[o];
    const double th = (e->kind == E_BA_MONO) ? 5.991 : 7.815;
    pb.obs_erase[o] = (e->chi2() > th || !e->isDepthPositive()) ? 1 : 0;
  }
  for (int i = 0; i < pb.P; ++i) {  // :898-911
    const SE3& T = vpose[i]->T;
    double* p = pb.poses + (size_t)i * 7;
    p[0] = T.r.x;
    p[1] = T.r.y;
    p[2] = T.r.z;
    p[3] = T.r.w;
    p[4] = T.t[0];
    p[5] = T.t[1];
    p[6] = T.t[2];
  }
  for (int l = 0; l < pb.L; ++l)  // :914-922
    for (int k = 0; k < 3; ++k) pb.points[l * 3 + k] = vpt[l]->p[k];
  return actual_iter;
}

}  // namespace

// ============================================================================
// C API for ctypes (tests / bench cpu_baseline)
// ============================================================================
extern "C" {

struct orc_camera {
  double fx, fy, cx, cy, bf;
  int32_t width, height;
};
struct orc_params {
  double neighbor_dist_thresh;
  float tri_lambda2, tri_str_thresh, ba_lambda2;
  int32_t tri_check_str_chi2, ba_first_as_prior;
  float sigma2_inv[8];
};

static Camera to_cam(const orc_camera* c) { return Camera{c->fx, c->fy, c->cx, c->cy, c->bf, c->width, c->height}; }
static Params to_prm(const orc_params* p) {
  Params q;
  q.neighbor_dist_thresh = p->neighbor_dist_thresh;
  q.tri_lambda2 = p->tri_lambda2;
  q.tri_str_thresh = p->tri_str_thresh;
  q.ba_lambda2 = p->ba_lambda2;
  q.tri_check_str_chi2 = p->tri_check_str_chi2;
  q.ba_first_as_prior = p->ba_first_as_prior;
  for (int i = 0; i < 8; ++i) q.sigma2_inv[i] = p->sigma2_inv[i];
  return q;
}
static SE3 to_se3(const double* p) { return se3_make(Quat{p[0], p[1], p[2], p[3]}, p + 4); }
static void from_se3(const SE3& T, double* p) {
  p[0] = T.r.x;
  p[1] = T.r.y;
  p[2] = T.r.z;
  p[3] = T.r.w;
  p[4] = T.t[0];
  p[5] = T.t[1];
  p[6] = T.t[2];
}

// frame::sigma2_inv table: init_config.hpp:60-79 (all float arithmetic)
void orc_default_params(orc_params* p) {
  p->neighbor_dist_thresh = 2.5;  // cfg/v1.yaml:27
  p->tri_lambda2 = 400.0f;        // cfg/v1.yaml:32
  p->tri_str_thresh = 0.0064f;    // cfg/v1.yaml:35
  p->ba_lambda2 = 400.0f;         // cfg/v1.yaml:37
  p->tri_check_str_chi2 = 1;
  p->ba_first_as_prior = 1;
  float sf[8], s2[8];
  sf[0] = 1.0f;
  s2[0] = 1.0f;
  p->sigma2_inv[0] = 1.0f;
  const float scale_factor = 1.2;
  for (int i = 1; i < 8; i++) {
    sf[i] = sf[i - 1] * scale_factor;
    s2[i] = sf[i] * sf[i];
    p->sigma2_inv[i] = 1.0f / s2[i];
  }
}

void* orc_gmm_create(const double* mean, const double* cov, int K) {
  OGmm* g = new OGmm();
  g->comps.resize(K);
  for (int k = 0; k < K; ++k) build_comp(mean + 3 * k, cov + 9 * k, g->comps[k]);
  return g;
}
void orc_gmm_destroy(void* h) { delete (OGmm*)h; }
int orc_gmm_count(void* h) { return (int)((OGmm*)h)->comps.size(); }

// field dump: cov_inv[K*9] det[K] scale[K*3] axis[K*9] sqrt_info[K*9] flags[K] (bit0 deg, bit1 salient)
void orc_gmm_get(void* h, double* cov_inv, double* det, double* scale, double* axis, double* sqrt_info,
                 uint8_t* flags) {
  OGmm* g = (OGmm*)h;
  for (size_t k = 0; k < g->comps.size(); ++k) {
    const Comp& c = g->comps[k];
    if (cov_inv) memcpy(cov_inv + 9 * k, c.cov_inv, 72);
    if (det) det[k] = c.det;
    if (scale) memcpy(scale + 3 * k, c.scale, 24);
    if (axis) memcpy(axis + 9 * k, c.axis, 72);
    if (sqrt_info) memcpy(sqrt_info + 9 * k, c.sqrt_info, 72);
    if (flags) flags[k] = (uint8_t)((c.is_degenerated ? 1 : 0) | (c.is_salient ? 2 : 0));
  }
}

// neighbour graph as CSR; returns nnz (call with col == NULL to size)
int orc_gmm_neighbours(void* h, double thresh, int32_t* row_ptr, int32_t* col, double* dist) {
  OGmm* g = (OGmm*)h;
  if (!g->nbs_built) build_neighbours(*g, thresh);
  int nnz = 0;
  for (size_t k = 0; k < g->comps.size(); ++k) {
    if (row_ptr) row_ptr[k] = nnz;
    for (size_t j = 0; j < g->comps[k].nbs.size(); ++j) {
      if (col) col[nnz] = g->comps[k].nbs[j];
      if (dist) dist[nnz] = g->comps[k].nbs_dist[j];
      ++nnz;
    }
  }
  if (row_ptr) row_ptr[g->comps.size()] = nnz;
  return nnz;
}
// rows [r0, r1) only (cheap partial check on big maps); returns nnz
int orc_gmm_neighbour_rows(void* h, double thresh, int r0, int r1, int32_t* row_ptr, int32_t* col, double* dist,
                           int cap) {
  OGmm* g = (OGmm*)h;
  const int K = (int)g->comps.size();
  int nnz = 0;
  for (int i = r0; i < r1; ++i) {
    row_ptr[i - r0] = nnz;
    for (int j = 0; j < K; ++j) {
      if (i == j) continue;
      const double d = bh3(g->comps[i], g->comps[j]);
      if (d < thresh) {
        if (nnz < cap) {
          col[nnz] = j;
          dist[nnz] = d;
        }
        ++nnz;
      }
    }
  }
  row_ptr[r1 - r0] = nnz;
  return nnz;
}

// exhaustive Mahalanobis argmin (north-star `associate`; A1 over all K), first index wins ties
void orc_associate3d(void* h, const double* pts, int N, int32_t* idx, double* d2) {
  OGmm* g = (OGmm*)h;
  const int K = (int)g->comps.size();
  for (int n = 0; n < N; ++n) {
    double best = std::numeric_limits<double>::infinity();
    int bi = -1;
    for (int k = 0; k < K; ++k) {
      const double d = chi2(g->comps[k].mean, g->comps[k].cov_inv, pts + 3 * n);
      if (d < best) {
        best = d;
        bi = k;
      }
    }
    idx[n] = bi;
    d2[n] = best;
  }
}
void orc_chi2(void* h, const int32_t* comp, const double* pts, int N, double* out) {
  OGmm* g = (OGmm*)h;
  for (int n = 0; n < N; ++n) out[n] = chi2(g->comps[comp[n]].mean, g->comps[comp[n]].cov_inv, pts + 3 * n);
}

// 5-NN on the 3-D means (GMM::queryPoint's kNN); idx5/dist5 are N x k
void orc_knn3d(void* h, const double* pts, int N, int k, int32_t* idx, double* dist, int32_t* cnt) {
  OGmm* g = (OGmm*)h;
  std::vector<int> ti(k);
  for (int n = 0; n < N; ++n) {
    const int c = knn_brute<3>(g->comps[0].mean, (int)g->comps.size(), (int)(sizeof(Comp) / sizeof(double)),
                               pts + 3 * n, k, ti.data(), dist + (size_t)n * k);
    for (int j = 0; j < k; ++j) idx[(size_t)n * k + j] = j < c ? ti[j] : -1;
    if (cnt) cnt[n] = c;
  }
}

// renderView; returns V and fills up to cap entries (sorted by depth desc)
int orc_render_view(void* h, const orc_camera* cam, const double* pose /*q xyzw, t*/, int cap, int32_t* id,
                    double* mean2d, double* cov2d, double* depth) {
  OGmm* g = (OGmm*)h;
  const Quat q{pose[0], pose[1], pose[2], pose[3]};
  render_view(*g, to_cam(cam), q, pose + 4);
  const int V = (int)g->comps2d.size();
  for (int i = 0; i < V && i < cap; ++i) {
    const Comp2d& c = g->comps2d[i];
    if (id) id[i] = c.id;
    if (mean2d) memcpy(mean2d + 2 * i, c.mean, 16);
    if (cov2d) memcpy(cov2d + 4 * i, c.cov, 32);
    if (depth) depth[i] = c.proj_d;
  }
  return V;
}

// GMM::searchCorrespondence(kpts, vector<comps>&, num): gaussian_mixture.cpp:484-534
// on the last rendered view. cand[N x k] = parent 3-D index (kNN order, gated), -1 padded.
void orc_search_correspondence(void* h, const double* uv, int N, int k, int32_t* cand, int32_t* ncand) {
  OGmm* g = (OGmm*)h;
  const int V = (int)g->comps2d.size();
  const double mdist2_thresh = 9.0;
  std::vector<int> ti(k);
  std::vector<double> td(k);
  for (int n = 0; n < N; ++n) {
    int c = 0;
    if (V > 0)
      c = knn_brute<2>(g->comps2d[0].mean, V, (int)(sizeof(Comp2d) / sizeof(double)), uv + 2 * n, k, ti.data(),
                       td.data());
    int m = 0;
    for (int j = 0; j < c; ++j) {
      const Comp2d& c2 = g->comps2d[ti[j]];
      if (mdist2_2d(c2.mean, c2.cov_inv, uv + 2 * n) < mdist2_thresh) cand[(size_t)n * k + m++] = c2.id;
    }
    ncand[n] = m;
    for (; m < k; ++m) cand[(size_t)n * k + m] = -1;
  }
}

// optimizePoint, batched; out: res[N], chi2_proj[N], chi2_str[N], pt_est[N x 3]
void orc_optimize_point(void* h, const orc_camera* cam, const orc_params* prm, int N, const double* pts,
                        const double* uvr, const int32_t* octave, const double* poses /*N x 7*/,
                        const int32_t* comp, const double* proj_z2, uint8_t* res, double* chi2_proj,
                        double* chi2_str, double* pt_est) {
  OGmm* g = (OGmm*)h;
  const Camera c = to_cam(cam);
  const Params p = to_prm(prm);
  for (int n = 0; n < N; ++n) {
    StrOptStat r = optimize_point(pts + 3 * n, uvr + 3 * n, octave[n], to_se3(poses + 7 * n), g->comps[comp[n]],
                                  proj_z2[n], c, p);
    res[n] = r.res;
    chi2_proj[n] = r.chi2_proj;
    chi2_str[n] = r.chi2_str;
    memcpy(pt_est + 3 * n, r.pt_est, 24);
  }
}

// checkMapAssociation, batched over the features of one key-frame (one pose)
void orc_check_map_association(void* h, const orc_camera* cam, const orc_params* prm, const double* pose, int N,
                               double* pts /*in/out*/, const double* uvr, const int32_t* octave,
                               const int32_t* cand /*N x k*/, const int32_t* ncand, int k, int32_t* out_comp) {
  OGmm* g = (OGmm*)h;
  const Camera c = to_cam(cam);
  const Params p = to_prm(prm);
  if (!g->nbs_built) build_neighbours(*g, p.neighbor_dist_thresh);
  const SE3 T = to_se3(pose);
  for (int n = 0; n < N; ++n)
    out_comp[n] = check_map_association(*g, pts + 3 * n, uvr + 3 * n, octave[n], T, cand + (size_t)n * k, ncand[n], c, p);
}

// optimizeTriangulationVec, batched
void orc_optimize_triangulation(void* h, const orc_camera* cam, const orc_params* prm, int N, double* x3d,
                                const double* pose1 /*N x 7*/, const double* uvr1, const int32_t* oct1,
                                const double* pose2, const double* uvr2, const int32_t* oct2,
                                const int32_t* cand1, const int32_t* n1, const int32_t* cand2, const int32_t* n2,
                                int k, int32_t* out_comp) {
  OGmm* g = (OGmm*)h;
  const Camera c = to_cam(cam);
  const Params p = to_prm(prm);
  for (int n = 0; n < N; ++n)
    out_comp[n] = optimize_triangulation(*g, x3d + 3 * n, to_se3(pose1 + 7 * n), uvr1 + 3 * n, oct1[n],
                                         uvr1[3 * n + 2] >= 0, to_se3(pose2 + 7 * n), uvr2 + 3 * n, oct2[n],
                                         uvr2[3 * n + 2] >= 0, cand1 + (size_t)n * k, n1[n], cand2 + (size_t)n * k,
                                         n2[n], c, p);
}

// ---- Localization::createMapPoints, the per-match block (localization_opt.cpp:286-420):
// parallax test -> linear triangulation (JacobiSVD<Matrix4d>, smallest right singular vector) or
// stereo unprojection -> optimizeTriangulationVec (B2) -> reprojection / scale-consistency checks.
// Smallest right singular vector by one-sided Jacobi (Eigen: two-sided Jacobi with QR preconditioner;
// the vector is defined up to sign and only vt[0..2] / vt[3] is consumed).
static void smallest_right_singular_vector4(const double* A /*4x4 row-major*/, double* v) {
  double U[16], V[16];
  for (int i = 0; i < 16; ++i) {
    U[i] = A[i];
    V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 4; ++k) {
          al += U[k * 4 + p] * U[k * 4 + p];
          be += U[k * 4 + q] * U[k * 4 + q];
          ga += U[k * 4 + p] * U[k * 4 + q];
        }
        if (ga == 0.0) continue;
        off = std::max(off, std::fabs(ga) / std::sqrt(al * be));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int k = 0; k < 4; ++k) {
          const double up = U[k * 4 + p], uq = U[k * 4 + q];
          U[k * 4 + p] = c * up - sn * uq;
          U[k * 4 + q] = sn * up + c * uq;
          const double vp = V[k * 4 + p], vq = V[k * 4 + q];
          V[k * 4 + p] = c * vp - sn * vq;
          V[k * 4 + q] = sn * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  int best = 0;
  double bn = 1e300;
  for (int j = 0; j < 4; ++j) {
    double nn = 0;
    for (int k = 0; k < 4; ++k) nn += U[k * 4 + j] * U[k * 4 + j];
    if (nn < bn) {
      bn = nn;
      best = j;
    }
  }
  for (int k = 0; k < 4; ++k) v[k] = V[k * 4 + best];
}

// type_out: 0 = no map point, 1 FromTriMono, 2 FromTriMonoGMM, 3 FromTriStereo, 4 FromTriStereoGMM (mappoint.h)
void orc_create_map_points(void* h, const orc_camera* cam, const orc_params* prm, float scale_factor, int N,
                           const double* pose1, const double* uvr1, const float* depth1, const int32_t* oct1,
                           const double* pose2, const double* uvr2, const float* depth2, const int32_t* oct2,
                           const int32_t* cand1, const int32_t* n1, const int32_t* cand2, const int32_t* n2, int k,
                           double* x3d_out, int32_t* type_out, int32_t* comp_out) {
  OGmm* g = (OGmm*)h;
  const Camera c = to_cam(cam);
  const Params p = to_prm(prm);
  float sf[8], sigma2[8];
  sf[0] = 1.0f;
  sigma2[0] = 1.0f;
  for (int i = 1; i < 8; ++i) {
    sf[i] = sf[i - 1] * scale_factor;
    sigma2[i] = sf[i] * sf[i];
  }
  const float fx = (float)c.fx, fy = (float)c.fy, cx = (float)c.cx, cy = (float)c.cy;  // const float fx1 = camera_->fx()
  const float invfx = 1.0f / fx, invfy = 1.0f / fy;
  const float mbf = (float)c.bf, mb = mbf / fx;  // frame.cpp:23-24
  const float ratio_factor = 1.5f * scale_factor;
  for (int n = 0; n < N; ++n) {
    type_out[n] = 0;
    comp_out[n] = -1;
    for (int i = 0; i < 3; ++i) x3d_out[3 * n + i] = 0.0;
    const SE3 Tcw1 = to_se3(pose1 + 7 * n), Tcw2 = to_se3(pose2 + 7 * n);
    const SE3 Twc1 = se3_inverse(Tcw1), Twc2 = se3_inverse(Tcw2);
    const double* k1 = uvr1 + 3 * n;
    const double* k2 = uvr2 + 3 * n;
    const float ur1 = (float)k1[2], ur2 = (float)k2[2];
    const bool bStereo1 = ur1 >= 0, bStereo2 = ur2 >= 0;
    const double xn1[3] = {(k1[0] - cx) * invfx, (k1[1] - cy) * invfy, 1.0};
    const double xn2[3] = {(k2[0] - cx) * invfx, (k2[1] - cy) * invfy, 1.0};
    double ray1[3], ray2[3];
    qrot(Twc1.r, xn1, ray1);
    qrot(Twc2.r, xn2, ray2);
    const double dot = ray1[0] * ray2[0] + ray1[1] * ray2[1] + ray1[2] * ray2[2];
    const double nr1 = std::sqrt(ray1[0] * ray1[0] + ray1[1] * ray1[1] + ray1[2] * ray1[2]);
    const double nr2 = std::sqrt(ray2[0] * ray2[0] + ray2[1] * ray2[1] + ray2[2] * ray2[2]);
    const float cosParallaxRays = dot / (nr1 * nr2);
    float cosParallaxStereo = cosParallaxRays + 1;
    float cps1 = cosParallaxStereo, cps2 = cosParallaxStereo;
    if (bStereo1)
      cps1 = std::cos(2 * std::atan2(mb / 2, depth1[n]));
    else if (bStereo2)
      cps2 = std::cos(2 * std::atan2(mb / 2, depth2[n]));
    cosParallaxStereo = std::min(cps1, cps2);
    double pt[3];
    bool from_mono = false;
    if (cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (bStereo1 || bStereo2 || cosParallaxRays < 0.9998)) {
      double R1[9], R2[9], A[16];
      qtoR(Tcw1.r, R1);
      qtoR(Tcw2.r, R2);
      for (int j = 0; j < 4; ++j) {
        const double r1[3] = {j < 3 ? R1[0 * 3 + j] : Tcw1.t[0], j < 3 ? R1[1 * 3 + j] : Tcw1.t[1],
                              j < 3 ? R1[2 * 3 + j] : Tcw1.t[2]};
        const double r2[3] = {j < 3 ? R2[0 * 3 + j] : Tcw2.t[0], j < 3 ? R2[1 * 3 + j] : Tcw2.t[1],
                              j < 3 ? R2[2 * 3 + j] : Tcw2.t[2]};
        A[0 * 4 + j] = xn1[0] * r1[2] - r1[0];
        A[1 * 4 + j] = xn1[1] * r1[2] - r1[1];
        A[2 * 4 + j] = xn2[0] * r2[2] - r2[0];
        A[3 * 4 + j] = xn2[1] * r2[2] - r2[1];
      }
      double vt[4];
      smallest_right_singular_vector4(A, vt);
      for (int i = 0; i < 3; ++i) pt[i] = vt[i] / vt[3];
      from_mono = true;
    } else if (bStereo1 && cps1 < cps2) {
      const double z = depth1[n];
      const double ptc[3] = {z * (k1[0] - c.cx) / c.fx, z * (k1[1] - c.cy) / c.fy, z};
      se3_map(Twc1, ptc, pt);
    } else if (bStereo2 && cps2 < cps1) {
      const double z = depth2[n];
      const double ptc[3] = {z * (k2[0] - c.cx) / c.fx, z * (k2[1] - c.cy) / c.fy, z};
      se3_map(Twc2, ptc, pt);
    } else {
      continue;  // no stereo and very low parallax
    }
    // optimizeTriangulationVec decides mono / stereo edges by kp.depth > 0 (:116-137)
    double b1[3] = {k1[0], k1[1], depth1[n] > 0 ? k1[2] : -1.0}, b2[3] = {k2[0], k2[1], depth2[n] > 0 ? k2[2] : -1.0};
    const int comp = optimize_triangulation(*g, pt, Tcw1, b1, oct1[n], depth1[n] > 0, Tcw2, b2, oct2[n], depth2[n] > 0,
                                            cand1 + (size_t)n * k, n1[n], cand2 + (size_t)n * k, n2[n], c, p);
    for (int i = 0; i < 3; ++i) x3d_out[3 * n + i] = pt[i];
    comp_out[n] = comp;
    // project3 into both key-frames
    auto project = [&](const SE3& Tcw, double* uvr) {
      double pc[3];
      se3_map(Tcw, pt, pc);
      if (pc[2] < 0.0) return false;
      const double rz = 1.0 / pc[2];
      const double u = c.fx * (pc[0] * rz) + c.cx, v = c.fy * (pc[1] * rz) + c.cy;
      if (!(u >= 0.0 && v >= 0.0 && u < (double)c.width && v < (double)c.height && pc[2] > 0.0)) return false;
      uvr[0] = u;
      uvr[1] = v;
      uvr[2] = u - mbf / pc[2];
      return true;
    };
    double p1[3], p2[3];
    if (!project(Tcw1, p1) || !project(Tcw2, p2)) continue;
    auto kp_error = [](const double* kp, float ur, const double* o) {  // Feature::error (feature.h:17-28)
      if (ur < 0.0f) return (kp[0] - o[0]) * (kp[0] - o[0]) + (kp[1] - o[1]) * (kp[1] - o[1]);
      const double d2 = (double)ur - o[2];
      return (kp[0] - o[0]) * (kp[0] - o[0]) + (kp[1] - o[1]) * (kp[1] - o[1]) + d2 * d2;
    };
    const float s2 = sigma2[oct1[n]];  // both checks use kp1's octave (:370-391)
    if (kp_error(k1, ur1, p1) > (bStereo1 ? 7.8 : 5.991) * s2) continue;
    if (kp_error(k2, ur2, p2) > (bStereo2 ? 7.8 : 5.991) * s2) continue;
    const double d1v[3] = {pt[0] - Twc1.t[0], pt[1] - Twc1.t[1], pt[2] - Twc1.t[2]};
    const double d2v[3] = {pt[0] - Twc2.t[0], pt[1] - Twc2.t[1], pt[2] - Twc2.t[2]};
    const float dist1 = std::sqrt(d1v[0] * d1v[0] + d1v[1] * d1v[1] + d1v[2] * d1v[2]);
    const float dist2 = std::sqrt(d2v[0] * d2v[0] + d2v[1] * d2v[1] + d2v[2] * d2v[2]);
    if (dist1 <= std::numeric_limits<float>::epsilon() || dist2 <= std::numeric_limits<float>::epsilon()) continue;
    const float ratio_dist = dist2 / dist1;
    const float ratio_octave = sf[oct1[n]] / sf[oct2[n]];
    if (ratio_dist * ratio_factor < ratio_octave || ratio_dist > ratio_octave * ratio_factor) continue;
    type_out[n] = from_mono ? (comp >= 0 ? 2 : 1) : (comp >= 0 ? 4 : 3);
  }
}

// optimizeCurrentPose for one frame; pose in/out; returns #inliers
int orc_optimize_current_pose(const orc_camera* cam, const orc_params* prm, double* pose, int N, const double* Xw,
                              const double* obs, const int32_t* octave, const uint8_t* has_pt,
                              uint8_t* is_outlier) {
  SE3 T = to_se3(pose);
  const int r = optimize_current_pose(T, N, Xw, obs, octave, has_pt, is_outlier, to_cam(cam), to_prm(prm));
  from_se3(T, pose);
  return r;
}

int orc_joint_optimization(void* h, const orc_camera* cam, const orc_params* prm, int P, int F, int L, int nobs,
                           double* poses, const uint8_t* has_prior, double* points, const int32_t* assoc,
                           const int32_t* obs_ptr, const int32_t* obs_pose, const double* obs_uvr,
                           const int32_t* obs_octave, uint8_t* assoc_dropped, uint8_t* obs_erase) {
  OGmm* g = (OGmm*)h;
  BAProblem pb{P, F, L, nobs, poses, has_prior, points, assoc, obs_ptr, obs_pose, obs_uvr, obs_octave, assoc_dropped, obs_erase};
  return joint_optimization(*g, pb, to_cam(cam), to_prm(prm), nullptr);
}

// the same with the stop word of gl_joint_optimization_stoppable: > 0 the flag is set on entry, < 0 it reads true after
// -value outer iterations (og::Optimizer::stopBudget), 0 no stop
int orc_joint_optimization_stop(void* h, const orc_camera* cam, const orc_params* prm, int P, int F, int L, int nobs,
                                double* poses, const uint8_t* has_prior, double* points, const int32_t* assoc,
                                const int32_t* obs_ptr, const int32_t* obs_pose, const double* obs_uvr,
                                const int32_t* obs_octave, uint8_t* assoc_dropped, uint8_t* obs_erase, int stop_value) {
  OGmm* g = (OGmm*)h;
  BAProblem pb{P, F, L, nobs, poses, has_prior, points, assoc, obs_ptr, obs_pose, obs_uvr, obs_octave, assoc_dropped, obs_erase};
  const bool set = stop_value > 0;
  return joint_optimization(*g, pb, to_cam(cam), to_prm(prm), set ? &set : nullptr, stop_value < 0 ? -stop_value : -1);
}

// switches of the oracle's declared deviations (og_math.hpp): which = 0 ldlt, 1 eig, 2 lambda_break, 3 tri_order
void orc_set_switch(int which, int value) {
  og::Switches& w = og::switches();
  if (which == 0) w.ldlt = value;
  else if (which == 1) w.eig = value;
  else if (which == 2) w.lambda_break = value;
  else if (which == 3) w.tri_order = value;
}

// SE3 helpers exposed for tests
// ---- ORBmatcher::searchByProjection(Frame&, mappts, stats, th) (orb_matcher.cpp:27-110) ---------
// with Frame::assignFeaturesToGrid (frame.cpp:54-79), Frame::getFeaturesInArea (frame.cpp:121-177),
// ORBmatcher::DescriptorDistance (orb_matcher.cpp:580-596), computeRadiusByViewingCos (:112-117),
// the float config scalars of init_config.hpp:50-54,63-79.  Sequential restatement, one frame.
//   feat_taken[idx] != 0  <=>  F.mappoints_[idx] && F.mappoints_[idx]->countObservations() > 0 on entry;
//   a map point assigned by this call has observations (it comes from the local map), so the feature it
//   takes is skipped by the later map points exactly as in the reference loop.
//   out feat_match[idx] = index of the map point the call assigned to feature idx, else -1.
int orc_search_by_projection(int width, int height, float scale_factor, int NF, const double* feat_uv,
                             const float* feat_ur, const int32_t* feat_oct, const uint8_t* feat_desc,
                             const uint8_t* feat_taken, int NP, const double* mp_uvr, const double* mp_level,
                             const double* mp_viewcos, const uint8_t* mp_valid, const uint8_t* mp_desc, float th,
                             float nn_ratio, int32_t* feat_match) {
  const int grid_cols = 64, grid_rows = 48;  // config.h:57
  const float col_inv = static_cast<float>(grid_cols) / width, row_inv = static_cast<float>(grid_rows) / height;
  float sf[8];
  sf[0] = 1.0f;
  for (int i = 1; i < 8; ++i) sf[i] = sf[i - 1] * scale_factor;
  std::vector<std::vector<int>> grid((size_t)grid_cols * grid_rows);
  for (int i = 0; i < NF; ++i) {
    if (feat_oct[i] < 0) continue;  // padding slot, not a feature
    const int px = (int)round((feat_uv[2 * i] - 0.0f) * col_inv), py = (int)round((feat_uv[2 * i + 1] - 0.0f) * row_inv);
    if (px < 0 || px >= grid_cols || py < 0 || py >= grid_rows) continue;
    grid[(size_t)px * grid_rows + py].push_back(i);
  }
  std::vector<uint8_t> taken(feat_taken, feat_taken + NF);
  for (int i = 0; i < NF; ++i) feat_match[i] = -1;
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int m = 0; m < NP; ++m) {
    if (!mp_valid[m]) continue;
    const int lvl_pred = (int)mp_level[m];
    float r = ((float)mp_viewcos[m] > 0.998) ? 2.5f : 4.0f;  // const float& viewCos
    if (bFactor) r *= th;
    const float x = (float)mp_uvr[3 * m], y = (float)mp_uvr[3 * m + 1], rr = r * sf[lvl_pred];
    const int minLevel = lvl_pred - 1, maxLevel = lvl_pred;
    // getFeaturesInArea
    const int x0 = std::max(0, (int)floor((x - 0.0f - rr) * col_inv));
    if (x0 >= grid_cols) continue;
    const int x1 = std::min(grid_cols - 1, (int)ceil((x - 0.0f + rr) * col_inv));
    if (x1 < 0) continue;
    const int y0 = std::max(0, (int)floor((y - 0.0f - rr) * row_inv));
    if (y0 >= grid_rows) continue;
    const int y1 = std::min(grid_rows - 1, (int)ceil((y - 0.0f + rr) * row_inv));
    if (y1 < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int ix = x0; ix <= x1; ++ix)
      for (int iy = y0; iy <= y1; ++iy)
        for (int idx : grid[(size_t)ix * grid_rows + iy]) {
          const int oc = feat_oct[idx];
          if (bCheckLevels) {
            if (oc < minLevel) continue;
            if (maxLevel >= 0 && oc > maxLevel) continue;
          }
          const float distx = feat_uv[2 * idx] - x, disty = feat_uv[2 * idx + 1] - y;
          if (!(fabs(distx) < rr && fabs(disty) < rr)) continue;
          // searchByProjection body
          if (taken[idx]) continue;
          if (feat_ur[idx] > 0) {
            const float er = fabs(mp_uvr[3 * m + 2] - feat_ur[idx]);
            if (er > r * sf[lvl_pred]) continue;
          }
          const int32_t* pa = (const int32_t*)(mp_desc + (size_t)m * 32);
          const int32_t* pb = (const int32_t*)(feat_desc + (size_t)idx * 32);
          int dist = 0;
          for (int w = 0; w < 8; ++w) {
            unsigned int v = pa[w] ^ pb[w];
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
          }
          if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestLevel2 = bestLevel;
            bestLevel = oc;
            bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = oc;
            bestDist2 = dist;
          }
        }
    if (bestDist <= 100) {  // TH_HIGH
      if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
      feat_match[bestIdx] = m;
      taken[bestIdx] = 1;
      nmatches++;
    }
  }
  return nmatches;
}

// ---- ORBmatcher::searchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
//      (orb_matcher.cpp:410-542) + computeThreeMaxima (:544-578), one frame pair.
//  pose_cw / pose_lw: (qx qy qz qw tx ty tz) = getTcw() of the current / last frame.
//  Eigen's Quaternion * Vector3 (uv = 2 q.vec x v; v + w uv + q.vec x uv) is restated literally.
//  last_valid[i] != 0  <=>  LastFrame.mappoints_[i] && !LastFrame.is_outlier_[i];  last_pt = its position.
//  out feat_match[idx] = i (index in the last frame) or -1; returns nmatches after the rotation check.
static void quat_rot(const double* q, const double* v, double* out) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  double uv[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  out[0] = v[0] + qw * uv[0] + (qy * uv[2] - qz * uv[1]);
  out[1] = v[1] + qw * uv[1] + (qz * uv[0] - qx * uv[2]);
  out[2] = v[2] + qw * uv[2] + (qx * uv[1] - qy * uv[0]);
}

int orc_search_by_projection_frame(const orc_camera* cam, float scale_factor, const double* pose_cw,
                                   const double* pose_lw, int NF, const double* feat_uv, const float* feat_ur,
                                   const int32_t* feat_oct, const float* feat_angle, const uint8_t* feat_desc,
                                   const uint8_t* feat_taken, int NL, const double* last_pt, const uint8_t* last_valid,
                                   const int32_t* last_oct, const float* last_angle, const uint8_t* last_desc, float th,
                                   int mono, int check_orientation, int32_t* feat_match) {
  const int grid_cols = 64, grid_rows = 48, HISTO_LENGTH = 30;
  const int width = cam->width, height = cam->height;
  const float col_inv = static_cast<float>(grid_cols) / width, row_inv = static_cast<float>(grid_rows) / height;
  const float cfx = (float)cam->fx, cfy = (float)cam->fy, ccx = (float)cam->cx, ccy = (float)cam->cy,
              cbf = (float)cam->bf;                     // config.h:38-48: float scalars
  const double fx = cfx, fy = cfy, cx = ccx, cy = ccy;  // PinholeCamera::fx() returns double
  const float mbf = cbf, mb = cbf / cfx;                // frame.cpp:23-24
  float sf[8];
  sf[0] = 1.0f;
  for (int i = 1; i < 8; ++i) sf[i] = sf[i - 1] * scale_factor;
  std::vector<std::vector<int>> grid((size_t)grid_cols * grid_rows);
  for (int i = 0; i < NF; ++i) {
    if (feat_oct[i] < 0) continue;
    const int px = (int)round((feat_uv[2 * i] - 0.0f) * col_inv), py = (int)round((feat_uv[2 * i + 1] - 0.0f) * row_inv);
    if (px < 0 || px >= grid_cols || py < 0 || py >= grid_rows) continue;
    grid[(size_t)px * grid_rows + py].push_back(i);
  }
  std::vector<uint8_t> taken(feat_taken, feat_taken + NF);
  for (int i = 0; i < NF; ++i) feat_match[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = HISTO_LENGTH / 360.0f;
  // Twc = Tcw.inverse(): r = conj(q), t = r * (-t);  tlc = Rlw * twc + tlw
  const double qc_inv[4] = {-pose_cw[0], -pose_cw[1], -pose_cw[2], pose_cw[3]};
  const double mt[3] = {pose_cw[4] * -1., pose_cw[5] * -1., pose_cw[6] * -1.};
  double twc[3], tlc[3];
  quat_rot(qc_inv, mt, twc);
  quat_rot(pose_lw, twc, tlc);
  tlc[0] += pose_lw[4];
  tlc[1] += pose_lw[5];
  tlc[2] += pose_lw[6];
  const bool bForward = tlc[2] > mb && !mono;
  const bool bBackward = -tlc[2] > mb && !mono;
  int nmatches = 0;
  for (int i = 0; i < NL; ++i) {
    if (!last_valid[i]) continue;
    double ptc[3];
    quat_rot(pose_cw, last_pt + 3 * i, ptc);
    ptc[0] += pose_cw[4];
    ptc[1] += pose_cw[5];
    ptc[2] += pose_cw[6];
    const float xc = ptc[0], yc = ptc[1], invzc = 1.0 / ptc[2];
    if (invzc < 0) continue;
    const float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
    if (u < 0 || u > width) continue;
    if (v < 0 || v > height) continue;
    const int oct = last_oct[i];
    const float radius = th * sf[oct];
    int minLevel, maxLevel;
    if (bForward) {
      minLevel = oct;
      maxLevel = -1;
    } else if (bBackward) {
      minLevel = 0;
      maxLevel = oct;
    } else {
      minLevel = oct - 1;
      maxLevel = oct + 1;
    }
    const float x = u, y = v, rr = radius;
    const int x0 = std::max(0, (int)floor((x - 0.0f - rr) * col_inv));
    if (x0 >= grid_cols) continue;
    const int x1 = std::min(grid_cols - 1, (int)ceil((x - 0.0f + rr) * col_inv));
    if (x1 < 0) continue;
    const int y0 = std::max(0, (int)floor((y - 0.0f - rr) * row_inv));
    if (y0 >= grid_rows) continue;
    const int y1 = std::min(grid_rows - 1, (int)ceil((y - 0.0f + rr) * row_inv));
    if (y1 < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestIdx2 = -1;
    for (int ix = x0; ix <= x1; ++ix)
      for (int iy = y0; iy <= y1; ++iy)
        for (int i2 : grid[(size_t)ix * grid_rows + iy]) {
          const int oc = feat_oct[i2];
          if (bCheckLevels) {
            if (oc < minLevel) continue;
            if (maxLevel >= 0 && oc > maxLevel) continue;
          }
          const float distx = feat_uv[2 * i2] - x, disty = feat_uv[2 * i2 + 1] - y;
          if (!(fabs(distx) < rr && fabs(disty) < rr)) continue;
          if (taken[i2]) continue;
          if (feat_ur[i2] > 0) {
            const float ur = u - mbf * invzc;
            const float er = fabs(ur - feat_ur[i2]);
            if (er > radius) continue;
          }
          const int32_t* pa = (const int32_t*)(last_desc + (size_t)i * 32);
          const int32_t* pb = (const int32_t*)(feat_desc + (size_t)i2 * 32);
          int dist = 0;
          for (int w = 0; w < 8; ++w) {
            unsigned int vv = pa[w] ^ pb[w];
            vv = vv - ((vv >> 1) & 0x55555555);
            vv = (vv & 0x33333333) + ((vv >> 2) & 0x33333333);
            dist += (((vv + (vv >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
          }
          if (dist < bestDist) {
            bestDist = dist;
            bestIdx2 = i2;
          }
        }
    if (bestDist <= 100) {
      feat_match[bestIdx2] = i;
      taken[bestIdx2] = 1;
      nmatches++;
      if (check_orientation) {
        float rot = last_angle[i] - feat_angle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);  // (assert in the reference)
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int sz = rotHist[i].size();
      if (sz > max1) {
        max3 = max2;
        max2 = max1;
        max1 = sz;
        ind3 = ind2;
        ind2 = ind1;
        ind1 = i;
      } else if (sz > max2) {
        max3 = max2;
        max2 = sz;
        ind3 = ind2;
        ind2 = i;
      } else if (sz > max3) {
        max3 = sz;
        ind3 = i;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1;
      ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) {
          feat_match[idx] = -1;
          nmatches--;
        }
  }
  return nmatches;
}

// ORBmatcher::searchForTriangulation (orb_matcher.cpp:141-293) with checkEpipolarDist (:119-139) and computeThreeMaxima
// (:544-578), one key-frame pair: the producer of createMapPoints' matches (localization_opt.cpp:266).
// The two DBoW2::FeatureVector maps (node id -> feature indices, std::map: ascending node id) arrive as CSR: node_id ascending,
// node_ptr (nn + 1), node_idx in list order.  fmat = MathUtils::computeFundamentalMatrix(Tcw1, K1, Tcw2, K2) (row-major) and the
// epipole (ex, ey) of :155-160 are inputs: they are built with the host's Eigen (K^-T E K^-1, quaternion products) and are not
// part of this path.  has_mp: the feature already has a map point (getMapPoint(idx) != nullptr).  match12: N1, idx2 or -1.
int orc_search_for_triangulation(float scale_factor, int N1, const double* uv1, const float* ur1, const int32_t* oct1, const float* angle1,
                                 const uint8_t* desc1, const uint8_t* has_mp1, int nn1, const int32_t* node_id1, const int32_t* node_ptr1,
                                 const int32_t* node_idx1, int N2, const double* uv2, const float* ur2, const int32_t* oct2,
                                 const float* angle2, const uint8_t* desc2, const uint8_t* has_mp2, int nn2, const int32_t* node_id2,
                                 const int32_t* node_ptr2, const int32_t* node_idx2, const double* fmat, const float* epipole,
                                 int only_stereo, int check_orientation, int32_t* match12) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;  // orb_matcher.cpp:21-22
  float scale_factors[8], sigma2[8];          // init_config.hpp:63-79
  scale_factors[0] = 1.0f;
  sigma2[0] = 1.0f;
  for (int i = 1; i < 8; ++i) {
    scale_factors[i] = scale_factors[i - 1] * scale_factor;
    sigma2[i] = scale_factors[i] * scale_factors[i];
  }
  const float ex = epipole[0], ey = epipole[1];
  auto F = [&](int r, int c) { return fmat[r * 3 + c]; };
  int nmatches = 0;
  std::vector<bool> matched2(N2, false);
  for (int i = 0; i < N1; ++i) match12[i] = -1;
  std::vector<int> rotHist[30];
  const float factor = HISTO_LENGTH / 360.0f;
  int i1n = 0, i2n = 0;
  while (i1n < nn1 && i2n < nn2) {
    if (node_id1[i1n] == node_id2[i2n]) {
      for (int a = node_ptr1[i1n]; a < node_ptr1[i1n + 1]; ++a) {
        const int idx1 = node_idx1[a];
        if (has_mp1[idx1]) continue;
        const bool bStereo1 = ur1[idx1] >= 0;
        if (only_stereo && !bStereo1) continue;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int b = node_ptr2[i2n]; b < node_ptr2[i2n + 1]; ++b) {
          const int idx2 = node_idx2[b];
          if (matched2[idx2] || has_mp2[idx2]) continue;
          const bool bStereo2 = ur2[idx2] >= 0;
          if (only_stereo && !bStereo2) continue;
          const int32_t* pa = (const int32_t*)(desc1 + (size_t)idx1 * 32);
          const int32_t* pb = (const int32_t*)(desc2 + (size_t)idx2 * 32);
          int dist = 0;
          for (int w = 0; w < 8; ++w) {  // DescriptorDistance (:580-596)
            unsigned int vv = pa[w] ^ pb[w];
            vv = vv - ((vv >> 1) & 0x55555555);
            vv = (vv & 0x33333333) + ((vv >> 2) & 0x33333333);
            dist += (((vv + (vv >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
          }
          if (dist > TH_LOW || dist > bestDist) continue;
          const double u2 = uv2[2 * idx2], v2 = uv2[2 * idx2 + 1];
          if (!bStereo1 && !bStereo2) {
            const float distex = ex - u2;
            const float distey = ey - v2;
            if (distex * distex + distey * distey < 100 * scale_factors[oct2[idx2]]) continue;
          }
          {  // checkEpipolarDist(kp1, kp2, fmat)
            const double u1 = uv1[2 * idx1], v1 = uv1[2 * idx1 + 1];
            const double ea = u1 * F(0, 0) + v1 * F(1, 0) + F(2, 0);
            const double eb = u1 * F(0, 1) + v1 * F(1, 1) + F(2, 1);
            const double ec = u1 * F(0, 2) + v1 * F(1, 2) + F(2, 2);
            const float num = ea * u2 + eb * v2 + ec;
            const float den = ea * ea + eb * eb;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if (!(dsqr < 3.84 * sigma2[oct2[idx2]])) continue;
          }
          bestIdx2 = idx2;
          bestDist = dist;
        }
        if (bestIdx2 >= 0) {
          match12[idx1] = bestIdx2;
          matched2[bestIdx2] = true;
          nmatches++;
          if (check_orientation) {
            float rot = angle1[idx1] - angle2[bestIdx2];
            if (rot < 0.0) rot += 360.0f;
            int bin = round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(idx1);  // (assert in the reference)
          }
        }
      }
      ++i1n;
      ++i2n;
    } else if (node_id1[i1n] < node_id2[i2n]) {
      i1n = (int)(std::lower_bound(node_id1, node_id1 + nn1, node_id2[i2n]) - node_id1);
    } else {
      i2n = (int)(std::lower_bound(node_id2, node_id2 + nn2, node_id1[i1n]) - node_id2);
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int sz = rotHist[i].size();
      if (sz > max1) {
        max3 = max2;
        max2 = max1;
        max1 = sz;
        ind3 = ind2;
        ind2 = ind1;
        ind1 = i;
      } else if (sz > max2) {
        max3 = max2;
        max2 = sz;
        ind3 = ind2;
        ind2 = i;
      } else if (sz > max3) {
        max3 = sz;
        ind3 = i;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1;
      ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) {
          match12[idx] = -1;
          nmatches--;
        }
  }
  return nmatches;
}

// ORBmatcher::searchByBoW (orb_matcher.cpp:295-408) with computeThreeMaxima (:544-578), one key-frame / frame pair: the matcher of
// Tracking::trackReferenceKeyFrame (tracking.cpp:303).  Key-frame side ("1"): angle, descriptors, has_mp = the feature has a
// map point that is valid (`pMP && !pMP->not_valid_`), its DBoW2 feature vector as CSR; frame side ("2"): angle, descriptors,
// feature vector.  match21: N2 entries, the KEY-FRAME FEATURE whose map point the reference stores in matches[realIdxF], or -1.
int orc_search_by_bow(float nn_ratio, int check_orientation, int N1, const float* angle1, const uint8_t* desc1, const uint8_t* has_mp1,
                      int nn1, const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* node_idx1, int N2, const float* angle2,
                      const uint8_t* desc2, int nn2, const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* node_idx2,
                      int32_t* match21) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;  // orb_matcher.cpp:21-22
  (void)N1;
  for (int i = 0; i < N2; ++i) match21[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[30];
  const float factor = HISTO_LENGTH / 360.0f;
  int KFit = 0, Fit = 0;
  while (KFit < nn1 && Fit < nn2) {
    if (node_id1[KFit] == node_id2[Fit]) {
      for (int iKF = node_ptr1[KFit]; iKF < node_ptr1[KFit + 1]; ++iKF) {
        const int realIdxKF = node_idx1[iKF];
        if (!has_mp1[realIdxKF]) continue;  // !pMP || pMP->not_valid_
        const int32_t* pa = (const int32_t*)(desc1 + (size_t)realIdxKF * 32);
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int iF = node_ptr2[Fit]; iF < node_ptr2[Fit + 1]; ++iF) {
          const int realIdxF = node_idx2[iF];
          if (match21[realIdxF] >= 0) continue;
          const int32_t* pb = (const int32_t*)(desc2 + (size_t)realIdxF * 32);
          int dist = 0;
          for (int w = 0; w < 8; ++w) {  // DescriptorDistance (:580-596)
            unsigned int vv = pa[w] ^ pb[w];
            vv = vv - ((vv >> 1) & 0x55555555);
            vv = (vv & 0x33333333) + ((vv >> 2) & 0x33333333);
            dist += (((vv + (vv >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
          }
          if (dist < bestDist1) {
            bestDist2 = bestDist1;
            bestDist1 = dist;
            bestIdxF = realIdxF;
          } else if (dist < bestDist2) {
            bestDist2 = dist;
          }
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < nn_ratio * static_cast<float>(bestDist2)) {
            match21[bestIdxF] = realIdxKF;
            if (check_orientation) {
              float rot = angle1[realIdxKF] - angle2[bestIdxF];
              if (rot < 0.0) rot += 360.0f;
              int bin = round(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdxF);  // (assert in the reference)
            }
            nmatches++;
          }
        }
      }
      KFit++;
      Fit++;
    } else if (node_id1[KFit] < node_id2[Fit]) {
      KFit = (int)(std::lower_bound(node_id1, node_id1 + nn1, node_id2[Fit]) - node_id1);
    } else {
      Fit = (int)(std::lower_bound(node_id2, node_id2 + nn2, node_id1[KFit]) - node_id2);
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;  // computeThreeMaxima
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int sz = rotHist[i].size();
      if (sz > max1) {
        max3 = max2;
        max2 = max1;
        max1 = sz;
        ind3 = ind2;
        ind2 = ind1;
        ind1 = i;
      } else if (sz > max2) {
        max3 = max2;
        max2 = sz;
        ind3 = ind2;
        ind2 = i;
      } else if (sz > max3) {
        max3 = sz;
        ind3 = i;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1;
      ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) {
          match21[idx] = -1;
          nmatches--;
        }
  }
  return nmatches;
}

// Localization::fuseObservations (localization.cpp:226-318), the matching half, one key-frame: for every candidate map point
// (mp_valid: non-null, valid, not yet observed by the key-frame, project3 and checkScaleAndVisible passed - the host's part; mp_uvr
// = project3's (u, v, u_right), mp_level = ProjStat::scale_pred) the most similar feature of Frame::getFeaturesInArea(u, v,
// th * scale_factors[level]) (frame.cpp:121-177, no level arguments) whose octave is level - 1 or level and whose
// Feature::error(uvr) * sigma2_inv[octave] (feature.h:17-28) is within 5.99 (mono) / 7.8 (stereo).  best_idx = the feature if its
// distance is <= TH_LOW = 50, else -1; best_dist = the distance found (256: none).  What the reference does with the match
// (addObservation / replaceMapPoint, :296-312) is the host's graph work.  Returns the number of map points with a match.
int orc_fuse_search(int width, int height, float scale_factor, int NF, const double* feat_uv, const float* feat_ur, const int32_t* feat_oct,
                    const uint8_t* feat_desc, int NP, const double* mp_uvr, const int32_t* mp_level, const uint8_t* mp_valid,
                    const uint8_t* mp_desc, float th, int32_t* best_idx_out, int32_t* best_dist_out) {
  const int grid_cols = 64, grid_rows = 48;
  const float col_inv = static_cast<float>(grid_cols) / width, row_inv = static_cast<float>(grid_rows) / height;
  float sf[8], sigma2_inv[8];
  sf[0] = 1.0f;
  sigma2_inv[0] = 1.0f;
  for (int i = 1; i < 8; ++i) {
    sf[i] = sf[i - 1] * scale_factor;
    const float sigma2 = sf[i] * sf[i];
    sigma2_inv[i] = 1.0f / sigma2;
  }
  std::vector<std::vector<int>> grid((size_t)grid_cols * grid_rows);
  for (int i = 0; i < NF; ++i) {
    if (feat_oct[i] < 0) continue;  // padding slot, not a feature
    const int px = (int)round((feat_uv[2 * i] - 0.0f) * col_inv), py = (int)round((feat_uv[2 * i + 1] - 0.0f) * row_inv);
    if (px < 0 || px >= grid_cols || py < 0 || py >= grid_rows) continue;
    grid[(size_t)px * grid_rows + py].push_back(i);
  }
  int num = 0;
  for (int m = 0; m < NP; ++m) {
    best_idx_out[m] = -1;
    best_dist_out[m] = 256;
    if (!mp_valid[m]) continue;
    const double uvr[3] = {mp_uvr[3 * m], mp_uvr[3 * m + 1], mp_uvr[3 * m + 2]};
    const int lvl_pred = mp_level[m];
    const float radius = th * sf[lvl_pred];
    const float x = uvr[0], y = uvr[1], r = radius;  // getFeaturesInArea(const float&, const float&, const float&)
    const int x0 = std::max(0, (int)floor((x - 0.0f - r) * col_inv));
    if (x0 >= grid_cols) continue;
    const int x1 = std::min(grid_cols - 1, (int)ceil((x - 0.0f + r) * col_inv));
    if (x1 < 0) continue;
    const int y0 = std::max(0, (int)floor((y - 0.0f - r) * row_inv));
    if (y0 >= grid_rows) continue;
    const int y1 = std::min(grid_rows - 1, (int)ceil((y - 0.0f + r) * row_inv));
    if (y1 < 0) continue;
    int best_dist = 256, best_idx = -1;
    for (int ix = x0; ix <= x1; ++ix)
      for (int iy = y0; iy <= y1; ++iy)
        for (int idx : grid[(size_t)ix * grid_rows + iy]) {
          const float distx = feat_uv[2 * idx] - x, disty = feat_uv[2 * idx + 1] - y;
          if (!(fabs(distx) < r && fabs(disty) < r)) continue;
          const int kpLevel = feat_oct[idx];
          if (kpLevel < lvl_pred - 1 || kpLevel > lvl_pred) continue;
          double err;
          {
            const double dx = feat_uv[2 * idx] - uvr[0], dy = feat_uv[2 * idx + 1] - uvr[1];
            if (feat_ur[idx] < 0.0f) {
              err = dx * dx + dy * dy;
            } else {
              const double dz = (double)feat_ur[idx] - uvr[2];
              err = dx * dx + dy * dy + dz * dz;
            }
          }
          err *= sigma2_inv[kpLevel];
          const double thresh = feat_ur[idx] >= 0 ? 7.8 : 5.99;
          if (err > thresh) continue;
          const int32_t* pa = (const int32_t*)(mp_desc + (size_t)m * 32);
          const int32_t* pb = (const int32_t*)(feat_desc + (size_t)idx * 32);
          int dist = 0;
          for (int w = 0; w < 8; ++w) {
            unsigned int v = pa[w] ^ pb[w];
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
          }
          if (dist < best_dist) {
            best_dist = dist;
            best_idx = idx;
          }
        }
    best_dist_out[m] = best_dist;
    if (best_dist <= 50) {  // TH_LOW
      best_idx_out[m] = best_idx;
      num++;
    }
  }
  return num;
}

// Frame::project3(pt, &uvr) (frame.cpp:98-119; g2o SE3Quat::map; PinholeCamera::project3 + evaluateProjectionResult,
// pinhole_camera.cpp:46-66, 128-150, kMinimumDepth = 0) followed by MapPoint::checkScaleAndVisible (mappoint.cpp:257-303), one frame,
// NP map points - the loop of Tracking::searchLocalPoints (tracking.cpp:233-256) / Localization::fuseObservations (:242-254).
// cam: fx fy cx cy bf as the float config scalars.  cand[m]: the host's tests in front of the projection.  Returns the in-view count.
// T_w_c.translation() of a pose T_cw (g2o SE3Quat::inverse: r = conj(q), t = r * (-t); Eigen's quaternion-vector product): what
// Tracking::searchLocalPoints reads from curr_frame_->T_w_c_ (tracking.cpp:233) after the pose was set
void orc_pose_twc(const double* pose_cw, double* t_wc) {
  const double qi[4] = {-pose_cw[0], -pose_cw[1], -pose_cw[2], pose_cw[3]};
  const double mt[3] = {pose_cw[4] * -1., pose_cw[5] * -1., pose_cw[6] * -1.};
  quat_rot(qi, mt, t_wc);
}

int orc_project_map_points(const orc_camera* cam, float scale_factor, const double* pose_cw, const double* t_wc, int NP, const double* pos,
                           const double* normal, const float* max_dist_, const float* min_dist_, const uint8_t* cand, double* uvr_out,
                           int32_t* level_out, double* viewcos_out, double* dist_out, uint8_t* inview_out) {
  const double fu = (float)cam->fx, fv = (float)cam->fy, cu = (float)cam->cx, cv = (float)cam->cy;
  const float mbf = (float)cam->bf;
  const float scale_factor_log = std::log(scale_factor);  // config.cpp:57
  const int num_levels = 8;
  int n = 0;
  for (int m = 0; m < NP; ++m) {
    uvr_out[3 * m] = uvr_out[3 * m + 1] = uvr_out[3 * m + 2] = 0.0;
    level_out[m] = 0;
    viewcos_out[m] = dist_out[m] = 0.0;
    inview_out[m] = 0;
    if (!cand[m]) continue;
    double ptc[3];
    quat_rot(pose_cw, pos + 3 * m, ptc);
    ptc[0] += pose_cw[4];
    ptc[1] += pose_cw[5];
    ptc[2] += pose_cw[6];
    if (ptc[2] < 0.0) continue;
    const double rz = static_cast<double>(1.0) / ptc[2];
    const double kx = ptc[0] * rz, ky = ptc[1] * rz;
    const double u = fu * kx + cu, v = fv * ky + cv;
    const bool visibility = u >= 0.0 && v >= 0.0 && u < static_cast<double>(cam->width) && v < static_cast<double>(cam->height);
    if (!(visibility && ptc[2] > 0.0)) continue;
    const double ur = u - mbf / ptc[2];
    // checkScaleAndVisible
    const float max_dist = 1.2f * max_dist_[m], min_dist = 0.8f * min_dist_[m];
    const double vx = pos[3 * m] - t_wc[0], vy = pos[3 * m + 1] - t_wc[1], vz = pos[3 * m + 2] - t_wc[2];
    const float dist = std::sqrt(vx * vx + vy * vy + vz * vz);
    if (dist < min_dist || dist > max_dist) continue;
    const float view_cos = (vx * normal[3 * m] + vy * normal[3 * m + 1] + vz * normal[3 * m + 2]) / dist;
    if (view_cos < 0.5f) continue;
    const float ratio = max_dist_[m] / dist;
    // std::log(float), std::ceil(float): mappoint.cpp has using namespace std.  inf / NaN (dist == 0): the int conversion is undefined,
    // x86 gives INT_MIN (-> level 0)
    const float lq = std::ceil(std::log(ratio) / scale_factor_log);
    int lvl_scale = std::isfinite(lq) ? (int)lq : INT_MIN;
    if (lvl_scale < 0) lvl_scale = 0;
    else if (lvl_scale >= num_levels) lvl_scale = num_levels - 1;
    uvr_out[3 * m] = u;
    uvr_out[3 * m + 1] = v;
    uvr_out[3 * m + 2] = ur;
    level_out[m] = lvl_scale;
    viewcos_out[m] = view_cos;
    dist_out[m] = dist;
    inview_out[m] = 1;
    ++n;
  }
  return n;
}

void orc_se3_exp(const double* u, double* pose) { from_se3(se3_exp(u), pose); }
void orc_se3_log(const double* pose, double* u) { se3_log(to_se3(pose), u); }
void orc_se3_mul(const double* a, const double* b, double* out) { from_se3(se3_mul(to_se3(a), to_se3(b)), out); }
void orc_se3_map(const double* pose, const double* x, double* out) { se3_map(to_se3(pose), x, out); }

}  // extern "C"
