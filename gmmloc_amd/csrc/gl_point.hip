// Point refinement against the GMM:
//   k_optimize_point        GMMLoc::optimizePoint              (gmmloc_opt.cpp:260-342)   B1   a thread per problem
//   k_check_map_association GMMLoc::checkMapAssociation        (gmmloc_opt.cpp:156-258)   A8   a DPP row (16 lanes) per feature
//   k_optimize_triangulation Localization::optimizeTriangulationVec (localization_opt.cpp:27-204) B2   a DPP row per match
// (A8 / B2: one candidate component per lane, the ordered choices of the reference as lexicographic row argmins.)
// Each problem is a 3-DoF Gauss-Newton (g2o OptimizationAlgorithmGaussNewton, BlockSolverX +
// LinearSolverEigen on one 3x3 block) over EdgeProjectXYZOnly{,Stereo} + EdgePt2GaussianDeg
// (factors.cpp:55-168).  The reference pays a g2o graph construction (heap allocations,
// virtual dispatch) per call; here the whole solve lives in registers.
// g2o semantic kept: e->chi2() after optimize(n) is the error of the LAST
// computeActiveErrors(), i.e. evaluated before the final update.
#include "gl_device.hpp"
#include "gl_internal.hpp"

using namespace gld;

namespace {

struct PtK {
  double fx, fy, cx, cy, bf;
  double s2inv[8];
  double tri_lambda2;      // (double) loc::tri_lambda2
  double str_thresh;       // (double)(tri_str_thresh * tri_lambda2), float product
  int check_str;
};

struct Plane {
  double n[3], mu[3];
};

struct FixedPose {
  double R[9], t[3];
};

GL_DEV FixedPose load_pose(const double* p) {
  const SE3 T = se3_load(p);
  FixedPose f;
  qtoR(T.r, f.R);
  f.t[0] = T.t[0];
  f.t[1] = T.t[1];
  f.t[2] = T.t[2];
  return f;
}

// 3x3 symmetric solve by LDL^T (SimplicialLDLT: fails on a zero pivot only)
GL_DEV bool solve3(const double* H, const double* b, double* x) { return ldlt_solve<3>(H, b, x, false); }

// accumulate one fixed-pose reprojection edge at point x: returns chi2 (= s |e|^2)
GL_DEV double reproj_edge(const PtK& k, const FixedPose& P, const double* x, const double* uvr, bool stereo, double s,
                          double* H, double* b) {
  const double X = P.R[0] * x[0] + P.R[1] * x[1] + P.R[2] * x[2] + P.t[0];
  const double Y = P.R[3] * x[0] + P.R[4] * x[1] + P.R[5] * x[2] + P.t[1];
  const double Z = P.R[6] * x[0] + P.R[7] * x[1] + P.R[8] * x[2] + P.t[2];
  double J[9], e[3];
  int D;
  if (stereo) {  // factors.cpp:116-131 (error), :133-168 (Jacobian)
    const double invz = 1.0 / Z;
    const double pu = X * invz * k.fx + k.cx, pv = Y * invz * k.fy + k.cy;
    e[0] = uvr[0] - pu;
    e[1] = uvr[1] - pv;
    e[2] = uvr[2] - (pu - k.bf * invz);
    const double z2 = Z * Z;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      J[0 * 3 + c] = -k.fx * P.R[0 * 3 + c] / Z + k.fx * X * P.R[2 * 3 + c] / z2;
      J[1 * 3 + c] = -k.fy * P.R[1 * 3 + c] / Z + k.fy * Y * P.R[2 * 3 + c] / z2;
      J[2 * 3 + c] = J[0 * 3 + c] - k.bf * P.R[2 * 3 + c] / z2;
    }
    D = 3;
  } else {  // factors.cpp:66-82 (error), :84-107 (Jacobian)
    e[0] = uvr[0] - (X / Z * k.fx + k.cx);
    e[1] = uvr[1] - (Y / Z * k.fy + k.cy);
    e[2] = 0.0;
    const double tmp[6] = {k.fx, 0.0, -X / Z * k.fx, 0.0, k.fy, -Y / Z * k.fy};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        J[r * 3 + c] = -1. / Z * (tmp[r * 3] * P.R[c] + tmp[r * 3 + 1] * P.R[3 + c] + tmp[r * 3 + 2] * P.R[6 + c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) J[6 + c] = 0.0;
    D = 2;
  }
  double chi = 0.0;
  for (int r = 0; r < D; ++r) chi += e[r] * (s * e[r]);
  if (H) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double bi = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) bi += J[r * 3 + i] * (s * e[r]);
      b[i] -= bi;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double h = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) h += J[r * 3 + i] * s * J[r * 3 + j];
        H[i * 3 + j] += h;
      }
    }
  }
  return chi;
}

// EdgePt2GaussianDeg (factors.cpp:55-64) with information `lam`
GL_DEV double plane_edge(const Plane& pl, const double* x, double lam, double* H, double* b) {
  const double es = pl.n[0] * (x[0] - pl.mu[0]) + pl.n[1] * (x[1] - pl.mu[1]) + pl.n[2] * (x[2] - pl.mu[2]);
  if (H) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      b[i] -= pl.n[i] * (lam * es);
#pragma unroll
      for (int j = 0; j < 3; ++j) H[i * 3 + j] += pl.n[i] * lam * pl.n[j];
    }
  }
  return es * (lam * es);
}

struct StrOptStat {  // types/map.h:30-35
  bool res;
  double chi2_proj, chi2_str;
  double pt[3];
};

// GMMLoc::optimizePoint
GL_DEV StrOptStat optimize_point(const PtK& k, const FixedPose& P, const double* pt, const double* uvr, int octave,
                                 const Plane& pl, double proj_z2) {
  const double s = k.s2inv[octave];
  const double lam = 1.0 * k.tri_lambda2 * proj_z2;
  StrOptStat r;
  double x[3] = {pt[0], pt[1], pt[2]};
  r.chi2_proj = r.chi2_str = 0.0;
  for (int it = 0; it < 5; ++it) {
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}, dx[3];
    r.chi2_proj = reproj_edge(k, P, x, uvr, true, s, H, b);
    r.chi2_str = plane_edge(pl, x, lam, H, b);
    if (!solve3(H, b, dx)) break;  // SolverResult::Fail ends optimize()
    x[0] += dx[0];
    x[1] += dx[1];
    x[2] += dx[2];
  }
  r.pt[0] = x[0];
  r.pt[1] = x[1];
  r.pt[2] = x[2];
  r.res = true;
  if (r.chi2_proj > 7.815) r.res = false;
  if (k.check_str && r.chi2_str > k.str_thresh) r.res = false;
  return r;
}

GL_DEV Plane load_plane(const double* __restrict__ rec12, const double* __restrict__ axis, int c) {
  Plane pl;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    pl.n[i] = axis[(size_t)c * 9 + i * 3];  // axis_.col(0)
    pl.mu[i] = rec12[(size_t)c * 12 + i];
  }
  return pl;
}

__global__ void k_optimize_point(PtK k, int N, int K, const double* __restrict__ rec12, const double* __restrict__ axis,
                                 const double* __restrict__ pts, const double* __restrict__ uvr,
                                 const int32_t* __restrict__ octave, const double* __restrict__ pose,
                                 const int32_t* __restrict__ comp, const double* __restrict__ proj_z2,
                                 uint8_t* __restrict__ res, double* __restrict__ chi2_proj,
                                 double* __restrict__ chi2_str, double* __restrict__ pt_est) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  // "no component" (-1, the marker used everywhere in this API) / an octave outside the pyramid: no problem to
  // solve - failure with the point untouched, never an out-of-range read of the component / sigma tables
  if (comp[n] < 0 || comp[n] >= K || octave[n] < 0 || octave[n] > 7) {
    res[n] = 0;
    chi2_proj[n] = 0.0;
    chi2_str[n] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) pt_est[(size_t)n * 3 + i] = pts[(size_t)n * 3 + i];
    return;
  }
  const FixedPose P = load_pose(pose + (size_t)n * 7);
  const Plane pl = load_plane(rec12, axis, comp[n]);
  const StrOptStat r = optimize_point(k, P, pts + (size_t)n * 3, uvr + (size_t)n * 3, octave[n], pl, proj_z2[n]);
  res[n] = r.res ? 1 : 0;
  chi2_proj[n] = r.chi2_proj;
  chi2_str[n] = r.chi2_str;
#pragma unroll
  for (int i = 0; i < 3; ++i) pt_est[(size_t)n * 3 + i] = r.pt[i];
}

// GMMLoc::checkMapAssociation, one thread per feature
// GMMLoc::checkMapAssociation (gmmloc_opt.cpp:156-258): one DPP row (16 lanes) per feature.  The candidate
// refinements (one optimizePoint each) run on one lane each, the neighbour scan and the nearest-mean fallback are
// dealt round the lanes; every "first minimum in order" of the sequential code is a lexicographic
// (value, position) argmin over the row.  With a thread per feature one key-frame (1 000 features) took 0.46 ms -
// the time of its slowest thread's chain of dependent loads - against 0.64 ms for 128 of them.
constexpr int CMA_LANES = 16;
template <int CTRL>
GL_DEV void row_lexmin(double& d, int& i) {
  const double od = dpp_f64<CTRL>(d);
  const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, false);
  const bool t = od < d || (od == d && oi < i);
  d = t ? od : d;
  i = t ? oi : i;
}
GL_DEV void row_argmin16(double& d, int& i) {
  row_lexmin<0xB1>(d, i);   // quad_perm [1,0,3,2]
  row_lexmin<0x4E>(d, i);   // quad_perm [2,3,0,1]
  row_lexmin<0x141>(d, i);  // row_half_mirror
  row_lexmin<0x140>(d, i);  // row_mirror
}
__global__ __launch_bounds__(256) void k_check_map_association(PtK k, int B, int N, int K, const double* __restrict__ rec12,
                                        const double* __restrict__ axis, const uint8_t* __restrict__ flags,
                                        const int32_t* __restrict__ nbs_ptr, const int32_t* __restrict__ nbs_idx,
                                        const double* __restrict__ pose_all, double* __restrict__ pts_all,
                                        const double* __restrict__ uvr_all, const int32_t* __restrict__ oct_all,
                                        const int32_t* __restrict__ cand_all, const int32_t* __restrict__ ncand_all,
                                        int kc, int32_t* __restrict__ out_comp) {
  const size_t tg = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t gid = tg / CMA_LANES;
  const int sub = (int)(tg % CMA_LANES);
  if (gid >= (size_t)B * N) return;  // the whole row leaves together
  const int row0 = (int)(threadIdx.x & 63 & ~(CMA_LANES - 1));
  const int f = (int)(gid / N);
  int result = -1;
  const int oc = oct_all[gid];
  const int nc = ncand_all[gid];
  if (oc >= 0 && nc > 0) {  // comps.empty() -> nullptr (:162-164)
    const SE3 T = se3_load(pose_all + (size_t)f * 7);
    FixedPose P;
    qtoR(T.r, P.R);
    P.t[0] = T.t[0];
    P.t[1] = T.t[1];
    P.t[2] = T.t[2];
    double* pt3d = pts_all + gid * 3;
    const double pt_init[3] = {pt3d[0], pt3d[1], pt3d[2]};
    const double* uvr = uvr_all + gid * 3;
    double ptc[3];
    qrot(T.r, pt_init, ptc);
    double proj_z = ptc[2] + T.t[2];
    proj_z = proj_z > 1.0 ? 1.0 : proj_z;  // :169-172
    const double proj_z2 = proj_z * proj_z;
    const int32_t* cand = cand_all + gid * kc;
    // candidates (:179-197): `r.res && r.chi2_proj < min_value` in order = lowest position among the minima
    double val = 1.7976931348623157e308, res[3] = {0, 0, 0};
    if (sub < nc && cand[sub] >= 0) {
      const StrOptStat r = optimize_point(k, P, pt_init, uvr, oc, load_plane(rec12, axis, cand[sub]), proj_z2);
      if (r.res && r.chi2_proj < 1.7976931348623157e308) {
        val = r.chi2_proj;
        res[0] = r.pt[0];
        res[1] = r.pt[1];
        res[2] = r.pt[2];
      }
    }
    double best = val;
    int min_idx = sub;
    row_argmin16(best, min_idx);
    double min_res[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) min_res[i] = __shfl(res[i], row0 + min_idx, 64);
    if (best < 1.7976931348623157e308) {
      const int g3d = cand[min_idx];
      // neighbour refinement (:203-217): the first neighbour with the smallest chi2, if that is below g3d's
      const int e0 = nbs_ptr[g3d], e1 = nbs_ptr[g3d + 1];
      double ln = __builtin_inf();
      int le = 0x7fffffff;
      for (int e = e0 + sub; e < e1; e += CMA_LANES) {
        const double v = chi2_rec(rec12 + (size_t)nbs_idx[e] * 12, min_res[0], min_res[1], min_res[2]);
        if (v < ln) {
          ln = v;
          le = e;
        }
      }
      row_argmin16(ln, le);
      double ll = chi2_rec(rec12 + (size_t)g3d * 12, min_res[0], min_res[1], min_res[2]);
      int str = g3d;
      if (ln < ll) {
        ll = ln;
        str = nbs_idx[le];
      }
      if (str != g3d) {  // :219-228 (every lane of the row repeats it)
        const StrOptStat r = optimize_point(k, P, pt_init, uvr, oc, load_plane(rec12, axis, str), proj_z2);
        if (r.res) {
          min_res[0] = r.pt[0];
          min_res[1] = r.pt[1];
          min_res[2] = r.pt[2];
        } else {
          str = g3d;
          ll = chi2_rec(rec12 + (size_t)g3d * 12, min_res[0], min_res[1], min_res[2]);
        }
      }
      if (!(ll > 9.0)) {  // :230-235
        if (sub == 0) {
          pt3d[0] = min_res[0];
          pt3d[1] = min_res[1];
          pt3d[2] = min_res[2];
        }
        result = str;
      }
    } else {
      // GMM::queryPoint: nearest mean (:237-256); moves the point but still returns nullptr
      double bd = __builtin_inf();
      int gi = 0x7fffffff;
      for (int c = sub; c < K; c += CMA_LANES) {
        const double d0 = pt_init[0] - rec12[(size_t)c * 12], d1 = pt_init[1] - rec12[(size_t)c * 12 + 1],
                     d2 = pt_init[2] - rec12[(size_t)c * 12 + 2];
        const double d = (d0 * d0 + d1 * d1) + d2 * d2;
        if (d < bd) {
          bd = d;
          gi = c;
        }
      }
      row_argmin16(bd, gi);
      if (gi != 0x7fffffff && (flags[gi] & 1)) {
        const StrOptStat r = optimize_point(k, P, pt_init, uvr, oc, load_plane(rec12, axis, gi), proj_z2);
        if (r.res && sub == 0) {
          pt3d[0] = r.pt[0];
          pt3d[1] = r.pt[1];
          pt3d[2] = r.pt[2];
        }
      }
    }
  }
  if (sub == 0) out_comp[gid] = result;
}

// Localization::optimizeTriangulationVec: one DPP row (16 lanes) per triangulated match, one candidate component
// per lane (at most 2 x kc <= 16 of them).  The 20 Gauss-Newton iterations of the candidates are independent;
// only the choice is ordered - `err_sum < min_value` in candidate order, i.e. the lowest candidate position
// among the minima - and that is a lexicographic (value, position) argmin over the row.  A thread per match ran
// the candidates one after the other: 0.73 ms for the 1 000 matches of one key-frame pair, against 0.4 ms for
// 64 000 of them (latency, not work).
constexpr int TRI_LANES = 16;
template <int CTRL>
GL_DEV void tri_lexmin(double& d, int& i) {
  const double od = dpp_f64<CTRL>(d);
  const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, false);
  const bool t = od < d || (od == d && oi < i);
  d = t ? od : d;
  i = t ? oi : i;
}
__global__ __launch_bounds__(256) void k_optimize_triangulation(PtK k, int N, const double* __restrict__ rec12,
                                         const double* __restrict__ axis, const uint8_t* __restrict__ flags,
                                         double* __restrict__ x3d_all, const double* __restrict__ pose1,
                                         const double* __restrict__ uvr1, const int32_t* __restrict__ oct1,
                                         const double* __restrict__ pose2, const double* __restrict__ uvr2,
                                         const int32_t* __restrict__ cand1, const int32_t* __restrict__ n1,
                                         const int32_t* __restrict__ cand2, const int32_t* __restrict__ n2, int kc,
                                         int32_t* __restrict__ out_comp) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = gid / TRI_LANES, ci = gid % TRI_LANES;
  if (n >= N) return;  // the whole row leaves together
  const FixedPose P1 = load_pose(pose1 + (size_t)n * 7), P2 = load_pose(pose2 + (size_t)n * 7);
  const double* u1 = uvr1 + (size_t)n * 3;
  const double* u2 = uvr2 + (size_t)n * 3;
  const bool st1 = !(u1[2] < 0), st2 = !(u2[2] < 0);  // kp.depth > 0 <=> u_right >= 0
  const double th1 = st1 ? 7.8 : 5.991, th2 = st2 ? 7.8 : 5.991;  // :120-136
  if (oct1[n] < 0 || oct1[n] > 7) {  // octave outside the pyramid: nothing to weigh the edges with -> no association
    if (ci == 0) out_comp[n] = -1;
    return;
  }
  const double s1 = k.s2inv[oct1[n]];  // both edges use kp1's sigma2_inv (:132,135)
  double* x3d = x3d_all + (size_t)n * 3;
  const double pt_init[3] = {x3d[0], x3d[1], x3d[2]};
  const int c1n = n1[n], c2n = n2[n];
  auto cand_at = [&](int i) { return i < c1n ? cand1[(size_t)n * kc + i] : cand2[(size_t)n * kc + (i - c1n)]; };
  int c = -1;
  if (ci < c1n + c2n) {
    c = cand_at(ci);
    for (int cj = 0; cj < ci; ++cj)  // the reference de-duplicates through an unordered_set (:143-152)
      if (c >= 0 && cand_at(cj) == c) c = -1;
    if (c >= 0 && !(flags[c] & 1)) c = -1;  // only degenerate components (:155-157)
  }
  double x[3] = {pt_init[0], pt_init[1], pt_init[2]};
  double val = 1.7976931348623157e308;  // "not a candidate": never below min_value's initial value
  if (c >= 0) {
    const Plane pl = load_plane(rec12, axis, c);
    double e1 = 0, e2 = 0, es = 0;
    for (int it = 0; it < 20; ++it) {  // :169-171
      double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}, dx[3];
      e1 = reproj_edge(k, P1, x, u1, st1, s1, H, b);
      e2 = reproj_edge(k, P2, x, u2, st2, s1, H, b);
      es = plane_edge(pl, x, 1.0 * k.tri_lambda2, H, b);
      if (!solve3(H, b, dx)) break;
      x[0] += dx[0];
      x[1] += dx[1];
      x[2] += dx[2];
    }
    bool ok = true;
    if (k.check_str && es > k.str_thresh) ok = false;
    const double err_sum = e1 + e2;
    if (e1 > th1 || e2 > th2) ok = false;
    if (ok && err_sum < 1.7976931348623157e308) val = err_sum;
  }
  double best = val;
  int bi = ci;
  tri_lexmin<0xB1>(best, bi);   // quad_perm [1,0,3,2]
  tri_lexmin<0x4E>(best, bi);   // quad_perm [2,3,0,1]
  tri_lexmin<0x141>(best, bi);  // row_half_mirror
  tri_lexmin<0x140>(best, bi);  // row_mirror
  const bool found = best < 1.7976931348623157e308;
  const int src = (int)(threadIdx.x & 63 & ~(TRI_LANES - 1)) + bi;  // the winning lane of this row
  const double wx = __shfl(x[0], src, 64), wy = __shfl(x[1], src, 64), wz = __shfl(x[2], src, 64);
  const int wc = __shfl(c, src, 64);
  if (ci == 0) {
    if (found) {
      x3d[0] = wx;
      x3d[1] = wy;
      x3d[2] = wz;
    }
    out_comp[n] = found ? wc : -1;
  }
}

// ---- Localization::createMapPoints, per-match block (localization_opt.cpp:286-420) ---------------------
struct TriK {
  double fx, fy, cx, cy;                 // PinholeCamera intrinsics (double)
  float ffx, ffy, fcx, fcy, invfx, invfy;  // `const float fx1 = camera_->fx()` ... (:227-232)
  float mbf, mb, ratio_factor;
  float sf[8], sigma2[8];
  int width, height;
};

// smallest right singular vector of a 4 x 4 matrix by one-sided Jacobi (the reference: JacobiSVD, V.col(3))
__device__ void smallest_rsv4(const double* A, double* v) {
  double U[16], V[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    U[i] = A[i];
    V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          al += U[k * 4 + p] * U[k * 4 + p];
          be += U[k * 4 + q] * U[k * 4 + q];
          ga += U[k * 4 + p] * U[k * 4 + q];
        }
        if (ga != 0.0) {
          off = fmax(off, fabs(ga) / sqrt(al * be));
          const double zeta = (be - al) / (2.0 * ga);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double up = U[k * 4 + p], uq = U[k * 4 + q];
            U[k * 4 + p] = c * up - sn * uq;
            U[k * 4 + q] = sn * up + c * uq;
            const double vp = V[k * 4 + p], vq = V[k * 4 + q];
            V[k * 4 + p] = c * vp - sn * vq;
            V[k * 4 + q] = sn * vp + c * vq;
          }
        }
      }
    if (off < 1e-15) break;
  }
  double nn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    nn[j] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) nn[j] += U[k * 4 + j] * U[k * 4 + j];
  }
  int best = 0;
  double bn = nn[0];
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (nn[j] < bn) {
      bn = nn[j];
      best = j;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = best == 0 ? V[k * 4] : (best == 1 ? V[k * 4 + 1] : (best == 2 ? V[k * 4 + 2] : V[k * 4 + 3]));
}

// stage 1: parallax test, triangulation / unprojection.  stage[n]: 0 = no point, 1 = from two views, 2 = from stereo.
// b1 / b2: the key-points as optimizeTriangulationVec reads them (u_right = -1 unless depth > 0, :116-137).
__global__ void k_tri_pre(TriK k, int N, const double* __restrict__ pose1, const double* __restrict__ uvr1,
                          const float* __restrict__ depth1, const double* __restrict__ pose2,
                          const double* __restrict__ uvr2, const float* __restrict__ depth2, double* __restrict__ x3d,
                          int32_t* __restrict__ stage, double* __restrict__ b1, double* __restrict__ b2) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const SE3 Tcw1 = se3_load(pose1 + (size_t)n * 7), Tcw2 = se3_load(pose2 + (size_t)n * 7);
  const SE3 Twc1 = se3_inverse(Tcw1), Twc2 = se3_inverse(Tcw2);
  const double* k1 = uvr1 + (size_t)n * 3;
  const double* k2 = uvr2 + (size_t)n * 3;
  const float ur1 = (float)k1[2], ur2 = (float)k2[2], dp1 = depth1[n], dp2 = depth2[n];
  const bool bStereo1 = ur1 >= 0, bStereo2 = ur2 >= 0;
  const double xn1[3] = {(k1[0] - k.fcx) * k.invfx, (k1[1] - k.fcy) * k.invfy, 1.0};
  const double xn2[3] = {(k2[0] - k.fcx) * k.invfx, (k2[1] - k.fcy) * k.invfy, 1.0};
  double ray1[3], ray2[3];
  qrot(Twc1.r, xn1, ray1);
  qrot(Twc2.r, xn2, ray2);
  const double dot = ray1[0] * ray2[0] + ray1[1] * ray2[1] + ray1[2] * ray2[2];
  const double nr1 = sqrt(ray1[0] * ray1[0] + ray1[1] * ray1[1] + ray1[2] * ray1[2]);
  const double nr2 = sqrt(ray2[0] * ray2[0] + ray2[1] * ray2[1] + ray2[2] * ray2[2]);
  const float cosRays = (float)(dot / (nr1 * nr2));
  float cps = cosRays + 1;
  float cps1 = cps, cps2 = cps;
  // cos(2 atan2(mb / 2, depth)) in FLOAT (localization_opt.cpp:311-315).  Whether `cosRays < cps` holds can be the last bit of
  // this value (one match in 48 million of the 50 000-round soak), and the last bit of a float cosine of a float arc tangent
  // is the math library's: glibc's cosf(2 atan2f()) and the device library's differ on 0.34 % of the depths.  Each of the two
  // functions is therefore evaluated in double and rounded once - the correctly rounded float function, which glibc's is on all
  // but 0.036 % of the depths.
  const float mbh = k.mb / 2;
  auto cps_of = [&](float dp) { return (float)cos(2.0 * (double)(float)atan2((double)mbh, (double)dp)); };
  if (bStereo1)
    cps1 = cps_of(dp1);
  else if (bStereo2)
    cps2 = cps_of(dp2);
  cps = fminf(cps1, cps2);
  double pt[3] = {0, 0, 0};
  int st = 0;
  if (cosRays < cps && cosRays > 0 && (bStereo1 || bStereo2 || cosRays < 0.9998)) {
    double R1[9], R2[9], A[16];
    qtoR(Tcw1.r, R1);
    qtoR(Tcw2.r, R2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double r1[3] = {j < 3 ? R1[0 * 3 + j] : Tcw1.t[0], j < 3 ? R1[1 * 3 + j] : Tcw1.t[1], j < 3 ? R1[2 * 3 + j] : Tcw1.t[2]};
      const double r2[3] = {j < 3 ? R2[0 * 3 + j] : Tcw2.t[0], j < 3 ? R2[1 * 3 + j] : Tcw2.t[1], j < 3 ? R2[2 * 3 + j] : Tcw2.t[2]};
      A[0 * 4 + j] = xn1[0] * r1[2] - r1[0];
      A[1 * 4 + j] = xn1[1] * r1[2] - r1[1];
      A[2 * 4 + j] = xn2[0] * r2[2] - r2[0];
      A[3 * 4 + j] = xn2[1] * r2[2] - r2[1];
    }
    double vt[4];
    smallest_rsv4(A, vt);
    for (int i = 0; i < 3; ++i) pt[i] = vt[i] / vt[3];
    st = 1;
  } else if (bStereo1 && cps1 < cps2) {
    const double z = dp1;
    const double ptc[3] = {z * (k1[0] - k.cx) / k.fx, z * (k1[1] - k.cy) / k.fy, z};
    double r[3];
    qrot(Twc1.r, ptc, r);
    for (int i = 0; i < 3; ++i) pt[i] = r[i] + Twc1.t[i];
    st = 2;
  } else if (bStereo2 && cps2 < cps1) {
    const double z = dp2;
    const double ptc[3] = {z * (k2[0] - k.cx) / k.fx, z * (k2[1] - k.cy) / k.fy, z};
    double r[3];
    qrot(Twc2.r, ptc, r);
    for (int i = 0; i < 3; ++i) pt[i] = r[i] + Twc2.t[i];
    st = 2;
  }
  stage[n] = st;
  for (int i = 0; i < 3; ++i) x3d[(size_t)n * 3 + i] = pt[i];
  b1[(size_t)n * 3] = k1[0];
  b1[(size_t)n * 3 + 1] = k1[1];
  b1[(size_t)n * 3 + 2] = dp1 > 0 ? k1[2] : -1.0;
  b2[(size_t)n * 3] = k2[0];
  b2[(size_t)n * 3 + 1] = k2[1];
  b2[(size_t)n * 3 + 2] = dp2 > 0 ? k2[2] : -1.0;
}

// stage 3: reprojection and scale-consistency checks (:366-404) -> MapPoint type
__global__ void k_tri_post(TriK k, int N, const double* __restrict__ pose1, const double* __restrict__ uvr1,
                           const int32_t* __restrict__ oct1, const double* __restrict__ pose2,
                           const double* __restrict__ uvr2, const int32_t* __restrict__ oct2,
                           const int32_t* __restrict__ stage, double* __restrict__ x3d, int32_t* __restrict__ comp,
                           int32_t* __restrict__ type) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int st = stage[n];
  if (st == 0) {
    type[n] = 0;
    comp[n] = -1;
    for (int i = 0; i < 3; ++i) x3d[(size_t)n * 3 + i] = 0.0;
    return;
  }
  const SE3 Tcw1 = se3_load(pose1 + (size_t)n * 7), Tcw2 = se3_load(pose2 + (size_t)n * 7);
  const SE3 Twc1 = se3_inverse(Tcw1), Twc2 = se3_inverse(Tcw2);
  const double pt[3] = {x3d[(size_t)n * 3], x3d[(size_t)n * 3 + 1], x3d[(size_t)n * 3 + 2]};
  const double* k1 = uvr1 + (size_t)n * 3;
  const double* k2 = uvr2 + (size_t)n * 3;
  const float ur1 = (float)k1[2], ur2 = (float)k2[2];
  const bool bStereo1 = ur1 >= 0, bStereo2 = ur2 >= 0;
  auto project = [&](const SE3& T, double* o) {
    double r[3];
    qrot(T.r, pt, r);
    const double pc[3] = {r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2]};
    if (pc[2] < 0.0) return false;
    const double rz = 1.0 / pc[2];
    const double u = k.fx * (pc[0] * rz) + k.cx, v = k.fy * (pc[1] * rz) + k.cy;
    if (!(u >= 0.0 && v >= 0.0 && u < (double)k.width && v < (double)k.height && pc[2] > 0.0)) return false;
    o[0] = u;
    o[1] = v;
    o[2] = u - k.mbf / pc[2];
    return true;
  };
  auto kp_error = [](const double* kp, float ur, const double* o) {
    const double e2 = (kp[0] - o[0]) * (kp[0] - o[0]) + (kp[1] - o[1]) * (kp[1] - o[1]);
    if (ur < 0.0f) return e2;
    const double d2 = (double)ur - o[2];
    return e2 + d2 * d2;
  };
  int ty = 0;
  double p1[3], p2[3];
  if (project(Tcw1, p1) && project(Tcw2, p2)) {
    const float s2 = k.sigma2[oct1[n] & 7];  // both checks use kp1's octave (:370-391)
    if (!(kp_error(k1, ur1, p1) > (bStereo1 ? 7.8 : 5.991) * s2) && !(kp_error(k2, ur2, p2) > (bStereo2 ? 7.8 : 5.991) * s2)) {
      const double a[3] = {pt[0] - Twc1.t[0], pt[1] - Twc1.t[1], pt[2] - Twc1.t[2]};
      const double b[3] = {pt[0] - Twc2.t[0], pt[1] - Twc2.t[1], pt[2] - Twc2.t[2]};
      const float dist1 = (float)sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
      const float dist2 = (float)sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
      if (!(dist1 <= 1.1920929e-07f || dist2 <= 1.1920929e-07f)) {
        const float ratio_dist = dist2 / dist1;
        const float ratio_octave = k.sf[oct1[n] & 7] / k.sf[oct2[n] & 7];
        if (!(ratio_dist * k.ratio_factor < ratio_octave || ratio_dist > ratio_octave * k.ratio_factor))
          ty = st == 1 ? (comp[n] >= 0 ? 2 : 1) : (comp[n] >= 0 ? 4 : 3);
      }
    }
  }
  type[n] = ty;
}

PtK make_ptk(const gl_camera* cam, const gl_params* prm) {
  PtK k;
  k.fx = cam->fx;
  k.fy = cam->fy;
  k.cx = cam->cx;
  k.cy = cam->cy;
  k.bf = cam->bf;
  for (int i = 0; i < 8; ++i) k.s2inv[i] = (double)prm->sigma2_inv[i];
  k.tri_lambda2 = (double)prm->tri_lambda2;
  k.str_thresh = (double)(prm->tri_str_thresh * prm->tri_lambda2);  // float product (gmmloc_opt.cpp:337)
  k.check_str = prm->tri_check_str_chi2;
  return k;
}

}  // namespace

extern "C" {

int gl_optimize_point(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int N,
                      const double* pts_dev, const double* uvr_dev, const int32_t* octave_dev,
                      const double* pose_dev, const int32_t* comp_dev, const double* proj_z2_dev, uint8_t* res_dev,
                      double* chi2_proj_dev, double* chi2_str_dev, double* pt_est_dev) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  if (N == 0) return GL_OK;
  GL_REQUIRE(N > 0 && pts_dev && uvr_dev && octave_dev && pose_dev && comp_dev && proj_z2_dev && res_dev &&
                 chi2_proj_dev && chi2_str_dev && pt_est_dev,
             "bad N / null buffer");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  k_optimize_point<<<(N + 127) / 128, 128, 0, c->stream>>>(make_ptk(cam, prm), N, g->K, g->rec12, g->axis, pts_dev, uvr_dev,
                                                           octave_dev, pose_dev, comp_dev, proj_z2_dev, res_dev,
                                                           chi2_proj_dev, chi2_str_dev, pt_est_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_check_map_association(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B,
                             int N, const double* pose_dev, double* pts_dev, const double* uvr_dev,
                             const int32_t* octave_dev, const int32_t* cand_dev, const int32_t* ncand_dev, int k,
                             int32_t* out_comp_dev) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  if (B == 0 || N == 0) return GL_OK;
  GL_REQUIRE(B > 0 && N > 0 && k >= 1 && k <= 8, "bad B / N / k");
  GL_REQUIRE(pose_dev && pts_dev && uvr_dev && octave_dev && cand_dev && ncand_dev && out_comp_dev, "null buffer");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  const int total = B * N;
  k_check_map_association<<<(unsigned)(((size_t)total * CMA_LANES + 255) / 256), 256, 0, c->stream>>>(make_ptk(cam, prm), B, N, g->K, g->rec12, g->axis,
                                                                   g->flags, g->nbs_ptr, g->nbs_idx, pose_dev, pts_dev,
                                                                   uvr_dev, octave_dev, cand_dev, ncand_dev, k,
                                                                   out_comp_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_optimize_triangulation(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int N,
                              double* x3d_dev, const double* pose1_dev, const double* uvr1_dev,
                              const int32_t* oct1_dev, const double* pose2_dev, const double* uvr2_dev,
                              const int32_t* oct2_dev, const int32_t* cand1_dev, const int32_t* n1_dev,
                              const int32_t* cand2_dev, const int32_t* n2_dev, int k, int32_t* out_comp_dev) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  if (N == 0) return GL_OK;
  GL_REQUIRE(N > 0 && k >= 1 && k <= 8, "bad N / k");
  GL_REQUIRE(x3d_dev && pose1_dev && uvr1_dev && oct1_dev && pose2_dev && uvr2_dev && oct2_dev && cand1_dev &&
                 n1_dev && cand2_dev && n2_dev && out_comp_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  (void)oct2_dev;  // kp2's octave is unused by the reference (it re-uses kp1's sigma, :132,135)
  k_optimize_triangulation<<<(int)(((size_t)N * TRI_LANES + 255) / 256), 256, 0, c->stream>>>(make_ptk(cam, prm), N, g->rec12, g->axis, g->flags,
                                                                x3d_dev, pose1_dev, uvr1_dev, oct1_dev, pose2_dev,
                                                                uvr2_dev, cand1_dev, n1_dev, cand2_dev, n2_dev, k,
                                                                out_comp_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}


int gl_create_map_points(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, float scale_factor,
                         int N, const double* pose1_dev, const double* uvr1_dev, const float* depth1_dev,
                         const int32_t* oct1_dev, const double* pose2_dev, const double* uvr2_dev,
                         const float* depth2_dev, const int32_t* oct2_dev, const int32_t* cand1_dev,
                         const int32_t* n1_dev, const int32_t* cand2_dev, const int32_t* n2_dev, int k, double* x3d_dev,
                         int32_t* type_dev, int32_t* comp_dev) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  if (N == 0) return GL_OK;
  GL_REQUIRE(N > 0 && k >= 1 && k <= 8, "bad N / k");
  GL_REQUIRE(cam->width > 0 && cam->height > 0, "camera width / height not set");
  GL_REQUIRE(pose1_dev && uvr1_dev && depth1_dev && oct1_dev && pose2_dev && uvr2_dev && depth2_dev && oct2_dev &&
                 cand1_dev && n1_dev && cand2_dev && n2_dev && x3d_dev && type_dev && comp_dev,
             "null buffer");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  void* scratch = nullptr;
  int rc = gl::ctx_scratch(c, (size_t)N * (2 * 24 + 4) + 64, &scratch);
  if (rc != GL_OK) return rc;
  double* b1 = (double*)scratch;
  double* b2 = b1 + (size_t)N * 3;
  int32_t* stage = (int32_t*)(b2 + (size_t)N * 3);
  TriK t;
  t.fx = cam->fx;
  t.fy = cam->fy;
  t.cx = cam->cx;
  t.cy = cam->cy;
  t.ffx = (float)cam->fx;
  t.ffy = (float)cam->fy;
  t.fcx = (float)cam->cx;
  t.fcy = (float)cam->cy;
  t.invfx = 1.0f / t.ffx;
  t.invfy = 1.0f / t.ffy;
  t.mbf = (float)cam->bf;
  t.mb = t.mbf / t.ffx;
  t.ratio_factor = 1.5f * scale_factor;
  t.sf[0] = 1.0f;
  t.sigma2[0] = 1.0f;
  for (int i = 1; i < 8; ++i) {
    t.sf[i] = t.sf[i - 1] * scale_factor;
    t.sigma2[i] = t.sf[i] * t.sf[i];
  }
  t.width = cam->width;
  t.height = cam->height;
  const int grid = (N + 63) / 64;
  k_tri_pre<<<grid, 64, 0, c->stream>>>(t, N, pose1_dev, uvr1_dev, depth1_dev, pose2_dev, uvr2_dev, depth2_dev, x3d_dev, stage,
                                        b1, b2);
  GL_HIP(hipGetLastError());
  k_optimize_triangulation<<<(int)(((size_t)N * TRI_LANES + 255) / 256), 256, 0, c->stream>>>(make_ptk(cam, prm), N, g->rec12, g->axis, g->flags, x3d_dev, pose1_dev,
                                                       b1, oct1_dev, pose2_dev, b2, cand1_dev, n1_dev, cand2_dev, n2_dev, k,
                                                       comp_dev);
  GL_HIP(hipGetLastError());
  k_tri_post<<<grid, 64, 0, c->stream>>>(t, N, pose1_dev, uvr1_dev, oct1_dev, pose2_dev, uvr2_dev, oct2_dev, stage, x3d_dev,
                                         comp_dev, type_dev);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

}  // extern "C"
