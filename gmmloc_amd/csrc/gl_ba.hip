// Structure-constrained refinement: Localization::jointOptimization
// (localization_opt.cpp:456-925) restricted to ONE free pose and L free,
// marginalised map points, each with one reprojection edge (mono / stereo,
// Huber) and at most one GMM edge (EdgePt2GaussianDeg x ba_lambda2, or
// EdgePt2Gaussian), optional EdgeSE3QuatPrior -- the north-star "Gauss-Newton /
// Schur reduction of reprojection + point-to-ellipsoid residuals".
//
// One persistent workgroup per frame, everything on-chip:
//   pass A (state, lambda): per point, in the CAMERA frame q = R p + t
//       A  = w Jpi^T Jpi,  a = w Jpi^T e            (reprojection, w = rho' / sigma^2)
//       Hc = R Hg R^T,     bc = R bg                (GMM edge)
//       D  = A + Hc + lambda I  -> L Delta L^T (ldl3_factor; g2o inverts the block by cofactors, which is noise for a badly scaled one)
//       Schur:  S += G^T (A - A D^-1 A) G,  g += G^T (a - A D^-1 (a + bc)),  G = [-[q]x | I]
//               (evaluated as M D^-1 A and M u - bc, M = Hc + lambda I: no cancellation)
//     (the orthogonal change of variables eps = R dp leaves lambda I and
//     computeScale() invariant), then ONE deterministic 28-value workgroup
//     reduction (gld::block_reduce), 6x6 LDL^T and exp() redundantly per thread;
//   pass B: back-substitution eps = D^-1 (b - A G dx), trial state, new chi2.
// Control flow = g2o OptimizationAlgorithmLevenberg::solve / SparseOptimizer::
// optimize + the 5 / gate GMM / 5 / gate reprojection / 40 schedule
// (localization_opt.cpp:770-828), incl. stale e->chi2() semantics.
#include <cstdlib>
#include <cstring>

#include "gl_ba_common.hpp"

#pragma clang fp contract(fast)

using namespace gld;
using namespace glba;

namespace {

// factors of the damped point block D (ldl3_factor: f), u = D^-1 b
GL_DEV void point_solve(const PtLin& o, double lambda, double* f, double* b, double* u) {
  double D[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) D[i] = o.A[i] + o.Hc[i];
  D[0] += lambda;
  D[3] += lambda;
  D[5] += lambda;
  ldl3_factor(D, f);
#pragma unroll
  for (int i = 0; i < 3; ++i) b[i] = o.a[i] + o.bc[i];
  ldl3_solve(f, b, u);
}

// acc[0..20] += upper(G^T C G), acc[21..26] += G^T c ; G = [-[q]x | I], C symmetric (full 3x3 given)
GL_DEV void accum_pose(const double* q, const double* Cf, const double* c, double* acc) {
  // M = Q C : column j of M = q x C[:,j]  -> rows m_i
  double M[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double col[3] = {Cf[0 * 3 + j], Cf[1 * 3 + j], Cf[2 * 3 + j]};
    double r[3];
    cross(q, col, r);
    M[0 * 3 + j] = r[0];
    M[1 * 3 + j] = r[1];
    M[2 * 3 + j] = r[2];
  }
  // TL row i = q x (row i of M)   (= -M Q)
  double TL[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double r[3];
    cross(q, &M[i * 3], r);
    TL[i * 3] = r[0];
    TL[i * 3 + 1] = r[1];
    TL[i * 3 + 2] = r[2];
  }
  // upper-triangle order of the 6x6: rows 0..2 -> [TL(i, i..2), TR(i, 0..2)], rows 3..5 -> [BR(i, i..2)]
  acc[0] += TL[0];
  acc[1] += TL[1];
  acc[2] += TL[2];
  acc[3] += M[0];
  acc[4] += M[1];
  acc[5] += M[2];
  acc[6] += TL[4];
  acc[7] += TL[5];
  acc[8] += M[3];
  acc[9] += M[4];
  acc[10] += M[5];
  acc[11] += TL[8];
  acc[12] += M[6];
  acc[13] += M[7];
  acc[14] += M[8];
  acc[15] += Cf[0];
  acc[16] += Cf[1];
  acc[17] += Cf[2];
  acc[18] += Cf[4];
  acc[19] += Cf[5];
  acc[20] += Cf[8];
  double qc[3];
  cross(q, c, qc);
  acc[21] += qc[0];
  acc[22] += qc[1];
  acc[23] += qc[2];
  acc[24] += c[0];
  acc[25] += c[1];
  acc[26] += c[2];
}

GL_DEV void unpack6(const double* acc, double* H) {
  int qi = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) {
      H[i * 6 + j] = acc[qi];
      H[j * 6 + i] = acc[qi];
      ++qi;
    }
}

// per-frame state in global memory
struct FrameView {
  int L;
  double* p;         // L x 3 current points (in/out)
  double* pn;        // L x 3 trial points (scratch)
  const double* obs; // L x 3
  const int32_t* oct;
  const int32_t* assoc;
  double* chi_r;     // stale e->chi2() of the reprojection edges
  uint8_t* lev;      // bit0: reprojection edge level, bit1: GMM edge level
};

// SparseOptimizer::optimize(iters) with the LM algorithm; returns cjIterations (-1: nothing active)
GL_DEV int ba_optimize(const BaK& k, const GmmDev& gm, FrameView& fv, SE3& T, bool pose_fixed, bool has_prior,
                       const SE3& prior_inv, bool robust, int iters, double* red) {
  double acc[32];
  // active census (initializeOptimization(0))
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  for (int l = threadIdx.x; l < fv.L; l += T_BA) {
    if (fv.oct[l] < 0) continue;
    const bool ar = !(fv.lev[l] & 1), ag = fv.assoc[l] >= 0 && !(fv.lev[l] & 2);
    if (ar) acc[0] += 1.0;
    if (ar || ag) acc[1] += 1.0;
  }
  block_reduce<2, NW_BA>(acc, red);
  const bool pose_active = !pose_fixed && ((int)acc[0] > 0 || has_prior);
  const bool any_point = (int)acc[1] > 0;
  if (!pose_active && !any_point) return -1;

  double lambda = 0.0, ni = 2.0;
  int cj = 0;
  for (int it = 0; it < iters; ++it) {
    double rho = 0.0;
    int qmax = 0;
    double currentChi = 0.0;
    double R[9];
    qtoR(T.r, R);
    if (it == 0) {  // computeLambdaInit: tau * max |H(j,j)| over pose and (world-frame) landmark blocks
      double md = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0;
      for (int l = threadIdx.x; l < fv.L; l += T_BA) {
        const int oc = fv.oct[l];
        if (oc < 0) continue;
        const bool ar = !(fv.lev[l] & 1), ag = !(fv.lev[l] & 2);
        GmmRef g;
        load_gmm(fv.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
        if (!(ar || (ag && g.has))) continue;
        PtLin o;
        lin_point(k, R, T.t, fv.p + (size_t)l * 3, fv.obs + (size_t)l * 3, oc, g, ar, ag, robust, o);
        double Hs[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) Hs[i] = o.A[i] + o.Hc[i];
        // diag of R^T Hs R
        const double Hf[9] = {Hs[0], Hs[1], Hs[2], Hs[1], Hs[3], Hs[4], Hs[2], Hs[4], Hs[5]};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double s = 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) s += R[a * 3 + j] * Hf[a * 3 + b] * R[b * 3 + j];
          md = fmax(md, fabs(s));
        }
        if (pose_active && o.act_r) {
          double Af[9];
          const double zero[3] = {0, 0, 0};
          Af[0] = o.A[0]; Af[1] = o.A[1]; Af[2] = o.A[2]; Af[3] = o.A[1]; Af[4] = o.A[3]; Af[5] = o.A[4];
          Af[6] = o.A[2]; Af[7] = o.A[4]; Af[8] = o.A[5];
          accum_pose(o.q, Af, zero, acc);
        }
      }
      block_reduce<21, NW_BA>(acc, red);
      if (pose_active) {
        double Hp[36], bp[6] = {0, 0, 0, 0, 0, 0};
        unpack6(acc, Hp);
        if (has_prior) prior_terms(prior_inv, T, true, Hp, bp);
        double mdp = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) mdp = fmax(mdp, fabs(Hp[i * 6 + i]));
        md = fmax(md, mdp);
      }
      md = block_max(md, red);
      lambda = 1e-5 * md;
      ni = 2.0;
    }
    do {
      // ---- pass A: linearise at (T, p), Schur-reduce with the current lambda -------------
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0;
      for (int l = threadIdx.x; l < fv.L; l += T_BA) {
        const int oc = fv.oct[l];
        if (oc < 0) continue;
        const bool ar = !(fv.lev[l] & 1), ag = !(fv.lev[l] & 2);
        GmmRef g;
        load_gmm(fv.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
        if (!(ar || (ag && g.has))) continue;
        PtLin o;
        lin_point(k, R, T.t, fv.p + (size_t)l * 3, fv.obs + (size_t)l * 3, oc, g, ar, ag, robust, o);
        if (o.act_r) fv.chi_r[l] = o.chi_r;  // computeActiveErrors
        acc[27] += o.rho0_r + o.chi_g;
        if (pose_active && o.act_r) {
          double Df[6], b[3], u[3];
          point_solve(o, lambda, Df, b, u);
          // A - A D^-1 A = M D^-1 A and a - A u = M u - bc with M = D - A = Hc + lambda I: products instead of the
          // subtraction, which loses most of its digits for a point held by its reprojection alone (gl_ba_fast_impl.hpp)
          double AD[9], Cf[9], c[3];
          {  // A D^-1: row r = D^-1 (row r of A)
            const double Af[9] = {o.A[0], o.A[1], o.A[2], o.A[1], o.A[3], o.A[4], o.A[2], o.A[4], o.A[5]};
#pragma unroll
            for (int r = 0; r < 3; ++r) ldl3_solve(Df, Af + r * 3, AD + r * 3);
          }
          const double Mf[9] = {o.Hc[0] + lambda, o.Hc[1], o.Hc[2], o.Hc[1], o.Hc[3] + lambda, o.Hc[4], o.Hc[2], o.Hc[4], o.Hc[5] + lambda};
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Cf[i * 3 + j] = Mf[i * 3] * AD[j * 3] + Mf[i * 3 + 1] * AD[j * 3 + 1] + Mf[i * 3 + 2] * AD[j * 3 + 2];
#pragma unroll
          for (int i = 0; i < 3; ++i) c[i] = (Mf[i * 3] * u[0] + Mf[i * 3 + 1] * u[1] + Mf[i * 3 + 2] * u[2]) - o.bc[i];
          {  // the product is symmetric in exact arithmetic only: the pose blocks get the mean of its two triangles
            const double s01 = 0.5 * (Cf[1] + Cf[3]), s02 = 0.5 * (Cf[2] + Cf[6]), s12 = 0.5 * (Cf[5] + Cf[7]);
            Cf[1] = Cf[3] = s01;
            Cf[2] = Cf[6] = s02;
            Cf[5] = Cf[7] = s12;
          }
          accum_pose(o.q, Cf, c, acc);
        }
      }
      block_reduce<28, NW_BA>(acc, red);
      double S[36], gvec[6], dx[6] = {0, 0, 0, 0, 0, 0};
      unpack6(acc, S);
#pragma unroll
      for (int i = 0; i < 6; ++i) gvec[i] = acc[21 + i];
      double chiA = acc[27];
      double bp_prior[6] = {0, 0, 0, 0, 0, 0};
      if (has_prior && pose_active) {
        chiA += prior_terms(prior_inv, T, true, S, bp_prior);
#pragma unroll
        for (int i = 0; i < 6; ++i) gvec[i] += bp_prior[i];
      }
      if (qmax == 0) currentChi = chiA;  // activeRobustChi2() at the start of solve()
      bool ok2 = true;
      if (pose_active) {
#pragma unroll
        for (int i = 0; i < 6; ++i) S[i * 6 + i] += lambda;
        ok2 = ldlt_solve<6>(S, gvec, dx, false);
      }
      // ---- pass B: back-substitute, trial state, new errors --------------------------------
      SE3 Tn = T;
      if (pose_active && ok2) Tn = se3_mul(se3_exp(dx), T);
      double Rn[9];
      qtoR(Tn.r, Rn);
      double tempChi = 1.7976931348623157e308, scale = 0.0, bdotx = 0.0;
      {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = 0.0;
        for (int l = threadIdx.x; l < fv.L; l += T_BA) {
          const int oc = fv.oct[l];
          if (oc < 0) continue;
          const bool ar = !(fv.lev[l] & 1), ag = !(fv.lev[l] & 2);
          GmmRef g;
          load_gmm(fv.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
          const double* p = fv.p + (size_t)l * 3;
          if (!(ar || (ag && g.has))) continue;
          PtLin o;
          lin_point(k, R, T.t, p, fv.obs + (size_t)l * 3, oc, g, ar, ag, robust, o);
          double Df[6], b[3], u[3];
          point_solve(o, lambda, Df, b, u);
          // eps = D^-1 (b - A (w x q + v))
          double gd[3], Agd[3], rhs[3], eps[3];
          cross(dx, o.q, gd);
          gd[0] += dx[3];
          gd[1] += dx[4];
          gd[2] += dx[5];
          if (o.act_r && pose_active) {
            sym3_mul_vec(o.A, gd, Agd);
            acc[2] += gd[0] * o.a[0] + gd[1] * o.a[1] + gd[2] * o.a[2];  // (G dx) . a
          } else {
            Agd[0] = Agd[1] = Agd[2] = 0.0;
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) rhs[i] = b[i] - Agd[i];
          ldl3_solve(Df, rhs, eps);
          // computeScale(): x (lambda x + b), landmark part (rotation invariant)
          acc[0] += eps[0] * (lambda * eps[0] + b[0]) + eps[1] * (lambda * eps[1] + b[1]) + eps[2] * (lambda * eps[2] + b[2]);
          double pn[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) pn[i] = p[i] + (R[0 * 3 + i] * eps[0] + R[1 * 3 + i] * eps[1] + R[2 * 3 + i] * eps[2]);
#pragma unroll
          for (int i = 0; i < 3; ++i) fv.pn[(size_t)l * 3 + i] = pn[i];
          // errors at the trial state
          if (o.act_r) {
            double qn[3], e[3], iz;
#pragma unroll
            for (int i = 0; i < 3; ++i) qn[i] = Rn[i * 3] * pn[0] + Rn[i * 3 + 1] * pn[1] + Rn[i * 3 + 2] * pn[2] + Tn.t[i];
            const bool stereo = !(fv.obs[(size_t)l * 3 + 2] < 0);
            const double c2 = reproj_err(k, qn, fv.obs + (size_t)l * 3, stereo, k.s2inv[oc], e, iz);
            fv.chi_r[l] = c2;
            double r0 = c2, r1;
            if (robust) huber(c2, stereo ? k.delta_stereo : k.delta_mono, r0, r1);
            acc[1] += r0;
          }
          if (o.act_g) acc[1] += gmm_chi2(k, g, pn);
        }
        block_reduce<3, NW_BA>(acc, red);
        scale = acc[0];
        bdotx = acc[2];
        if (ok2) tempChi = acc[1] + ((has_prior && pose_active) ? prior_terms(prior_inv, Tn, false, nullptr, nullptr) : 0.0);
      }
      // pose block of computeScale(): x_p . (lambda x_p + b_p), b_p = sum G^T a (+ prior)
      double scale_p = 0.0;
      if (pose_active) {
        scale_p = bdotx;
#pragma unroll
        for (int i = 0; i < 6; ++i) scale_p += dx[i] * (lambda * dx[i] + bp_prior[i]);
      }
      scale += scale_p + 1e-3;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        const double uu = 2 * rho - 1;
        double alpha = 1. - uu * uu * uu;
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        T = Tn;
        qtoR(T.r, R);
        // p <- pn for active points
        for (int l = threadIdx.x; l < fv.L; l += T_BA) {
          if (fv.oct[l] < 0) continue;
          const bool ar = !(fv.lev[l] & 1), ag = fv.assoc[l] >= 0 && !(fv.lev[l] & 2);
          if (!(ar || ag)) continue;
#pragma unroll
          for (int i = 0; i < 3; ++i) fv.p[(size_t)l * 3 + i] = fv.pn[(size_t)l * 3 + i];
        }
      } else {
        lambda *= ni;
        ni *= 2;
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    ++cj;
    if (qmax == 10 || rho == 0) break;
  }
  return cj;
}

__global__ __launch_bounds__(T_BA) void k_ba1(BaK k, GmmDev gm, int B, int L, double* __restrict__ pose_io,
                                              const uint8_t* __restrict__ has_prior_all, double* __restrict__ pts_io,
                                              const double* __restrict__ obs_all, const int32_t* __restrict__ oct_all,
                                              int32_t* __restrict__ assoc_all, const double* __restrict__ d2_all,
                                              double* __restrict__ pn_all, double* __restrict__ chi_all,
                                              uint8_t* __restrict__ lev_all, uint8_t* __restrict__ dropped_all,
                                              uint8_t* __restrict__ erase_all, int32_t* __restrict__ iters_out) {
  __shared__ double red[NW_BA * 32];
  const int f = blockIdx.x;
  if (f >= B) return;
  FrameView fv;
  fv.L = L;
  fv.p = pts_io + (size_t)f * L * 3;
  fv.pn = pn_all + (size_t)f * L * 3;
  fv.obs = obs_all + (size_t)f * L * 3;
  fv.oct = oct_all + (size_t)f * L;
  fv.assoc = assoc_all + (size_t)f * L;
  fv.chi_r = chi_all + (size_t)f * L;
  fv.lev = lev_all + (size_t)f * L;

  // gl_track_frames: association gate chi2 <= 9 (checkMapAssociation, gmmloc_opt.cpp:230-232)
  for (int l = threadIdx.x; l < L; l += T_BA) {
    fv.lev[l] = 0;
    fv.chi_r[l] = 0.0;
    if (d2_all && k.gate_chi2 >= 0) {
      if (!(d2_all[(size_t)f * L + l] <= k.gate_chi2)) assoc_all[(size_t)f * L + l] = -1;
    }
    if (fv.oct[l] < 0) assoc_all[(size_t)f * L + l] = -1;
  }
  __syncthreads();

  SE3 T = se3_load(pose_io + (size_t)f * 7);
  const bool prior_flag = has_prior_all ? (has_prior_all[f] != 0) : false;
  const bool has_prior = prior_flag && k.first_as_prior;
  const bool pose_fixed = prior_flag && !k.first_as_prior;  // vSE3->setFixed(idx_ == 0)
  const SE3 prior_inv = se3_inverse(T);                     // e->setMeasurement(kfi->getTcw())

  // optimize(5)  (:770-771)
  ba_optimize(k, gm, fv, T, pose_fixed, has_prior, prior_inv, true, 5, red);
  __syncthreads();
  // gate the degenerate GMM edges on a FRESH error (:773-786)
  for (int l = threadIdx.x; l < L; l += T_BA) {
    if (fv.oct[l] < 0) continue;
    GmmRef g;
    load_gmm(fv.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
    if (g.has && g.deg && gmm_chi2(k, g, fv.p + (size_t)l * 3) > k.str_thresh) fv.lev[l] |= 2;
  }
  __syncthreads();
  ba_optimize(k, gm, fv, T, pose_fixed, has_prior, prior_inv, true, 5, red);  // :788-789
  __syncthreads();
  // gate the reprojection edges: STALE chi2, fresh depth test; robust kernels off (:799-825)
  {
    double R[9];
    qtoR(T.r, R);
    for (int l = threadIdx.x; l < L; l += T_BA) {
      if (fv.oct[l] < 0) continue;
      const double* p = fv.p + (size_t)l * 3;
      const double z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + T.t[2];
      const bool stereo = !(fv.obs[(size_t)l * 3 + 2] < 0);
      if (fv.chi_r[l] > (stereo ? 7.815 : 5.991) || !(z > 0.0)) fv.lev[l] |= 1;
    }
  }
  __syncthreads();
  const int it3 = ba_optimize(k, gm, fv, T, pose_fixed, has_prior, prior_inv, false, 40, red);  // :827-828
  __syncthreads();
  // outputs (:837-879)
  {
    double R[9];
    qtoR(T.r, R);
    for (int l = threadIdx.x; l < L; l += T_BA) {
      uint8_t dr = 0, er = 0;
      if (fv.oct[l] >= 0) {
        GmmRef g;
        load_gmm(fv.assoc[l], gm.axis, gm.rec12, gm.sqrt_info, gm.flags, g);
        const double* p = fv.p + (size_t)l * 3;
        if (g.has && g.deg && gmm_chi2(k, g, p) > k.str_thresh) dr = 1;
        const double z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + T.t[2];
        const bool stereo = !(fv.obs[(size_t)l * 3 + 2] < 0);
        if (fv.chi_r[l] > (stereo ? 7.815 : 5.991) || !(z > 0.0)) er = 1;
      }
      if (dropped_all) dropped_all[(size_t)f * L + l] = dr;
      if (erase_all) erase_all[(size_t)f * L + l] = er;
      if (!dropped_all && dr) assoc_all[(size_t)f * L + l] = -1;  // gl_track_frames: final association
    }
  }
  if (threadIdx.x == 0) {
    if (!pose_fixed) se3_store(T, pose_io + (size_t)f * 7);
    if (iters_out) iters_out[f] = it3;
  }
}

}  // namespace

namespace gl {

bool ba1_fast_supported(int L);
int launch_ba1_fast(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int L, double* pose,
                    double* pts, const double* obs, const int32_t* oct, int32_t* assoc, const double* d2, double gate,
                    uint8_t* dropped, uint8_t* erase, int32_t* iters, void* scratch, const uint8_t* prior, const TrackFixed* fixed);

// gl_track_frames_anchored with fixed observer key-frames (gl_ba_gen.hip: the general kernel does those)
int track_frames_fixed(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B, int M,
                       double* pose_dev, double* Xw_dev, const double* obs_dev, const int32_t* octave_dev, int32_t* assoc_dev,
                       double* d2_dev, const gl_track_anchor* anchor);

// Single-free-pose jointOptimization for B frames; assoc in/out; scratch from ctx.
int launch_ba1(Ctx* c, const Gmm* g, const gl_camera* cam, const gl_params* prm, int B, int L, double* pose,
               const uint8_t* has_prior, double* pts, const double* obs, const int32_t* oct, int32_t* assoc,
               const double* d2, double gate, uint8_t* dropped, uint8_t* erase, int32_t* iters, void* scratch) {
  BaK k = make_bak(cam, prm, gate);
  GmmDev gm{g->rec12, g->axis, g->sqrt_info, g->hgw, g->flags, g->plane4};
  char* s = (char*)scratch;
  double* pn = (double*)s;
  s += (size_t)B * L * 24;
  double* chi = (double*)s;
  s += (size_t)B * L * 8;
  uint8_t* lev = (uint8_t*)s;
  {
    TimerScope ts(c, GL_TIMER_BA);
    k_ba1<<<B, T_BA, 0, c->stream>>>(k, gm, B, L, pose, has_prior, pts, obs, oct, assoc, d2, pn, chi, lev, dropped, erase,
                                     iters);
  }
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// fast path: plane records (32 B) + normalised observations (24 B) + permutation, flags, gated association (12 B) per
// point; general kernel: trial points, chi2, levels (33 B); + per frame: 2 x 4 x 32 x 2 exchange words of the latency
// shape, 12 doubles of the prior edge's inverse measurement, and for small batches the staging area of the latency shape's
// results (points 24 B + association 4 B per point, pose 64 B per frame) (gl_ba_fast.hip)
// F > 0 (fixed observer key-frames on the on-chip path): behind all that, per key-frame its pose {R, t} and per point and
// key-frame the normalised observation, the octave word and the stale chi2 (96 + L x 36 bytes)
size_t ba1_scratch_bytes(int B, int L, int F) {
  size_t n = (size_t)B * L * 36 + (size_t)B * (8192 + 8 + 96) + 512;
  if ((size_t)B * ((L + 255) / 256) <= 1024) n += (size_t)B * L * 28 + (size_t)B * 64 + 64;  // a batch the latency shape may take: its staging area
  if (F > 0) n = ((n + 63) / 64) * 64 + (size_t)B * F * (96 + (size_t)L * 36) + 64;
  return n;
}

}  // namespace gl

static int track_frames_impl(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B, int M,
                             double* pose_dev, double* Xw_dev, const double* obs_dev, const int32_t* octave_dev,
                             int32_t* assoc_dev, double* d2_dev, const uint8_t* prior_dev, const gl::TrackFixed* fixed = nullptr) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  if (B == 0 || M == 0) return GL_OK;
  GL_REQUIRE(B > 0 && M > 0, "bad B / M");
  GL_REQUIRE(pose_dev && Xw_dev && obs_dev && octave_dev && assoc_dev, "null buffer");
  gl::Ctx* c = gl::C(ctx);
  gl::Gmm* g = gl::G(gmm);
  GL_HIP(hipSetDevice(c->device));
  const size_t n = (size_t)B * M;
  // scratch: [d2 (if the caller does not want it)] [ba1 scratch]; the association kernel's own
  // partial buffers come from the same context scratch, so carve everything from one block.
  // Layout: | assoc partials (used first, dead afterwards) ... reused by ba1 | d2 |
  void* scratch = nullptr;
  const size_t ba_bytes = gl::ba1_scratch_bytes(B, M, fixed ? fixed->F : 0);
  const bool use_grid = g->grid.enabled && c->opt.assoc_grid != 0;
  const size_t assoc_bytes =
      use_grid ? gl::assoc_index_scratch_bytes(g->K, (int)n, d2_dev != nullptr) : gl::assoc_scratch_bytes(g->K, (int)n);
  const size_t work = ba_bytes > assoc_bytes ? ba_bytes : assoc_bytes;
  int rc = gl::ctx_scratch(c, work + n * 8 + 64, &scratch);
  if (rc != GL_OK) return rc;
  double* d2 = d2_dev ? d2_dev : (double*)((char*)scratch + ((work + 63) / 64) * 64);
  // Only chi2 <= 9 survives the gate below, so the points the cell index cannot resolve (minimum
  // above 9) need their exact argmin only when the caller asked for the chi2 values.
  if (use_grid)
    rc = gl::launch_assoc_index(c, g, Xw_dev, (int)n, assoc_dev, d2, d2_dev != nullptr, scratch);
  else
    rc = gl::launch_assoc_brute(c, g, Xw_dev, (int)n, assoc_dev, d2);
  if (rc != GL_OK) return rc;
  // on-chip fast path (gl_ba_fast.hip) for M <= 2000; option ba_slow forces the general kernel
  if ((c->opt.ba_slow == 0 || fixed) && gl::ba1_fast_supported(M))
    return gl::launch_ba1_fast(c, g, cam, prm, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, assoc_dev, d2, 9.0,
                               nullptr, nullptr, nullptr, scratch, prior_dev, fixed);
  return gl::launch_ba1(c, g, cam, prm, B, M, pose_dev, prior_dev, Xw_dev, obs_dev, octave_dev, assoc_dev, d2, 9.0,
                        nullptr, nullptr, nullptr, scratch);
}

extern "C" int gl_track_frames(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B,
                               int M, double* pose_dev, double* Xw_dev, const double* obs_dev,
                               const int32_t* octave_dev, int32_t* assoc_dev, double* d2_dev) {
  return track_frames_impl(ctx, gmm, cam, prm, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, assoc_dev, d2_dev, nullptr);
}

extern "C" int gl_track_frames_anchored(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int B,
                                        int M, double* pose_dev, double* Xw_dev, const double* obs_dev,
                                        const int32_t* octave_dev, int32_t* assoc_dev, double* d2_dev,
                                        const gl_track_anchor* anchor) {
  GL_REQUIRE(anchor, "null anchor");
  GL_REQUIRE(anchor->F >= 0 && anchor->F <= GL_TRACK_MAX_FIXED, "bad number of fixed observer key-frames");
  if (anchor->F == 0)
    return track_frames_impl(ctx, gmm, cam, prm, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, assoc_dev, d2_dev, anchor->prior_dev);
  GL_REQUIRE(anchor->fixed_pose_dev && anchor->fixed_obs_dev && anchor->fixed_oct_dev, "null buffer of the fixed observer key-frames");
  // up to 4 fixed observers on frames of up to 2 000 points: on chip, inside the per-frame refine (gl_ba_fast.hip: the kFixed
  // instances); beyond that - or with option ba_fixed_pack = 1 (A/B, tests) - packed into flat problems for the general kernel
  if (ctx && anchor->F <= 4 && gl::ba1_fast_supported(M) && gl::C(ctx)->opt.ba_fixed_pack == 0 && gl::C(ctx)->opt.ba_slow == 0) {
    const gl::TrackFixed fx{anchor->F, anchor->fixed_pose_dev, anchor->fixed_obs_dev, anchor->fixed_oct_dev, anchor->fixed_erase_dev};
    return track_frames_impl(ctx, gmm, cam, prm, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, assoc_dev, d2_dev, anchor->prior_dev, &fx);
  }
  return gl::track_frames_fixed(ctx, gmm, cam, prm, B, M, pose_dev, Xw_dev, obs_dev, octave_dev, assoc_dev, d2_dev, anchor);
}

// One frame, host buffers in and out: | pose | Xw | assoc || obs | octave | staged through the context's page-locked
// buffer; only the part before || comes back.
static int track_frame_host_impl(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int M,
                                 double* pose_host, double* Xw_host, const double* obs_host, const int32_t* octave_host,
                                 int32_t* assoc_host, int with_prior) {
  GL_REQUIRE(ctx && gmm && cam && prm, "null argument");
  GL_REQUIRE(M >= 0, "bad M");
  if (M == 0) return GL_OK;
  GL_REQUIRE(pose_host && Xw_host && obs_host && octave_host && assoc_host, "null buffer");
  gl::Ctx* c = gl::C(ctx);
  GL_HIP(hipSetDevice(c->device));
  const size_t oX = 64, oA = oX + (size_t)M * 24, oO = oA + (((size_t)M * 4 + 7) / 8) * 8, oC = oO + (size_t)M * 24,
               total = oC + (size_t)M * 4;
  if (c->stage_bytes < total) {
    GL_HIP(hipStreamSynchronize(c->stream));
    if (c->dev_stage) GL_HIP(hipFree(c->dev_stage));
    if (c->host_stage) GL_HIP(hipHostFree(c->host_stage));
    c->dev_stage = c->host_stage = nullptr;
    c->stage_bytes = 0;
    const size_t want = total + total / 4 + 64;
    GL_HIP(hipMalloc(&c->dev_stage, want));
    GL_HIP(hipHostMalloc(&c->host_stage, want, hipHostMallocDefault));
    c->stage_bytes = want;
  }
  char* st = (char*)c->host_stage;
  char* dv = (char*)c->dev_stage;
  memcpy(st, pose_host, 56);
  st[56] = with_prior ? 1 : 0;  // (the 8 bytes between the pose and the points: the frame's anchor flag)
  memcpy(st + oX, Xw_host, (size_t)M * 24);
  memcpy(st + oO, obs_host, (size_t)M * 24);
  memcpy(st + oC, octave_host, (size_t)M * 4);
  GL_HIP(hipMemcpyAsync(dv, st, total, hipMemcpyHostToDevice, c->stream));
  const int rc = track_frames_impl(ctx, gmm, cam, prm, 1, M, (double*)dv, (double*)(dv + oX), (const double*)(dv + oO),
                                   (const int32_t*)(dv + oC), (int32_t*)(dv + oA), nullptr,
                                   with_prior ? (const uint8_t*)(dv + 56) : nullptr);
  if (rc != GL_OK) {
    (void)hipStreamSynchronize(c->stream);  // the copy above may still read the staging buffer: the next call rewrites it
    return rc;
  }
  GL_HIP(hipMemcpyAsync(st, dv, oO, hipMemcpyDeviceToHost, c->stream));
  GL_HIP(hipStreamSynchronize(c->stream));
  memcpy(pose_host, st, 56);
  memcpy(Xw_host, st + oX, (size_t)M * 24);
  memcpy(assoc_host, st + oA, (size_t)M * 4);
  return GL_OK;
}

extern "C" int gl_track_frame_host(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int M,
                                   double* pose_host, double* Xw_host, const double* obs_host, const int32_t* octave_host,
                                   int32_t* assoc_host) {
  return track_frame_host_impl(ctx, gmm, cam, prm, M, pose_host, Xw_host, obs_host, octave_host, assoc_host, 0);
}
extern "C" int gl_track_frame_host_anchored(gl_ctx_t* ctx, const gl_gmm_t* gmm, const gl_camera* cam, const gl_params* prm, int M,
                                            double* pose_host, double* Xw_host, const double* obs_host,
                                            const int32_t* octave_host, int32_t* assoc_host) {
  return track_frame_host_impl(ctx, gmm, cam, prm, M, pose_host, Xw_host, obs_host, octave_host, assoc_host, 1);
}
