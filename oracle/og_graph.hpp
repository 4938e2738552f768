// ============================================================================
// TEST INFRASTRUCTURE ONLY (see og_math.hpp header).
//
// "mini-g2o": a sequential fp64 restatement of the g2o semantics the reference
// hot path relies on: SparseOptimizer::
// initializeOptimization(level) / optimize(n), OptimizationAlgorithmGaussNewton
// and OptimizationAlgorithmLevenberg::solve, BlockSolver::buildSystem / solve
// (incl. the Schur complement over marginalised points), Huber robust kernel,
// the stock SBA edges and the reference's own edges
// (gmmloc/include/gmmloc/gmm/factors.h:17-140, gmmloc/src/gmm/factors.cpp:5-168).
// g2o itself is an un-vendored, unpinned dependency (hyhuang1995/g2o_catkin,
// .gmmloc_https.install:16-18) -> PARITY UNPINNED against real g2o; the
// reference call sites are tracking_opt.cpp:23-217, gmmloc_opt.cpp:268-330,
// localization_opt.cpp:45-171,533-828.
// ============================================================================
#pragma once
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
#include <vector>

#include "og_math.hpp"

namespace og {

enum VertexKind { V_SE3 = 0, V_XYZ = 1 };

struct Edge;

struct Vertex {
  int kind = V_SE3;
  bool fixed = false;
  bool marginalized = false;
  int hessianIndex = -1;
  int colInHessian = -1;  // offset inside its (pose|landmark) segment
  SE3 T;                  // V_SE3 estimate
  double p[3] = {0, 0, 0};  // V_XYZ estimate
  struct Saved {
    SE3 T;
    double p[3];
  };
  std::vector<Saved> stack;
  double A[36];  // own diagonal Hessian block, dim x dim row-major
  double b[6];
  std::vector<Edge*> edges;
  int dim() const { return kind == V_SE3 ? 6 : 3; }
  void clearQuadraticForm() {
    for (int i = 0; i < 6; ++i) b[i] = 0.0;
  }
  void push() {
    Saved s;
    s.T = T;
    s.p[0] = p[0];
    s.p[1] = p[1];
    s.p[2] = p[2];
    stack.push_back(s);
  }
  void pop() {
    T = stack.back().T;
    p[0] = stack.back().p[0];
    p[1] = stack.back().p[1];
    p[2] = stack.back().p[2];
    stack.pop_back();
  }
  void discardTop() { stack.pop_back(); }
  // VertexSE3Expmap::oplusImpl: T <- exp(u) * T ; VertexSBAPointXYZ: p += u
  void oplus(const double* u) {
    if (kind == V_SE3) {
      T = se3_mul(se3_exp(u), T);
    } else {
      p[0] += u[0];
      p[1] += u[1];
      p[2] += u[2];
    }
  }
};

enum EdgeKind {
  E_POSE_MONO = 0,    // g2o::EdgeSE3ProjectXYZOnlyPose        (unary pose, D=2)
  E_POSE_STEREO,      // g2o::EdgeStereoSE3ProjectXYZOnlyPose  (unary pose, D=3)
  E_XYZ_MONO,         // g2o::EdgeProjectXYZOnly        factors.cpp:66-107  (unary point, D=2)
  E_XYZ_STEREO,       // g2o::EdgeProjectXYZOnlyStereo  factors.cpp:116-166 (unary point, D=3)
  E_PT2GAUSS,         // g2o::EdgePt2Gaussian           factors.cpp:5-17    (unary point, D=3)
  E_PT2GAUSS_DEG,     // g2o::EdgePt2GaussianDeg        factors.cpp:55-64   (unary point, D=1)
  E_SE3_PRIOR,        // g2o::EdgeSE3QuatPrior          factors.cpp:19-53   (unary pose, D=6)
  E_BA_MONO,          // g2o::EdgeSE3ProjectXYZ         (binary: v0 point, v1 pose, D=2)
  E_BA_STEREO         // g2o::EdgeStereoSE3ProjectXYZ   (binary, D=3)
};

struct Edge {
  int kind = 0;
  int D = 0;
  int level = 0;
  int id = 0;
  bool robust = false;
  double delta = 0.0;  // Huber delta
  Vertex* v[2] = {nullptr, nullptr};
  int nv = 1;
  double info[36];  // D x D row-major
  double meas[3] = {0, 0, 0};
  double error[6] = {0, 0, 0, 0, 0, 0};
  // parameters
  double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
  double Xw[3] = {0, 0, 0};        // *OnlyPose edges
  Quat rot{0, 0, 0, 1};            // EdgeProjectXYZOnly*: rot_c_w_
  double t[3] = {0, 0, 0};         //                      t_c_w_
  double normal[3] = {0, 0, 0};    // EdgePt2GaussianDeg
  double mean[3] = {0, 0, 0};      // EdgePt2Gaussian{,Deg}
  double sqrt_info[9];             // EdgePt2Gaussian: comp_->sqrt_info_ (lower L)
  SE3 inv_meas;                    // EdgeSE3QuatPrior::_inverseMeasurement
  // linearisation
  double Ji[6 * 6];  // D x dim(v0)
  double Jj[6 * 6];  // D x dim(v1)

  void setInformationIdentity(double s) {
    for (int i = 0; i < D * D; ++i) info[i] = 0.0;
    for (int i = 0; i < D; ++i) info[i * D + i] = s;
  }
  double chi2() const {  // _error.dot(information() * _error)
    double s = 0.0;
    for (int i = 0; i < D; ++i) {
      double r = 0.0;
      for (int j = 0; j < D; ++j) r += info[i * D + j] * error[j];
      s += error[i] * r;
    }
    return s;
  }
  bool allVerticesFixed() const {
    for (int i = 0; i < nv; ++i)
      if (!v[i]->fixed) return false;
    return true;
  }

  void camProjectMono(const double* pc, double* uv) const {
    // project2d then * f + c   (g2o types_six_dof_expmap.cpp / factors.cpp:70-74)
    const double px = pc[0] / pc[2], py = pc[1] / pc[2];
    uv[0] = px * fx + cx;
    uv[1] = py * fy + cy;
  }
  void camProjectStereo(const double* pc, double* uvr) const {
    // factors.cpp:116-123 (number_t invz = 1.0f / z)
    const double invz = 1.0 / pc[2];
    uvr[0] = pc[0] * invz * fx + cx;
    uvr[1] = pc[1] * invz * fy + cy;
    uvr[2] = uvr[0] - bf * invz;
  }

  void computeError() {
    double pc[3], pr[3];
    switch (kind) {
      case E_POSE_MONO: {
        se3_map(v[0]->T, Xw, pc);
        camProjectMono(pc, pr);
        error[0] = meas[0] - pr[0];
        error[1] = meas[1] - pr[1];
      } break;
      case E_POSE_STEREO: {
        se3_map(v[0]->T, Xw, pc);
        camProjectStereo(pc, pr);
        for (int i = 0; i < 3; ++i) error[i] = meas[i] - pr[i];
      } break;
      case E_XYZ_MONO: {  // factors.cpp:76-82
        double r[3];
        qrot(rot, v[0]->p, r);
        for (int i = 0; i < 3; ++i) pc[i] = r[i] + t[i];
        camProjectMono(pc, pr);
        error[0] = meas[0] - pr[0];
        error[1] = meas[1] - pr[1];
      } break;
      case E_XYZ_STEREO: {  // factors.cpp:125-131
        double r[3];
        qrot(rot, v[0]->p, r);
        for (int i = 0; i < 3; ++i) pc[i] = r[i] + t[i];
        camProjectStereo(pc, pr);
        for (int i = 0; i < 3; ++i) error[i] = meas[i] - pr[i];
      } break;
      case E_PT2GAUSS: {  // factors.cpp:5-11: sqrt_info^T (x - mean)
        double d[3];
        for (int i = 0; i < 3; ++i) d[i] = v[0]->p[i] - mean[i];
        for (int i = 0; i < 3; ++i) {
          double s = 0.0;
          for (int k = 0; k < 3; ++k) s += sqrt_info[k * 3 + i] * d[k];
          error[i] = s;
        }
      } break;
      case E_PT2GAUSS_DEG: {  // factors.cpp:55-60
        double s = 0.0;
        for (int i = 0; i < 3; ++i) s += normal[i] * (v[0]->p[i] - mean[i]);
        error[0] = s;
      } break;
      case E_SE3_PRIOR: {  // factors.cpp:19-28
        SE3 d = se3_mul(inv_meas, v[0]->T);
        se3_log(d, error);
      } break;
      case E_BA_MONO: {
        se3_map(v[1]->T, v[0]->p, pc);
        camProjectMono(pc, pr);
        error[0] = meas[0] - pr[0];
        error[1] = meas[1] - pr[1];
      } break;
      case E_BA_STEREO: {
        se3_map(v[1]->T, v[0]->p, pc);
        camProjectStereo(pc, pr);
        for (int i = 0; i < 3; ++i) error[i] = meas[i] - pr[i];
      } break;
    }
  }

  bool isDepthPositive() const {
    double pc[3];
    if (kind == E_BA_MONO || kind == E_BA_STEREO) {
      se3_map(v[1]->T, v[0]->p, pc);
    } else {
      double r[3];
      qrot(rot, v[0]->p, r);
      pc[2] = r[2] + t[2];
    }
    return pc[2] > 0.0;
  }

  // pose Jacobian rows of g2o's projection edges
  static void poseJac(double x, double y, double z, double fx, double fy, double bf, bool stereo, double* J) {
    const double invz = 1.0 / z, invz_2 = invz * invz;
    J[0] = x * y * invz_2 * fx;
    J[1] = -(1 + (x * x * invz_2)) * fx;
    J[2] = y * invz * fx;
    J[3] = -invz * fx;
    J[4] = 0;
    J[5] = x * invz_2 * fx;
    J[6 + 0] = (1 + y * y * invz_2) * fy;
    J[6 + 1] = -x * y * invz_2 * fy;
    J[6 + 2] = -x * invz * fy;
    J[6 + 3] = 0;
    J[6 + 4] = -invz * fy;
    J[6 + 5] = y * invz_2 * fy;
    if (stereo) {
      J[12 + 0] = J[0] - bf * y * invz_2;
      J[12 + 1] = J[1] + bf * x * invz_2;
      J[12 + 2] = J[2];
      J[12 + 3] = J[3];
      J[12 + 4] = 0;
      J[12 + 5] = J[5] - bf * invz_2;
    }
  }
  // point Jacobian rows (factors.cpp:84-107 mono, :133-168 stereo)
  static void pointJac(double x, double y, double z, const double* R, double fx, double fy, double bf,
                       bool stereo, double* J) {
    const double z_2 = z * z;
    if (!stereo) {
      // -1/z * tmp * R,  tmp = [fx 0 -x/z fx; 0 fy -y/z fy]
      const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
      double tr[6];
      matmul(tmp, R, tr, 2, 3, 3);
      for (int i = 0; i < 6; ++i) J[i] = -1. / z * tr[i];
    } else {
      for (int c = 0; c < 3; ++c) {
        J[0 * 3 + c] = -fx * R[0 * 3 + c] / z + fx * x * R[2 * 3 + c] / z_2;
        J[1 * 3 + c] = -fy * R[1 * 3 + c] / z + fy * y * R[2 * 3 + c] / z_2;
        J[2 * 3 + c] = J[0 * 3 + c] - bf * R[2 * 3 + c] / z_2;
      }
    }
  }

  void linearizeOplus() {
    double pc[3], R[9];
    switch (kind) {
      case E_POSE_MONO:
      case E_POSE_STEREO:
        se3_map(v[0]->T, Xw, pc);
        poseJac(pc[0], pc[1], pc[2], fx, fy, bf, kind == E_POSE_STEREO, Ji);
        break;
      case E_XYZ_MONO:
      case E_XYZ_STEREO: {
        double r[3];
        qrot(rot, v[0]->p, r);
        for (int i = 0; i < 3; ++i) pc[i] = r[i] + t[i];
        qtoR(rot, R);
        pointJac(pc[0], pc[1], pc[2], R, fx, fy, bf, kind == E_XYZ_STEREO, Ji);
      } break;
      case E_PT2GAUSS:  // J = sqrt_info^T
        transpose(sqrt_info, Ji, 3, 3);
        break;
      case E_PT2GAUSS_DEG:
        for (int i = 0; i < 3; ++i) Ji[i] = normal[i];
        break;
      case E_SE3_PRIOR: {  // factors.cpp:30-53
        SE3 d = se3_mul(inv_meas, v[0]->T);
        double dvec[6];
        se3_log(d, dvec);
        double Jr[36];
        for (int i = 0; i < 36; ++i) Jr[i] = 0.0;
        double phi_s[9], luo_s[9];
        skew(dvec, phi_s);
        skew(dvec + 3, luo_s);
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) {
            Jr[i * 6 + j] = phi_s[i * 3 + j];
            Jr[(i + 3) * 6 + (j + 3)] = phi_s[i * 3 + j];
            Jr[i * 6 + (j + 3)] = luo_s[i * 3 + j];
          }
        for (int i = 0; i < 36; ++i) Jr[i] *= 0.5;
        for (int i = 0; i < 6; ++i) Jr[i * 6 + i] = 1.0 + Jr[i * 6 + i];
        double Adj[36];
        se3_adj(se3_inverse(v[0]->T), Adj);
        matmul(Jr, Adj, Ji, 6, 6, 6);
      } break;
      case E_BA_MONO:
      case E_BA_STEREO: {
        se3_map(v[1]->T, v[0]->p, pc);
        qtoR(v[1]->T.r, R);
        const bool st = kind == E_BA_STEREO;
        if (!st) {
          pointJac(pc[0], pc[1], pc[2], R, fx, fy, bf, false, Ji);
          // g2o writes the pose rows with x/z_2 style divisions (types_six_dof_expmap.cpp)
          const double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z;
          Jj[0] = x * y / z_2 * fx;
          Jj[1] = -(1 + (x * x / z_2)) * fx;
          Jj[2] = y / z * fx;
          Jj[3] = -1. / z * fx;
          Jj[4] = 0;
          Jj[5] = x / z_2 * fx;
          Jj[6 + 0] = (1 + y * y / z_2) * fy;
          Jj[6 + 1] = -x * y / z_2 * fy;
          Jj[6 + 2] = -x / z * fy;
          Jj[6 + 3] = 0;
          Jj[6 + 4] = -1. / z * fy;
          Jj[6 + 5] = y / z_2 * fy;
        } else {
          pointJac(pc[0], pc[1], pc[2], R, fx, fy, bf, true, Ji);
          const double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z;
          Jj[0] = x * y / z_2 * fx;
          Jj[1] = -(1 + (x * x / z_2)) * fx;
          Jj[2] = y / z * fx;
          Jj[3] = -1. / z * fx;
          Jj[4] = 0;
          Jj[5] = x / z_2 * fx;
          Jj[6 + 0] = (1 + y * y / z_2) * fy;
          Jj[6 + 1] = -x * y / z_2 * fy;
          Jj[6 + 2] = -x / z * fy;
          Jj[6 + 3] = 0;
          Jj[6 + 4] = -1. / z * fy;
          Jj[6 + 5] = y / z_2 * fy;
          Jj[12 + 0] = Jj[0] - bf * y / z_2;
          Jj[12 + 1] = Jj[1] + bf * x / z_2;
          Jj[12 + 2] = Jj[2];
          Jj[12 + 3] = Jj[3];
          Jj[12 + 4] = 0;
          Jj[12 + 5] = Jj[5] - bf / z_2;
        }
      } break;
    }
  }
};

// g2o::RobustKernelHuber::robustify
inline void huber(double e, double delta, double* rho) {
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    rho[0] = e;
    rho[1] = 1.;
    rho[2] = 0.;
  } else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

enum Algorithm { ALG_GN = 0, ALG_LM = 1 };
enum LinearSolver { LS_DENSE = 0, LS_EIGEN = 1 };
enum SolverResult { R_TERMINATE = 2, R_OK = 1, R_FAIL = -1 };

struct Optimizer {
  std::vector<std::unique_ptr<Vertex>> vstore;
  std::vector<std::unique_ptr<Edge>> estore;
  std::vector<Vertex*> vertices;
  std::vector<Edge*> edges;
  std::vector<Edge*> activeEdges;
  std::vector<Vertex*> activeVertices;
  std::vector<Vertex*> ivMap;
  int algorithm = ALG_LM;
  int linearSolver = LS_DENSE;
  const bool* forceStop = nullptr;
  // test hook with the semantics of gl_joint_optimization_stoppable's negative stop word: the flag reads true once
  // `stopBudget` outer iterations have completed in total (deterministic stand-in for another thread raising *forceStop)
  int stopBudget = -1, itersDone = 0;

  // solver state
  bool doSchur = false;
  int numPoses = 0, numLandmarks = 0, sizePoses = 0, sizeLandmarks = 0;
  std::vector<double> Hpp_backup_diag, Hll_backup_diag;
  std::vector<double> bvec, xvec;
  struct PL {  // one Hpl block: pose segment offset + 6x3 (dimPose x 3) block
    Vertex* pose;
    double W[18];
  };
  std::vector<std::vector<PL>> hpl;  // per landmark (index = hessianIndex - numPoses)
  double currentLambda = -1.0, ni = 2.0;
  int levenbergIterations = 0;
  bool trace = std::getenv("OG_TRACE") != nullptr;  // per-trial (currentChi, tempChi, lambda, rho) on stderr (debugging aid)

  Vertex* addVertex(int kind) {
    vstore.emplace_back(new Vertex());
    Vertex* v = vstore.back().get();
    v->kind = kind;
    vertices.push_back(v);
    return v;
  }
  Edge* addEdge(int kind, Vertex* v0, Vertex* v1 = nullptr) {
    estore.emplace_back(new Edge());
    Edge* e = estore.back().get();
    e->kind = kind;
    e->v[0] = v0;
    e->v[1] = v1;
    e->nv = v1 ? 2 : 1;
    static const int Ds[] = {2, 3, 2, 3, 3, 1, 6, 2, 3};
    e->D = Ds[kind];
    e->id = (int)edges.size();
    edges.push_back(e);
    v0->edges.push_back(e);
    if (v1) v1->edges.push_back(e);
    return e;
  }
  // HyperGraph::removeEdge (used by optimizeTriangulationVec, localization_opt.cpp:161-164)
  bool removeEdge(Edge* e) {
    auto it = std::find(edges.begin(), edges.end(), e);
    if (it == edges.end()) return false;
    edges.erase(it);
    for (int i = 0; i < e->nv; ++i) {
      auto& ve = e->v[i]->edges;
      ve.erase(std::remove(ve.begin(), ve.end(), e), ve.end());
    }
    return true;
  }

  bool terminate() const { return (forceStop && *forceStop) || (stopBudget >= 0 && itersDone >= stopBudget); }

  // SparseOptimizer::initializeOptimization(level) over all vertices
  bool initializeOptimization(int level = 0) {
    if (edges.empty()) return false;
    activeVertices.clear();
    activeEdges.clear();
    ivMap.clear();
    for (Vertex* v : vertices) {
      int levelEdges = 0;
      for (Edge* e : v->edges)
        if (level < 0 || e->level == level)
          if (!e->allVerticesFixed()) ++levelEdges;
      if (levelEdges) activeVertices.push_back(v);
    }
    for (Edge* e : edges)  // == set sorted by edge id
      if ((level < 0 || e->level == level) && !e->allVerticesFixed()) activeEdges.push_back(e);
    // buildIndexMapping: non-marginalised first, then marginalised
    int i = 0;
    for (int s = 0; s < 2; ++s)
      for (Vertex* v : activeVertices) {
        if (!v->fixed) {
          if ((int)v->marginalized == s) {
            v->hessianIndex = i++;
            ivMap.push_back(v);
          }
        } else {
          v->hessianIndex = -1;
        }
      }
    return true;
  }

  void computeActiveErrors() {
    for (Edge* e : activeEdges) e->computeError();
  }
  double activeRobustChi2() const {
    double rho[3];
    double chi = 0.0;
    for (const Edge* e : activeEdges) {
      if (e->robust) {
        huber(e->chi2(), e->delta, rho);
        chi += rho[0];
      } else {
        chi += e->chi2();
      }
    }
    return chi;
  }
  void push() {
    for (Vertex* v : ivMap) v->push();
  }
  void pop() {
    for (Vertex* v : ivMap) v->pop();
  }
  void discardTop() {
    for (Vertex* v : ivMap) v->discardTop();
  }
  void update(const double* u) {
    for (Vertex* v : ivMap) {
      v->oplus(u);
      u += v->dim();
    }
  }

  // BlockSolver::buildStructure (layout only)
  void buildStructure() {
    numPoses = numLandmarks = sizePoses = sizeLandmarks = 0;
    for (Vertex* v : ivMap) {
      if (!v->marginalized) {
        v->colInHessian = sizePoses;
        sizePoses += v->dim();
        ++numPoses;
      } else {
        v->colInHessian = sizeLandmarks;
        sizeLandmarks += v->dim();
        ++numLandmarks;
      }
    }
    bvec.assign(sizePoses + sizeLandmarks, 0.0);
    xvec.assign(sizePoses + sizeLandmarks, 0.0);
    hpl.assign(numLandmarks, {});
  }

  static void addJtWJ(double* A, int lda, const double* Ja, int da, const double* W, const double* Jb, int db,
                      int D) {
    // A(da x db) += Ja^T (D x da)^T * W (D x D) * Jb (D x db)
    for (int i = 0; i < da; ++i)
      for (int j = 0; j < db; ++j) {
        double s = 0.0;
        for (int r = 0; r < D; ++r) {
          double wj = 0.0;
          for (int c = 0; c < D; ++c) wj += W[r * D + c] * Jb[c * db + j];
          s += Ja[r * da + i] * wj;
        }
        A[i * lda + j] += s;
      }
  }

  PL& plBlock(Vertex* point, Vertex* pose) {
    auto& col = hpl[point->hessianIndex - numPoses];
    for (auto& b : col)
      if (b.pose == pose) return b;
    PL nb;
    nb.pose = pose;
    for (int i = 0; i < 18; ++i) nb.W[i] = 0.0;
    col.push_back(nb);
    return col.back();
  }

  // BlockSolver::buildSystem + Base{Unary,Binary}Edge::constructQuadraticForm
  void buildSystem() {
    for (Vertex* v : ivMap) {
      v->clearQuadraticForm();
      for (int i = 0; i < 36; ++i) v->A[i] = 0.0;
    }
    for (auto& col : hpl) col.clear();
    for (Edge* e : activeEdges) {
      e->linearizeOplus();
      const int D = e->D;
      double rho[3] = {0, 1, 0};
      double wOmega[36];
      if (e->robust) huber(e->chi2(), e->delta, rho);
      for (int i = 0; i < D * D; ++i) wOmega[i] = rho[1] * e->info[i];  // robustInformation
      double omega_r[6];  // -Omega * error (times rho' when robust)
      for (int i = 0; i < D; ++i) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s += e->info[i * D + j] * e->error[j];
        omega_r[i] = -s * (e->robust ? rho[1] : 1.0);
      }
      Vertex* v0 = e->v[0];
      Vertex* v1 = e->nv == 2 ? e->v[1] : nullptr;
      const bool f0 = !v0->fixed, f1 = v1 && !v1->fixed;
      if (f0) {
        const int d0 = v0->dim();
        for (int i = 0; i < d0; ++i) {
          double s = 0.0;
          for (int r = 0; r < D; ++r) s += e->Ji[r * d0 + i] * omega_r[r];
          v0->b[i] += s;
        }
        addJtWJ(v0->A, d0, e->Ji, d0, wOmega, e->Ji, d0, D);
      }
      if (f1) {
        const int d1 = v1->dim();
        for (int i = 0; i < d1; ++i) {
          double s = 0.0;
          for (int r = 0; r < D; ++r) s += e->Jj[r * d1 + i] * omega_r[r];
          v1->b[i] += s;
        }
        addJtWJ(v1->A, d1, e->Jj, d1, wOmega, e->Jj, d1, D);
      }
      if (f0 && f1) {
        // v0 = point (marginalised), v1 = pose: Hpl(pose, point) += Jj^T W Ji  (6 x 3)
        PL& blk = plBlock(v0, v1);
        addJtWJ(blk.W, 3, e->Jj, 6, wOmega, e->Ji, 3, D);
      }
    }
    for (Vertex* v : ivMap) {
      const int base = v->colInHessian + (v->marginalized ? sizePoses : 0);
      for (int i = 0; i < v->dim(); ++i) bvec[base + i] = v->b[i];
    }
  }

  void setLambda(double lambda) {  // backup=true
    Hpp_backup_diag.clear();
    for (Vertex* v : ivMap) {
      const int d = v->dim();
      for (int i = 0; i < d; ++i) {
        Hpp_backup_diag.push_back(v->A[i * d + i]);
        v->A[i * d + i] += lambda;
      }
    }
  }
  void restoreDiagonal() {
    size_t k = 0;
    for (Vertex* v : ivMap) {
      const int d = v->dim();
      for (int i = 0; i < d; ++i) v->A[i * d + i] = Hpp_backup_diag[k++];
    }
  }

  // BlockSolver::solve
  bool solveLinear() {
    const bool pos = (linearSolver == LS_DENSE);
    if (!doSchur) {
      // Hpp is block diagonal (no pose-pose edges on this path)
      std::vector<double> H((size_t)sizePoses * sizePoses, 0.0);
      for (Vertex* v : ivMap) {
        const int d = v->dim(), o = v->colInHessian;
        for (int i = 0; i < d; ++i)
          for (int j = 0; j < d; ++j) H[(size_t)(o + i) * sizePoses + (o + j)] = v->A[i * d + j];
      }
      return ldlt_solve(H.data(), bvec.data(), xvec.data(), sizePoses, pos);
    }
    std::vector<double> S((size_t)sizePoses * sizePoses, 0.0), coeff(sizePoses, 0.0);
    for (Vertex* v : ivMap)
      if (!v->marginalized) {
        const int d = v->dim(), o = v->colInHessian;
        for (int i = 0; i < d; ++i)
          for (int j = 0; j < d; ++j) S[(size_t)(o + i) * sizePoses + (o + j)] = v->A[i * d + j];
      }
    std::vector<double> Dinvs((size_t)numLandmarks * 9);
    for (Vertex* v : ivMap) {
      if (!v->marginalized) continue;
      const int li = v->hessianIndex - numPoses;
      double* Dinv = &Dinvs[(size_t)li * 9];
      inv3(v->A, Dinv);
      double db[3];
      matmul(Dinv, &bvec[sizePoses + v->colInHessian], db, 3, 3, 1);
      auto& col = hpl[li];
      // g2o walks the column sorted by pose row; order only affects rounding
      std::sort(col.begin(), col.end(), [](const PL& a, const PL& b) { return a.pose->colInHessian < b.pose->colInHessian; });
      for (size_t a = 0; a < col.size(); ++a) {
        const int o1 = col[a].pose->colInHessian;
        double BDinv[18], Bb[6];
        matmul(col[a].W, Dinv, BDinv, 6, 3, 3);
        matmul(col[a].W, db, Bb, 6, 3, 1);
        for (int i = 0; i < 6; ++i) coeff[o1 + i] += Bb[i];
        for (size_t c = a; c < col.size(); ++c) {
          const int o2 = col[c].pose->colInHessian;
          for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
              double s = 0.0;
              for (int k = 0; k < 3; ++k) s += BDinv[i * 3 + k] * col[c].W[j * 3 + k];
              S[(size_t)(o1 + i) * sizePoses + (o2 + j)] -= s;
            }
        }
      }
    }
    // the linear solver reads the upper triangle: mirror it
    for (int i = 0; i < sizePoses; ++i)
      for (int j = i + 1; j < sizePoses; ++j) S[(size_t)j * sizePoses + i] = S[(size_t)i * sizePoses + j];
    std::vector<double> bsch(sizePoses);
    for (int i = 0; i < sizePoses; ++i) bsch[i] = bvec[i] - coeff[i];
    if (!ldlt_solve(S.data(), bsch.data(), xvec.data(), sizePoses, pos)) return false;
    // landmarks: xl = Dinv (bl - Hpl^T xp)
    for (Vertex* v : ivMap) {
      if (!v->marginalized) continue;
      const int li = v->hessianIndex - numPoses;
      double cl[3];
      for (int i = 0; i < 3; ++i) cl[i] = bvec[sizePoses + v->colInHessian + i];
      for (auto& blk : hpl[li]) {
        const int o = blk.pose->colInHessian;
        for (int k = 0; k < 3; ++k) {
          double s = 0.0;
          for (int i = 0; i < 6; ++i) s += blk.W[i * 3 + k] * (-xvec[o + i]);
          cl[k] += s;
        }
      }
      matmul(&Dinvs[(size_t)li * 9], cl, &xvec[sizePoses + v->colInHessian], 3, 3, 1);
    }
    return true;
  }

  // OptimizationAlgorithmGaussNewton::solve
  int solveGN(int iteration) {
    computeActiveErrors();
    if (iteration == 0) buildStructure();
    buildSystem();
    const bool ok = solveLinear();
    if (ok) update(xvec.data());  // (on failure g2o applies an undefined x; we skip it)
    return ok ? R_OK : R_FAIL;
  }

  double computeLambdaInit() const {
    double maxDiagonal = 0.;
    for (const Vertex* v : ivMap) {
      const int d = v->dim();
      for (int j = 0; j < d; ++j) maxDiagonal = std::max(std::fabs(v->A[j * d + j]), maxDiagonal);
    }
    return 1e-5 * maxDiagonal;  // _tau
  }
  double computeScale() const {
    double scale = 0.;
    for (size_t j = 0; j < xvec.size(); ++j) scale += xvec[j] * (currentLambda * xvec[j] + bvec[j]);
    return scale;
  }

  // OptimizationAlgorithmLevenberg::solve
  int solveLM(int iteration) {
    if (iteration == 0) buildStructure();
    computeActiveErrors();
    double currentChi = activeRobustChi2();
    double tempChi = currentChi;
    buildSystem();
    if (iteration == 0) {
      currentLambda = computeLambdaInit();
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    const int maxTrials = 10;
    do {
      push();
      setLambda(currentLambda);
      const bool ok2 = solveLinear();
      update(xvec.data());
      restoreDiagonal();
      computeActiveErrors();
      tempChi = activeRobustChi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = (currentChi - tempChi);
      double scale = computeScale();
      scale += 1e-3;
      rho /= scale;
      if (trace) {
        std::fprintf(stderr, "OGTRACE %.17g %.17g %.17g %.17g", currentChi, tempChi, currentLambda, rho);
        for (int i = 0; i < sizePoses && i < 6; ++i) std::fprintf(stderr, " %.17g", xvec[i]);
        std::fprintf(stderr, "\n");
      }
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double scaleFactor = std::max(1. / 3., alpha);
        currentLambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
        discardTop();
      } else {
        currentLambda *= ni;
        ni *= 2;
        pop();
        if (switches().lambda_break && !std::isfinite(currentLambda)) break;  // newer g2o: `if (!g2o_isfinite(_currentLambda)) break;`
      }
      qmax++;
    } while (rho < 0 && qmax < maxTrials && !terminate());
    levenbergIterations = qmax;
    if (qmax == maxTrials || rho == 0) return R_TERMINATE;
    return R_OK;
  }

  // SparseOptimizer::optimize
  int optimize(int iterations) {
    if (ivMap.empty()) return -1;
    doSchur = false;
    for (Vertex* v : activeVertices)
      if (v->marginalized) {
        doSchur = true;
        break;
      }
    int cj = 0;
    bool ok = true;
    int result = R_OK;
    for (int i = 0; i < iterations && !terminate() && ok; ++i) {
      result = (algorithm == ALG_GN) ? solveGN(i) : solveLM(i);
      ok = (result == R_OK);
      ++cj;
      ++itersDone;
    }
    if (result == R_FAIL) return 0;
    return cj;
  }
};

}  // namespace og
