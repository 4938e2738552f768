// gmm_adapter.hpp -- header-only C++ host mirror of the reference's hot-path interface over
// the C-ABI (include/gmmloc_hip.h).  This is the adapter a maintainer of HyHuang1995/gmmloc
// adds to the host (INTEGRATION.md shows the call-site patch); it depends only on the STL so
// that it also compiles where Eigen / g2o are absent.  Method names follow the reference:
//   gmmloc::GMM::renderView + searchCorrespondence  (gaussian_mixture.cpp:271-371, :484-534)
//   gmmloc::GMM::queryPoint                         (gaussian_mixture.cpp:545-576)
//   GMMUtility::loadGMMModel                        (gmm_utils.cpp:9-67)
//   GMMLoc::optimizePoint / checkMapAssociation     (gmmloc_opt.cpp:156-342)
//   Tracking::optimizeCurrentPose                   (tracking_opt.cpp:21-217)
//   Localization::jointOptimization                 (localization_opt.cpp:456-925)
//   north-star per-frame path: associate + structure-constrained refinement of one frame (trackFrame)
// Host buffers in, host buffers out: each call stages through device memory owned by the
// adapter (gl_malloc / gl_memcpy_*), so the host code never sees a HIP type.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "gmmloc_hip.h"

namespace gmmloc_hip {

// 3: trackFrame is the unanchored refine again, trackFrameAnchored the anchored one (2: trackFrame(..., anchored = true))
constexpr int kAdapterVersion = 3;

inline void check(int rc, const char* what) {
  if (rc != GL_OK) throw std::runtime_error(std::string(what) + ": " + gl_last_error_string());
}

// RAII device buffer
class DevBuf {
 public:
  DevBuf(gl_ctx_t* ctx, size_t bytes) : ctx_(ctx), bytes_(bytes) { check(gl_malloc(ctx, bytes, &p_), "gl_malloc"); }
  ~DevBuf() { gl_free(ctx_, p_); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  template <class T>
  T* as() { return static_cast<T*>(p_); }
  void upload(const void* src) { check(gl_memcpy_h2d(ctx_, p_, src, bytes_), "h2d"); }
  void download(void* dst) { check(gl_memcpy_d2h(ctx_, dst, p_, bytes_), "d2h"); }
  void upload(const void* src, size_t n) { check(gl_memcpy_h2d(ctx_, p_, src, n), "h2d"); }
  void download(void* dst, size_t n) { check(gl_memcpy_d2h(ctx_, dst, p_, n), "d2h"); }
  size_t bytes() const { return bytes_; }

 private:
  gl_ctx_t* ctx_;
  void* p_ = nullptr;
  size_t bytes_;
};

struct Pose {  // g2o::SE3Quat T_cw: Eigen quaternion coefficients (x y z w) + translation
  double q[4], t[3];
};

class GMM {
 public:
  // loadGMMModel(path, model): false on failure like the reference (message via last_error())
  static bool loadGMMModel(const std::string& path, GMM& model, int device = 0) {
    model.release();
    if (gl_ctx_create(device, nullptr, &model.ctx_) != GL_OK) return false;
    gl_default_params(&model.prm_);
    if (gl_gmm_load_file(model.ctx_, path.c_str(), &model.prm_, &model.gmm_) != GL_OK) return false;
    return true;
  }
  ~GMM() { release(); }
  size_t countComponents() const { return (size_t)gl_gmm_count(gmm_); }
  void setCamera(const gl_camera& cam) { cam_ = cam; }
  static const char* last_error() { return gl_last_error_string(); }

  // renderView(rot_c_w, t_c_w) followed by searchCorrespondence(kpts, comps, num): one call, because
  // the rendered list only exists to be searched (gmmloc_opt.cpp:121-134).  uv: N x 2.
  // comps[i] = component indices (GaussianComponent2d::parent_ via getComponent3d(idx)).
  void renderViewAndSearch(const Pose& Tcw, const std::vector<double>& uv, std::vector<std::vector<int32_t>>& comps,
                           int num = 5) {
    const int N = (int)(uv.size() / 2);
    DevBuf dpose(ctx_, 56), duv(ctx_, uv.size() * 8 + 8), dcand(ctx_, (size_t)N * num * 4 + 4), dn(ctx_, (size_t)N * 4 + 4);
    dpose.upload(&Tcw);
    if (N) duv.upload(uv.data());
    check(gl_search2d(ctx_, gmm_, &cam_, 1, dpose.as<double>(), N, duv.as<double>(), nullptr, num, dcand.as<int32_t>(),
                      dn.as<int32_t>(), 0, nullptr, nullptr),
          "gl_search2d");
    check(gl_ctx_synchronize(ctx_), "sync");
    std::vector<int32_t> cand((size_t)N * num + 1), n(N + 1);
    dcand.download(cand.data());
    dn.download(n.data());
    comps.assign(N, {});
    for (int i = 0; i < N; ++i) comps[i].assign(cand.begin() + (size_t)i * num, cand.begin() + (size_t)i * num + n[i]);
  }

  // queryPoint(pt, res): res = {nearest component} (the reference pushes ret_index[0])
  void queryPoint(const double pt[3], std::vector<int>& res) {
    DevBuf dp(ctx_, 24), di(ctx_, 8);
    dp.upload(pt);
    check(gl_associate3d(ctx_, gmm_, dp.as<double>(), 1, GL_ASSOC_KNN5_EUCLID, di.as<int32_t>(), nullptr), "queryPoint");
    check(gl_ctx_synchronize(ctx_), "sync");
    int32_t idx[2];
    di.download(idx);
    res.clear();
    if (idx[0] >= 0) res.push_back(idx[0]);
  }

  // exhaustive Mahalanobis association of N points (north-star `associate`)
  void associate(const std::vector<double>& pts, std::vector<int32_t>& idx, std::vector<double>& d2) {
    const int N = (int)(pts.size() / 3);
    idx.assign(N, -1);
    d2.assign(N, 0.0);
    if (!N) return;
    DevBuf dp(ctx_, pts.size() * 8), di(ctx_, (size_t)N * 4), dd(ctx_, (size_t)N * 8);
    dp.upload(pts.data());
    check(gl_associate3d(ctx_, gmm_, dp.as<double>(), N, GL_ASSOC_BRUTE, di.as<int32_t>(), dd.as<double>()), "associate");
    check(gl_ctx_synchronize(ctx_), "sync");
    di.download(idx.data());
    dd.download(d2.data());
  }

  // Tracking::optimizeCurrentPose for one frame: Xw / obs are M x 3, octave[i] < 0 = no map point.
  // Returns the inlier count; pose and is_outlier are updated like the reference does: the flag of a feature WITH a
  // map point is rewritten, the flags of the others are left as the host had them (tracking_opt.cpp:63-69).
  int optimizeCurrentPose(Pose& Tcw, const std::vector<double>& Xw, const std::vector<double>& obs,
                          const std::vector<int32_t>& octave, std::vector<uint8_t>& is_outlier) {
    const int M = (int)octave.size();
    is_outlier.resize(M, 0);
    // one pooled device buffer, one page-locked staging buffer, one transfer each way, one synchronize:
    //   pose | ninlier | is_outlier || Xw | obs | octave
    const size_t oN = 64, oF = 72, oX = oF + (((size_t)M + 7) / 8) * 8 + 8, oO = oX + (size_t)M * 24, oC = oO + (size_t)M * 24,
                 total = oC + (size_t)M * 4 + 8;
    DevBuf& d = pooled(1, total);
    char* st = stage(total);
    std::memcpy(st, &Tcw, 56);
    std::memset(st + oN, 0, 8);
    if (M) {
      std::memcpy(st + oF, is_outlier.data(), M);
      std::memcpy(st + oX, Xw.data(), (size_t)M * 24);
      std::memcpy(st + oO, obs.data(), (size_t)M * 24);
      std::memcpy(st + oC, octave.data(), (size_t)M * 4);
    }
    char* base = d.as<char>();
    check(gl_memcpy_h2d_async(ctx_, base, st, total), "h2d");
    check(gl_optimize_current_pose(ctx_, &cam_, &prm_, 1, M, reinterpret_cast<double*>(base), reinterpret_cast<const double*>(base + oX),
                                   reinterpret_cast<const double*>(base + oO), reinterpret_cast<const int32_t*>(base + oC),
                                   reinterpret_cast<uint8_t*>(base + oF), reinterpret_cast<int32_t*>(base + oN)),
          "gl_optimize_current_pose");
    check(gl_memcpy_d2h_async(ctx_, st, base, oX), "d2h");
    check(gl_ctx_synchronize(ctx_), "sync");
    std::memcpy(&Tcw, st, 56);
    if (M) std::memcpy(is_outlier.data(), st + oF, M);
    int32_t n;
    std::memcpy(&n, st + oN, 4);
    return n;
  }

  // North-star per-frame path for ONE frame (gl_track_frames, B = 1): exact Mahalanobis association of the
  // frame's M map points (kept iff chi2 <= 9, gmmloc_opt.cpp:230-232) + jointOptimization restricted to the
  // frame (1 free pose, M marginalised points, 5 / 5 / 40 Levenberg schedule).  Tcw and Xw are updated;
  // assoc[i] = component after the final gates or -1; octave[i] < 0 = no map point.
  // trackFrame: the pose is held by the map's Gaussians alone (rounds 1 - 2; the default again since adapter version 3 - version 2
  // silently switched this entry point to the anchored refine).  trackFrameAnchored: with the reference's gauge
  // anchor of key-frame 0 - an EdgeSE3QuatPrior on the pose the call starts from (sigma 2 deg / 1 cm; the pose is FIXED when
  // gl_params.ba_first_as_prior == 0), localization_opt.cpp:556-581: the reference never runs its structure BA without one, so
  // this is the entry point a port of Localization's per-frame refine wants; the two give DIFFERENT poses and points.
  void trackFrame(Pose& Tcw, std::vector<double>& Xw, const std::vector<double>& obs, const std::vector<int32_t>& octave,
                  std::vector<int32_t>& assoc, bool anchored = false) {
    const int M = (int)octave.size();
    assoc.assign(M, -1);
    if (!M) return;
    // the per-frame caller: the library keeps a page-locked staging buffer and its device mirror in the context, one
    // transfer each way on the context's stream and ONE synchronize per frame (gl_track_frame_host[_anchored])
    if (anchored)
      check(gl_track_frame_host_anchored(ctx_, gmm_, &cam_, &prm_, M, reinterpret_cast<double*>(&Tcw), Xw.data(), obs.data(),
                                         octave.data(), assoc.data()),
            "gl_track_frame_host_anchored");
    else
      check(gl_track_frame_host(ctx_, gmm_, &cam_, &prm_, M, reinterpret_cast<double*>(&Tcw), Xw.data(), obs.data(), octave.data(),
                                assoc.data()),
            "gl_track_frame_host");
  }
  void trackFrameAnchored(Pose& Tcw, std::vector<double>& Xw, const std::vector<double>& obs, const std::vector<int32_t>& octave,
                          std::vector<int32_t>& assoc) {
    trackFrame(Tcw, Xw, obs, octave, assoc, true);
  }

  // Localization::jointOptimization on one flattened local window (layout: gmmloc_hip.h, gl_joint_optimization):
  // poses [0,P) free, [P,P+F) fixed; observations in CSR order by point.
  struct LocalWindow {
    int P = 0, F = 0;
    std::vector<Pose> poses;            // P + F, in/out for [0,P)
    std::vector<uint8_t> prior;         // P: key-frame idx_ 0 (prior edge)
    std::vector<double> points;         // L x 3, in/out
    std::vector<int32_t> assoc;         // L: component or -1
    std::vector<int32_t> obs_ptr;       // L + 1
    std::vector<int32_t> obs_pose;      // NOBS
    std::vector<double> obs_uvr;        // NOBS x 3
    std::vector<int32_t> obs_oct;       // NOBS
    std::vector<uint8_t> assoc_dropped; // out, L   (:837-853)
    std::vector<uint8_t> obs_erase;     // out, NOBS (:855-879)
    int iters = 0;                      // out: actual_iter of the last optimize(40)
  };
  // stop_flag: the reference's pbStopFlag (setForceStopFlag, localization_opt.cpp:541-542) as one int32 in memory from
  // gl_malloc_host (another thread raises it to 1) or NULL: the window returns with the state of the last accepted step
  void jointOptimization(LocalWindow& w, const int32_t* stop_flag = nullptr) {
    const int L = (int)w.assoc.size(), NOBS = (int)w.obs_pose.size(), NP = w.P + w.F;
    w.assoc_dropped.assign(L, 0);
    w.obs_erase.assign(NOBS, 0);
    w.iters = 0;
    if (!L || !NOBS || !w.P) return;
    DevBuf dposes(ctx_, (size_t)NP * 56), dprior(ctx_, (size_t)w.P + 8), dpts(ctx_, (size_t)L * 24), dassoc(ctx_, (size_t)L * 4),
        dptr(ctx_, (size_t)(L + 1) * 4), dop(ctx_, (size_t)NOBS * 4), duvr(ctx_, (size_t)NOBS * 24), doct(ctx_, (size_t)NOBS * 4),
        ddrop(ctx_, (size_t)L + 8), derase(ctx_, (size_t)NOBS + 8), dit(ctx_, 8);
    std::vector<uint8_t> pr(w.P + 8, 0);
    std::memcpy(pr.data(), w.prior.data(), w.P);
    dposes.upload(w.poses.data());
    dprior.upload(pr.data());
    dpts.upload(w.points.data());
    dassoc.upload(w.assoc.data());
    dptr.upload(w.obs_ptr.data());
    dop.upload(w.obs_pose.data());
    duvr.upload(w.obs_uvr.data());
    doct.upload(w.obs_oct.data());
    check(gl_joint_optimization_stoppable(ctx_, gmm_, &cam_, &prm_, 1, w.P, w.F, L, NOBS, dposes.as<double>(), dprior.as<uint8_t>(),
                                          dpts.as<double>(), dassoc.as<int32_t>(), dptr.as<int32_t>(), dop.as<int32_t>(),
                                          duvr.as<double>(), doct.as<int32_t>(), ddrop.as<uint8_t>(), derase.as<uint8_t>(),
                                          dit.as<int32_t>(), stop_flag),
          "gl_joint_optimization");
    check(gl_ctx_synchronize(ctx_), "sync");
    dposes.download(w.poses.data());
    dpts.download(w.points.data());
    std::vector<uint8_t> t1(L + 8), t2(NOBS + 8);
    ddrop.download(t1.data());
    derase.download(t2.data());
    std::memcpy(w.assoc_dropped.data(), t1.data(), L);
    std::memcpy(w.obs_erase.data(), t2.data(), NOBS);
    int32_t it[2];
    dit.download(it);
    w.iters = it[0];
  }

  gl_ctx_t* ctx() { return ctx_; }
  gl_gmm_t* handle() { return gmm_; }
  gl_params& params() { return prm_; }
  const gl_camera& camera() const { return cam_; }

 private:
  void release() {
    pool_.clear();
    if (stage_ && ctx_) gl_free_host(ctx_, stage_);
    stage_ = nullptr;
    stage_bytes_ = 0;
    if (gmm_) gl_gmm_destroy(gmm_);
    if (ctx_) gl_ctx_destroy(ctx_);
    gmm_ = nullptr;
    ctx_ = nullptr;
  }
  DevBuf& pooled(size_t slot, size_t bytes) {
    if (pool_.size() <= slot) pool_.resize(slot + 1);
    if (!pool_[slot] || pool_[slot]->bytes() < bytes) pool_[slot].reset(new DevBuf(ctx_, bytes + bytes / 4 + 64));
    return *pool_[slot];
  }
  char* stage(size_t bytes) {  // page-locked, grown on demand
    if (stage_bytes_ < bytes) {
      if (stage_) gl_free_host(ctx_, stage_);
      stage_ = nullptr;
      stage_bytes_ = 0;
      void* p = nullptr;
      check(gl_malloc_host(ctx_, bytes + bytes / 4 + 64, &p), "gl_malloc_host");
      stage_ = static_cast<char*>(p);
      stage_bytes_ = bytes + bytes / 4 + 64;
    }
    return stage_;
  }
  std::vector<std::unique_ptr<DevBuf>> pool_;
  char* stage_ = nullptr;
  size_t stage_bytes_ = 0;
  gl_ctx_t* ctx_ = nullptr;
  gl_gmm_t* gmm_ = nullptr;
  gl_params prm_;
  gl_camera cam_{};
};

}  // namespace gmmloc_hip
