// Drives include/gmmloc_hip/gmm_adapter.hpp (the C++ host mirror a gmmloc maintainer links) on one frame dumped
// by tests/test_gpu_adapter.py and writes the results back for comparison with the Python host.
//   adapter_check <map.gmm> <frame.bin> <out.bin>
// frame.bin: int32 M, N, width, height; double fx fy cx cy bf; pose[7]; Xw[M*3]; obs[M*3]; int32 octave[M]; double uv[N*2]
// out.bin:   track: double pose[7], Xw[M*3], int32 assoc[M];  pose-only: double pose[7], int32 ninlier, uint8 outl[M];
//            associate: int32 idx[M], double d2[M];  search: int32 n[N], cand[N*5] (-1 padded);  int32 queryPoint
// then (optional) a local window: int32 P, F, L, NOBS; double poses[(P+F)*7]; uint8 prior[P]; double points[L*3];
//   int32 assoc[L], obs_ptr[L+1], obs_pose[NOBS]; double obs_uvr[NOBS*3]; int32 obs_oct[NOBS]
//   -> out: double poses[(P+F)*7], points[L*3]; uint8 dropped[L], erase[NOBS]; int32 iters
#include <chrono>
#include <cstdio>
#include <vector>

#include "gmmloc_hip/gmm_adapter.hpp"

template <class T>
static void rd(FILE* f, T* p, size_t n) {
  if (n && fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short read");
}
template <class T>
static void wr(FILE* f, const T* p, size_t n) {
  if (n && fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("short write");
}

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  try {
    gmmloc_hip::GMM gmm;
    if (!gmmloc_hip::GMM::loadGMMModel(argv[1], gmm)) {
      fprintf(stderr, "loadGMMModel: %s\n", gmmloc_hip::GMM::last_error());
      return 1;
    }
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 1;
    int32_t MN[4];
    double intr[5];
    rd(f, MN, 4);
    rd(f, intr, 5);
    const int M = MN[0], N = MN[1];
    gl_camera cam{intr[0], intr[1], intr[2], intr[3], intr[4], MN[2], MN[3]};
    gmm.setCamera(cam);
    gmmloc_hip::Pose pose0;
    std::vector<double> Xw(M * 3), obs(M * 3), uv(N * 2);
    std::vector<int32_t> oct(M);
    rd(f, reinterpret_cast<double*>(&pose0), 7);
    rd(f, Xw.data(), Xw.size());
    rd(f, obs.data(), obs.size());
    rd(f, oct.data(), oct.size());
    rd(f, uv.data(), uv.size());
    gmmloc_hip::GMM::LocalWindow w;
    int32_t hdr[4] = {0, 0, 0, 0};
    const bool has_window = fread(hdr, sizeof(int32_t), 4, f) == 4;
    if (has_window) {
      w.P = hdr[0];
      w.F = hdr[1];
      const int L = hdr[2], NOBS = hdr[3];
      w.poses.resize(w.P + w.F);
      w.prior.resize(w.P);
      w.points.resize((size_t)L * 3);
      w.assoc.resize(L);
      w.obs_ptr.resize(L + 1);
      w.obs_pose.resize(NOBS);
      w.obs_uvr.resize((size_t)NOBS * 3);
      w.obs_oct.resize(NOBS);
      rd(f, reinterpret_cast<double*>(w.poses.data()), (size_t)(w.P + w.F) * 7);
      rd(f, w.prior.data(), w.prior.size());
      rd(f, w.points.data(), w.points.size());
      rd(f, w.assoc.data(), w.assoc.size());
      rd(f, w.obs_ptr.data(), w.obs_ptr.size());
      rd(f, w.obs_pose.data(), w.obs_pose.size());
      rd(f, w.obs_uvr.data(), w.obs_uvr.size());
      rd(f, w.obs_oct.data(), w.obs_oct.size());
    }
    fclose(f);
    FILE* o = fopen(argv[3], "wb");
    if (!o) return 1;
    for (int anchored = 0; anchored < 2; ++anchored) {  // north-star path: unanchored, then with the prior edge on the pose
      gmmloc_hip::Pose p = pose0;
      std::vector<double> X = Xw;
      std::vector<int32_t> assoc;
      gmm.trackFrame(p, X, obs, oct, assoc, anchored != 0);
      wr(o, reinterpret_cast<double*>(&p), 7);
      wr(o, X.data(), X.size());
      wr(o, assoc.data(), assoc.size());
    }
    {
      gmmloc_hip::Pose p = pose0;
      std::vector<uint8_t> outl;
      const int32_t nin = gmm.optimizeCurrentPose(p, Xw, obs, oct, outl);
      wr(o, reinterpret_cast<double*>(&p), 7);
      wr(o, &nin, 1);
      wr(o, outl.data(), outl.size());
    }
    {
      std::vector<int32_t> idx;
      std::vector<double> d2;
      gmm.associate(Xw, idx, d2);
      wr(o, idx.data(), idx.size());
      wr(o, d2.data(), d2.size());
    }
    {
      std::vector<std::vector<int32_t>> comps;
      gmm.renderViewAndSearch(pose0, uv, comps, 5);
      std::vector<int32_t> n(N), cand((size_t)N * 5, -1);
      for (int i = 0; i < N; ++i) {
        n[i] = (int32_t)comps[i].size();
        for (size_t j = 0; j < comps[i].size(); ++j) cand[(size_t)i * 5 + j] = comps[i][j];
      }
      wr(o, n.data(), n.size());
      wr(o, cand.data(), cand.size());
    }
    {
      std::vector<int> res;
      gmm.queryPoint(Xw.data(), res);
      const int32_t q = res.empty() ? -1 : res[0];
      wr(o, &q, 1);
    }
    if (has_window) {  // Localization::jointOptimization on one local window
      gmm.jointOptimization(w);
      wr(o, reinterpret_cast<double*>(w.poses.data()), w.poses.size() * 7);
      wr(o, w.points.data(), w.points.size());
      wr(o, w.assoc_dropped.data(), w.assoc_dropped.size());
      wr(o, w.obs_erase.data(), w.obs_erase.size());
      const int32_t it = w.iters;
      wr(o, &it, 1);
    }
    {  // host-buffer call time of the frame-at-a-time path (staging included)
      gmmloc_hip::Pose p = pose0;
      std::vector<double> X = Xw;
      std::vector<int32_t> assoc;
      gmm.trackFrame(p, X, obs, oct, assoc);
      const auto t0 = std::chrono::steady_clock::now();
      const int reps = 20;
      for (int r = 0; r < reps; ++r) {
        p = pose0;
        X = Xw;
        gmm.trackFrame(p, X, obs, oct, assoc);
      }
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
      printf("trackFrame host-to-host %.3f ms per call (M = %d)\n", ms, M);
      std::vector<uint8_t> outl;
      gmm.optimizeCurrentPose(p, Xw, obs, oct, outl);
      const auto t1 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) {
        p = pose0;
        gmm.optimizeCurrentPose(p, Xw, obs, oct, outl);
      }
      const double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() / reps;
      printf("optimizeCurrentPose host-to-host %.3f ms per call (M = %d)\n", ms2, M);
    }
    fclose(o);
    printf("components %zu\n", gmm.countComponents());
  } catch (const std::exception& e) {
    fprintf(stderr, "adapter_check: %s\n", e.what());
    return 1;
  }
  return 0;
}
