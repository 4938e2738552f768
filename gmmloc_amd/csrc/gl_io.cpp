// .gmm map stream reader / writer (host C++, no protobuf dependency).
// Framing (gmmloc/src/utils/protobuf_utils.cpp:12-29,42-80; writer :31-40,82-100):
//   varint32 count, then count x { varint32 size, ComponentProto bytes }.
// ComponentProto (gmmloc/proto/gmmloc/GMM.proto:5-14, proto2):
//   1 bool is_degenerated, 2 bool is_salient,
//   3 repeated double mean [packed], 4 repeated double covariance [packed].
// The reference loader ignores fields 1-2 and recomputes them
// (gmm_utils.cpp:50-61); covariance is read row-major (:54-59).
#include <cstdio>
#include <cstring>
#include <vector>

#include "gl_internal.hpp"

namespace gl {

namespace {
bool get_varint(const std::vector<uint8_t>& b, size_t& pos, uint64_t& out) {
  out = 0;
  for (int shift = 0; shift < 64; shift += 7) {
    if (pos >= b.size()) return false;
    const uint8_t c = b[pos++];
    out |= (uint64_t)(c & 0x7F) << shift;
    if (!(c & 0x80)) return true;
  }
  return false;
}
void put_varint(std::vector<uint8_t>& b, uint64_t v) {
  while (v >= 0x80) {
    b.push_back((uint8_t)(v | 0x80));
    v >>= 7;
  }
  b.push_back((uint8_t)v);
}
int fail(const char* msg) {
  set_error("%s", msg);
  return GL_ERR_FORMAT;
}
}  // namespace

int read_gmm_file(const char* path, std::vector<double>& mean, std::vector<double>& cov) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    set_error("Could not open protobuf file to load layer: %s", path);  // gmm_utils.cpp:19-22
    return GL_ERR_IO;
  }
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  size_t pos = 0;
  uint64_t count = 0;
  if (!get_varint(buf, pos, count)) return fail("failed read number of messages.");  // gmm_utils.cpp:27-30
  if (count == 0) return fail("protobuf file empty!");                               // gmm_utils.cpp:32-35
  mean.clear();
  cov.clear();
  for (uint64_t i = 0; i < count; ++i) {
    uint64_t size = 0;
    if (!get_varint(buf, pos, size) || size == 0 || pos + size > buf.size())
      return fail("failed to read component message.");  // gmm_utils.cpp:44-47
    const size_t end = pos + size;
    size_t nm = 0, nc = 0;
    double m[3] = {0, 0, 0}, c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto take = [&](int field, double v) {
      if (field == 3) {
        if (nm < 3) m[nm] = v;
        ++nm;
      } else if (field == 4) {
        if (nc < 9) c[nc] = v;
        ++nc;
      }
    };
    while (pos < end) {
      uint64_t tag = 0;
      if (!get_varint(buf, pos, tag)) return fail("truncated tag");
      const int field = (int)(tag >> 3), wt = (int)(tag & 7);
      if (wt == 0) {
        uint64_t v;
        if (!get_varint(buf, pos, v)) return fail("truncated varint");
      } else if (wt == 2) {
        uint64_t len = 0;
        if (!get_varint(buf, pos, len) || pos + len > end) return fail("bad length-delimited field");
        if (field == 3 || field == 4) {
          if (len % 8) return fail("bad packed double field");
          for (uint64_t o = 0; o < len; o += 8) {
            double v;
            memcpy(&v, &buf[pos + o], 8);
            take(field, v);
          }
        }
        pos += len;
      } else if (wt == 1) {  // unpacked double
        if (pos + 8 > end) return fail("truncated double");
        double v;
        memcpy(&v, &buf[pos], 8);
        pos += 8;
        take(field, v);
      } else if (wt == 5) {
        if (pos + 4 > end) return fail("truncated fixed32");
        pos += 4;
      } else {
        return fail("unsupported wire type");
      }
    }
    if (pos != end) return fail("Could not consume protobuf message.");
    if (nm != 3 || nc != 9)  // CHECK_EQ(mean_size, 3) / CHECK_EQ(covariance_size, 9), gmm_utils.cpp:50-51
      return fail("component with mean_size != 3 or covariance_size != 9");
    mean.insert(mean.end(), m, m + 3);
    cov.insert(cov.end(), c, c + 9);
  }
  return GL_OK;
}

// saveGMMModel (gmm_utils.cpp:69-119): is_degenerated, is_salient, mean[3],
// covariance[9] ("cov(i)" walks Eigen's column-major storage, :102-104).
int write_gmm_file(const char* path, const double* mean, const double* cov, const uint8_t* flags, int K) {
  std::vector<uint8_t> out;
  put_varint(out, (uint64_t)K);
  for (int k = 0; k < K; ++k) {
    std::vector<uint8_t> msg;
    msg.push_back(0x08);
    msg.push_back((flags[k] & 1) ? 1 : 0);
    msg.push_back(0x10);
    msg.push_back((flags[k] & 2) ? 1 : 0);
    msg.push_back(0x1A);
    put_varint(msg, 24);
    const uint8_t* p = (const uint8_t*)(mean + 3 * k);
    msg.insert(msg.end(), p, p + 24);
    msg.push_back(0x22);
    put_varint(msg, 72);
    for (int i = 0; i < 9; ++i) {  // column-major walk of the row-major array
      const double v = cov[9 * k + (i % 3) * 3 + (i / 3)];
      const uint8_t* q = (const uint8_t*)&v;
      msg.insert(msg.end(), q, q + 8);
    }
    put_varint(out, msg.size());
    out.insert(out.end(), msg.begin(), msg.end());
  }
  FILE* f = fopen(path, "wb");
  if (!f) {
    set_error("fail to save model, path: %s", path);
    return GL_ERR_IO;
  }
  const size_t w = fwrite(out.data(), 1, out.size(), f);
  fclose(f);
  if (w != out.size()) {
    set_error("short write");
    return GL_ERR_IO;
  }
  return GL_OK;
}

}  // namespace gl

extern "C" {

int gl_gmm_file_read(const char* path, double* mean, double* cov, int cap, int* K_out) {
  GL_REQUIRE(path && K_out, "null argument");
  std::vector<double> m, c;
  const int rc = gl::read_gmm_file(path, m, c);
  if (rc != GL_OK) return rc;
  const int K = (int)(m.size() / 3);
  *K_out = K;
  if (mean || cov) {
    GL_REQUIRE(cap >= K, "output capacity too small");
    if (mean) memcpy(mean, m.data(), sizeof(double) * 3 * K);
    if (cov) memcpy(cov, c.data(), sizeof(double) * 9 * K);
  }
  return GL_OK;
}

int gl_gmm_file_write(const char* path, const double* mean, const double* cov, const uint8_t* flags, int K) {
  GL_REQUIRE(path && mean && cov && flags && K > 0, "bad argument");
  return gl::write_gmm_file(path, mean, cov, flags, K);
}

// Map::summarize (map.cpp:162-188): one line per frame, `timestamp tx ty tz qx qy qz qw` of T_wc in TUM
// format -- std::fixed, 6 digits for the stamp, 9 for the pose, single spaces, '\n'.
int gl_write_tum_trajectory(const char* path, const double* stamps, const double* pose_wc, int N) {
  GL_REQUIRE(path && (N == 0 || (stamps && pose_wc)) && N >= 0, "bad argument");
  FILE* f = fopen(path, "w");
  if (!f) {
    gl::set_error("gl_write_tum_trajectory: cannot open %s", path);
    return GL_ERR_IO;
  }
  for (int i = 0; i < N; ++i) {
    const double* p = pose_wc + (size_t)i * 7;  // qx qy qz qw tx ty tz
    fprintf(f, "%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", stamps[i], p[4], p[5], p[6], p[0], p[1], p[2], p[3]);
  }
  fclose(f);
  return GL_OK;
}

}  // extern "C"
