// io.cpp -- see include/ddamg_hip_io.h.  Host code only.
#include "../../include/ddamg_hip_io.h"
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdarg>
#include <string>
#include <vector>
#include <fcntl.h>
#include <unistd.h>

namespace {

thread_local char g_err[512] = "";
int fail(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return -1;
}

void swap8(void* p, size_t n) {
  unsigned char* b = static_cast<unsigned char*>(p);
  for (size_t i = 0; i < n; i++, b += 8)
    for (int k = 0; k < 4; k++) { unsigned char t = b[k]; b[k] = b[7 - k]; b[7 - k] = t; }
}
void swap4(void* p, size_t n) {
  unsigned char* b = static_cast<unsigned char*>(p);
  for (size_t i = 0; i < n; i++, b += 4)
    for (int k = 0; k < 2; k++) { unsigned char t = b[k]; b[k] = b[3 - k]; b[3 - k] = t; }
}

struct Part {
  int G[4], L[4], O[4];   // global extents, local extents, origin of this process
  size_t Vloc, Vglob;
};
bool make_part(const int G[4], const int P[4], const int C[4], Part& p) {
  p.Vloc = 1; p.Vglob = 1;
  for (int mu = 0; mu < 4; mu++) {
    const int np = P[mu] < 1 ? 1 : P[mu];
    if (G[mu] < 1 || G[mu] % np || C[mu] < 0 || C[mu] >= np) return false;
    p.G[mu] = G[mu]; p.L[mu] = G[mu] / np; p.O[mu] = C[mu] * p.L[mu];
    p.Vloc *= p.L[mu]; p.Vglob *= G[mu];
  }
  return true;
}

// rows of L[X] consecutive sites: the unit the reference moves as well (read_size = 4*18*ll[X], bar_size = 24*ll[X])
template <typename F>
int for_rows(const Part& p, F&& f) {
  size_t row = 0;
  for (int t = 0; t < p.L[0]; t++) for (int z = 0; z < p.L[1]; z++) for (int y = 0; y < p.L[2]; y++, row++) {
    const size_t gsite = (((size_t)(p.O[0] + t) * p.G[1] + (p.O[1] + z)) * p.G[2] + (p.O[2] + y)) * p.G[3] + p.O[3];
    if (int rc = f(row, gsite)) return rc;
  }
  return 0;
}

int pread_all(int fd, void* buf, size_t n, off_t off) {
  char* b = static_cast<char*>(buf);
  while (n) { ssize_t r = pread(fd, b, n, off); if (r <= 0) return -1; b += r; off += r; n -= (size_t)r; }
  return 0;
}
int pwrite_all(int fd, const void* buf, size_t n, off_t off) {
  const char* b = static_cast<const char*>(buf);
  while (n) { ssize_t r = pwrite(fd, b, n, off); if (r <= 0) return -1; b += r; off += r; n -= (size_t)r; }
  return 0;
}

const size_t CONF_HEADER = 4 * sizeof(int32_t) + sizeof(double);

// length of the "<header> ... </header>\n" block at the start of the file, 0 if there is none, -1 if it does not end
long header_length(int fd) {
  std::string text;
  char buf[4096];
  off_t off = 0;
  const char* open_tag = "<header>\n";
  const char* close_tag = "</header>\n";
  for (;;) {
    ssize_t r = pread(fd, buf, sizeof buf, off);
    if (r < 0) return -1;
    text.append(buf, (size_t)r); off += r;
    if (text.size() >= strlen(open_tag) && text.compare(0, strlen(open_tag), open_tag) != 0) return 0;
    size_t pos = text.find(close_tag);
    if (pos != std::string::npos) return (long)(pos + strlen(close_tag));
    if (r == 0) return text.size() < strlen(open_tag) ? 0 : -1;
    if (text.size() > (1u << 22)) return -1;
  }
}

std::string format_header(const ddamg_hip_vector_header& h, const Part& p, int n) {
  // write_header, src/io.c:671-702 (BASIS0 is the reference's default Clifford basis, src/clifford.h:41)
  char line[512];
  std::string s = "<header>\n";
  auto add = [&](const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(line, sizeof line, fmt, ap); va_end(ap); s += line; };
  add("%s\n", h.vector_type ? h.vector_type : "");
  add("clifford basis: %s\n", "BASIS0:OPENQCD/DD-HMC BASIS");
  add("m0: %.14lf\n", h.m0);
  add("csw: %.14lf\n", h.csw);
  add("clov plaq: %.14lf\n", h.clov_plaq);
  add("hopp plaq: %.14lf\n", h.hopp_plaq);
  add("clov conf name: %s\n", h.clov_conf_name ? h.clov_conf_name : "");
  add("hopp conf name: %s\n", h.hopp_conf_name ? h.hopp_conf_name : "");
  add("X: %d\n", p.G[3]); add("Y: %d\n", p.G[2]); add("Z: %d\n", p.G[1]); add("T: %d\n", p.G[0]);
  add("X local: %d\n", p.L[3]); add("Y local: %d\n", p.L[2]); add("Z local: %d\n", p.L[1]); add("T local: %d\n", p.L[0]);
  add("number of vectors: %d\n", n);
  add("krylov subspace size: %d\n", 100);
  add("clifford basis: %s\n", "BASIS0:OPENQCD/DD-HMC BASIS");
  if (h.has_eigenvalues && h.eigenvalues) {
    s += "eigenvalues: ";
    for (int i = 0; i < 2 * n; i++) add("%.16lf ", h.eigenvalues[i]);
    s += "\n";
  }
  s += "</header>\n";
  return s;
}

bool at_origin(const Part& p) { return p.O[0] == 0 && p.O[1] == 0 && p.O[2] == 0 && p.O[3] == 0; }

}  // namespace

extern "C" {

const char* ddamg_hip_io_last_error(void) { return g_err; }

int ddamg_hip_conf_info(const char* path, int big_endian, int lattice_out[4], double* plaq_out) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) return fail("cannot open configuration '%s'", path);
  int32_t dims[4]; double plaq;
  const bool ok = pread_all(fd, dims, sizeof dims, 0) == 0 && pread_all(fd, &plaq, sizeof plaq, sizeof dims) == 0;
  close(fd);
  if (!ok) return fail("'%s' is shorter than a configuration header", path);
  if (big_endian) { swap4(dims, 4); swap8(&plaq, 1); }
  for (int mu = 0; mu < 4; mu++) lattice_out[mu] = dims[mu];
  if (plaq_out) *plaq_out = plaq;
  return 0;
}

int ddamg_hip_read_conf(const char* path, const int G[4], const int P[4], const int C[4], int big_endian, double* gauge_local, double* plaq_out) {
  Part p;
  if (!make_part(G, P, C, p)) return fail("read_conf: process grid does not divide the lattice / coordinates outside the grid");
  int dims[4]; double plaq;
  if (ddamg_hip_conf_info(path, big_endian, dims, &plaq)) return -1;
  for (int mu = 0; mu < 4; mu++)
    if (dims[mu] != G[mu]) return fail("configuration '%s' is %dx%dx%dx%d, expected %dx%dx%dx%d (T,Z,Y,X)", path, dims[0], dims[1], dims[2], dims[3], G[0], G[1], G[2], G[3]);
  int fd = open(path, O_RDONLY);
  if (fd < 0) return fail("cannot open configuration '%s'", path);
  const size_t row_doubles = (size_t)72 * p.L[3];
  int rc = for_rows(p, [&](size_t row, size_t gsite) {
    double* dst = gauge_local + row * row_doubles;
    if (pread_all(fd, dst, row_doubles * sizeof(double), (off_t)(CONF_HEADER + gsite * 72 * sizeof(double)))) return fail("configuration '%s' ends early", path);
    if (big_endian) swap8(dst, row_doubles);
    return 0;
  });
  close(fd);
  if (rc == 0 && plaq_out) *plaq_out = plaq;
  return rc;
}

// read_conf_multi (src/io.c:566-668): one file per process of the grid, named <base>.pt<T>pz<Z>py<Y>px<X>; each file holds the
// header of the GLOBAL lattice (extents asserted against it, :619-620; the plaquette must agree between the files, :630-632)
// followed by that process's links in its local lexicographic order
static std::string multi_name(const char* base, const int C[4]) {
  char post[128];
  snprintf(post, sizeof post, ".pt%dpz%dpy%dpx%d", C[0], C[1], C[2], C[3]);
  return std::string(base) + post;
}

int ddamg_hip_read_conf_multi(const char* base, const int G[4], const int P[4], const int C[4], int big_endian, double* gauge_local, double* plaq_out) {
  Part p;
  if (!make_part(G, P, C, p)) return fail("read_conf_multi: process grid does not divide the lattice / coordinates outside the grid");
  const std::string path = multi_name(base, C);
  int dims[4]; double plaq;
  if (ddamg_hip_conf_info(path.c_str(), big_endian, dims, &plaq)) return -1;
  for (int mu = 0; mu < 4; mu++)
    if (dims[mu] != G[mu]) return fail("configuration part '%s' belongs to a %dx%dx%dx%d lattice, expected %dx%dx%dx%d (T,Z,Y,X)", path.c_str(), dims[0], dims[1], dims[2], dims[3], G[0], G[1], G[2], G[3]);
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return fail("cannot open configuration part '%s'", path.c_str());
  const size_t n = p.Vloc * 72;
  const off_t size = lseek(fd, 0, SEEK_END);
  int rc = 0;
  if (size != (off_t)(CONF_HEADER + n * sizeof(double))) rc = fail("configuration part '%s' has %lld bytes, a %dx%dx%dx%d part has %lld", path.c_str(), (long long)size, p.L[0], p.L[1], p.L[2], p.L[3], (long long)(CONF_HEADER + n * sizeof(double)));
  else if (pread_all(fd, gauge_local, n * sizeof(double), (off_t)CONF_HEADER)) rc = fail("configuration part '%s' ends early", path.c_str());
  close(fd);
  if (rc == 0 && big_endian) swap8(gauge_local, n);
  if (rc == 0 && plaq_out) *plaq_out = plaq;
  return rc;
}

int ddamg_hip_write_conf_multi(const char* base, const int G[4], const int P[4], const int C[4], int big_endian, const double* gauge_local, double plaq) {
  Part p;
  if (!make_part(G, P, C, p)) return fail("write_conf_multi: process grid does not divide the lattice / coordinates outside the grid");
  const std::string path = multi_name(base, C);
  int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return fail("cannot create '%s'", path.c_str());
  int32_t dims[4] = {G[0], G[1], G[2], G[3]}; double pl = plaq;
  if (big_endian) { swap4(dims, 4); swap8(&pl, 1); }
  const size_t n = p.Vloc * 72;
  int rc = 0;
  if (pwrite_all(fd, dims, sizeof dims, 0) || pwrite_all(fd, &pl, sizeof pl, sizeof dims)) rc = fail("write to '%s' failed", path.c_str());
  if (rc == 0) {
    std::vector<double> tmp;
    const double* src = gauge_local;
    if (big_endian) { tmp.assign(gauge_local, gauge_local + n); swap8(tmp.data(), n); src = tmp.data(); }
    if (pwrite_all(fd, src, n * sizeof(double), (off_t)CONF_HEADER)) rc = fail("write to '%s' failed", path.c_str());
  }
  close(fd);
  return rc;
}

int ddamg_hip_write_conf(const char* path, const int G[4], const int P[4], const int C[4], int big_endian, const double* gauge_local, double plaq) {
  Part p;
  if (!make_part(G, P, C, p)) return fail("write_conf: process grid does not divide the lattice / coordinates outside the grid");
  int fd = open(path, O_WRONLY | O_CREAT, 0644);
  if (fd < 0) return fail("cannot create '%s'", path);
  int rc = 0;
  if (at_origin(p)) {
    int32_t dims[4] = {G[0], G[1], G[2], G[3]}; double pl = plaq;
    if (big_endian) { swap4(dims, 4); swap8(&pl, 1); }
    if (pwrite_all(fd, dims, sizeof dims, 0) || pwrite_all(fd, &pl, sizeof pl, sizeof dims)) rc = fail("write to '%s' failed", path);
  }
  const size_t row_doubles = (size_t)72 * p.L[3];
  std::vector<double> tmp(big_endian ? row_doubles : 0);
  if (rc == 0) rc = for_rows(p, [&](size_t row, size_t gsite) {
    const double* src = gauge_local + row * row_doubles;
    if (big_endian) { memcpy(tmp.data(), src, row_doubles * sizeof(double)); swap8(tmp.data(), row_doubles); src = tmp.data(); }
    if (pwrite_all(fd, src, row_doubles * sizeof(double), (off_t)(CONF_HEADER + gsite * 72 * sizeof(double)))) return fail("write to '%s' failed", path);
    return 0;
  });
  close(fd);
  return rc;
}

int ddamg_hip_read_vectors(const char* path, const int G[4], const int P[4], const int C[4], int n, int big_endian, double* vectors_local) {
  Part p;
  if (!make_part(G, P, C, p)) return fail("read_vectors: process grid does not divide the lattice / coordinates outside the grid");
  if (n < 1) return fail("read_vectors: n must be positive");
  int fd = open(path, O_RDONLY);
  if (fd < 0) return fail("cannot open vector file '%s'", path);
  const long hl = header_length(fd);
  if (hl < 0) { close(fd); return fail("'%s': header does not end with </header>", path); }
  if (hl == 0 && n != 1) { close(fd); return fail("'%s': a file without header holds a single vector (src/io.c:735-743)", path); }
  const off_t fsize = lseek(fd, 0, SEEK_END);
  if ((size_t)fsize < (size_t)hl + (size_t)n * p.Vglob * 24 * sizeof(double)) { close(fd); return fail("'%s' holds fewer than %d vectors of this lattice", path, n); }
  const size_t row_doubles = (size_t)24 * p.L[3];
  int rc = 0;
  for (int j = 0; j < n && rc == 0; j++)
    rc = for_rows(p, [&](size_t row, size_t gsite) {
      double* dst = vectors_local + ((size_t)j * p.Vloc * 24) + row * row_doubles;
      if (pread_all(fd, dst, row_doubles * sizeof(double), (off_t)(hl + ((size_t)j * p.Vglob + gsite) * 24 * sizeof(double)))) return fail("'%s' ends early", path);
      if (big_endian) swap8(dst, row_doubles);
      return 0;
    });
  close(fd);
  return rc;
}

int ddamg_hip_write_vectors(const char* path, const int G[4], const int P[4], const int C[4], int n, int big_endian,
                            const ddamg_hip_vector_header* header, const double* vectors_local) {
  Part p;
  if (!make_part(G, P, C, p)) return fail("write_vectors: process grid does not divide the lattice / coordinates outside the grid");
  if (n < 1) return fail("write_vectors: n must be positive");
  if (!header && n != 1) return fail("write_vectors: several vectors need a header (vector_io_single_file always writes one)");
  // every process formats the same header text, so all of them know where the data starts without talking to each other
  const std::string h = header ? format_header(*header, p, n) : std::string();
  int fd = open(path, O_WRONLY | O_CREAT, 0644);
  if (fd < 0) return fail("cannot create '%s'", path);
  int rc = 0;
  if (at_origin(p) && !h.empty() && pwrite_all(fd, h.data(), h.size(), 0)) rc = fail("write to '%s' failed", path);
  const size_t row_doubles = (size_t)24 * p.L[3];
  std::vector<double> tmp(big_endian ? row_doubles : 0);
  for (int j = 0; j < n && rc == 0; j++)
    rc = for_rows(p, [&](size_t row, size_t gsite) {
      const double* src = vectors_local + ((size_t)j * p.Vloc * 24) + row * row_doubles;
      if (big_endian) { memcpy(tmp.data(), src, row_doubles * sizeof(double)); swap8(tmp.data(), row_doubles); src = tmp.data(); }
      if (pwrite_all(fd, src, row_doubles * sizeof(double), (off_t)(h.size() + ((size_t)j * p.Vglob + gsite) * 24 * sizeof(double)))) return fail("write to '%s' failed", path);
      return 0;
    });
  close(fd);
  return rc;
}

}  // extern "C"
