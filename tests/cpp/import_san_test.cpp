// Sanitizer harness for the HOST half of hifamd_load / hifamd_add_level / hifamd_set_dense / hifamd_finalize
// (hifir_amd/csrc/import.hpp + host.hpp -- the very code engine.hip compiles, minus the uploads).
//
// Why: round 1 saw an intermittent corruption of two entries of E's row pointer on the host (complex `young1c`,
// handle created by hifamd_load while a first handle was alive), attributed to a second OpenMP runtime but never
// shown.  The GPU pool has no sanitizers, so the host path is exercised here, on the CPU, the way that test did:
//   * several handles alive at once, loaded from the same file, derived arrays compared bit for bit;
//   * the dense factorization (complex QRCP 228^2 for young1c), the explicit operators and every block inverse
//     with the thread pool active;
//   * two loads running concurrently on two threads (distinct handles from distinct threads);
//   * the adjoint hierarchy;
//   * truncated and bit-flipped files: must be refused with HIFAMD_BAD_PREC / MISMATCHED_SIZES, never read out of bounds.
// Build: g++ -std=c++17 -O1 -g -fsanitize=address,undefined  |  -fsanitize=thread ; -DHIFAMD_TEST_OPENMP -fopenmp
// replaces the std::thread loop by the OpenMP loops the library had before commit b9dba02.
// Usage: import_san_test file.hifamd [more files...]; exit code 0 = clean.
#include "import.hpp"

#include <thread>

using namespace hifamd;

template <class T>
struct Sink {  // what Engine<T> does on import, without a device
  HostHierarchy<T> host;
  BandOptions opt;
  bool analyze = true;
  std::vector<LevelAnalysis<T>> cached;  // the analysis trailer of the file being loaded (load_bytes), if adopted
  int adopted = 0;
  void add_level(int64_t m, int64_t n, const int64_t *Lcp, const int32_t *Lri, const T *Lv, const int64_t *Ucp,
                 const int32_t *Uri, const T *Uv, const int64_t *Ecp, const int32_t *Eri, const T *Ev, int64_t F_ncols,
                 const int64_t *Fcp, const int32_t *Fri, const T *Fv, const T *d, const double *s, const double *t,
                 const int32_t *p, const int32_t *p_inv, const int32_t *q, const int32_t *q_inv) {
    const int64_t parent_nm = host.levels.empty() ? -1 : host.levels.back().n - host.levels.back().m;
    HostLevel<T> H = import_level<T>(parent_nm, m, n, Lcp, Lri, Lv, Ucp, Uri, Uv, Ecp, Eri, Ev, F_ncols, Fcp, Fri, Fv, d, s, t,
                                     p, p_inv, q, q_inv);
    const size_t li = host.levels.size();
    if (li < cached.size() && adopt_analysis(H, cached[li], opt))
      ++adopted;
    else if (analyze)
      analyze_level(H, opt);
    seal_level(H);  // (Engine::add_level does)
    host.levels.push_back(std::move(H));
  }
  // (hostile-file runs -- analyze == false -- only care that the block arrives with the right size)
  void set_dense(int64_t nd, const T *mat, double cond) {
    if (analyze) dense_factorize(host.dense, mat, nd, cond);
    host.has_dense = true;
  }
  void set_dense_symm(int64_t nd, const T *mat, int spd) {
    if (analyze) dense_factorize_symm(host.dense, mat, nd, spd);
    host.has_dense = true;
  }
  void set_dense_lup(int64_t nd, const T *mat) {
    if (analyze) dense_factorize_lup(host.dense, mat, nd);
    host.has_dense = true;
  }
};

static std::vector<unsigned char> slurp(const char *path) {
  std::FILE *f = std::fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  std::vector<unsigned char> b;
  unsigned char buf[1 << 16];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + k);
  std::fclose(f);
  return b;
}

template <class T>
static void load_bytes(const std::vector<unsigned char> &bytes, Sink<T> &S, bool trailer = false) {
  std::FILE *f = fmemopen((void *)bytes.data(), bytes.size(), "rb");
  if (!f) throw std::runtime_error("fmemopen failed");
  try {
    if (std::fseek(f, 16, SEEK_SET) != 0) throw Error(kBadPrec, "truncated hierarchy file");
    if (trailer) load_analysis<T>(f, S.opt, S.cached);  // (what Engine::load does first)
    load_hierarchy<T>(f, S);
    S.cached.clear();
  } catch (...) {
    std::fclose(f);
    throw;
  }
  std::fclose(f);
}

template <class V>
static bool same(const std::vector<V> &a, const std::vector<V> &b) {
  return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(V)) == 0);
}
template <class T>
static bool same_csr(const Csr<T> &a, const Csr<T> &b) {
  return same(a.ptr, b.ptr) && same(a.col, b.col) && same(a.val, b.val) && same(a.rowid, b.rowid);
}
static bool same_plan(const BandPlan &a, const BandPlan &b) {
  return same(a.order, b.order) && same(a.grp_slot_ptr, b.grp_slot_ptr) && same(a.wg_grp_ptr, b.wg_grp_ptr) &&
         same(a.band_wg_ptr, b.band_wg_ptr) && same(a.srcslot, b.srcslot) && same(a.split, b.split) &&
         same(a.band_dense, b.band_dense) && same(a.band_fused, b.band_fused) && same(a.blk_slot0, b.blk_slot0) &&
         same(a.blk_slot1, b.blk_slot1) && same(a.blk_inv_off, b.blk_inv_off) && same(a.band_blk_ptr, b.band_blk_ptr) &&
         same(a.grp_inv_off, b.grp_inv_off) && same(a.band_prefix, b.band_prefix) && same(a.band_cd, b.band_cd) &&
         same(a.band_old, b.band_old) && same(a.csplit, b.csplit) && same(a.mid_k, b.mid_k) && same(a.mid_lrow, b.mid_lrow) &&
         same(a.cd_desc, b.cd_desc) && same(a.own_k, b.own_k) && same(a.own_lsrc, b.own_lsrc) && same(a.own_lvl, b.own_lvl) &&
         same(a.own_rptr, b.own_rptr) && a.cd_sparse == b.cd_sparse;
}
template <class T>
static int compare(const Sink<T> &A, const Sink<T> &B, const char *what) {
  int bad = 0;
  if (A.host.levels.size() != B.host.levels.size()) return 1;
  for (size_t l = 0; l < A.host.levels.size(); ++l) {
    const HostLevel<T> &x = A.host.levels[l], &y = B.host.levels[l];
    if (!same_csr(x.Lr, y.Lr) || !same_csr(x.Ur, y.Ur) || !same_csr(x.Er, y.Er) || !same_csr(x.Fr, y.Fr)) ++bad;
    if (!same_plan(x.Lp, y.Lp) || !same_plan(x.Up, y.Up)) ++bad;
    if (!same(x.d, y.d) || !same(x.s, y.s) || !same(x.t, y.t) || !same(x.p, y.p) || !same(x.q_inv, y.q_inv)) ++bad;
  }
  if (A.host.has_dense != B.host.has_dense) ++bad;
  if (A.host.has_dense)
    if (A.host.dense.rank != B.host.dense.rank || !same(A.host.dense.jpvt0, B.host.dense.jpvt0) || !same(A.host.dense.qr, B.host.dense.qr) ||
        !same(A.host.dense.QH, B.host.dense.QH) || !same(A.host.dense.Rinv, B.host.dense.Rinv))
      ++bad;
  if (bad) std::fprintf(stderr, "MISMATCH between two handles built from the same file (%s): %d arrays\n", what, bad);
  return bad;
}

// the host part of hifamd_finalize: invariants, then every block inverse (thread pool active)
template <class T>
static int finalize_host(Sink<T> &S, bool adjoint = false) {
  int bad = 0;
  for (size_t l = 0; l < S.host.levels.size(); ++l) bad += verify_level(S.host.levels[l], l, adjoint);  // (0: untouched)
  for (size_t l = 0; l < S.host.levels.size(); ++l) check_level_invariants(S.host.levels[l], l, adjoint);
  std::vector<double> ops;
  for (auto &H : S.host.levels)
    for (int tri = 0; tri < 2; ++tri) {
      BandPlan &P = tri ? H.Up : H.Lp;
      const Csr<T> &A = tri ? H.Ur : H.Lr;
      for (size_t q = 0; q < P.blk_slot0.size(); ++q) {
        ops.assign((size_t)dense_block_elems(P.blk_slot1[q] - P.blk_slot0[q], sizeof(T) != sizeof(double)), 0.0);
        if (!(build_dense_block(P, A, q, ops.data()) >= 1.0)) ++bad;
      }
    }
  for (size_t l = 0; l < S.host.levels.size(); ++l) check_level_invariants(S.host.levels[l], l, adjoint);
  for (auto &H : S.host.levels) {  // the streamed-sink form of sparse-own U triangles (k_band_us): built and re-derived
    UsPlan<T> U;
    build_us_plan(H.Up, H.Ur, U);
    if (U.any) check_us_plan(H.Up, H.Ur, U);
  }
  return bad;
}

template <class T>
static int run_file(const char *path, const std::vector<unsigned char> &bytes, int64_t dense_block) {
  int bad = 0;
  Sink<T> A, B, C;
  A.opt.dense_block = B.opt.dense_block = C.opt.dense_block = dense_block;
  A.opt.dense_min_rows = B.opt.dense_min_rows = C.opt.dense_min_rows = 8;  // small fixtures still get block-dense bands
  A.opt.thin_rows = B.opt.thin_rows = C.opt.thin_rows = 12;
  load_bytes(bytes, A);          // first handle, stays alive
  bad += finalize_host(A);
  load_bytes(bytes, B);          // second handle while the first one is alive (the round-1 scenario)
  bad += finalize_host(B);
  {  // sparse-own components on every shallow triangle (level 0 of the large hierarchies), here on the small fixtures
    Sink<T> W;
    W.opt = A.opt;
    W.opt.cd_sparse_min_rows = 0;
    load_bytes(bytes, W);
    bad += finalize_host(W);
  }
  {  // the host copy changed behind the library's back (rounds 1 and 4 met two overwritten row pointers of E): E / F's row
     // forms are rebuilt from the imported arrays and reported, anything else is refused
    Sink<T> V;
    V.opt = A.opt;
    load_bytes(bytes, V);
    for (size_t l = 0; l < V.host.levels.size(); ++l) {
      HostLevel<T> &H = V.host.levels[l];
      if (H.Er.ptr.size() >= 4) {
        const Csr<T> keep = H.Er;
        H.Er.ptr[H.Er.ptr.size() - 3] = 0;
        H.Er.ptr[H.Er.ptr.size() - 2] = 1072693248;  // (the high word of 1.0)
        if (verify_level(H, l, false) != 1 || !same_csr(H.Er, keep)) ++bad, std::fprintf(stderr, "%s: level %zu: E not rebuilt\n", path, l);
        if (verify_level(H, l, false) != 0) ++bad, std::fprintf(stderr, "%s: level %zu: still different after the rebuild\n", path, l);
      }
      if (!H.Fr.val.empty()) {
        const Csr<T> keep = H.Fr;
        H.Fr.val[H.Fr.val.size() / 2] = T(12345.0);
        if (verify_level(H, l, false) != 1 || !same_csr(H.Fr, keep)) ++bad, std::fprintf(stderr, "%s: level %zu: F not rebuilt\n", path, l);
      }
      if (!H.s.empty()) {
        const double was = H.s[0];
        H.s[0] = was + 1.0;
        bool thrown = false;
        try {
          verify_level(H, l, false);
        } catch (const Error &e) {
          thrown = std::string(e.what()).find(" s ") != std::string::npos || std::string(e.what()).find(": s") != std::string::npos;
        }
        if (!thrown) ++bad, std::fprintf(stderr, "%s: level %zu: a changed scale vector was not refused by name\n", path, l);
        H.s[0] = was;
      }
      if (!H.E.vals.empty() && !H.Er.val.empty()) {  // imported AND derived array changed: nothing to rebuild from
        H.E.vals[0] = T(777.0);
        H.Er.val[0] = T(777.0);
        bool thrown = false;
        try {
          verify_level(H, l, false);
        } catch (const Error &) {
          thrown = true;
        }
        if (!thrown) ++bad, std::fprintf(stderr, "%s: level %zu: damaged imported arrays accepted\n", path, l);
      }
    }
  }
  bad += compare(A, B, "sequential");
  // two more loads concurrently on two threads
  Sink<T> D;
  D.opt = C.opt;
  std::exception_ptr e1, e2;
  std::thread t1([&] { try { load_bytes(bytes, C); finalize_host(C); } catch (...) { e1 = std::current_exception(); } });
  std::thread t2([&] { try { load_bytes(bytes, D); finalize_host(D); } catch (...) { e2 = std::current_exception(); } });
  t1.join();
  t2.join();
  if (e1) std::rethrow_exception(e1);
  if (e2) std::rethrow_exception(e2);
  bad += compare(A, C, "concurrent 1") + compare(A, D, "concurrent 2");
  // adjoint hierarchy of the first handle
  if (!A.host.levels[0].q.empty() && !A.host.levels[0].p_inv.empty()) {
    Sink<T> J;
    J.opt = A.opt;
    for (const auto &P : A.host.levels) {
      HostLevel<T> H = adjoint_level(P);
      analyze_level(H, J.opt);
      J.host.levels.push_back(std::move(H));
    }
    bad += finalize_host(J, true);
    if (A.host.has_dense && A.host.dense.kind == 0) {
      HostDense<T> Dn;
      Dn.n = A.host.dense.n, Dn.rank = A.host.dense.rank, Dn.qr = A.host.dense.qr, Dn.tau = A.host.dense.tau;
      Dn.jpvt0 = A.host.dense.jpvt0;
      dense_adjoint_ops(Dn);
    }
  }
  bad += compare(A, B, "after the adjoint build");
  // save -> identical bytes
  {
    std::vector<unsigned char> out(bytes.size() + 64);
    std::FILE *f = fmemopen(out.data(), out.size(), "wb");
    save_hierarchy(f, A.host);
    const long len = std::ftell(f);
    std::fclose(f);
    if (len != (long)bytes.size() - 16 || std::memcmp(out.data(), bytes.data() + 16, (size_t)len) != 0) {
      std::fprintf(stderr, "%s: save_hierarchy does not reproduce the file\n", path);
      ++bad;
    }
  }
  // the analysis trailer (hifamd_save_ex): adopted -> the same derived arrays as analyzed; then hostile trailers whose
  // CHECKSUM IS RIGHT (so that the size / range checks behind it are what refuses them): words of the trailer overwritten
  // with nasty values -- a load must either ignore the trailer, adopt a still-consistent one, or refuse at the
  // invariants of finalize; never read out of bounds
  {
    std::vector<unsigned char> out(bytes.size() * 6 + (1 << 20));
    std::FILE *f = fmemopen(out.data(), out.size(), "wb");
    if (std::fwrite(bytes.data(), 1, 16, f) != 16) throw std::runtime_error("fmemopen write failed");
    save_hierarchy(f, A.host);
    save_analysis(f, A.host, A.opt);
    const long len = std::ftell(f);
    std::fclose(f);
    out.resize((size_t)len);
    Sink<T> K;
    K.opt = A.opt;
    load_bytes(out, K, true);
    if (K.adopted != (int)A.host.levels.size()) {
      std::fprintf(stderr, "%s: analysis trailer adopted for %d of %zu levels\n", path, K.adopted, A.host.levels.size());
      ++bad;
    }
    bad += finalize_host(K);
    bad += compare(A, K, "adopted analysis");
    // a cached component must fit the LDS block the kernels size from the planner OPTIONS (a re-sealed trailer with a
    // larger one used to be adopted): the same plan is refused under options one row short of its largest component, and
    // check_band_plan -- what finalize runs -- refuses it as well
    for (size_t l = 0; l < K.host.levels.size(); ++l)
      for (int tri = 0; tri < 2; ++tri) {
        const HostLevel<T> &H = K.host.levels[l];
        const BandPlan &P = tri ? H.Up : H.Lp;
        const Csr<T> &M = tri ? H.Ur : H.Lr;
        int32_t mx = 0;
        for (size_t b = 0; b < P.band_cd.size(); ++b)
          if (P.band_cd[b])
            for (int32_t c = P.wg_grp_ptr[(size_t)P.band_wg_ptr[b]]; c < P.wg_grp_ptr[(size_t)P.band_wg_ptr[b + 1]]; ++c)
              mx = std::max(mx, P.grp_slot_ptr[(size_t)c + 1] - P.grp_slot_ptr[(size_t)c]);
        if (mx < 2) continue;
        BandOptions small = A.opt;
        (P.cd_sparse ? small.cd_sparse_rows : small.cd_rows) = mx - 1;
        if (!cached_plan_ok(P, M, H.m, (int64_t)M.col.size(), A.opt)) ++bad, std::fprintf(stderr, "%s: a computed plan is not accepted as a cached one\n", path);
        if (cached_plan_ok(P, M, H.m, (int64_t)M.col.size(), small)) ++bad, std::fprintf(stderr, "%s: oversized cached component accepted\n", path);
        bool thrown = false;
        try {
          check_band_plan(P, M, "test", l, &small);
        } catch (const Error &) {
          thrown = true;
        }
        if (!thrown) ++bad, std::fprintf(stderr, "%s: check_band_plan accepts a component larger than the options allow\n", path);
      }
    const size_t t0 = bytes.size(), t1 = out.size() - 24;  // the trailer's bytes
    auto reseal = [&](std::vector<unsigned char> &b) {
      HashIo h{nullptr};
      h.mix(b.data() + t0, t1 - t0);
      const int64_t hv = (int64_t)h.value();
      std::memcpy(&b[b.size() - 8], &hv, 8);
    };
    {  // (the footer's checksum is the one reseal computes)
      std::vector<unsigned char> chk = out;
      reseal(chk);
      if (chk != out) ++bad, std::fprintf(stderr, "%s: trailer checksum not reproduced\n", path);
    }
    int t_adopted = 0, t_ignored = 0, t_refused = 0;
    uint64_t rs = 88172645463325252ull;
    auto rnd = [&] { rs ^= rs << 13, rs ^= rs >> 7, rs ^= rs << 17; return rs; };
    const int32_t nasty32[] = {-1, 0, 1, 255, 256, 1 << 30, 2147483647, 7};
    std::vector<unsigned char> mut = out;
    for (int trial = 0; trial < 800; ++trial) {
      const size_t off = t0 + (size_t)(rnd() % (t1 - t0 - 4)) & ~(size_t)3;
      unsigned char keep[4];
      std::memcpy(keep, &mut[off], 4);
      const int32_t v = (trial & 7) == 7 ? (int32_t)(keep[0] + 1) : nasty32[rnd() % 8];
      std::memcpy(&mut[off], &v, 4);
      reseal(mut);
      Sink<T> S;
      S.opt = A.opt;
      S.analyze = false;  // (a level whose trailer is refused stays unanalyzed here: analyze_level has its own runs above)
      try {
        load_bytes(mut, S, true);
        if (S.adopted == (int)S.host.levels.size())
          for (size_t l = 0; l < S.host.levels.size(); ++l) check_level_invariants(S.host.levels[l], l);
        (S.adopted ? t_adopted : t_ignored)++;
      } catch (const Error &e) {
        if (e.code < 1 || e.code > 4) ++bad;
        ++t_refused;
      }
      std::memcpy(&mut[off], keep, 4);
    }
    std::fprintf(stderr, "%s: hostile trailers with a valid checksum: %d ignored, %d adopted, %d refused at the invariants\n", path,
                 t_ignored, t_adopted, t_refused);
  }
  // hostile files: truncation at ~200 points, and every 8-byte word of the first 4 KB + a sample of the rest
  // overwritten with a few nasty values -- must throw Error (or load cleanly when the word is plain data)
  int refused = 0, accepted = 0;
  auto attempt = [&](const std::vector<unsigned char> &b) {
    Sink<T> S;
    S.analyze = false;
    try {
      load_bytes(b, S);
      ++accepted;
    } catch (const Error &e) {
      if (e.code < 1 || e.code > 4) ++bad;
      ++refused;
    }
  };
  const size_t step = std::max<size_t>(8, bytes.size() / 200);
  for (size_t cut = 16; cut < bytes.size(); cut += step) attempt(std::vector<unsigned char>(bytes.begin(), bytes.begin() + cut));
  const int64_t nasty[] = {-1, 0, 1, (int64_t)1 << 33, (int64_t)1 << 62, 7};
  std::vector<unsigned char> mut = bytes;
  for (size_t off = 16; off + 8 <= bytes.size(); off += (off < 4096 ? 8 : std::max<size_t>(8, (bytes.size() / 300) & ~(size_t)7))) {
    int64_t keep;
    std::memcpy(&keep, &mut[off], 8);
    for (int64_t v : nasty) {
      std::memcpy(&mut[off], &v, 8);
      attempt(mut);
    }
    std::memcpy(&mut[off], &keep, 8);
  }
  {  // a few dozen bytes that CLAIM a 2^31-row level: every count is bounded by what the file still holds, so this is
     // refused before anything of that size is allocated (under ASan a 16 GB request would abort the run)
    std::vector<unsigned char> tiny(bytes.begin(), bytes.begin() + 16);
    const int64_t big = 2147483647;
    const int64_t words[] = {1, 0, big, big, 0, big, big, big + 1, 0, 0, 0, 0};
    tiny.insert(tiny.end(), (const unsigned char *)words, (const unsigned char *)words + sizeof(words));
    const int before = refused;
    attempt(tiny);
    if (refused != before + 1) ++bad;
  }
  std::fprintf(stderr, "%s: levels=%zu dense=%d blocks(L0)=%zu  hostile files: %d refused, %d loaded  -> %s\n", path,
               A.host.levels.size(), (int)A.host.has_dense, A.host.levels[0].Lp.blk_slot0.size(), refused, accepted,
               bad ? "FAILED" : "ok");
  return bad;
}

int main(int argc, char **argv) {
  int bad = 0;
  try {
    for (int a = 1; a < argc; ++a) {
      const std::vector<unsigned char> bytes = slurp(argv[a]);
      if (bytes.size() < 16 || std::memcmp(bytes.data(), "HIFAMD1", 8) != 0) throw std::runtime_error("not a hierarchy file");
      int64_t vt;
      std::memcpy(&vt, &bytes[8], 8);
      for (int64_t db : {(int64_t)64, (int64_t)0})
        bad += vt == 1 ? run_file<zdouble>(argv[a], bytes, db) : run_file<double>(argv[a], bytes, db);
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
    return 2;
  }
  return bad ? 1 : 0;
}
