// Development tool (host only, no GPU): loads a hierarchy file written by hifamd_save, analyzes every level with the
// engine's default planner options (or the HIFIR_AMD_* overrides below) and writes, per band and per component, the
// quantities the band-time model of tests/band_model.py prices: rows, entries the band kernel walks itself, entries
// carried by the previous launch, distinct sources, 16x4 coefficient tiles, the longest wave chunk, own nonzeros.
//   g++ -O2 -std=c++17 -pthread -I hifir_amd/csrc tests/cpp/plan_model.cpp -o /tmp/plan_model
//   /tmp/plan_model hier.hifamd > plan.jsonl
#include "import.hpp"
#include <set>
using namespace hifamd;

static int env_int(const char *name, int dflt) {
  const char *e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}

struct Sink {
  BandOptions opt;
  int64_t parent_nm = -1;
  size_t level_no = 0;
  void add_level(int64_t m, int64_t n, const int64_t *Lcp, const int32_t *Lri, const double *Lv, const int64_t *Ucp,
                 const int32_t *Uri, const double *Uv, const int64_t *Ecp, const int32_t *Eri, const double *Ev, int64_t fn,
                 const int64_t *Fcp, const int32_t *Fri, const double *Fv, const double *d, const double *s, const double *t,
                 const int32_t *p, const int32_t *p_inv, const int32_t *q, const int32_t *q_inv) {
    HostLevel<double> H = import_level<double>(parent_nm, m, n, Lcp, Lri, Lv, Ucp, Uri, Uv, Ecp, Eri, Ev, fn, Fcp, Fri, Fv, d, s, t, p,
                                               p_inv, q, q_inv);
    parent_nm = n - m;
    analyze_level(H, opt, false, level_no);
    dump(H);
    ++level_no;
  }
  void set_dense(int64_t nd, const double *, double) { std::printf("{\"dense_n\": %ld}\n", (long)nd); }
  void set_dense_symm(int64_t nd, const double *, int) { std::printf("{\"dense_n\": %ld}\n", (long)nd); }
  void set_dense_lup(int64_t nd, const double *) { std::printf("{\"dense_n\": %ld}\n", (long)nd); }

  void dump(const HostLevel<double> &H) {
    // distinct (row block, column) pairs of E and F for blocks of 16 / 32 / 64 rows: what a tiled Schur product gathers
    long dist[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int which = 0; which < 2; ++which) {
      const Csr<double> &M = which ? H.Fr : H.Er;
      for (int bi = 0; bi < 3; ++bi) {
        const int64_t bs = 16 << bi;
        std::vector<int32_t> u;
        for (int64_t r0 = 0; r0 < M.nrows; r0 += bs) {
          const int64_t r1 = std::min<int64_t>(M.nrows, r0 + bs);
          u.assign(M.col.begin() + M.ptr[(size_t)r0], M.col.begin() + M.ptr[(size_t)r1]);
          std::sort(u.begin(), u.end());
          dist[which][bi] += (long)(std::unique(u.begin(), u.end()) - u.begin());
        }
      }
    }
    std::printf("{\"level\": %zu, \"m\": %ld, \"n\": %ld, \"nnzE\": %zu, \"nnzF\": %zu, \"top_n\": %ld, \"distinctE\": [%ld, %ld, %ld], "
                "\"distinctF\": [%ld, %ld, %ld]}\n", level_no, (long)H.m, (long)H.n, H.Er.col.size(), H.Fr.col.size(), (long)H.top_n,
                dist[0][0], dist[0][1], dist[0][2], dist[1][0], dist[1][1], dist[1][2]);
    for (int tri = 0; tri < 2; ++tri) {
      const BandPlan &P = tri ? H.Up : H.Lp;
      const Csr<double> &A = tri ? H.Ur : H.Lr;
      CtTiles Tw;
      build_ct_tiles(P, A, Tw);
      for (int64_t b = 0; b < P.nbands(); ++b) {
        const int32_t g0 = P.band_wg_ptr[(size_t)b], g1 = P.band_wg_ptr[(size_t)b + 1];
        const int32_t c0 = P.wg_grp_ptr[(size_t)g0], c1 = P.wg_grp_ptr[(size_t)g1];
        const int32_t s0 = P.grp_slot_ptr[(size_t)c0], s1 = P.grp_slot_ptr[(size_t)c1];
        int64_t carried = 0;
        for (int32_t s = s0; s < s1; ++s) carried += P.split[(size_t)s] - A.ptr[(size_t)s];
        std::printf("{\"level\": %zu, \"tri\": \"%c\", \"band\": %ld, \"rows\": %d, \"nnz\": %d, \"wgs\": %d, \"cd\": %d, \"dense\": %d, "
                    "\"prefix\": %d, \"fused\": %d, \"sparse\": %d, \"carried\": %ld",
                    level_no, tri ? 'U' : 'L', (long)b, s1 - s0, A.ptr[(size_t)s1] - A.ptr[(size_t)s0], g1 - g0, (int)P.band_cd[(size_t)b],
                    (int)P.band_dense[(size_t)b], (int)P.band_prefix[(size_t)b], (int)P.band_fused[(size_t)b], (int)P.cd_sparse, (long)carried);
        {  // what the band's launch reads of explicit inverses: per component the lower strips of its strip-major operand
          double inv_bytes = 0.0;
          if (P.band_cd[(size_t)b] && !P.cd_sparse)
            for (int32_t c = c0; c < c1; ++c) {
              const int32_t nb = P.grp_slot_ptr[(size_t)c + 1] - P.grp_slot_ptr[(size_t)c];
              for (int32_t st = 0; st * 16 < nb; ++st) inv_bytes += 16.0 * 8.0 * (double)(((std::min(nb, 16 * (st + 1)) + 31) / 32) * 32);
            }
          std::printf(", \"inv_bytes\": %.0f, \"top_band\": %d", inv_bytes, (int)((tri ? H.top_bandU : H.top_bandL) == (int32_t)b && H.top_n > 0));
        }
        if (!P.band_cd[(size_t)b]) {
          std::printf("}\n");
          continue;
        }
        if (!Tw.sptr.empty()) {  // coefficient tiles of the band (what k_band_ct multiplies) and its longest wave
          int64_t tw = 0;
          for (int32_t c = c0; c < c1; ++c) {
            const int32_t sp0 = Tw.desc[(size_t)c * kCdDescWords + 20];
            const int32_t S = (P.grp_slot_ptr[(size_t)c + 1] - P.grp_slot_ptr[(size_t)c] + 15) / 16;
            tw += Tw.sptr[(size_t)(sp0 + S)] - Tw.sptr[(size_t)sp0];
          }
          std::printf(", \"ct_tiles\": %ld, \"ct_wave_max\": %d", (long)tw, Tw.band_wave_tiles[(size_t)b]);
        }
        // per component: [rows, walked entries, distinct sources, tiles (16-row strips x 4 distinct sources), tiles of 32-row
        // strips, longest wave chunk, own nonzeros, depth levels (sparse plans)]
        std::printf(", \"comps\": [");
        std::vector<int32_t> u;
        for (int32_t c = c0; c < c1; ++c) {
          const int32_t a = P.grp_slot_ptr[(size_t)c], e = P.grp_slot_ptr[(size_t)c + 1], nb = e - a;
          int64_t walked = 0, own = 0, tiles16 = 0, tiles32 = 0;
          std::set<int32_t> all;
          for (int32_t r0 = a; r0 < e; r0 += 16) {
            u.clear();
            for (int32_t s = r0; s < std::min(e, r0 + 16); ++s)
              for (int32_t k = P.split[(size_t)s]; k < P.csplit[(size_t)s]; ++k) u.push_back(P.srcslot[(size_t)k]);
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            tiles16 += ((int64_t)u.size() + 3) / 4;
            all.insert(u.begin(), u.end());
          }
          for (int32_t r0 = a; r0 < e; r0 += 32) {
            u.clear();
            for (int32_t s = r0; s < std::min(e, r0 + 32); ++s)
              for (int32_t k = P.split[(size_t)s]; k < P.csplit[(size_t)s]; ++k) u.push_back(P.srcslot[(size_t)k]);
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            tiles32 += ((int64_t)u.size() + 3) / 4;
          }
          for (int32_t s = a; s < e; ++s) {
            walked += P.csplit[(size_t)s] - P.split[(size_t)s];
            own += A.ptr[(size_t)s + 1] - P.csplit[(size_t)s];
          }
          int32_t ck = 0;
          const uint16_t *wm = reinterpret_cast<const uint16_t *>(&P.cd_desc[(size_t)c * kCdDescWords + 11]);
          for (int q = 0; q < 16; ++q) ck = std::max<int32_t>(ck, wm[q + 1] - wm[q]);
          const int32_t nl = P.cd_sparse ? P.cd_desc[(size_t)c * kCdDescWords + 24] : 0;
          std::printf("%s[%d,%ld,%zu,%ld,%ld,%d,%ld,%d]", c == c0 ? "" : ",", nb, (long)walked, all.size(), (long)tiles16, (long)tiles32, ck, (long)own, nl);
        }
        // workgroup -> number of components (bags share a workgroup)
        std::printf("], \"wg_comps\": [");
        for (int32_t g = g0; g < g1; ++g) std::printf("%s%d", g == g0 ? "" : ",", P.wg_grp_ptr[(size_t)g + 1] - P.wg_grp_ptr[(size_t)g]);
        std::printf("]}\n");
      }
    }
  }
};

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  char magic[8];
  int64_t vt = -1;
  if (std::fread(magic, 8, 1, f) != 1 || std::fread(&vt, 8, 1, f) != 1 || vt != 0) return 4;
  Sink S;
  BandOptions &o = S.opt;
  o.max_wg_rows = 16384;
  o.dense_block = env_int("HIFIR_AMD_DENSE_BLOCK", 2048);
  o.fuse_reorder = o.dense_block > 0;
  o.fuse_max_wgs = env_int("HIFIR_AMD_BAND_FUSE_WGS", 512);
  o.cd_fuse_max_wgs = env_int("HIFIR_AMD_CD_FUSE_WGS", 600);
  o.cd_rows = env_int("HIFIR_AMD_CD_ROWS", 128);
  o.cd_max_nnz = env_int("HIFIR_AMD_CD_NNZ", 4000);
  o.cd_sparse_rows = env_int("HIFIR_AMD_CD_SPARSE_ROWS", 192);
  o.top_max = env_int("HIFIR_AMD_TOP_ROWS", 4096);
  o.top_few_wgs = env_int("HIFIR_AMD_TOP_WGS", 96);
  try {
    load_hierarchy<double>(f, S);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
