// CPU test of the host-side analysis (hifir_amd/csrc/host.hpp): CCS -> CSR, level schedule, band plan, slot
// permutation, finish_band_plan (incl. the nonzero reordering of block-dense bands), block cutting and block
// inverses, on synthetic strict triangles with long thin tails.  The analysis re-checks every dependency against
// the executed order itself (finish_band_plan throws otherwise); this program additionally verifies the block
// inverses against the rows they were built from.  Built and run by tests/test_abi_and_host.py (g++, no GPU);
// meant to be run under -fsanitize=address,undefined as well.
#include "host.hpp"
#include <random>
using namespace hifamd;

template <class T>
static Ccs<T> make_lower(int64_t m, int fan, uint64_t seed) {
  // strict lower triangle by columns: column j feeds rows j+1.. (a chain j -> j+1 keeps the schedule deep near the
  // end, random longer links keep it wide at the start)
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> u(-0.5, 0.5);
  Ccs<T> A;
  A.nrows = A.ncols = m;
  A.colptr.assign(1, 0);
  for (int64_t j = 0; j < m; ++j) {
    std::vector<int32_t> rows;
    if (j + 1 < m && j > m / 2) rows.push_back((int32_t)(j + 1));
    for (int f = 0; f < fan; ++f) {
      const int64_t r = j + 1 + (int64_t)(g() % (uint64_t)std::max<int64_t>(1, (m - j - 1)));
      if (r < m) rows.push_back((int32_t)r);
    }
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    for (int32_t r : rows) {
      A.rowind.push_back(r);
      A.vals.push_back(T(u(g)));
    }
    A.colptr.push_back((int64_t)A.rowind.size());
  }
  return A;
}

template <class T>
static int run(int64_t m, int fan, int64_t dense_block) {
  BandOptions opt;
  opt.dense_block = dense_block;
  opt.max_wg_rows = 16384;
  Ccs<T> L = make_lower<T>(m, fan, 17 + m);
  Csr<T> Lr = ccs_to_csr(L, false);
  Schedule S = level_schedule(Lr, true);
  BandPlan P = plan_bands(Lr, S, true, opt);
  Csr<T> Ls = permute_rows(Lr, P.order);
  finish_band_plan(P, Ls);
  const int64_t elems = plan_dense_blocks<T>(P, opt);
  int bad = 0;
  // every block inverse times the block's unit triangle is the identity
  std::vector<double> ops;
  int64_t ndense = 0;
  for (size_t q = 0; q < P.blk_slot0.size(); ++q) {
    const int64_t r0 = P.blk_slot0[q], nb = P.blk_slot1[q] - r0;
    ops.assign((size_t)dense_block_elems(nb, sizeof(T) != sizeof(double)), 0.0);
    const double growth = build_dense_block(P, Ls, q, ops.data());
    if (!(growth >= 1.0)) ++bad;
    ++ndense;
  }
  // block-dense bands: after the reordering, [split, end) of every row holds in-band sources only
  for (int64_t b = 0; b < P.nbands(); ++b) {
    if (!P.band_dense[(size_t)b]) continue;
    const int32_t g = P.band_wg_ptr[(size_t)b];
    const int32_t s0 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g]], s1 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g + 1]];
    for (int32_t s = s0; s < s1; ++s) {
      for (int32_t k = Ls.ptr[(size_t)s]; k < P.split[(size_t)s]; ++k)
        if (P.srcslot[(size_t)k] >= s0) ++bad;
      for (int32_t k = P.split[(size_t)s]; k < Ls.ptr[(size_t)s + 1]; ++k)
        if (P.srcslot[(size_t)k] < s0) ++bad;
    }
  }
  // bands with a carried prefix: [ptr, split) holds only sources older than the previous band, and after the
  // reordering (fast mode) [split, end) holds none of them
  int64_t nfused = 0;
  for (int64_t b = 1; b < P.nbands(); ++b) {
    if (!P.band_fused[(size_t)b]) continue;
    ++nfused;
    const int32_t prev0 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)P.band_wg_ptr[(size_t)b - 1]]];
    for (int32_t g = P.band_wg_ptr[(size_t)b]; g < P.band_wg_ptr[(size_t)b + 1]; ++g)
      for (int32_t s = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g]]; s < P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g + 1]]; ++s) {
        for (int32_t k = Ls.ptr[(size_t)s]; k < P.split[(size_t)s]; ++k)
          if (P.srcslot[(size_t)k] >= prev0) ++bad;
        for (int32_t k = P.split[(size_t)s]; k < Ls.ptr[(size_t)s + 1]; ++k)
          if (P.srcslot[(size_t)k] < prev0) ++bad;
      }
  }
  std::printf("(%ld bands with a carried prefix) ", (long)nfused);
  std::printf("m=%ld fan=%d cplx=%d: %ld wavefronts, %ld bands, %ld dense blocks (%ld operand doubles), bad=%d\n", (long)m, fan,
              (int)(sizeof(T) != sizeof(double)), (long)S.nwf(), (long)P.nbands(), (long)ndense, (long)elems, bad);
  return bad;
}

// The component-dense plan (host.hpp plan_bands_cd), lower and upper, checked NUMERICALLY on the host: the plan is
// executed band by band exactly as the device does it -- right-hand sides minus the sources outside the component
// ([ptr, csplit)), then the product with the component's explicit inverse read from the operand layout the kernels
// use; dense rest bands block by block -- and compared with the plain sequential substitution.
template <class T>
static int run_cd(int64_t m, int fan, int64_t cd_rows, int64_t dense_block, bool sparse = false) {
  BandOptions opt;
  opt.cd_rows = cd_rows;
  opt.cd_sparse_rows = cd_rows;
  opt.dense_block = dense_block;
  opt.max_wg_rows = 16384;
  int bad = 0;
  for (int upper = 0; upper < 2; ++upper) {
    Ccs<T> L = make_lower<T>(m, fan, 91 + m);
    Csr<T> R;
    if (!upper) {
      R = ccs_to_csr(L, false);
    } else {  // the transposed pattern as a strict UPPER triangle in descending-column row form
      Ccs<T> U;
      U.nrows = U.ncols = m;
      U.colptr.assign((size_t)m + 1, 0);
      Csr<T> Lr = ccs_to_csr(L, false);  // rows of L = columns of U
      for (int64_t j = 0; j < m; ++j) U.colptr[(size_t)j + 1] = U.colptr[(size_t)j] + (Lr.ptr[(size_t)j + 1] - Lr.ptr[(size_t)j]);
      U.rowind.assign(Lr.col.begin(), Lr.col.end());
      U.vals.assign(Lr.val.begin(), Lr.val.end());
      R = ccs_to_csr(U, true);
    }
    Schedule S = level_schedule(R, !upper);
    BandPlan P = plan_bands_cd(R, S, !upper, opt, nullptr, sparse);
    Csr<T> Rs = permute_rows(R, P.order);
    finish_band_plan(P, Rs, opt);  // (re-checks every dependency against the executed order, throws otherwise)
    const int64_t elems = plan_dense_blocks<T>(P, opt);
    build_cd_streams(P, Rs.ptr);
    std::vector<double> ops((size_t)elems, 0.0);
    for (size_t q = 0; q < P.blk_slot0.size(); ++q) build_dense_block(P, Rs, q, ops.data() + P.blk_inv_off[q], true);
    // reference: plain substitution in the natural order
    std::mt19937_64 g(5 + m);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::vector<T> b((size_t)m), xr, x;
    for (auto &v : b) v = T(u(g));
    xr = b;
    for (int64_t ii = 0; ii < m; ++ii) {
      const int64_t i = upper ? m - 1 - ii : ii;
      T acc = xr[(size_t)i];
      for (int32_t k = R.ptr[(size_t)i]; k < R.ptr[(size_t)i + 1]; ++k) acc -= R.val[(size_t)k] * xr[(size_t)R.col[(size_t)k]];
      xr[(size_t)i] = acc;
    }
    x = b;
    auto tinv_at = [&](size_t q, int32_t r, int32_t k) {
      const int32_t nb = P.blk_slot1[q] - P.blk_slot0[q];
      const int64_t lda = round_up32(nb), plane = plane_elems(nb, lda);
      const int64_t e = ((int64_t)(r >> 4) * lda + k) * 16 + (r & 15);
      const double *o = ops.data() + P.blk_inv_off[q];
      return sizeof(T) == sizeof(double) ? T(o[e]) : T(zdouble(o[e], o[plane + e]).real());
    };
    // the packed gather streams and descriptors of the component-dense bands (build_cd_streams): every component's
    // stream is exactly the [split, csplit) ranges of its rows in order, and the 16 wave chunks tile rows and entries
    for (int64_t bnd = 0; bnd < P.nbands(); ++bnd) {
      if (!P.band_cd[(size_t)bnd]) continue;
      for (int32_t gg = P.band_wg_ptr[(size_t)bnd]; gg < P.band_wg_ptr[(size_t)bnd + 1]; ++gg)
        for (int32_t c = P.wg_grp_ptr[(size_t)gg]; c < P.wg_grp_ptr[(size_t)gg + 1]; ++c) {
          const int32_t *dsc = &P.cd_desc[(size_t)c * kCdDescWords];
          const int32_t s0 = dsc[0], nb = dsc[1], mid0 = dsc[2], nmid = dsc[3];
          int64_t off;
          std::memcpy(&off, &dsc[4], 8);
          const uint8_t *wrow = reinterpret_cast<const uint8_t *>(&dsc[6]);
          const uint16_t *wmid = reinterpret_cast<const uint16_t *>(&dsc[11]);
          if (s0 != P.grp_slot_ptr[(size_t)c] || nb != P.grp_slot_ptr[(size_t)c + 1] - s0 || off != P.grp_inv_off[(size_t)c]) ++bad;
          if (wrow[0] != 0 || wrow[16] != nb || wmid[0] != 0 || wmid[16] != nmid) ++bad;
          int32_t e = mid0;
          for (int w = 0; w < 16; ++w) {
            if (wrow[w + 1] < wrow[w] || wrow[w + 1] - wrow[w] > 64 || wmid[w] != e - mid0) ++bad;
            for (int32_t r = wrow[w]; r < wrow[w + 1]; ++r)
              for (int32_t k = P.split[(size_t)(s0 + r)]; k < P.csplit[(size_t)(s0 + r)]; ++k, ++e)
                if (e >= mid0 + nmid || P.mid_k[(size_t)e] != k || P.mid_lrow[(size_t)e] != r) ++bad;
          }
          if (e != mid0 + nmid) ++bad;
        }
    }
    // the tile form of the same entries (build_ct_tiles, kernel k_band_ct), read the way the kernel reads it: the four
    // waves' strip masks partition the component's strips, and per strip  sum over its tiles of coef[16 x 4] * x[4 sources]
    // equals the rows' entry sums; every tile's sources are rows finished before the component
    if (!sparse) {
      if constexpr (std::is_same<T, double>::value) {
        CtTiles Tl;
        build_ct_tiles(P, Rs, Tl);
        std::vector<double> xs((size_t)m);
        std::mt19937_64 gt(313 + m);
        for (auto &vv : xs) vv = u(gt);
        int64_t ntile_seen = 0, ncomp_ct = 0;
        for (int64_t bnd = 0; bnd < P.nbands(); ++bnd) {
          if (!P.band_cd[(size_t)bnd]) continue;
          for (int32_t gg = P.band_wg_ptr[(size_t)bnd]; gg < P.band_wg_ptr[(size_t)bnd + 1]; ++gg)
            for (int32_t c = P.wg_grp_ptr[(size_t)gg]; c < P.wg_grp_ptr[(size_t)gg + 1]; ++c, ++ncomp_ct) {
              const int32_t *dsc = &Tl.desc[(size_t)c * kCdDescWords];
              const int32_t s0 = dsc[0], nb = dsc[1], sp0 = dsc[20], S = (nb + 15) / 16;
              const uint32_t masks[4] = {(uint32_t)dsc[22] & 0xffffu, (uint32_t)dsc[22] >> 16, (uint32_t)dsc[23] & 0xffffu, (uint32_t)dsc[23] >> 16};
              uint32_t seen = 0;
              for (int w = 0; w < 4; ++w) {
                if (seen & masks[w]) ++bad;  // (a strip owned twice)
                seen |= masks[w];
              }
              if (seen != (S >= 32 ? 0xffffffffu : ((1u << S) - 1u))) ++bad;
              for (int32_t st = 0; st < S; ++st) {
                const int32_t t0 = Tl.sptr[(size_t)(sp0 + st)], t1 = Tl.sptr[(size_t)(sp0 + st + 1)];
                if (t1 < t0 || (int64_t)t1 > Tl.ntiles) {
                  ++bad;
                  continue;
                }
                ntile_seen += t1 - t0;
                double acc[16];
                for (double &a : acc) a = 0.0;
                for (int32_t t = t0; t < t1; ++t)
                  for (int k = 0; k < 4; ++k) {
                    const int32_t src = Tl.src[(size_t)(4 * t + k)];
                    bool older = false;  // a source row: finished before this component (its slot lies in front of it)
                    for (int32_t r = 16 * st; r < std::min(nb, 16 * st + 16) && !older; ++r)
                      for (int32_t kk = P.split[(size_t)(s0 + r)]; kk < P.csplit[(size_t)(s0 + r)]; ++kk)
                        if (Rs.col[(size_t)kk] == src) older = true;
                    if (!older) {  // (a padding slot repeats the tile's last real source with zero coefficients)
                      for (int r = 0; r < 16; ++r)
                        if (Tl.coef[(size_t)(64 * t + (k << 4) + r)] != 0.0) ++bad;
                    }
                    for (int r = 0; r < 16; ++r) acc[r] += Tl.coef[(size_t)(64 * t + (k << 4) + r)] * xs[(size_t)src];
                  }
                for (int32_t r = 16 * st; r < std::min(nb, 16 * st + 16); ++r) {
                  double ref = 0.0;
                  for (int32_t kk = P.split[(size_t)(s0 + r)]; kk < P.csplit[(size_t)(s0 + r)]; ++kk) ref += Rs.val[(size_t)kk] * xs[(size_t)Rs.col[(size_t)kk]];
                  if (std::fabs(ref - acc[r - 16 * st]) > 1e-12 * (1.0 + std::fabs(ref))) ++bad;
                }
                for (int32_t r = std::min(nb, 16 * st + 16); r < 16 * st + 16; ++r)
                  if (acc[r - 16 * st] != 0.0) ++bad;  // (rows behind the component's end carry no coefficient)
              }
            }
        }
        if (ncomp_ct && ntile_seen != Tl.ntiles) ++bad;
      }
    }
    // S5 fused into the L streams (build_cd_streams_fused): every row's stream entries are its [split, csplit) L entries
    // followed by its F entries shifted to the child's rows, wave chunk by wave chunk with the plain streams' ownership;
    // checked numerically too: rhs - sum (stream entries) == rhs - L_old x - F y for random x, y
    if (!upper) {  // (cd_f_fusable decides about USING them; the streams of the component bands can always be built)
      const int64_t ncols = 37;
      Csr<T> F;
      F.nrows = m, F.ncols = ncols;
      F.ptr.assign((size_t)m + 1, 0);
      std::mt19937_64 gf(77 + m);
      for (int64_t i = 0; i < m; ++i) {
        const int cnt = (int)(gf() % 4);
        for (int k = 0; k < cnt; ++k) F.col.push_back((int32_t)(gf() % ncols)), F.val.push_back(T(1.0 + (double)(gf() % 7)));
        F.ptr[(size_t)i + 1] = (int32_t)F.col.size();
      }
      CdFusedStreams<T> Sf;
      const int64_t src0 = 2 * m + 11;
      if (!build_cd_streams_fused(P, Rs, F, src0, Sf)) ++bad;
      int64_t nchecked = 0;
      for (int64_t bnd = 0; bnd < P.nbands(); ++bnd)
        for (int32_t gg = P.band_wg_ptr[(size_t)bnd]; P.band_cd[(size_t)bnd] && gg < P.band_wg_ptr[(size_t)bnd + 1]; ++gg)
          for (int32_t c = P.wg_grp_ptr[(size_t)gg]; c < P.wg_grp_ptr[(size_t)gg + 1]; ++c, ++nchecked) {
            const int32_t *d0 = &P.cd_desc[(size_t)c * kCdDescWords], *d1 = &Sf.desc[(size_t)c * kCdDescWords];
            const uint8_t *wrow = reinterpret_cast<const uint8_t *>(&d1[6]);
            const uint16_t *wmid = reinterpret_cast<const uint16_t *>(&d1[11]);
            if (d0[0] != d1[0] || d0[1] != d1[1] || std::memcmp(&d0[4], &d1[4], 8) != 0) ++bad;
            int32_t e = d1[2];
            for (int w = 0; w < 16; ++w) {
              if (wmid[w] != e - d1[2]) ++bad;
              for (int32_t r = wrow[w]; r < wrow[w + 1]; ++r) {
                const int32_t sl = d1[0] + r;
                for (int32_t k = P.split[(size_t)sl]; k < P.csplit[(size_t)sl]; ++k, ++e)
                  if (Sf.col[(size_t)e] != Rs.col[(size_t)k] || Sf.val[(size_t)e] != Rs.val[(size_t)k] || Sf.lrow[(size_t)e] != r) ++bad;
                const int32_t i = Rs.rowid[(size_t)sl];
                for (int32_t k = F.ptr[(size_t)i]; k < F.ptr[(size_t)i + 1]; ++k, ++e)
                  if (Sf.col[(size_t)e] != src0 + F.col[(size_t)k] || Sf.val[(size_t)e] != F.val[(size_t)k] || Sf.lrow[(size_t)e] != r) ++bad;
              }
            }
            if (e != d1[2] + d1[3] || wmid[16] != d1[3]) ++bad;
          }
      if (!nchecked) ++bad;
    }
    int64_t ncd = 0, ncomp = 0, maxcomp = 0;
    for (int64_t bnd = 0; bnd < P.nbands(); ++bnd) {
      if (!P.band_cd[(size_t)bnd] && !P.band_dense[(size_t)bnd]) {
        ++bad;  // the component-dense planner emits nothing else
        continue;
      }
      ncd += P.band_cd[(size_t)bnd];
      for (int32_t q = P.band_blk_ptr[(size_t)bnd]; q < P.band_blk_ptr[(size_t)bnd + 1]; ++q) {
        const int32_t r0 = P.blk_slot0[(size_t)q], nb = P.blk_slot1[(size_t)q] - r0;
        if (P.band_cd[(size_t)bnd]) ++ncomp, maxcomp = std::max<int64_t>(maxcomp, nb);
        std::vector<T> t((size_t)nb);
        for (int32_t r = 0; r < nb; ++r) {
          const int32_t s = r0 + r;
          T acc = x[(size_t)Rs.rowid[(size_t)s]];
          for (int32_t k = Rs.ptr[(size_t)s]; k < Rs.ptr[(size_t)s + 1]; ++k)
            if (P.srcslot[(size_t)k] < r0) {
              if (P.band_cd[(size_t)bnd] && k >= P.csplit[(size_t)s]) ++bad;
              acc -= Rs.val[(size_t)k] * x[(size_t)Rs.col[(size_t)k]];
            } else if (P.band_cd[(size_t)bnd] && k < P.csplit[(size_t)s])
              ++bad;
          t[(size_t)r] = acc;
        }
        for (int32_t r = 0; r < nb; ++r) {
          T acc = T(0);
          for (int32_t k = 0; k <= r; ++k) acc += tinv_at((size_t)q, r, k) * t[(size_t)k];
          x[(size_t)Rs.rowid[(size_t)(r0 + r)]] = acc;
        }
      }
      if (P.band_cd[(size_t)bnd] && P.cd_sparse) {
        // sparse-own components: old sources, then substitution inside the component level by level, exactly as the
        // kernel walks the own stream (build_cd_streams): every source of a level's rows lies in an earlier level
        for (int32_t gg = P.band_wg_ptr[(size_t)bnd]; gg < P.band_wg_ptr[(size_t)bnd + 1]; ++gg)
          for (int32_t c = P.wg_grp_ptr[(size_t)gg]; c < P.wg_grp_ptr[(size_t)gg + 1]; ++c) {
            const int32_t *dsc = &P.cd_desc[(size_t)c * kCdDescWords];
            const int32_t r0 = dsc[0], nb = dsc[1], own0 = dsc[20], nown = dsc[21], orp0 = dsc[22], lvl0 = dsc[23], nlvl = dsc[24];
            ++ncomp, maxcomp = std::max<int64_t>(maxcomp, nb);
            std::vector<T> t((size_t)nb);
            for (int32_t r = 0; r < nb; ++r) {
              const int32_t s = r0 + r;
              T acc = x[(size_t)Rs.rowid[(size_t)s]];
              for (int32_t k = Rs.ptr[(size_t)s]; k < P.csplit[(size_t)s]; ++k) acc -= Rs.val[(size_t)k] * x[(size_t)Rs.col[(size_t)k]];
              t[(size_t)r] = acc;
            }
            if (P.own_rptr[(size_t)orp0] != 0 || P.own_rptr[(size_t)(orp0 + nb)] != nown || P.own_lvl[(size_t)lvl0] != 0 ||
                P.own_lvl[(size_t)(lvl0 + nlvl)] != nb)
              ++bad;
            std::vector<uint8_t> done((size_t)nb, 0);
            for (int32_t lv = 0; lv < nlvl; ++lv) {
              const int32_t lo = P.own_lvl[(size_t)(lvl0 + lv)], hi = P.own_lvl[(size_t)(lvl0 + lv + 1)];
              std::vector<T> fresh;
              for (int32_t r = lo; r < hi; ++r) {
                T acc = t[(size_t)r];
                for (int32_t e = P.own_rptr[(size_t)(orp0 + r)]; e < P.own_rptr[(size_t)(orp0 + r + 1)]; ++e) {
                  const int32_t q = P.own_lsrc[(size_t)(own0 + e)];
                  if (!done[(size_t)q] || P.srcslot[(size_t)P.own_k[(size_t)(own0 + e)]] != r0 + q) ++bad;  // (an earlier level)
                  acc -= Rs.val[(size_t)P.own_k[(size_t)(own0 + e)]] * t[(size_t)q];
                }
                fresh.push_back(acc);
              }
              for (int32_t r = lo; r < hi; ++r) t[(size_t)r] = fresh[(size_t)(r - lo)], done[(size_t)r] = 1;
            }
            for (int32_t r = 0; r < nb; ++r) x[(size_t)Rs.rowid[(size_t)(r0 + r)]] = t[(size_t)r];
          }
      }
    }
    double err = 0.0, nrm = 0.0;
    for (int64_t i = 0; i < m; ++i) err = std::max(err, abs_(x[(size_t)i] - xr[(size_t)i])), nrm = std::max(nrm, abs_(xr[(size_t)i]));
    if (!(err <= 1e-12 * nrm) || maxcomp > cd_rows) ++bad;
    std::printf("cd plan m=%ld fan=%d %s cd_rows=%ld: %ld wavefronts -> %ld bands (%ld component-dense, %ld components, largest %ld rows), "
                "relerr %.2e, bad=%d\n", (long)m, fan, upper ? "upper" : "lower", (long)cd_rows, (long)S.nwf(), (long)P.nbands(),
                (long)ncd, (long)ncomp, (long)maxcomp, err / nrm, bad);
  }
  return bad;
}

// The combined top operator (choose_top / build_top_operator): a triangle pair with the same pattern, both planned around
// the top set T; G t_T must equal what plain substitution gives on T when t_T is the L right-hand side of the T rows
// with everything below T already subtracted.
static int run_top(int64_t m, int fan, int64_t top_max) {
  typedef double T;
  BandOptions opt;
  opt.cd_rows = 64;
  opt.dense_block = 2048;
  opt.max_wg_rows = 16384;
  opt.top_max = top_max;
  int bad = 0;
  Ccs<T> L = make_lower<T>(m, fan, 311 + m);
  Csr<T> Lr = ccs_to_csr(L, false);
  Ccs<T> U;
  U.nrows = U.ncols = m;
  U.colptr.assign((size_t)m + 1, 0);
  for (int64_t j = 0; j < m; ++j) U.colptr[(size_t)j + 1] = U.colptr[(size_t)j] + (Lr.ptr[(size_t)j + 1] - Lr.ptr[(size_t)j]);
  U.rowind.assign(Lr.col.begin(), Lr.col.end());
  std::mt19937_64 g(5 + m);
  std::uniform_real_distribution<double> u(-0.5, 0.5);
  for (size_t k = 0; k < Lr.val.size(); ++k) U.vals.push_back(u(g));
  Csr<T> Ur = ccs_to_csr(U, true);
  const Csr<T> L0 = Lr, U0 = Ur;  // natural order, for the reference substitution
  std::vector<T> d((size_t)m), b((size_t)m);
  for (auto &v : d) v = 1.5 + u(g);
  for (auto &v : b) v = u(g);
  Schedule Ls = level_schedule(Lr, true), Us = level_schedule(Ur, false);
  BandPlan Lp0 = plan_bands_cd(Lr, Ls, true, opt);
  std::vector<uint8_t> top = choose_top(Lr, Ur, Lp0, opt.top_max, opt.top_few_wgs);
  int64_t nt = 0;
  for (uint8_t t : top) nt += t;
  if (top.empty() || nt == 0) {
    std::printf("top operator m=%ld fan=%d: no top set chosen\n", (long)m, fan);
    return 1;  // (the cases below are built to have one)
  }
  BandPlan Lp = plan_bands_cd(Lr, Ls, true, opt, &top), Up = plan_bands_cd(Ur, Us, false, opt, &top);
  Lr = permute_rows(Lr, Lp.order);
  Ur = permute_rows(Ur, Up.order);
  finish_band_plan(Lp, Lr, opt);
  finish_band_plan(Up, Ur, opt);
  const int32_t bL = (int32_t)Lp.nbands() - 1, bU = 0;
  const int32_t r0L = Lp.grp_slot_ptr[(size_t)Lp.wg_grp_ptr[(size_t)Lp.band_wg_ptr[(size_t)bL]]];
  const int32_t r0U = Up.grp_slot_ptr[(size_t)Up.wg_grp_ptr[(size_t)Up.band_wg_ptr[(size_t)bU]]];
  std::vector<T> G;
  const double growth = build_top_operator(Lr, Lp, r0L, Ur, Up, r0U, nt, d, G);
  // reference: y = L^{-1} b, z = y / d, x = U^{-1} z in the natural order
  std::vector<T> y = b;
  for (int64_t i = 0; i < m; ++i)
    for (int32_t k = L0.ptr[(size_t)i]; k < L0.ptr[(size_t)i + 1]; ++k) y[(size_t)i] -= L0.val[(size_t)k] * y[(size_t)L0.col[(size_t)k]];
  std::vector<T> x((size_t)m);
  for (int64_t i = 0; i < m; ++i) x[(size_t)i] = y[(size_t)i] / d[(size_t)i];
  for (int64_t i = m - 1; i >= 0; --i)
    for (int32_t k = U0.ptr[(size_t)i]; k < U0.ptr[(size_t)i + 1]; ++k) x[(size_t)i] -= U0.val[(size_t)k] * x[(size_t)U0.col[(size_t)k]];
  // t_T: the L right-hand side of the T rows with the rows below T subtracted (what the prefix pass delivers), L slot order
  std::vector<T> tT((size_t)nt);
  for (int64_t r = 0; r < nt; ++r) {
    const int32_t i = Lr.rowid[(size_t)(r0L + r)];
    if (!top[(size_t)i]) ++bad;
    T acc = b[(size_t)i];
    for (int32_t k = L0.ptr[(size_t)i]; k < L0.ptr[(size_t)i + 1]; ++k)
      if (!top[(size_t)L0.col[(size_t)k]]) acc -= L0.val[(size_t)k] * y[(size_t)L0.col[(size_t)k]];
    tT[(size_t)r] = acc;
  }
  // every U source of a T row lies in T (the set is closed)
  for (int64_t i = 0; i < m; ++i)
    if (top[(size_t)i])
      for (int32_t k = U0.ptr[(size_t)i]; k < U0.ptr[(size_t)i + 1]; ++k)
        if (!top[(size_t)U0.col[(size_t)k]]) ++bad;
  double err = 0.0, nrm = 0.0;
  for (int64_t r = 0; r < nt; ++r) {
    T acc = 0;
    for (int64_t c = 0; c < nt; ++c) acc += G[(size_t)(r + c * nt)] * tT[(size_t)c];
    const T want = x[(size_t)Lr.rowid[(size_t)(r0L + r)]];
    err = std::max(err, std::abs(acc - want));
    nrm = std::max(nrm, std::abs(want));
  }
  if (!(err <= 1e-11 * nrm)) ++bad;
  std::printf("top operator m=%ld fan=%d: |T| = %ld of %ld rows, growth %.1f, relerr %.2e, bad=%d\n", (long)m, fan, (long)nt, (long)m,
              growth, err / nrm, bad);
  return bad;
}

// Tiled form of a coupling block (build_spmm_tiles): the tiles reproduce A x -- every nonzero sits in exactly one
// (block, group, column-in-group, row-in-block) cell, padding columns carry zero coefficients
static int run_tiles(int64_t nrows, int64_t ncols, int per_row) {
  Csr<double> A;
  A.nrows = nrows, A.ncols = ncols;
  A.ptr.assign(1, 0);
  std::mt19937_64 g(17 + nrows);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int64_t i = 0; i < nrows; ++i) {
    std::vector<int32_t> cols;
    const int64_t base = (i / 16) * 7 % std::max<int64_t>(1, ncols - 40);  // neighbouring rows share columns
    for (int k = 0; k < per_row; ++k) cols.push_back((int32_t)(base + (int64_t)(g() % 40)));
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    for (int32_t c : cols) A.col.push_back(c), A.val.push_back(u(g));
    A.ptr.push_back((int32_t)A.col.size());
  }
  A.rowid.resize((size_t)nrows);
  for (int64_t i = 0; i < nrows; ++i) A.rowid[(size_t)i] = (int32_t)i;
  const SpmmTiles Tl = build_spmm_tiles(A);
  std::vector<double> x((size_t)ncols), y((size_t)nrows, 0.0), yt((size_t)nrows, 0.0);
  for (auto &v : x) v = u(g);
  for (int64_t i = 0; i < nrows; ++i)
    for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) y[(size_t)i] += A.val[(size_t)k] * x[(size_t)A.col[(size_t)k]];
  int bad = 0;
  if (Tl.nblk != (nrows + 15) / 16 || (int64_t)Tl.blk_gptr.size() != Tl.nblk + 1) ++bad;
  for (int64_t b = 0; b < Tl.nblk; ++b)
    for (int32_t gq = Tl.blk_gptr[(size_t)b]; gq < Tl.blk_gptr[(size_t)b + 1]; ++gq)
      for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 16; ++r) {
          const double a = Tl.coef[(size_t)(64 * (int64_t)gq + (k << 4) + r)];
          const int64_t i = 16 * b + r;
          if (i >= nrows) {
            if (a != 0.0) ++bad;
            continue;
          }
          const int32_t c = Tl.ucol[(size_t)(4 * (int64_t)gq + k)];
          if (c < 0 || c >= ncols) ++bad; else yt[(size_t)i] += a * x[(size_t)c];
        }
  double err = 0.0, nrm = 0.0;
  for (int64_t i = 0; i < nrows; ++i) err = std::max(err, std::abs(y[(size_t)i] - yt[(size_t)i])), nrm = std::max(nrm, std::abs(y[(size_t)i]));
  if (!(err <= 1e-13 * nrm) || !(Tl.reuse >= 1.0)) ++bad;
  std::printf("spmm tiles %ld x %ld, %d per row: %ld blocks, %d groups, reuse %.2f, relerr %.2e, bad=%d\n", (long)nrows, (long)ncols, per_row,
              (long)Tl.nblk, Tl.blk_gptr.back(), Tl.reuse, err / nrm, bad);
  return bad;
}

int main() {
  int bad = 0;
  bad += run_tiles(1000, 5000, 12);
  bad += run_tiles(77, 300, 30);
  bad += run_top(3000, 2, 4096);
  bad += run_top(9000, 3, 8192);
  bad += run_cd<double>(3000, 2, 192, 2048);
  bad += run_cd<double>(9000, 3, 64, 512);
  bad += run_cd<double>(20000, 6, 192, 2048);
  bad += run_cd<double>(700, 1, 32, 64);
  bad += run_cd<double>(20000, 2, 192, 2048, true);  // sparse-own components (thin triangles)
  bad += run_cd<double>(6000, 1, 64, 512, true);
  bad += run<double>(3000, 2, 2048);
  bad += run<double>(9000, 3, 2048);
  bad += run<double>(9000, 3, 512);
  bad += run<zdouble>(4000, 2, 1024);
  bad += run<double>(500, 1, 0);
  bad += run<double>(60000, 6, 2048);
  std::printf(bad ? "FAIL\n" : "OK\n");
  return bad ? 1 : 0;
}
