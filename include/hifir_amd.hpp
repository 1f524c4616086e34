// hifir_amd.hpp -- header-only C++11 facade over the C ABI (hifir_amd.h) that mirrors the apply-side
// interface of hif::HIF<ValueType, int, std::ptrdiff_t> (reference: src/hif/builder.hpp:107-513), so
// that header-only users of the reference keep their call sites:
//
//     hif::HIF<double> M;  M.factorize(A, params);          // host, unchanged (builder.hpp:264)
//     hifamd::HIF<double> G;  G.attach(M);  G.set_matrix(A); // once: hierarchy + matrix to HBM
//     G.solve(b, x);                                         // was M.solve(b, x)           :409-423
//     G.solve(b, x, true);                                   // was M.solve(b, x, true)     (prec_solve_tran)
//     G.solve_mrhs(B, X);                                    // was M.solve_mrhs(B, X)      :433-445
//     G.hifir(A, b, N, x);  G.hifir(A, b, N, betas, x);      // was M.hifir(...)            :459-489
//     G.mmultiply(x, y);                                     // was M.mmultiply(x, y)       :503-513
//
// Same names, argument meaning and defaults; errors are thrown as std::runtime_error carrying
// hifamd_last_error() (the reference built with HIF_THROW throws std::runtime_error too).  Array
// arguments are anything with data() and size() (hif::Array, std::vector, ...); multi-RHS blocks are
// arrays of std::array<T, Nrhs> like hif::Array<std::array<T, Nrhs>> (CompressedStorage.hpp:2127).
// Not thread-safe per object, like hif::HIF (mutable work space, builder.hpp:579).
#ifndef HIFIR_AMD_HPP
#define HIFIR_AMD_HPP

#include <array>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "hifir_amd.h"

namespace hifamd {

namespace detail {
template <class T>
struct value_tag;
template <>
struct value_tag<double> {
  static HifAmdValueType value() { return HIFAMD_D; }
};
template <>
struct value_tag<std::complex<double>> {
  static HifAmdValueType value() { return HIFAMD_Z; }
};
inline void check(HifAmdStatus st) {
  if (st == HIFAMD_SUCCESS) return;
  const char *msg = hifamd_last_error();
  throw std::runtime_error(std::string("hifir_amd: ") + (msg ? msg : "error without message"));
}
inline std::int64_t rank_arg(std::size_t r) {  // size_type(-1) = full rank (builder.hpp:461)
  return r == static_cast<std::size_t>(-1) ? -1 : static_cast<std::int64_t>(r);
}
}  // namespace detail

template <class ValueType>
class HIF {
 public:
  typedef ValueType value_type;
  typedef std::size_t size_type;

  explicit HIF(int device = -1) : _h(nullptr), _device(device), _has_A(false) {}
  ~HIF() { clear(); }
  HIF(const HIF &) = delete;
  HIF &operator=(const HIF &) = delete;
  HIF(HIF &&o) noexcept : _h(o._h), _device(o._device), _has_A(o._has_A) { o._h = nullptr; }

  /// Ship the hierarchy of a factorized hif::HIF (or anything with the same precs() interface,
  /// alg/Prec.hpp:309-357) to HBM.  max_nrhs sizes the device work arena (wider batches are tiled).
  /// spd: Options::spd of the factorization (only read for is_symm hierarchies: truncation rule of SYEIG).
  template <class RefHif>
  void attach(const RefHif &M, const size_type max_nrhs = 64, const int spd = 0) {
    clear();
    detail::check(hifamd_create(detail::value_tag<value_type>::value(), _device, &_h));
    try {
      for (auto itr = M.precs().cbegin(); itr != M.precs().cend(); ++itr) {
        const auto &p = *itr;
        // CCS accessors: ds/CompressedStorage.hpp:1910-1915; pointer type may be any integer
        const std::vector<std::int64_t> Lp(p.L_B.col_start().cbegin(), p.L_B.col_start().cend()),
            Up(p.U_B.col_start().cbegin(), p.U_B.col_start().cend()),
            Ep(p.E.col_start().cbegin(), p.E.col_start().cend()), Fp(p.F.col_start().cbegin(), p.F.col_start().cend());
        const std::vector<std::int32_t> Li(p.L_B.row_ind().cbegin(), p.L_B.row_ind().cend()),
            Ui(p.U_B.row_ind().cbegin(), p.U_B.row_ind().cend()), Ei(p.E.row_ind().cbegin(), p.E.row_ind().cend()),
            Fi(p.F.row_ind().cbegin(), p.F.row_ind().cend());
        const std::vector<std::int32_t> pp(p.p.cbegin(), p.p.cend()), pi(p.p_inv.cbegin(), p.p_inv.cend()),
            qq(p.q.cbegin(), p.q.cend()), qi(p.q_inv.cbegin(), p.q_inv.cend());
        const std::vector<double> ss(p.s.cbegin(), p.s.cend()), tt(p.t.cbegin(), p.t.cend());
        const std::int64_t F_ncols = static_cast<std::int64_t>(p.F.ncols());
        detail::check(hifamd_add_level(_h, (std::int64_t)p.m, (std::int64_t)p.n, Lp.data(), Li.data(), p.L_B.vals().data(),
                                       Up.data(), Ui.data(), p.U_B.vals().data(), Ep.data(), Ei.data(), p.E.vals().data(),
                                       F_ncols, F_ncols ? Fp.data() : nullptr, Fi.data(), p.F.vals().data(),
                                       p.d_B.data(), ss.data(), tt.data(), pp.data(), pi.data(), qq.data(), qi.data()));
        if (!p.dense_solver.empty()) {  // the UNFACTORED block, Prec::inquire_or_export_dense (Prec.hpp:275-293)
          const auto &D = p.dense_solver.mat_backup();
          if (std::strcmp(p.dense_solver.method(), "LUP") == 0)  // reference built with HIF_DENSE_MODE=0
            detail::check(hifamd_set_dense_lup(_h, (std::int64_t)D.nrows(), D.data()));
          else
            detail::check(hifamd_set_dense(_h, (std::int64_t)D.nrows(), D.data(), 0.0));
        } else if (!p.symm_dense_solver.empty()) {  // is_symm factorizations (symm_factor.hpp:654-657): SYEIG
          const auto &D = p.symm_dense_solver.mat_backup();
          detail::check(hifamd_set_dense_symm(_h, (std::int64_t)D.nrows(), D.data(), spd));
        }
      }
      detail::check(hifamd_finalize(_h, (std::int64_t)max_nrhs));
    } catch (...) {
      clear();
      throw;
    }
  }

  /// The matrix of the iterative-refinement / GMRES loops (0- or 1-based CRS, copied to HBM).
  void set_matrix(const size_type n, const std::int64_t *indptr, const std::int32_t *indices, const value_type *vals) {
    require();
    detail::check(hifamd_set_matrix(_h, (std::int64_t)n, indptr, indices, vals));
    _has_A = true;
  }
  /// ... from a hif::CRS-like object (row_start(), col_ind(), vals(); CompressedStorage.hpp:812-817)
  template <class Crs>
  void set_matrix(const Crs &A) {
    const std::vector<std::int64_t> ip(A.row_start().cbegin(), A.row_start().cend());
    const std::vector<std::int32_t> ci(A.col_ind().cbegin(), A.col_ind().cend());
    set_matrix(A.nrows(), ip.data(), ci.data(), A.vals().data());
  }

  // ---- queries (builder.hpp:136-199) --------------------------------------------------------------
  bool empty() const { return !_h; }
  size_type levels() const { return _h ? (size_type)hifamd_levels(_h) : 0u; }
  size_type nnz() const { return _h ? (size_type)hifamd_nnz(_h) : 0u; }
  size_type nrows() const { return _h ? (size_type)hifamd_nrows(_h) : 0u; }
  size_type ncols() const { return nrows(); }
  size_type schur_rank() const { return _h ? (size_type)hifamd_schur_rank(_h) : 0u; }
  size_type schur_size() const { return _h ? (size_type)hifamd_schur_size(_h) : 0u; }
  size_type rank() const { return empty() ? 0u : nrows() - (schur_size() - schur_rank()); }
  void clear() {
    if (_h) hifamd_destroy(_h);
    _h = nullptr;
    _has_A = false;
  }
  HifAmdHdl handle() const { return _h; }  ///< for the device-pointer entry points of hifir_amd.h
  /// HIF::nsp / nsp_tran with NspFilter::set_nsp_const (NspFilter.hpp:118-125); start > end removes it
  void set_nsp_const(const size_type start = 0, const size_type end = static_cast<size_type>(-1), const bool trans = false) {
    require();
    detail::check(hifamd_set_nsp_const(_h, trans ? HIFAMD_SH : HIFAMD_S, (std::int64_t)start,
                                       end == static_cast<size_type>(-1) ? -1 : (std::int64_t)end));
  }

  // ---- x = M^{-1} b / M^{-H} b (builder.hpp:409-423) ---------------------------------------------
  template <class RhsType, class SolType>
  void solve(const RhsType &b, SolType &x, const bool trans = false, const size_type r = 0u) const {
    require();
    if (b.size() != x.size()) throw std::runtime_error("hifir_amd: unmatched sizes");
    detail::check(hifamd_apply_batch(_h, trans ? HIFAMD_SH : HIFAMD_S, b.data(), 1, x.data(), 1, 1, 1, nullptr,
                                     detail::rank_arg(r), nullptr));
  }

  // ---- X = M^{-1} B, B and X arrays of std::array<T, Nrhs> (builder.hpp:433-445) -------------------
  template <class RhsBlock, class SolBlock>
  void solve_mrhs(const RhsBlock &b, SolBlock &x, const size_type r = 0u) const {
    require();
    typedef typename std::remove_cv<typename std::remove_reference<decltype(b.data()[0])>::type>::type row_type;
    const std::int64_t nrhs = (std::int64_t)std::tuple_size<row_type>::value;
    if (b.size() != x.size()) throw std::runtime_error("hifir_amd: unmatched sizes");
    detail::check(hifamd_solve_batch(_h, b.data(), nrhs, x.data(), nrhs, nrhs, detail::rank_arg(r)));
  }

  // ---- iterative refinement (builder.hpp:459-489).  A is the matrix handed to set_matrix(); when none
  //      was attached yet and A looks like a hif::CRS it is attached on the fly. -------------------------
  template <class Matrix, class RhsType, class SolType>
  void hifir(const Matrix &A, const RhsType &b, const size_type N, SolType &x, const bool trans = false,
             const size_type r = static_cast<size_type>(-1)) {
    ensure_matrix(A);
    detail::check(hifamd_apply_batch(_h, trans ? HIFAMD_SH : HIFAMD_S, b.data(), 1, x.data(), 1, 1, (int)N, nullptr,
                                     detail::rank_arg(r), nullptr));
  }
  template <class Matrix, class RhsType, class SolType>
  std::pair<size_type, int> hifir(const Matrix &A, const RhsType &b, const size_type N, const double *betas, SolType &x,
                                  const bool trans = false, const size_type r = static_cast<size_type>(-1)) {
    ensure_matrix(A);
    int st[2] = {0, 0};
    detail::check(hifamd_apply_batch(_h, trans ? HIFAMD_SH : HIFAMD_S, b.data(), 1, x.data(), 1, 1, (int)N, betas,
                                     detail::rank_arg(r), st));
    return std::make_pair((size_type)st[0], st[1]);
  }

  // ---- y = M x / M^H x (builder.hpp:503-513) ------------------------------------------------------
  template <class RhsType, class SolType>
  void mmultiply(const RhsType &x, SolType &y, const bool trans = false, const size_type r = 0u) const {
    require();
    if (y.size() != x.size()) throw std::runtime_error("hifir_amd: unmatched sizes");
    detail::check(hifamd_apply_batch(_h, trans ? HIFAMD_MH : HIFAMD_M, x.data(), 1, y.data(), 1, 1, 1, nullptr,
                                     detail::rank_arg(r), nullptr));
  }

  // ---- the example driver of examples/advanced/gmres.hpp:19-123 on the device (real or complex; Hermitian inner product) ----
  template <class Matrix, class ArrayType>
  std::tuple<ArrayType, int, int> gmres(const Matrix &A, const ArrayType &b, const int restart, const double rtol,
                                        const int maxit, const bool full_rank = false) {
    ensure_matrix(A);
    ArrayType x(b.size());
    int flag = 0, iters = 0;
    detail::check(hifamd_gmres_batch(_h, b.data(), 1, x.data(), 1, 1, restart, rtol, maxit, full_rank ? -1 : 0, &flag, &iters));
    return std::make_tuple(std::move(x), flag, iters);
  }

 private:
  void require() const {
    if (!_h) throw std::runtime_error("hifir_amd: MILU-Prec is empty!");  // builder.hpp:412
  }
  template <class Matrix>
  void ensure_matrix(const Matrix &A) {
    require();
    if (!_has_A) set_matrix(A);
  }

  HifAmdHdl _h;
  int _device;
  bool _has_A;
};

}  // namespace hifamd

#endif  // HIFIR_AMD_HPP
