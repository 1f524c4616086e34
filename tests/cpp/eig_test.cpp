// CPU unit test of the host-side symmetric / Hermitian eigensolver and SYEIG factorization (hifir_amd/csrc/host.hpp
// herm_eig, dense_factorize_symm): A V = V diag(w), V^H V = I, ascending w, and a solve through the two operators the
// device uses.  Built and run by tests/test_abi_and_host.py::test_host_eigensolver (g++, no GPU).
#include "host.hpp"
#include <random>
using namespace hifamd;
template <class T> double run(int64_t n, bool clustered) {
  std::mt19937_64 g(n * 7 + clustered);
  std::uniform_real_distribution<double> u(-1, 1);
  std::vector<T> A((size_t)(n * n));
  for (int64_t j = 0; j < n; ++j) for (int64_t i = j; i < n; ++i) {
    T v; if constexpr (sizeof(T) == 16) v = T(u(g), i == j ? 0.0 : u(g)); else v = T(u(g));
    if (clustered && i != j) v *= 1e-3;
    A[(size_t)(i + j * n)] = v; A[(size_t)(j + i * n)] = conj_(v);
  }
  std::vector<T> V = A; std::vector<double> w;
  herm_eig(n, V, w);
  double err = 0, orth = 0;
  for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < n; ++i) {
    T av = T(0); for (int64_t k = 0; k < n; ++k) av += A[(size_t)(i + k * n)] * V[(size_t)(k + j * n)];
    err = std::max(err, abs_(av - V[(size_t)(i + j * n)] * w[(size_t)j]));
    T o = T(0); for (int64_t k = 0; k < n; ++k) o += conj_(V[(size_t)(k + i * n)]) * V[(size_t)(k + j * n)];
    orth = std::max(orth, abs_(o - (i == j ? T(1) : T(0))));
  }
  for (int64_t j = 1; j < n; ++j) if (w[j] < w[j-1]) err = 1e9;
  printf("n=%ld cplx=%d clustered=%d  |AV-VW|=%.2e |V^HV-I|=%.2e\n", (long)n, (int)(sizeof(T)==16), (int)clustered, err, orth);
  return std::max(err, orth);
}
int main() {
  double m = 0;
  for (int64_t n : {1, 2, 3, 10, 57, 200}) for (int c = 0; c < 2; ++c) { m = std::max(m, run<double>(n, c)); m = std::max(m, run<zdouble>(n, c)); }
  // dense_factorize_symm: solve check
  { int64_t n = 40; std::vector<double> A(n*n); std::mt19937_64 g(5); std::uniform_real_distribution<double> u(-1,1);
    for (int64_t j=0;j<n;++j) for (int64_t i=j;i<n;++i){ double v=u(g); if(i==j) v+=0.0; A[i+j*n]=v; A[j+i*n]=v; }
    HostDense<double> D; dense_factorize_symm(D, A.data(), n, 0);
    std::vector<double> b(n), t(n), x(n), r(n);
    for (auto &v : b) v = u(g);
    for (int64_t i=0;i<n;++i){ double a=0; for(int64_t k=0;k<n;++k) a+=D.QH[i+k*n]*b[k]; t[i]=a; }
    for (int64_t i=0;i<n;++i){ double a=0; for(int64_t k=0;k<n;++k) a+=D.Q[i+k*n]*t[k]; x[i]=a; }
    double res=0; for (int64_t i=0;i<n;++i){ double a=0; for(int64_t k=0;k<n;++k) a+=A[i+k*n]*x[k]; res=std::max(res,std::fabs(a-b[i])); }
    printf("symm solve residual %.2e rank %ld\n", res, (long)D.rank); m = std::max(m, res); }
  // dense_factorize_lup: A * inverse = I, and the adjoint operators are the plain transpose / conjugate transpose
  for (int cx = 0; cx < 2; ++cx) {
    const int64_t n = 53;
    std::mt19937_64 g(9);
    std::uniform_real_distribution<double> u(-1, 1);
    double res = 0;
    if (!cx) {
      std::vector<double> A(n * n);
      for (auto &v : A) v = u(g);
      HostDense<double> D;
      dense_factorize_lup(D, A.data(), n);
      for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < n; ++i) {
        double a = 0; for (int64_t k = 0; k < n; ++k) a += A[i + k * n] * D.QH[k + j * n];
        res = std::max(res, std::fabs(a - (i == j ? 1.0 : 0.0)));
      }
    } else {
      std::vector<zdouble> A(n * n);
      for (auto &v : A) v = zdouble(u(g), u(g));
      HostDense<zdouble> D;
      dense_factorize_lup(D, A.data(), n);
      for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < n; ++i) {
        zdouble a = 0; for (int64_t k = 0; k < n; ++k) a += A[i + k * n] * D.QH[k + j * n];
        res = std::max(res, std::abs(a - (i == j ? zdouble(1) : zdouble(0))));
      }
      dense_lup_ops(D, true);
      for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < n; ++i)
        res = std::max(res, std::abs(D.SymMul[i + j * n] - std::conj(A[j + i * n])));
    }
    printf("lup cplx=%d |A inv(A) - I| = %.2e\n", cx, res);
    m = std::max(m, res);
  }
  printf(m < 1e-9 ? "OK\n" : "FAIL\n");
  return m < 1e-9 ? 0 : 1;
}
