// finite-difference linearisation columns: float64 arithmetic over an fp32 state.
#include "mjb_kernels.hpp"
namespace mjb {
template <>
hipError_t launch_fd<double, float>(int G, const DevModel<double>* m, const Lay* Ldev, const Lay& L, const DevData<float>& d, int ncol, int nv, int nu, int chunk, double eps, double* y, int* valid, hipStream_t stream) {
  MJB_DISPATCH_G(G, return (launch_fd_g<double, float, GG>(m, Ldev, L, d, ncol, nv, nu, chunk, eps, y, valid, stream)));
  return hipErrorInvalidValue;
}
template <>
hipError_t launch_jac<double, float>(int G, const DevModel<double>* m, const Lay* Ldev, const Lay& L, const DevData<float>& d, int nreq, const int* kinds, const int* ids, double* out_p, double* out_r, hipStream_t stream) {
  MJB_DISPATCH_G(G, return (launch_jac_g<double, float, GG>(m, Ldev, L, d, nreq, kinds, ids, out_p, out_r, stream)));
  return hipErrorInvalidValue;
}
}  // namespace mjb
