// float64 validation path: same kernels instantiated in double.
#include "mjb_kernels.hpp"
namespace mjb {
template <>
hipError_t launch_step<double, double>(int G, const DevModel<double>* m, const Lay* Ldev, const Lay& L, const DevData<double>& d, const DevDebug<double>& dbg,
                                       const StepArgs& a, const ObsSpecDev& obs, double* obs_out, hipStream_t stream) {
  MJB_DISPATCH_G(G, return (launch_step_g<double, double, GG>(m, Ldev, L, d, dbg, a, obs, obs_out, stream)));
  return hipErrorInvalidValue;
}
}  // namespace mjb
