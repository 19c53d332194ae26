// float64 validation path: same kernels instantiated in double.
#include "mjb_kernels.hpp"
namespace mjb {
template <>
hipError_t launch_step<double, double>(int G, const DevModel<double>* m, const Lay* Ldev, const Lay& L, const DevData<double>& d, const DevDebug<double>& dbg,
                                       const StepArgs& a, const ObsSpecDev& obs, double* obs_out, hipStream_t stream) {
  MJB_DISPATCH_G(G, return (launch_step_g<double, double, GG>(m, Ldev, L, d, dbg, a, obs, obs_out, stream)));
  return hipErrorInvalidValue;
}
template <>
int step_blocks_per_cu<double, double>(int G, const Lay& L) {
  switch (G) {
    case 8: return step_blocks_per_cu_g<double, double, 8>(L);
    case 16: return step_blocks_per_cu_g<double, double, 16>(L);
    case 64: return step_blocks_per_cu_g<double, double, 64>(L);
  }
  return 0;
}
}  // namespace mjb
