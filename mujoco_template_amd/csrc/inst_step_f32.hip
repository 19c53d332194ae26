// fp32 product path: step / forward / rollout kernels.
#include "mjb_kernels.hpp"
namespace mjb {
template <>
hipError_t launch_step<float, float>(int G, const DevModel<float>* m, const Lay* Ldev, const Lay& L, const DevData<float>& d, const DevDebug<float>& dbg,
                                     const StepArgs& a, const ObsSpecDev& obs, float* obs_out, hipStream_t stream) {
  MJB_DISPATCH_G(G, return (launch_step_g<float, float, GG>(m, Ldev, L, d, dbg, a, obs, obs_out, stream)));
  return hipErrorInvalidValue;
}
template <>
hipError_t launch_step2<float, float>(const DevModel<float>* m, const Lay* Ldev, const Lay& L, const DevData<float>& d, const StepArgs& a, const ObsSpecDev& obs, float* obs_out, hipStream_t stream) {
  return launch_step2_impl<float, float>(m, Ldev, L, d, a, obs, obs_out, stream);
}
template <>
int step_blocks_per_cu<float, float>(int G, const Lay& L) {
  switch (G) {
    case 8: return step_blocks_per_cu_g<float, float, 8>(L);
    case 16: return step_blocks_per_cu_g<float, float, 16>(L);
    case 64: return step_blocks_per_cu_g<float, float, 64>(L);
  }
  return 0;
}
}  // namespace mjb
