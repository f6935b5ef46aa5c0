// Reference-arithmetic (double) instantiation: the parity path.  This file is compiled with -ffp-contract=off so
// that no multiply-add is fused: the reference (g++ -O2, baseline x86-64) rounds every product and sum separately.
#include "rt_launch_impl.h"

namespace rtamd {

template hipError_t launch_render<double>(const SceneT<double> &, const CameraT<double> &, const RenderLaunch &, void *,
                                          unsigned long long *, hipStream_t);
template hipError_t launch_trace<double>(const SceneT<double> &, const rt_ray *, const double *, size_t, rt_hit *, uint32_t,
                                         uint32_t, int, hipStream_t);

} // namespace rtamd
