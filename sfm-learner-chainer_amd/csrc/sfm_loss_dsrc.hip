// The gradient kernels of a launch that also produces the optional dL/d(src) (SfmLossDesc.d_src; loss_kernel_dsrc, the LDS
// accumulation window of sfm_ssim_pass.h).  A translation unit of its own so that it compiles next to sfm_loss.hip (make -j).
#include "sfm_loss_kernels.h"

namespace sfm {

template <bool LOSS>
static const void* pick_dsrc(bool ssim, bool expl, int smode, bool hwc, bool warped) {
#define SFM_KPICK(...)                                                                                                                   \
  do {                                                                                                                                   \
    if constexpr (LOSS) {                                                                                                                \
      if (warped) return hwc ? (const void*)&loss_kernel_dsrc<__VA_ARGS__, true, true> : (const void*)&loss_kernel_dsrc<__VA_ARGS__, false, true>; \
    }                                                                                                                                    \
    return hwc ? (const void*)&loss_kernel_dsrc<__VA_ARGS__, true, false> : (const void*)&loss_kernel_dsrc<__VA_ARGS__, false, false>;   \
  } while (0)
  if (expl) {
    if (smode == 0) SFM_KPICK(false, LOSS, true, 0);
    else if (smode == 1) SFM_KPICK(false, LOSS, true, 1);
    else SFM_KPICK(false, LOSS, true, 2);
  } else if (ssim) {
    if (smode == 0) SFM_KPICK(true, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(true, LOSS, false, 1);
    else SFM_KPICK(true, LOSS, false, 2);
  } else {
    if (smode == 0) SFM_KPICK(false, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(false, LOSS, false, 1);
    else SFM_KPICK(false, LOSS, false, 2);
  }
#undef SFM_KPICK
}

const void* kernel_ptr_dsrc(bool loss, bool ssim, bool expl, int smode, bool hwc, bool warped) {
  return loss ? pick_dsrc<true>(ssim, expl, smode, hwc, warped) : pick_dsrc<false>(ssim, expl, smode, hwc, warped);
}

}  // namespace sfm
