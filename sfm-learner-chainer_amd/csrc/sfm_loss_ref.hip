// The fused-loss kernels of SfmLossDesc.projection = SFM_PROJECTION_REFERENCE_ORDER: the instantiations of loss_body with the
// per-pixel projection in the reference's own evaluation order (sfm_ssim_pass.h, ref_position; models/transform.py:105-108,122-131,189),
// one for every (entry point, loss mode, smoothness form, layout, warped output) the product's projection has -- except the
// three-waves-per-SIMD builds of the small L1 launches (loss_kernel_wide) and the launches that also produce dL/d(src).
// A translation unit of its own so that it compiles next to sfm_loss.hip (make -j).
#include "sfm_loss_kernels.h"

namespace sfm {

template <bool GRAD, bool LOSS>
static const void* pick_ref(bool ssim, bool expl, int smode, bool hwc, bool warped) {
  // (WARPED only exists for the LOSS entry points)
#define SFM_KPICK(...)                                                                                                              \
  do {                                                                                                                              \
    if constexpr (LOSS) {                                                                                                           \
      if (warped) return hwc ? (const void*)&loss_kernel_ref<__VA_ARGS__, true, true> : (const void*)&loss_kernel_ref<__VA_ARGS__, false, true>; \
    }                                                                                                                               \
    return hwc ? (const void*)&loss_kernel_ref<__VA_ARGS__, true, false> : (const void*)&loss_kernel_ref<__VA_ARGS__, false, false>; \
  } while (0)
  if (expl) {
    if (smode == 0) SFM_KPICK(false, GRAD, LOSS, true, 0);
    else if (smode == 1) SFM_KPICK(false, GRAD, LOSS, true, 1);
    else SFM_KPICK(false, GRAD, LOSS, true, 2);
  } else if (ssim) {
    if (smode == 0) SFM_KPICK(true, GRAD, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(true, GRAD, LOSS, false, 1);
    else SFM_KPICK(true, GRAD, LOSS, false, 2);
  } else {
    if (smode == 0) SFM_KPICK(false, GRAD, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(false, GRAD, LOSS, false, 1);
    else SFM_KPICK(false, GRAD, LOSS, false, 2);
  }
#undef SFM_KPICK
}

const void* kernel_ptr_ref(bool grad, bool loss, bool ssim, bool expl, int smode, bool hwc, bool warped) {
  if (grad && loss) return pick_ref<true, true>(ssim, expl, smode, hwc, warped);
  if (grad) return pick_ref<true, false>(ssim, expl, smode, hwc, warped);
  return pick_ref<false, true>(ssim, expl, smode, hwc, warped);
}

}  // namespace sfm
