// The SSIM kernels that walk TWO sources per pass (loss_kernel_pair; sfm_ssim_pair.h): pixel-interleaved layout, the product's
// projection, gradient launches of an even number of sources without the warped output.  A translation unit of its own (compiles
// next to sfm_loss.hip).
#include "sfm_loss_kernels.h"

namespace sfm {

template <bool GRAD, bool LOSS>
static const void* pick_pair(int smode) {
  if (smode == 0) return (const void*)&loss_kernel_pair<GRAD, LOSS, 0>;
  if (smode == 1) return (const void*)&loss_kernel_pair<GRAD, LOSS, 1>;
  return (const void*)&loss_kernel_pair<GRAD, LOSS, 2>;
}

const void* kernel_ptr_pair(bool grad, bool loss, int smode) {
  if (grad && loss) return pick_pair<true, true>(smode);
  if (grad) return pick_pair<true, false>(smode);
  return nullptr;     // (the forward-only launches keep one source per pass: four waves per SIMD at 128 registers)
}

}  // namespace sfm
