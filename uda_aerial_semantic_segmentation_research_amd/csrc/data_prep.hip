// Device-side input pipeline: what the reference does per sample on the host between the decoded
// uint8 image and the model input --
//   src/data/dataset.py:116-138      uint8 RGB HWC image + uint8 label mask -> (float image, int64 mask)
//   src/models/augmentation.py:11-13 RandomRotate90 / Flip / Transpose: together an element of the dihedral group D4,
//                                    applied identically to image and mask
//   src/models/augmentation.py:36    A.Normalize(): (x - 255*mean) * (1 / (255*std)), fp32, ImageNet mean/std
// -- as ONE pass that writes the padded NHWC tensor the stem convolution reads (fp32 or bf16) and the int64 mask.
// HBM-bound byte shuffling: 4 B/pixel read, 4*cpad (or 2*cpad) + 8 B/pixel written.
#include "common.h"

namespace udaseg {

// D4 code: bit0 = transpose, bit1 = vertical flip (rows), bit2 = horizontal flip (columns), applied in that order:
//   out = fliph^b2( flipv^b1( transpose^b0( in ) ) )
template <bool BF16>
__global__ __launch_bounds__(256) void prepare_batch_kernel(const uint8_t* __restrict__ images, const uint8_t* __restrict__ masks,
                                                            const int32_t* __restrict__ d4, int h, int w, float m0, float m1,
                                                            float m2, float r0, float r1, float r2, void* __restrict__ out,
                                                            int cpad, int64_t* __restrict__ out_masks) {
  const int ni = blockIdx.y;
  const int code = d4 ? d4[ni] : 0;
  const int hw = h * w;
  const uint8_t* img = images + (size_t)ni * hw * 3;
  const uint8_t* msk = masks ? masks + (size_t)ni * hw : nullptr;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256) {
    int y = p / w, x = p - y * w;
    if (code & 2) y = h - 1 - y;
    if (code & 4) x = w - 1 - x;
    const int sp = (code & 1) ? x * w + y : y * w + x;        // transpose needs h == w (checked by the caller)
    const float v0 = ((float)img[sp * 3 + 0] - m0) * r0;
    const float v1 = ((float)img[sp * 3 + 1] - m1) * r1;
    const float v2 = ((float)img[sp * 3 + 2] - m2) * r2;
    const size_t o = ((size_t)ni * hw + p) * cpad;
    if (BF16) {
      __bf16* dst = reinterpret_cast<__bf16*>(out) + o;
      dst[0] = (__bf16)v0;
      dst[1] = (__bf16)v1;
      dst[2] = (__bf16)v2;
      for (int k = 3; k < cpad; ++k) dst[k] = (__bf16)0.f;
    } else {
      float* dst = reinterpret_cast<float*>(out) + o;
      *reinterpret_cast<f32x4*>(dst) = f32x4{v0, v1, v2, 0.f};
      for (int k = 4; k < cpad; ++k) dst[k] = 0.f;
    }
    if (msk) out_masks[(size_t)ni * hw + p] = (int64_t)msk[sp];
  }
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_prepare_batch_u8(const uint8_t* images, const uint8_t* masks, const int32_t* d4, int n, int h, int w,
                                       const float* mean255, const float* inv_std255, void* out_images, int cpad, int out_bf16,
                                       int64_t* out_masks, int square_checked, void* stream) {
  UDASEG_CHECK_ARG(images && out_images && mean255 && inv_std255 && n > 0 && h > 0 && w > 0, "prepare_batch_u8: bad arguments");
  UDASEG_CHECK_ARG(cpad >= 4 && cpad % (out_bf16 ? 8 : 4) == 0, "prepare_batch_u8: cpad must be a multiple of %d",
                   out_bf16 ? 8 : 4);
  UDASEG_CHECK_ARG((masks == nullptr) == (out_masks == nullptr), "prepare_batch_u8: masks and out_masks go together");
  UDASEG_CHECK_ARG(d4 == nullptr || h == w || square_checked, "prepare_batch_u8: transposing codes need square images; pass "
                   "square_checked=1 after making sure no code has bit 0 set");
  UDASEG_CHECK_ARG((int64_t)h * w < (1LL << 30), "prepare_batch_u8: image too large");
  const int gx = (h * w + 255) / 256 > 1024 ? 1024 : (h * w + 255) / 256;
  hipStream_t st = as_stream(stream);
  if (out_bf16)
    hipLaunchKernelGGL(prepare_batch_kernel<true>, dim3(gx, n), dim3(256), 0, st, images, masks, d4, h, w, mean255[0], mean255[1],
                       mean255[2], inv_std255[0], inv_std255[1], inv_std255[2], out_images, cpad, out_masks);
  else
    hipLaunchKernelGGL(prepare_batch_kernel<false>, dim3(gx, n), dim3(256), 0, st, images, masks, d4, h, w, mean255[0], mean255[1],
                       mean255[2], inv_std255[0], inv_std255[1], inv_std255[2], out_images, cpad, out_masks);
  UDASEG_LAUNCH_CHECK("prepare_batch launch");
  return UDASEG_OK;
}
