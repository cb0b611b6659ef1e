// On-device CutMix / MixUp of an image batch.
//
// Reference call site: /root/reference/training/trainer.py:84-96 -- `cutmix_or_mixup(images, class_labels, use_cutmix,
// use_mixup, cutmix_alpha, mixup_alpha) -> (images, labels_a, labels_b, lam)`, imported from data/transforms.py,
// which is NOT part of the reference checkout (SURVEY.md section 2 lists it as missing).  The arithmetic below is
// therefore the published definition of the two augmentations (MixUp: Zhang et al. 2018, CutMix: Yun et al. 2019),
// "parity unpinned" against the reference, pinned against a torch restatement in tests/test_gpu_augment.py:
//   mixup : out[b] = lam * x[b] + (1 - lam) * x[perm[b]]
//   cutmix: out[b] = x[b] outside the box, x[perm[b]] inside rows [y0,y1) x cols [x0,x1)
// One pass, out-of-place (the source batch is read through perm, so in-place would race), float4 accesses:
// HBM-bound, 2 reads + 1 write of the batch for mixup, ~1 read + 1 write for cutmix.
#include "common.h"

namespace {

struct MixArgs {
  const float* x; float* out; const long long* perm;
  int C, H, W4;              // W4 = W / 4 (float4 columns)
  int y0, y1, x0, x1;        // cutmix box (pixels); empty for mixup
  float lam;
  int mode;                  // 0 = mixup, 1 = cutmix
  size_t total4;             // B*C*H*W4
};

__global__ __launch_bounds__(256) void mix_images_kernel(const MixArgs a) {
  const size_t per_img = (size_t)a.C * a.H * a.W4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.total4; i += (size_t)gridDim.x * 256) {
    const size_t b = i / per_img, r = i - b * per_img;
    const int w4 = (int)(r % a.W4);
    const int y = (int)((r / a.W4) % a.H);
    const float4 own = reinterpret_cast<const float4*>(a.x)[i];
    float4 o = own;
    if (a.mode == 0) {
      const float4 oth = reinterpret_cast<const float4*>(a.x)[(size_t)a.perm[b] * per_img + r];
      const float l = a.lam, m = 1.f - a.lam;
      o.x = l * own.x + m * oth.x; o.y = l * own.y + m * oth.y; o.z = l * own.z + m * oth.z; o.w = l * own.w + m * oth.w;
    } else if (y >= a.y0 && y < a.y1 && 4 * w4 + 3 >= a.x0 && 4 * w4 < a.x1) {
      const float4 oth = reinterpret_cast<const float4*>(a.x)[(size_t)a.perm[b] * per_img + r];
      const int c = 4 * w4;
      if (c + 0 >= a.x0 && c + 0 < a.x1) o.x = oth.x;
      if (c + 1 >= a.x0 && c + 1 < a.x1) o.y = oth.y;
      if (c + 2 >= a.x0 && c + 2 < a.x1) o.z = oth.z;
      if (c + 3 >= a.x0 && c + 3 < a.x1) o.w = oth.w;
    }
    reinterpret_cast<float4*>(a.out)[i] = o;
  }
}

}  // namespace

extern "C" int rovit_mix_images(const float* images, float* out, const long long* perm, int batch, int channels, int height,
                                int width, int mode, float lam, int y0, int y1, int x0, int x1, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(images && out && perm, ROVIT_ERR_NULL, "mix_images: null pointer");
  ROVIT_CHECK_ARG(images != out, ROVIT_ERR_SHAPE, "mix_images: must be out of place");
  ROVIT_CHECK_ARG(batch > 0 && channels > 0 && height > 0 && width > 0 && width % 4 == 0, ROVIT_ERR_SHAPE,
                  "mix_images: bad shape (width must be a multiple of 4)");
  ROVIT_CHECK_ARG(rovit_aligned16(images) && rovit_aligned16(out), ROVIT_ERR_ALIGN, "mix_images: alignment");
  ROVIT_CHECK_ARG(mode == 0 || mode == 1, ROVIT_ERR_SHAPE, "mix_images: mode 0 (mixup) or 1 (cutmix)");
  if (mode == 1)
    ROVIT_CHECK_ARG(0 <= y0 && y0 <= y1 && y1 <= height && 0 <= x0 && x0 <= x1 && x1 <= width, ROVIT_ERR_SHAPE,
                    "mix_images: box outside the image");
  MixArgs a{images, out, perm, channels, height, width / 4, y0, y1, x0, x1, lam, mode,
            (size_t)batch * channels * height * (width / 4)};
  const size_t blocks = (a.total4 + 255) / 256;
  hipLaunchKernelGGL(mix_images_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("mix_images_kernel");
  return ROVIT_OK;
}
