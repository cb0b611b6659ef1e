// Ablation lab for the fp32-matrix-core GEMM of csrc/vit_f32.hip (developer tool, not part of the library): times the product kernel and
// its ablations (no global loads in the loop / no restaging / no epilogue) on the four GEMM shapes of one 256-image block, so that the gap
// between the measured duration and the matrix-pipe time can be attributed.  Build + run (GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -fno-vectorize -I rovit-*/csrc -I include tools/lab/f32_gemm_lab.hip \
//         rovit-*/lib/librovit_hip.so -o /tmp/f32_gemm_lab && /tmp/f32_gemm_lab
#include "vit_f32.hip"
namespace { constexpr int GBN = 192; }      // the lab's LDS-DMA kernel keeps the 128 x 192 tile
#include <cstdio>
#include <vector>

namespace {
// ---- the same GEMM with the operand tiles brought in by LDS-DMA and a 64 x 96 tile per wave -------------------------------------------
// The fp32 matrix instruction runs at the rate of the vector FMA: it uses the whole register-file bandwidth of its SIMD, and every other
// register access takes cycles from it (tools/lab/f32_gemm_lab.hip: the kernel above needs 1.24 x the time of its MFMAs alone for the
// fragment reads -- 4 ds_read_b128 per 12 MFMAs -- and 1.5 x with the register-staged global -> LDS copy on top).  So this kernel
//   * moves global -> LDS by global_load_lds_dwordx4 (no register passes through the copy),
//   * gives a wave 2 x 3 accumulator tiles (96 registers): 5 ds_read_b128 feed 24 MFMAs instead of 4 feeding 12,
//   * runs 4 waves per workgroup (wm, wn in 0..1), the same 128 x 192 output tile, two workgroups per CU (2 x 80 KB of LDS).
// LDS: ring of 2 stages x 320 rows (128 of A, 192 of W) x 32 floats, rows UNPADDED (the DMA writes lane-linear: 64 lanes x 16 bytes), bank
// conflicts avoided by an XOR of the 16-byte chunk index with (row >> 1) & 7 on the DMA's source address and on the fragment read (the
// same swizzle as gemm_ws_dma_kernel; 16 consecutive rows x one logical chunk cover the 64 banks once).
// Per stage: wait for the wave's own pieces of stage t (vmcnt 0: nothing younger is in flight yet), barrier (every piece of t landed, every
// wave done with t - 1), request stage t + 1 into the other slot, compute t.  Operand order and summation order are those of the kernel above:
// bit-identical results.
constexpr int DSTAGE = (GBM + GBN) * GBK;                          // floats per ring slot (40 KB)
constexpr size_t DMA_LDS = (size_t)2 * DSTAGE * sizeof(float);    // 80 KB
template <int EPI, int LAB = 0>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N,
                                                             int K, const float* __restrict__ pos) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  // DMA pieces of this wave: piece j = w + 4 i (i < 10) = stage rows 8 j .. 8 j + 7; lane = (row 8 j + (lane >> 3), physical chunk lane & 7)
  const float* src[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int rr = 8 * (w + 4 * i) + (lane >> 3), cc = (lane & 7) ^ ((rr >> 1) & 7);       // logical chunk this lane fetches
    if (i < 4) {                                                                           // pieces 0 .. 15: the A rows
      int m = m0 + rr;
      m = m < M ? m : M - 1;                                                               // clamped: rows beyond M are computed and never stored
      if (EPI == F_PATCH) {
        const int b = m / (T - 1), pch = m - b * (T - 1);
        src[i] = A + ((size_t)b * 3 * 224 + (pch / 14) * 16) * 224 + (pch % 14) * 16 + 4 * (cc & 3) + 224 * (cc >> 2);   // + stage term below
      } else {
        src[i] = A + (size_t)m * lda + 4 * cc;
      }
    } else {
      src[i] = W + (size_t)(n0 + rr - GBM) * K + 4 * cc;
    }
  }
  auto dma = [&](int k0, int slot) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      // PatchEmbed gather: k = c * 256 + py * 16 + px; a stage is 32 consecutive k = two pixel rows of one channel (chunk cc: row cc >> 2)
      const size_t koff = (EPI == F_PATCH && i < 4) ? ((size_t)(k0 >> 8) * 224 + ((k0 >> 4) & 15)) * 224 : (size_t)k0;
      const float* sp = src[i] + koff;
      if ((LAB & 16) && i < 4)      // (lab: A as if stored stage-tiled, [M / 128][K / 32][128 x 32] -- 16 KB contiguous per stage; timing only)
        sp = A + ((size_t)blockIdx.y * (K / GBK) + (k0 / GBK)) * (GBM * GBK) + (w + 4 * i) * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                       (__attribute__((address_space(3))) void*)(dsm + slot * DSTAGE + (w + 4 * i) * 256), 16, 0, 0);
    }
  };
  dma(0, 0);
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
  // fragment reads: row R (A: 64 wm + 32 i + l31, W: 96 wn + 32 t + l31; (R >> 1) & 7 = (l31 >> 1) & 7 for all of them), logical chunk 2 c + lh
  const int swz = (l31 >> 1) & 7;
  int xo[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) xo[c] = ((2 * c + lh) ^ swz) * 4;
  const int arow = (64 * wm + l31) * GBK, wrow = (GBM + 96 * wn + l31) * GBK;
  const int nst = K / GBK;
  for (int st = 0; st < nst; ++st) {
    const float* S = dsm + (st & 1) * DSTAGE;
    if (!(LAB & 2) || st == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(LAB & 3) && st + 1 < nst) dma((st + 1) * GBK, (st + 1) & 1);
    }
    float4 fa[2][2], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[0][i] = *(const float4*)&S[arow + 32 * i * GBK + xo[0]];
#pragma unroll
    for (int t = 0; t < 3; ++t) fb[0][t] = *(const float4*)&S[wrow + 32 * t * GBK + xo[0]];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cur = c & 1, nxt = cur ^ 1;
      if (c < 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[nxt][i] = *(const float4*)&S[arow + 32 * i * GBK + xo[c + 1]];
#pragma unroll
        for (int t = 0; t < 3; ++t) fb[nxt][t] = *(const float4*)&S[wrow + 32 * t * GBK + xo[c + 1]];
      }
      __builtin_amdgcn_sched_barrier(0);                           // (the scheduler would sink the reads behind the MFMAs)
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float av[4] = {fa[cur][i].x, fa[cur][i].y, fa[cur][i].z, fa[cur][i].w};
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const float bv[4] = {fb[cur][t].x, fb[cur][t].y, fb[cur][t].z, fb[cur][t].w};
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sidx], bv[sidx], acc[i][t], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // accumulator tile: column n = lane & 31, row m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): a register is two 128-byte row segments
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int n = n0 + 96 * wn + 32 * t + l31;
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        if ((LAB & 4) && acc[i][t][r] != 12345.678f) continue;
        float v = acc[i][t][r] + bn;
        if (EPI == F_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));      // exact-erf GELU (timm default)
        if (EPI == F_PATCH) {
          const int b = m / (T - 1), pch = m - b * (T - 1);
          C[((size_t)b * T + 1 + pch) * ldc + n] = v + pos[(size_t)(1 + pch) * N + n];
        } else if (EPI == F_RESID) {
          C[(size_t)m * ldc + n] += v;
        } else {
          C[(size_t)m * ldc + n] = v;
        }
      }
  }
}

}  // namespace

template <int EPI, int LAB>
static float time_one(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(gemm_f32_grid(M, N, 192));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, LAB>), grid, dim3(512), 0, 0, A, lda, W, bias, C, ldc, M, N, K, nullptr, F32Ln{nullptr, nullptr, nullptr, nullptr, 0.f});
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, LAB>), grid, dim3(512), 0, 0, A, lda, W, bias, C, ldc, M, N, K, nullptr, F32Ln{nullptr, nullptr, nullptr, nullptr, 0.f});
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

template <int EPI, int LAB>
static float time_dma(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(N / GBN, (M + GBM - 1) / GBM);
  rovit_set_max_lds((const void*)gemm_f32_dma_kernel<EPI, LAB>, DMA_LDS);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_f32_dma_kernel<EPI, LAB>), grid, dim3(256), DMA_LDS, 0, A, lda, W, bias, C, ldc, M, N, K, nullptr);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_f32_dma_kernel<EPI, LAB>), grid, dim3(256), DMA_LDS, 0, A, lda, W, bias, C, ldc, M, N, K, nullptr);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}

template <int EPI, int WN, int NT, int LAB = 0>
static float time_tile(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(gemm_f32_grid(M, N, 32 * NT * WN));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, LAB, WN, NT>), grid, dim3(256 * WN), 0, 0, A, lda, W, bias, C, ldc, M, N, K, nullptr, F32Ln{nullptr, nullptr, nullptr, nullptr, 0.f});
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, LAB, WN, NT>), grid, dim3(256 * WN), 0, 0, A, lda, W, bias, C, ldc, M, N, K, nullptr, F32Ln{nullptr, nullptr, nullptr, nullptr, 0.f});
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}

int main() {
  const int M = 256 * 197;
  float *A, *W, *bias, *C;
  hipMalloc(&A, (size_t)M * 768 * 4); hipMalloc(&W, (size_t)768 * 768 * 4); hipMalloc(&bias, 768 * 4); hipMalloc(&C, (size_t)M * 768 * 4);
  std::vector<float> h((size_t)M * 768);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 1024.f - 0.5f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)768 * 768 * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), 768 * 4, hipMemcpyHostToDevice);
  hipMemset(C, 0, (size_t)M * 768 * 4);
  float* C2;
  hipMalloc(&C2, (size_t)M * 768 * 4);
  std::vector<float> h2(h.size());
  struct Shape { const char* name; int N, K; } shapes[] = {{"qkv  N576 K192", 576, 192}, {"proj N192 K192", 192, 192}, {"fc1  N768 K192", 768, 192}, {"fc2  N192 K768", 192, 768}};
  for (const Shape& s : shapes) {
    const double peak_us = 2.0 * M * s.N * s.K / 157.3e12 * 1e6;
    const float t0 = time_one<F_NONE, 0>(A, s.K, W, bias, C, s.N, M, s.N, s.K, 20);
    const float t1 = time_one<F_NONE, 1>(A, s.K, W, bias, C, s.N, M, s.N, s.K, 20);
    const float t2 = time_one<F_NONE, 2>(A, s.K, W, bias, C, s.N, M, s.N, s.K, 20);
    const float t4 = time_one<F_NONE, 4>(A, s.K, W, bias, C, s.N, M, s.N, s.K, 20);
    const float t6 = time_one<F_NONE, 6>(A, s.K, W, bias, C, s.N, M, s.N, s.K, 20);
    const float t14 = time_one<F_NONE, 14>(A, s.K, W, bias, C, s.N, M, s.N, s.K, 20);
    printf("%s  matrix-pipe floor %6.1f us | product %6.1f | no loop loads %6.1f | no restaging %6.1f | no epilogue %6.1f | neither %6.1f | MFMAs alone %6.1f\n", s.name, peak_us, t0, t1, t2,
           t4, t6, t14);
    hipMemset(C2, 0, (size_t)M * s.N * 4);
    time_one<F_NONE, 0>(A, s.K, W, bias, C, s.N, M, s.N, s.K, 1);
    const float d0 = time_dma<F_NONE, 0>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
    hipMemcpy(h.data(), C, (size_t)M * s.N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), C2, (size_t)M * s.N * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    float worst = 0.f;
    for (size_t i = 0; i < (size_t)M * s.N; ++i) { bad += h[i] != h2[i]; worst = fmaxf(worst, fabsf(h[i] - h2[i])); }
    printf("   max abs difference %g\n", worst);
    const float d1 = time_dma<F_NONE, 1>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
    const float d2 = time_dma<F_NONE, 2>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
    const float d4 = time_dma<F_NONE, 4>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
    const float d6 = time_dma<F_NONE, 6>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
    const float d16 = time_dma<F_NONE, 16>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
    const float d20 = time_dma<F_NONE, 20>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
    {
      hipMemset(C2, 0, (size_t)M * s.N * 4);
      const float q12 = time_tile<F_NONE, 1, 2>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      hipMemcpy(h2.data(), C2, (size_t)M * s.N * 4, hipMemcpyDeviceToHost);
      size_t bad2 = 0;
      for (size_t i = 0; i < (size_t)M * s.N; ++i) bad2 += h[i] != h2[i];
      const float q13 = time_tile<F_NONE, 1, 3>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      const float q11 = time_tile<F_NONE, 1, 1>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      const float qr = time_tile<F_RESID, 2, 3>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      const float qr12 = time_tile<F_RESID, 1, 2>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      const float l1 = time_tile<F_NONE, 1, 2, 1>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      const float l4 = time_tile<F_NONE, 1, 2, 4>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      const float l6 = time_tile<F_NONE, 1, 2, 6>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      const float l14 = time_tile<F_NONE, 1, 2, 14>(A, s.K, W, bias, C2, s.N, M, s.N, s.K, 20);
      printf("   128 x 64 tile: no loop loads %6.1f | no epilogue %6.1f | neither, no restaging %6.1f | MFMAs alone %6.1f\n", l1, l4, l6, l14);
      printf("   product kernel with a 128 x 64 tile (4 waves): %6.1f us, %zu results differ | 128 x 96: %6.1f | 128 x 32: %6.1f | residual epilogue: 128 x 192 %6.1f, 128 x 64 %6.1f\n", q12, bad2, q13, q11, qr, qr12);
    }
    printf("   LDS-DMA kernel: %zu of %zu results differ from the kernel above | product %6.1f | no loop DMA %6.1f | no restaging %6.1f | no epilogue %6.1f | neither %6.1f | A stage-tiled %6.1f, and no epilogue %6.1f\n",
           bad, (size_t)M * s.N, d0, d1, d2, d4, d6, d16, d20);
  }
  return 0;
}
