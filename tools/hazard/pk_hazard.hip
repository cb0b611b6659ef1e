// Minimal-repro attempt for the hazard described in DESIGN.md section 5: a 12-term fp32 sum written exactly as the
// failing build of the residual+LayerNorm epilogue had it (v_pk_add_f32 with op_sel, v_mov_b32 into one half of a
// 64-bit operand right before the packed add that reads the pair), executed while the same wave has global loads in
// flight.  Every lane recomputes the sum with plain adds in the same association order and counts disagreements.
// Round 2: the same sequence with MFMA bursts issued by the waves that share the SIMD (mode "mfma"): tools/hazard/diag_ln.py
// showed that the real failure needs a second workgroup on the CU AND its MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/hazard/pk_hazard.hip -o tools/hazard/build/pk_hazard && tools/hazard/build/pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256, 2) void pk_hazard_kernel(const float* __restrict__ in, const float4* __restrict__ junk, size_t junk_n,
                                                        unsigned* __restrict__ bad, unsigned* __restrict__ bad_lane, float* __restrict__ sink, float4* __restrict__ scratch,
                                                        int iters, int with_traffic, int with_mfma) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float x[12];
  for (int i = 0; i < 12; ++i) x[i] = in[(size_t)gid * 12 + i];
  float acc_junk = 0.f;
  unsigned nbad = 0;
  bf16x8 ma, mb;
  for (int i = 0; i < 8; ++i) { ma[i] = (__bf16)(x[i] * 0.25f); mb[i] = (__bf16)(x[11 - i] * 0.25f); }
  f32x4 macc[6];
  for (int i = 0; i < 6; ++i) macc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    float4 j0 = make_float4(0, 0, 0, 0), j1 = j0, j2 = j0;
    if (with_traffic) {                       // loads in flight while the packed adds execute (like the tile prefetch)
      const size_t o = ((size_t)it * gridDim.x * 256 * 3 + (size_t)gid * 3) % (junk_n - 3);
      j0 = junk[o]; j1 = junk[o + 1]; j2 = junk[o + 2];
    }
    // MFMA burst (24 back-to-back v_mfma_f32_16x16x32_bf16, like one GEMM tile); odd waves run it before the packed sum,
    // even waves after it, so a wave's packed adds meet its SIMD partner's MFMAs
#define MFMA_BURST()                                                                                         \
  asm volatile(                                                                                              \
      "v_mfma_f32_16x16x32_bf16 v[120:123], %0, %1, v[120:123]\n\tv_mfma_f32_16x16x32_bf16 v[124:127], %0, %1, v[124:127]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[128:131], %0, %1, v[128:131]\n\tv_mfma_f32_16x16x32_bf16 v[132:135], %0, %1, v[132:135]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[136:139], %0, %1, v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], %0, %1, v[140:143]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[120:123], %0, %1, v[120:123]\n\tv_mfma_f32_16x16x32_bf16 v[124:127], %0, %1, v[124:127]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[128:131], %0, %1, v[128:131]\n\tv_mfma_f32_16x16x32_bf16 v[132:135], %0, %1, v[132:135]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[136:139], %0, %1, v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], %0, %1, v[140:143]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[120:123], %0, %1, v[120:123]\n\tv_mfma_f32_16x16x32_bf16 v[124:127], %0, %1, v[124:127]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[128:131], %0, %1, v[128:131]\n\tv_mfma_f32_16x16x32_bf16 v[132:135], %0, %1, v[132:135]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[136:139], %0, %1, v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], %0, %1, v[140:143]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[120:123], %0, %1, v[120:123]\n\tv_mfma_f32_16x16x32_bf16 v[124:127], %0, %1, v[124:127]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[128:131], %0, %1, v[128:131]\n\tv_mfma_f32_16x16x32_bf16 v[132:135], %0, %1, v[132:135]\n\t" \
      "v_mfma_f32_16x16x32_bf16 v[136:139], %0, %1, v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], %0, %1, v[140:143]\n\t" \
      "s_nop 15\n\t"                                                                                          \
      :: "v"(ma), "v"(mb) : "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135", \
         "v136","v137","v138","v139","v140","v141","v142","v143")
    if (with_mfma && ((wave + blockIdx.x) & 1)) { MFMA_BURST(); }
    // perturb the inputs every iteration so nothing is hoisted
    const float d = (float)(it & 7) * 0.125f;
    f2 P0 = {x[0] + d, x[1]}, P1 = {x[2], x[3] - d}, P2 = {x[4], x[5] + d}, P3 = {x[6] - d, x[7]}, P4 = {x[8], x[9] + d}, P5 = {x[10] - d, x[11]};
    float S;
    float4* sp = scratch + (size_t)gid * 3;
    asm volatile(
        // operands into the registers the failing build used: x0 = v[106:109], x1 = v[102:105], x2 = v[98:101]
        "v_mov_b32 v106, %[a0]\n\tv_mov_b32 v107, %[a1]\n\tv_mov_b32 v108, %[a2]\n\tv_mov_b32 v109, %[a3]\n\t"
        "v_mov_b32 v102, %[b0]\n\tv_mov_b32 v103, %[b1]\n\tv_mov_b32 v104, %[b2]\n\tv_mov_b32 v105, %[b3]\n\t"
        "v_mov_b32 v98, %[c0]\n\tv_mov_b32 v99, %[c1]\n\tv_mov_b32 v100, %[c2]\n\tv_mov_b32 v101, %[c3]\n\t"
        "v_mov_b32 v164, 0\n\t"
        "v_mov_b32 v116, %[plo]\n\tv_mov_b32 v117, %[phi]\n\t"
        "s_nop 1\n\t"
        "global_store_dwordx4 v[116:117], v[106:109], off\n\t"
        "global_store_dwordx4 v[116:117], v[102:105], off offset:16\n\t"
        "global_store_dwordx4 v[116:117], v[98:101], off offset:32\n\t"
        "v_pk_add_f32 v[112:113], v[106:107], v[106:107] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 v[114:115], v[102:103], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 v[112:113], v[112:113], v[108:109]\n\t"
        "v_pk_add_f32 v[114:115], v[114:115], v[104:105]\n\t"
        "v_mov_b32 v113, v98\n\t"
        "v_mov_b32 v116, v109\n\t"
        "v_mov_b32 v117, v99\n\t"
        "v_pk_add_f32 v[114:115], v[114:115], v[104:105] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 v[112:113], v[112:113], v[116:117]\n\t"
        "v_mov_b32 v165, v100\n\t"
        "v_pk_add_f32 v[112:113], v[112:113], v[164:165]\n\t"
        "v_mov_b32 v115, v101\n\t"
        "v_pk_add_f32 v[112:113], v[112:113], v[114:115]\n\t"
        "s_nop 0\n\t"
        "v_add_f32 %[s], v112, v113\n\t"
        "s_nop 1\n\t"
        : [s] "=v"(S)
        : [a0] "v"(P0.x), [a1] "v"(P0.y), [a2] "v"(P1.x), [a3] "v"(P1.y), [b0] "v"(P2.x), [b1] "v"(P2.y), [b2] "v"(P3.x), [b3] "v"(P3.y),
          [c0] "v"(P4.x), [c1] "v"(P4.y), [c2] "v"(P5.x), [c3] "v"(P5.y),
          [plo] "v"((unsigned)(uintptr_t)sp), [phi] "v"((unsigned)((uintptr_t)sp >> 32))
        : "memory", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v112", "v113", "v114", "v115",
          "v116", "v117", "v164", "v165");
    // the same sum, same association order, plain adds
    const float alo = ((((P0.x + P0.y) + P1.x) + P1.y) + 0.f) + (((P2.x + P2.y) + P3.x) + P3.y);
    const float ahi = ((P4.x + P4.y) + P5.x) + P5.y;
    const float E = alo + ahi;
    if (S != E) { ++nbad; atomicAdd(&bad_lane[lane], 1u); }
    if (with_mfma && !((wave + blockIdx.x) & 1)) { MFMA_BURST(); }
  }
  if (nbad) atomicAdd(bad, nbad);
  for (int i = 0; i < 6; ++i) acc_junk += macc[i][0] + macc[i][3];
  if (acc_junk == 123.456f) sink[0] = acc_junk;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int blocks = argc > 2 ? atoi(argv[2]) : 256 * 2;      // 2 workgroups (8 waves) per CU: two waves per SIMD
  const size_t n = (size_t)blocks * 256 * 12;
  std::vector<float> h(n);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 4.f - 2.f;
  float *din, *sink; float4* junk; unsigned *bad, *bad_lane;
  const size_t junk_n = (size_t)1 << 26;      // 1 GiB of float4
  hipMalloc(&din, n * 4); hipMalloc(&sink, 4); hipMalloc(&junk, junk_n * 16); hipMalloc(&bad, 4); hipMalloc(&bad_lane, 64 * 4);
  float4* scratch; hipMalloc(&scratch, (size_t)blocks * 256 * 48);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(junk, 0, junk_n * 16);
  for (int mode = 0; mode < 4; ++mode) {
    const int traffic = mode & 1, mfma = mode >> 1;
    hipMemset(bad, 0, 4); hipMemset(bad_lane, 0, 64 * 4);
    hipLaunchKernelGGL(pk_hazard_kernel, dim3(blocks), dim3(256), 0, 0, din, junk, junk_n, bad, bad_lane, sink, scratch, iters, traffic, mfma);
    hipDeviceSynchronize();
    unsigned hb = 0, hl[64];
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hl, bad_lane, 256, hipMemcpyDeviceToHost);
    printf("mfma=%d traffic=%d: %u mismatching sums of %.3g  per 16-lane group:", mfma, traffic, hb, (double)blocks * 256 * iters);
    for (int g = 0; g < 4; ++g) { unsigned s = 0; for (int l = 0; l < 16; ++l) s += hl[16 * g + l]; printf(" %u", s); }
    printf("\n");
  }
  return 0;
}
