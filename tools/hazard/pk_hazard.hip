// Minimal-repro attempt for the hazard described in DESIGN.md section 5: a 12-term fp32 sum written exactly as the
// failing build of the residual+LayerNorm epilogue had it (v_pk_add_f32 with op_sel, v_mov_b32 into one half of a
// 64-bit operand right before the packed add that reads the pair), executed while the same wave has global loads in
// flight.  Every lane recomputes the sum with plain adds in the same association order and counts disagreements.
//   hipcc --offload-arch=gfx950 -O3 tools/hazard/pk_hazard.hip -o tools/hazard/pk_hazard && tools/hazard/pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pk_hazard_kernel(const float* __restrict__ in, const float4* __restrict__ junk, size_t junk_n,
                                                        unsigned* __restrict__ bad, unsigned* __restrict__ bad_lane, float* __restrict__ sink,
                                                        int iters, int with_traffic) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float x[12];
  for (int i = 0; i < 12; ++i) x[i] = in[(size_t)gid * 12 + i];
  float acc_junk = 0.f;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    float4 j0 = make_float4(0, 0, 0, 0), j1 = j0, j2 = j0;
    if (with_traffic) {                       // loads in flight while the packed adds execute (like the tile prefetch)
      const size_t o = ((size_t)it * gridDim.x * 256 * 3 + (size_t)gid * 3) % (junk_n - 3);
      j0 = junk[o]; j1 = junk[o + 1]; j2 = junk[o + 2];
    }
    // perturb the inputs every iteration so nothing is hoisted
    const float d = (float)(it & 7) * 0.125f;
    f2 P0 = {x[0] + d, x[1]}, P1 = {x[2], x[3] - d}, P2 = {x[4], x[5] + d}, P3 = {x[6] - d, x[7]}, P4 = {x[8], x[9] + d}, P5 = {x[10] - d, x[11]};
    float S;
    asm volatile(
        "v_mov_b32 v206, 0\n\t"
        "v_pk_add_f32 v[200:201], %[p0], %[p0] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 v[202:203], %[p2], %[p2] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 v[200:201], v[200:201], %[p1]\n\t"
        "v_pk_add_f32 v[202:203], v[202:203], %[p3]\n\t"
        "v_mov_b32 v201, %[x8]\n\t"
        "v_mov_b32 v204, %[x3]\n\t"
        "v_mov_b32 v205, %[x9]\n\t"
        "v_pk_add_f32 v[202:203], v[202:203], %[p3] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 v[200:201], v[200:201], v[204:205]\n\t"
        "v_mov_b32 v207, %[x10]\n\t"
        "v_pk_add_f32 v[200:201], v[200:201], v[206:207]\n\t"
        "v_mov_b32 v203, %[x11]\n\t"
        "v_pk_add_f32 v[200:201], v[200:201], v[202:203]\n\t"
        "s_nop 1\n\t"
        "v_add_f32 %[s], v200, v201\n\t"
        : [s] "=v"(S)
        : [p0] "v"(P0), [p1] "v"(P1), [p2] "v"(P2), [p3] "v"(P3), [x8] "v"(P4.x), [x3] "v"(P1.y), [x9] "v"(P4.y), [x10] "v"(P5.x),
          [x11] "v"(P5.y)
        : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");
    // the same sum, same association order, plain adds
    const float alo = ((((P0.x + P0.y) + P1.x) + P1.y) + 0.f) + (((P2.x + P2.y) + P3.x) + P3.y);
    const float ahi = ((P4.x + P4.y) + P5.x) + P5.y;
    const float E = alo + ahi;
    if (S != E) { ++nbad; atomicAdd(&bad_lane[lane], 1u); }
    acc_junk += j0.x + j1.y + j2.z;
  }
  if (nbad) atomicAdd(bad, nbad);
  if (acc_junk == 123.456f) sink[0] = acc_junk;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int blocks = 256 * 8;
  const size_t n = (size_t)blocks * 256 * 12;
  std::vector<float> h(n);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 4.f - 2.f;
  float *din, *sink; float4* junk; unsigned *bad, *bad_lane;
  const size_t junk_n = (size_t)1 << 26;      // 1 GiB of float4
  hipMalloc(&din, n * 4); hipMalloc(&sink, 4); hipMalloc(&junk, junk_n * 16); hipMalloc(&bad, 4); hipMalloc(&bad_lane, 64 * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(junk, 0, junk_n * 16);
  for (int traffic = 0; traffic < 2; ++traffic) {
    hipMemset(bad, 0, 4); hipMemset(bad_lane, 0, 64 * 4);
    hipLaunchKernelGGL(pk_hazard_kernel, dim3(blocks), dim3(256), 0, 0, din, junk, junk_n, bad, bad_lane, sink, iters, traffic);
    hipDeviceSynchronize();
    unsigned hb = 0, hl[64];
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hl, bad_lane, 256, hipMemcpyDeviceToHost);
    printf("traffic=%d: %u mismatching sums of %.3g  per 16-lane group:", traffic, hb, (double)blocks * 256 * iters);
    for (int g = 0; g < 4; ++g) { unsigned s = 0; for (int l = 0; l < 16; ++l) s += hl[16 * g + l]; printf(" %u", s); }
    printf("\n");
  }
  return 0;
}
