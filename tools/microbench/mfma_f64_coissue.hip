// Does fp64 MFMA buy issue bandwidth next to fp64 VALU on gfx950?  ("issue the Te / Le
// contractions as v_mfma_f64_16x16x4 - a different port; measure, keep only if faster".)
//
// Streams timed on all CUs, 4 waves per SIMD (the occupancy of k_se_patch_tiled<2>), 8 independent chains:
//   valu        v_fma_f64 only                                   (256 FMA-lanes per instruction ... 64 lanes x 1)
//   mfma        v_mfma_f64_16x16x4_f64 only                      (1024 FMAs per instruction)
//   mixed-wave  every wave alternates 4 v_fma_f64 : 1 v_mfma     (same wave feeds both pipes)
//   split-wave  two waves of a SIMD run VALU, two run MFMA       (different waves feed the two pipes)
// and the price of getting lane-per-cell data into / out of the MFMA operand layout:
//   transpose   per 64 cells: A operand (16 x 4 per instruction) gathered from 6 per-lane values with
//               ds_bpermute, D (16 x 16, 4 values per lane) scattered back to the owning lanes - the data
//               movement the Le contraction [64 cells x 6] . [6 x 9] would need around its 8 MFMAs.
// Output: ns per instruction and SIMD; FMA-equivalents per ns and SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_f64_coissue.hip -o build_exp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;

// MODE 0 valu, 1 mfma, 2 mixed in one wave, 3 split by wave parity, 4 transposes only, 5 VALU Le (54 FMA) per block,
// 6 MFMA Le incl. transposes per block
template <int MODE>
__global__ void __launch_bounds__(1024) k_stream(double* out, double seed)
{
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    a[i] = seed + threadIdx.x * 1e-9 + i;
  const double b = seed * 0.999999, c = seed * 1e-7;
  double4_t acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool mfma_wave = (MODE == 1) || (MODE == 3 && (wave & 1));
  const bool valu_wave = (MODE == 0) || (MODE == 3 && !(wave & 1));
  for (int it = 0; it < ITER; ++it)
  {
    if constexpr (MODE <= 3)
    {
      if (MODE == 2 || valu_wave)
      {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          a[u] = __builtin_fma(a[u], b, c);
      }
      if (MODE == 2 || mfma_wave)
      {
        // MODE 2: 2 MFMAs per 8 FMAs; pure MFMA stream: 8 per iteration
        constexpr int NM = (MODE == 2) ? 2 : 8;
#pragma unroll
        for (int u = 0; u < NM; ++u)
          acc[u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u & 7], b, acc[u & 1], 0, 0, 0);
      }
    }
    else if constexpr (MODE == 5)
    {
      // the contraction as the kernel does it: 54 FMAs per lane on 9 accumulators
      double s[9];
#pragma unroll
      for (int h = 0; h < 9; ++h)
        s[h] = 0.0;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc)
#pragma unroll
        for (int h = 0; h < 9; ++h)
          s[h] = __builtin_fma(a[cc], b + h * 1e-3 + cc, s[h]);
#pragma unroll
      for (int h = 0; h < 8; ++h)
        a[h] += s[h] + ((h == 0) ? s[8] : 0.0);
    }
    else
    {
      // MFMA route for 64 cells x 6 inputs -> 9 outputs: 4 row blocks x 2 k-chunks = 8 MFMAs.
      // A operand of (row block rb, chunk kc): lane (i + 16 k) needs input (4 kc + k) of cell (16 rb + i).
      double res[9];
#pragma unroll
      for (int h = 0; h < 9; ++h)
        res[h] = 0.0;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
      {
        double4_t d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
        {
          const int kk = lane >> 4, src = 16 * rb + (lane & 15);
          // the value sits in register (4 kc + kk) of lane src: select among the candidates, then move it
          const int idx = 4 * kc + kk;
          double v = (idx == 0) ? a[0] : (idx == 1) ? a[1] : (idx == 2) ? a[2] : (idx == 3) ? a[3] : (idx == 4) ? a[4] : a[5];
          if (idx > 5)
            v = 0.0;
          const double av = __shfl(v, src, 64);
          if constexpr (MODE == 6)
            d = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, d, 0, 0, 0);
          else
            d[kc] += av;
        }
        // D: lane l holds rows 4 (l / 16) + r, column l % 16; cell (16 rb + row) wants its 9 columns
#pragma unroll
        for (int h = 0; h < 9; ++h)
        {
          const int row = lane & 15; // the row this lane's cell has inside block rb (if it is in the block)
          const int from = 16 * (row >> 2) + h;
          const int r = row & 3;
          const double dv = (r == 0) ? d[0] : (r == 1) ? d[1] : (r == 2) ? d[2] : d[3];
          const double t = __shfl(dv, from, 64);
          if ((lane >> 4) == rb)
            res[h] = t;
        }
      }
#pragma unroll
      for (int h = 0; h < 8; ++h)
        a[h] += res[h] + ((h == 0) ? res[8] : 0.0);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    s += a[i];
  s += acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
  if (s == 1.2345e300)
    out[0] = s;
}

template <int MODE>
static double run()
{
  double* d = nullptr;
  (void)hipMalloc(&d, 8);
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_stream<MODE>), dim3(p.multiProcessorCount), dim3(1024), 0, 0, d, 1.0000001);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_stream<MODE>), dim3(p.multiProcessorCount), dim3(1024), 0, 0, d, 1.0000001);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return ms;
}

int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs; 4 waves per SIMD, %d iterations\n", p.name, p.multiProcessorCount, ITER);
  const double it = ITER;
  // per SIMD: 4 waves
  const double t0 = run<0>(), t1 = run<1>(), t2 = run<2>(), t3 = run<3>();
  printf("valu only : %.3f ms  %.2f ns per v_fma_f64 and SIMD            -> %.1f FMA-lanes/ns/SIMD\n", t0,
         t0 * 1e6 / (it * 8 * 4), 64.0 * it * 8 * 4 / (t0 * 1e6));
  printf("mfma only : %.3f ms  %.2f ns per v_mfma_f64_16x16x4 and SIMD    -> %.1f FMA-lanes/ns/SIMD\n", t1,
         t1 * 1e6 / (it * 8 * 4), 1024.0 * it * 8 * 4 / (t1 * 1e6));
  printf("mixed wave: %.3f ms  (8 FMA + 2 MFMA per wave and iteration)    -> %.1f FMA-lanes/ns/SIMD; the same work "
         "on separate streams would take %.3f ms\n",
         t2, (64.0 * 8 + 1024.0 * 2) * it * 4 / (t2 * 1e6), t0 + t1 * 2.0 / 8.0);
  printf("split wave: %.3f ms  (2 waves x 8 FMA, 2 waves x 8 MFMA)        -> %.1f FMA-lanes/ns/SIMD; serialised it "
         "would take %.3f ms\n",
         t3, (64.0 * 8 * 2 + 1024.0 * 8 * 2) * it / (t3 * 1e6), 0.5 * (t0 + t1));
  const double t4 = run<4>(), t5 = run<5>(), t6 = run<6>();
  printf("Le contraction of 64 cells ([64 x 6] . [6 x 9]) per wave-block, 4 waves per SIMD:\n");
  printf("  VALU (54 FMA per lane)                 : %.2f ns per wave-block and SIMD\n", t5 * 1e6 / (it * 4));
  printf("  operand transposes alone (no MFMA)     : %.2f ns\n", t4 * 1e6 / (it * 4));
  printf("  8 MFMA + operand transposes            : %.2f ns\n", t6 * 1e6 / (it * 4));
  return 0;
}
