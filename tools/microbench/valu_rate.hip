// VALU issue rates on gfx950 measured, not assumed: cycles per wave64 instruction and SIMD for fp64 FMA / MUL /
// ADD, 32-bit DPP moves and v_cndmask, at 1, 2 and 4 waves per SIMD, with 8 independent dependency chains per
// lane (issue-bound) and with one chain (latency-bound).  Calibrates the "VALU floor" of DESIGN.md section 7.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rate.hip -o build_exp/valu_rate && build_exp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 4096;

template <int KIND, int CHAINS>
__global__ void __launch_bounds__(1024) k_rate(double* out, double seed, long long* cyc)
{
  const long long t0 = clock64(); // s_memtime: shader clock
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    a[i] = seed + threadIdx.x * 1e-9 + i;
  const double b = seed * 0.999999, c = seed * 1e-7;
  int m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    m[i] = threadIdx.x + i;
  for (int it = 0; it < ITER; ++it)
  {
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      const int i = u % CHAINS;
      if constexpr (KIND == 0)
        a[i] = __builtin_fma(a[i], b, c);
      else if constexpr (KIND == 1)
        a[i] = a[i] * b;
      else if constexpr (KIND == 2)
        a[i] = a[i] + c;
      else if constexpr (KIND == 3)
        m[i] = __builtin_amdgcn_update_dpp(0, m[i], 0x111, 0xf, 0xf, true); // row_shr:1
      else if constexpr (KIND == 4)
      {
        int t = m[i];
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t) : "v"(it));
        m[i] = t;
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    s += a[i] + m[i];
  if (s == 1.2345e300)
    out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    cyc[0] = clock64() - t0;
}

template <int KIND, int CHAINS>
static void run(const char* name, int waves_per_simd, double ghz)
{
  double* d = nullptr;
  (void)hipMalloc(&d, 8);
  long long* dc = nullptr;
  (void)hipMalloc(&dc, 8);
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int threads = 64 * 4 * waves_per_simd; // one workgroup per CU
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<KIND, CHAINS>), dim3(cus), dim3(threads), 0, 0, d, 1.0000001, dc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_rate<KIND, CHAINS>), dim3(cus), dim3(threads), 0, 0, d, 1.0000001, dc);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long cyc = 0;
  (void)hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  const double instr_per_simd = (double)ITER * 8 * waves_per_simd;
  // clock64() = s_memtime ticks; on gfx9 it counts at a fixed 100 MHz reference, so the shader clock follows from
  // the known 4-cycle issue of independent fp64 FMAs, not from this counter
  printf("%-10s chains %d waves/SIMD %d: %.3f ms = %.3f ns per wave64 instruction and SIMD (%lld s_memtime ticks)\n", name,
         CHAINS, waves_per_simd, ms, ms * 1e6 / instr_per_simd, cyc);
  (void)ghz;
  (void)hipFree(dc);
  (void)hipFree(d);
}

int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  printf("%s, %d CUs, clockRate %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
  for (int w : {1, 2, 4})
  {
    run<0, 8>("fma_f64", w, ghz);
    run<0, 1>("fma_f64", w, ghz);
    run<1, 8>("mul_f64", w, ghz);
    run<2, 8>("add_f64", w, ghz);
    run<3, 8>("mov_dpp", w, ghz);
    run<3, 1>("mov_dpp", w, ghz);
    run<4, 8>("cndmask", w, ghz);
  }
  return 0;
}
