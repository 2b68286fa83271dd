// Micro-benchmarks behind DESIGN.md's VALU accounting (gfx950): issue rate of plain and packed fp32 VALU instructions,
// DPP whole-wave shifts (semantics + rate), LDS dword reads. Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_valu.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096, UNROLL = 16;

__global__ __launch_bounds__(256) void k_mul(float *out, float a)
{
  float v[UNROLL];
  for (int i = 0; i < UNROLL; i++) v[i] = threadIdx.x + i;
  for (int it = 0; it < ITER; it++)
#pragma unroll
    for (int i = 0; i < UNROLL; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
  float s = 0;
  for (int i = 0; i < UNROLL; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pkmul(float *out, float a)
{
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v v[UNROLL], av = {a, a};
  for (int i = 0; i < UNROLL; i++) v[i] = float2v{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < ITER; it++)
#pragma unroll
    for (int i = 0; i < UNROLL; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
  float s = 0;
  for (int i = 0; i < UNROLL; i++) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_dpp(float *out)
{
  int v[UNROLL];
  for (int i = 0; i < UNROLL; i++) v[i] = threadIdx.x + i;
  for (int it = 0; it < ITER; it++)
#pragma unroll
    for (int i = 0; i < UNROLL; i++) v[i] = __builtin_amdgcn_update_dpp(v[i], v[i], 0x138, 0xf, 0xf, false);
  int s = 0;
  for (int i = 0; i < UNROLL; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_lds(float *out)
{
  __shared__ float sm[256 * 5];
  for (int i = threadIdx.x; i < 256 * 5; i += 256) sm[i] = i;
  __syncthreads();
  float s = 0;
  int idx = threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s += sm[idx + (i & 3) * 256 + (i >> 2)];
    asm volatile("" : "+v"(idx));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dpp_sem(int *out)
{
  const int x = threadIdx.x * 10;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);        // wave_shr:1, old = -1
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);    // old = own value
  out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, true);   // bound_ctrl
}

template <class F> static double time_ms(F &&launch)
{
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; i++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main()
{
  int ncu = 0;
  CHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  int clk = 0;
  CHK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  printf("CUs %d, clock %d kHz\n", ncu, clk);
  float *out;
  const int blocks = ncu * 8, threads = 256; // 8 blocks x 4 waves per CU = 8 waves / SIMD
  CHK(hipMalloc(&out, (size_t)blocks * threads * 4));
  int *sem;
  CHK(hipMalloc(&sem, 256 * 4));
  hipLaunchKernelGGL(k_dpp_sem, dim3(1), dim3(64), 0, 0, sem);
  std::vector<int> h(256);
  CHK(hipMemcpy(h.data(), sem, 256 * 4, hipMemcpyDeviceToHost));
  printf("wave_shr:1 (old=-1): lane0 %d lane1 %d lane16 %d lane32 %d lane63 %d\n", h[0], h[1], h[16], h[32], h[63]);
  printf("wave_shl:1 (old=-1): lane0 %d lane15 %d lane31 %d lane62 %d lane63 %d\n", h[64], h[64 + 15], h[64 + 31], h[64 + 62], h[64 + 63]);
  printf("wave_shr:1 (old=own): lane0 %d lane1 %d\n", h[128], h[129]);
  printf("wave_shr:1 bound_ctrl: lane0 %d lane1 %d\n", h[192], h[193]);
  const double n_inst = (double)blocks * (threads / 64) * ITER * UNROLL; // wave-instructions
  const double simds = ncu * 4.0;
  struct { const char *name; double ms; } r[4];
  r[0] = {"v_mul_f32", time_ms([&] { hipLaunchKernelGGL(k_mul, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f); })};
  r[1] = {"v_pk_mul_f32", time_ms([&] { hipLaunchKernelGGL(k_pkmul, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f); })};
  r[2] = {"v_mov_b32_dpp wave_shr:1", time_ms([&] { hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(threads), 0, 0, out); })};
  r[3] = {"ds_read_b32", time_ms([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), 0, 0, out); })};
  for (auto &x : r)
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at %.2f GHz\n", x.name, x.ms, x.ms * 1e-3 * clk * 1e3 * simds / n_inst,
           clk * 1e-6);
  return 0;
}
