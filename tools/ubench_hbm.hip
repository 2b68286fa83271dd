// Practical HBM ceilings of an MI355X with the row-marching kernels' OWN access pattern (gfx950): what bench.py prints next to the
// runtime's device copy and what DESIGN.md's "fraction of the ceiling" figures refer to.
//
// Pattern: one wavefront owns a 64-column strip (64 lanes x 16 B = 1 KiB per row) and walks up a segment of rows, issuing the row
// r + 1 loads before it consumes row r (software prefetch, as the kernels do); the rows are cut into eight bands, one per XCD
// (workgroup id % 8 = XCD, MI355X_MICROARCH.md), each band into segments of `seg` rows; four waves per workgroup on neighbouring strips.
//   read   : float4 loads only (one dword per wave stored at the end so that the loads are kept)
//   write  : float4 stores only
//   copy   : load row r, store row r (one stream in, one out)
//   mix    : the wet iteration's stream count and widths with no arithmetic: loads 16 + 4 + 16 + 4 + 8 + 4 + 4 B, stores 16 + 4 + 16 +
//            4 + 4 + 8 B per cell = 108 B/cell over seven input and six output planes (the "skeleton" of k_march_wet)
// Build + run: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm tools/ubench_hbm.hip && tools/ubench_hbm [X Y seg reps]
// Prints ONE JSON line: GB/s of bytes moved (read + written), median of `reps` launches timed with HIP events.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                               \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                              \
    }                                                                        \
  } while (0)

struct Shape {
  int X, Y, seg, n_strips, segs_per_band;
};

__device__ __forceinline__ bool my_rows(const Shape &s, int &strip, int &y_lo, int &y_hi)
{
  const int k = blockIdx.x & 7, j = blockIdx.x >> 3, wave = threadIdx.x >> 6;
  const int groups = (s.n_strips + 3) / 4;
  const int sg = j / groups;
  strip = (j - sg * groups) * 4 + wave;
  if (sg >= s.segs_per_band || strip >= s.n_strips) return false;
  const int band_lo = (int)(((long long)k * s.Y) >> 3), band_hi = (int)(((long long)(k + 1) * s.Y) >> 3);
  y_lo = band_lo + sg * s.seg;
  y_hi = min(y_lo + s.seg, band_hi);
  return y_lo < y_hi;
}

__global__ __launch_bounds__(256) void k_read(Shape s, const float4 *__restrict__ a, float *__restrict__ sink)
{
  int strip, y_lo, y_hi;
  if (!my_rows(s, strip, y_lo, y_hi)) return;
  const int col = strip * 64 + (threadIdx.x & 63);
  float4 pf = a[(size_t)y_lo * s.X + col], acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int y = y_lo; y < y_hi; y++) {
    const float4 cur = pf;
    if (y + 1 < y_hi) pf = a[(size_t)(y + 1) * s.X + col];
    acc.x += cur.x;
    acc.y += cur.y;
    acc.z += cur.z;
    acc.w += cur.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x; // (never true: keeps the loads alive)
}

// The same read stream through LDS-DMA (global_load_lds_dwordx4: the row lands in LDS without passing through registers), DEPTH rows in
// flight per wave in a wave-private ring -- the "more bytes in flight without registers" lever of MI355X_MICROARCH.md for a kernel whose
// occupancy is bound by registers. Reported next to the register-prefetch stream above (read_lds1 / read_lds2 / read_lds3).
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_lds(Shape s, const float4 *__restrict__ a, float *__restrict__ sink)
{
  __shared__ float4 ring[4][4][64]; // [wave][slot][lane]
  int strip, y_lo, y_hi;
  if (!my_rows(s, strip, y_lo, y_hi)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = strip * 64 + lane;
  typedef const __attribute__((address_space(1))) void *GP;
  typedef __attribute__((address_space(3))) void *LP;
  auto issue = [&](int y) { __builtin_amdgcn_global_load_lds((GP)(a + (size_t)y * s.X + col), (LP)&ring[wave][(y - y_lo) & 3][0], 16, 0, 0); };
  for (int k = 0; k < DEPTH; k++)
    if (y_lo + k < y_hi) issue(y_lo + k);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int y = y_lo; y < y_hi; y++) {
    if (y + DEPTH < y_hi) issue(y + DEPTH);
    // rows still allowed in flight behind row y: those issued after it
    const int behind = min(DEPTH, y_hi - 1 - y);
    if (behind >= 3) __builtin_amdgcn_s_waitcnt(0x0F73);
    else if (behind == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
    else if (behind == 1) __builtin_amdgcn_s_waitcnt(0x0F71);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f cur = *(volatile v4f *)&ring[wave][(y - y_lo) & 3][lane];
    acc.x += cur.x;
    acc.y += cur.y;
    acc.z += cur.z;
    acc.w += cur.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void k_write(Shape s, float4 *__restrict__ b)
{
  int strip, y_lo, y_hi;
  if (!my_rows(s, strip, y_lo, y_hi)) return;
  const int col = strip * 64 + (threadIdx.x & 63);
  const float4 v = make_float4((float)col, 1.f, 2.f, 3.f);
  for (int y = y_lo; y < y_hi; y++) b[(size_t)y * s.X + col] = v;
}

__global__ __launch_bounds__(256) void k_copy(Shape s, const float4 *__restrict__ a, float4 *__restrict__ b)
{
  int strip, y_lo, y_hi;
  if (!my_rows(s, strip, y_lo, y_hi)) return;
  const int col = strip * 64 + (threadIdx.x & 63);
  float4 pf = a[(size_t)y_lo * s.X + col];
  for (int y = y_lo; y < y_hi; y++) {
    const float4 cur = pf;
    if (y + 1 < y_hi) pf = a[(size_t)(y + 1) * s.X + col];
    b[(size_t)y * s.X + col] = cur;
  }
}

struct MixIn {
  const float4 *base, *water;
  const int *wall;
  const float *lx, *l0y, *l0x;
  const float2 *lzw;
};
struct MixOut {
  float4 *base, *water;
  int *wall;
  float *lx, *ly;
  float2 *lzw;
};
__global__ __launch_bounds__(256) void k_mix(Shape s, MixIn in, MixOut out)
{
  int strip, y_lo, y_hi;
  if (!my_rows(s, strip, y_lo, y_hi)) return;
  const int col = strip * 64 + (threadIdx.x & 63);
  size_t i = (size_t)y_lo * s.X + col;
  float4 pb = in.base[i], pq = in.water[i];
  int pw = in.wall[i];
  float plx = in.lx[i], ply = in.l0y[i], pl0 = in.l0x[i];
  float2 pzw = in.lzw[i];
  for (int y = y_lo; y < y_hi; y++) {
    const float4 b = pb, q = pq;
    const int w = pw;
    const float lx = plx, ly = ply, l0 = pl0;
    const float2 zw = pzw;
    const size_t o = (size_t)y * s.X + col;
    if (y + 1 < y_hi) {
      i = o + s.X;
      pb = in.base[i];
      pw = in.wall[i];
      pq = in.water[i];
      plx = in.lx[i];
      pzw = in.lzw[i];
      ply = in.l0y[i];
      pl0 = in.l0x[i];
    }
    out.base[o] = b;
    out.water[o] = q;
    out.wall[o] = w;
    out.lx[o] = lx + l0;
    out.ly[o] = ly;
    out.lzw[o] = zw;
  }
}

template <class F> static int timed(const char *name, double bytes, int reps, F launch, double *gbps)
{
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) launch();
  CHK(hipDeviceSynchronize());
  std::vector<float> ms(reps);
  for (int i = 0; i < reps; i++) {
    CHK(hipEventRecord(a, 0));
    launch();
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms[i], a, b));
  }
  CHK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  *gbps = bytes / (ms[reps / 2] * 1e-3) / 1e9;
  (void)name;
  hipEventDestroy(a);
  hipEventDestroy(b);
  return 0;
}

int main(int argc, char **argv)
{
  Shape s;
  s.X = argc > 1 ? atoi(argv[1]) : 16384;
  s.Y = argc > 2 ? atoi(argv[2]) : 2048;
  s.seg = argc > 3 ? atoi(argv[3]) : 64;
  const int reps = argc > 4 ? atoi(argv[4]) : 21;
  if (s.X % 64 || s.Y % 8 || s.seg < 1) {
    fprintf(stderr, "X must be a multiple of 64, Y of 8\n");
    return 1;
  }
  s.n_strips = s.X / 64;
  s.segs_per_band = (s.Y / 8 + s.seg - 1) / s.seg;
  const size_t n = (size_t)s.X * s.Y;
  float4 *a, *b, *c, *d;
  float *sink;
  CHK(hipMalloc(&a, n * 16));
  CHK(hipMalloc(&b, n * 16));
  CHK(hipMalloc(&c, n * 16));
  CHK(hipMalloc(&d, n * 16));
  CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(a, 0, n * 16));
  CHK(hipMemset(b, 0, n * 16));
  CHK(hipMemset(c, 0, n * 16));
  CHK(hipMemset(d, 0, n * 16));
  // the small planes of the mix: carved out of two more 16-byte-per-cell allocations
  char *e, *f;
  CHK(hipMalloc(&e, n * 24));
  CHK(hipMalloc(&f, n * 20));
  CHK(hipMemset(e, 0, n * 24));
  CHK(hipMemset(f, 0, n * 20));
  const dim3 grid(8 * ((s.n_strips + 3) / 4) * s.segs_per_band), block(256);
  double r = 0, w = 0, cp = 0, mix = 0;
  if (timed("read", (double)n * 16, reps, [&] { hipLaunchKernelGGL(k_read, grid, block, 0, 0, s, a, sink); }, &r)) return 1;
  double rl1 = 0, rl2 = 0, rl3 = 0;
  if (timed("read_lds1", (double)n * 16, reps, [&] { hipLaunchKernelGGL(k_read_lds<1>, grid, block, 0, 0, s, a, sink); }, &rl1)) return 1;
  if (timed("read_lds2", (double)n * 16, reps, [&] { hipLaunchKernelGGL(k_read_lds<2>, grid, block, 0, 0, s, a, sink); }, &rl2)) return 1;
  if (timed("read_lds3", (double)n * 16, reps, [&] { hipLaunchKernelGGL(k_read_lds<3>, grid, block, 0, 0, s, a, sink); }, &rl3)) return 1;
  if (timed("write", (double)n * 16, reps, [&] { hipLaunchKernelGGL(k_write, grid, block, 0, 0, s, b); }, &w)) return 1;
  if (timed("copy", (double)n * 32, reps, [&] { hipLaunchKernelGGL(k_copy, grid, block, 0, 0, s, a, b); }, &cp)) return 1;
  MixIn in{a, c, (const int *)e, (const float *)(e + n * 4), (const float *)(e + n * 8), (const float *)(e + n * 12), (const float2 *)(e + n * 16)};
  MixOut out{b, d, (int *)f, (float *)(f + n * 4), (float *)(f + n * 8), (float2 *)(f + n * 12)};
  if (timed("mix", (double)n * (56 + 52), reps, [&] { hipLaunchKernelGGL(k_mix, grid, block, 0, 0, s, in, out); }, &mix)) return 1;
  printf("{\"X\": %d, \"Y\": %d, \"seg_rows\": %d, \"waves\": %d, \"read_GBps\": %.1f, \"read_lds_dma_GBps\": [%.1f, %.1f, %.1f], \"write_GBps\": %.1f, \"copy_GBps\": %.1f, "
         "\"wet_stream_mix_GBps\": %.1f, \"wet_stream_mix_ms\": %.4f, \"pattern\": \"64-lane x 16 B rows, row-marching waves with a one-row prefetch, 8 XCD row bands\"}\n",
         s.X, s.Y, s.seg, s.n_strips * s.segs_per_band * 8, r, rl1, rl2, rl3, w, cp, mix, (double)n * 108 / mix / 1e6);
  return 0;
}
