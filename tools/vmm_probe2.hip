// vmm_probe2: is the "level" of the placement lottery a property of PHYSICAL CHUNKS that a library can test one batch at a time?
// (follow-up of tools/vmm_probe.hip, whose header explains the question)
//
// The marching wet kernel streams 13 planes at once, and WHERE those planes lie in physical memory is worth +-8 % (LABNOTES section 4:
// the level follows the position in the 288 GB; no allocation sequence steers it, so wx_tune_placement re-rolls whole allocations --
// up to 3x the state's memory while it searches). hipMemCreate / hipMemMap let a library keep ONE pool of physical chunks and choose
// which chunk backs which piece of which plane. This probe runs the wet kernel's stream skeleton (the k_mix of tools/ubench_hbm.hip:
// 7 input + 6 output planes, 108 B/cell, no arithmetic) over
//   (a) planes from hipMalloc, one allocation each, re-rolled TRIES times (the lottery as the library sees it today),
//   (b) planes mapped from ONE pool of CHUNK-sized physical handles (1.25x the state), with the chunk -> plane assignment permuted TRIES
//       times: identity, plane order reversed, chunk-interleaved between planes, random permutations -- the SAME physical memory every time.
// If (b) spreads like (a), a library can search assignments inside one pool before any data is uploaded (a few ms per probe, no copies,
// 1.25x memory); if (b) does not move, the level is a property of the physical range and only a different range helps.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/vmm_probe tools/vmm_probe.hip && /tmp/vmm_probe [X Y chunk_MiB tries]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHK(x)                                                                       \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s (line %d): %s\n", #x, __LINE__, hipGetErrorString(e_));    \
      return 1;                                                                      \
    }                                                                                \
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
struct Planes { // byte sizes per cell: in 16 16 4 4 4 4 8, out 16 16 4 4 4 8
  char *p[13];
};
static const int kBytes[13] = {16, 16, 4, 4, 4, 4, 8, 16, 16, 4, 4, 4, 8};
__global__ __launch_bounds__(256) void k_mix(Shape s, Planes q)
{
  int strip, y_lo, y_hi;
  if (!my_rows(s, strip, y_lo, y_hi)) return;
  const int col = strip * 64 + (threadIdx.x & 63);
  const float4 *ib = (const float4 *)q.p[0], *iq = (const float4 *)q.p[1];
  const int *iw = (const int *)q.p[2];
  const float *ilx = (const float *)q.p[3], *ily = (const float *)q.p[4], *il0 = (const float *)q.p[5];
  const float2 *izw = (const float2 *)q.p[6];
  float4 *ob = (float4 *)q.p[7], *oq = (float4 *)q.p[8];
  int *ow = (int *)q.p[9];
  float *olx = (float *)q.p[10], *oly = (float *)q.p[11];
  float2 *ozw = (float2 *)q.p[12];
  size_t i = (size_t)y_lo * s.X + col;
  float4 pb = ib[i], pq = iq[i];
  int pw = iw[i];
  float plx = ilx[i], ply = ily[i], pl0 = il0[i];
  float2 pzw = izw[i];
  for (int y = y_lo; y < y_hi; y++) {
    const float4 b = pb, w4 = pq;
    const int w = pw;
    const float lx = plx, ly = ply, l0 = pl0;
    const float2 zw = pzw;
    const size_t o = (size_t)y * s.X + col;
    if (y + 1 < y_hi) {
      i = o + s.X;
      pb = ib[i];
      pw = iw[i];
      pq = iq[i];
      plx = ilx[i];
      pzw = izw[i];
      ply = ily[i];
      pl0 = il0[i];
    }
    ob[o] = b;
    oq[o] = w4;
    ow[o] = w;
    olx[o] = lx + l0;
    oly[o] = ly;
    ozw[o] = zw;
  }
}

static int time_mix(const Shape &s, const Planes &q, int reps, float *ms_out)
{
  const dim3 grid(8 * ((s.n_strips + 3) / 4) * s.segs_per_band), block(256);
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  for (int i = 0; i < 12; i++) hipLaunchKernelGGL(k_mix, grid, block, 0, 0, s, q); // (clocks)
  CHK(hipDeviceSynchronize());
  std::vector<float> ms(reps);
  for (int i = 0; i < reps; i++) {
    CHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_mix, grid, block, 0, 0, s, q);
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms[i], a, b));
  }
  CHK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  *ms_out = ms[reps / 2];
  hipEventDestroy(a);
  hipEventDestroy(b);
  return 0;
}

static int map_planes(const Shape &s, Planes &q, const std::vector<hipMemGenericAllocationHandle_t> &chunks, size_t chunk, int dev, bool first_time, std::vector<size_t> &need)
{
  const size_t n = (size_t)s.X * s.Y;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t c = 0;
  for (int k = 0; k < 13; k++) {
    need[k] = (n * kBytes[k] + chunk - 1) / chunk;
    if (first_time) {
      void *va = nullptr;
      CHK(hipMemAddressReserve(&va, need[k] * chunk, 0, nullptr, 0));
      q.p[k] = (char *)va;
    }
    for (size_t i = 0; i < need[k]; i++, c++) {
      if (c >= chunks.size()) {
        fprintf(stderr, "not enough chunks\n");
        return 1;
      }
      CHK(hipMemMap(q.p[k] + i * chunk, chunk, 0, chunks[c], 0));
    }
    CHK(hipMemSetAccess(q.p[k], need[k] * chunk, &acc, 1));
  }
  return 0;
}
static int unmap_planes(Planes &q, size_t chunk, const std::vector<size_t> &need)
{
  for (int k = 0; k < 13; k++) CHK(hipMemUnmap(q.p[k], need[k] * chunk));
  return 0;
}

int main(int argc, char **argv)
{
  const int n_batches = argc > 1 ? atoi(argv[1]) : 32;
  const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
  int dev = 0;
  CHK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  // the probe grid: 4096 x 2048 cells x 108 B = 0.91 GB = one batch of 16 chunks (15 needed); the full grid: 16384 x 2048 = 3.62 GB = 4 batches
  Shape sp;
  sp.X = 4096;
  sp.Y = 2048;
  sp.seg = 52;
  sp.n_strips = sp.X / 64;
  sp.segs_per_band = (sp.Y / 8 + sp.seg - 1) / sp.seg;
  Shape sf = sp;
  sf.X = 16384;
  sf.n_strips = sf.X / 64;
  const int per_batch = 17; // (every plane is rounded up to whole chunks: 4 x 2 + 7 x 1 + 2 x 1)
  printf("# %d batches of %d x %zu MiB physical chunks; every batch is probed ALONE with the wet stream skeleton on a %d x %d grid (0.91 GB), then full-size\n"
         "# plane sets (16384 x 2048, 3.62 GB = 4 batches) are assembled from the fastest and from the slowest batches\n", n_batches, per_batch, chunk >> 20, sp.X, sp.Y);
  std::vector<std::vector<hipMemGenericAllocationHandle_t>> batch(n_batches);
  std::vector<float> level(n_batches, 0.f);
  Planes qp, qf;
  std::vector<size_t> need_p(13), need_f(13);
  for (int b = 0; b < n_batches; b++) {
    batch[b].resize(per_batch);
    for (int i = 0; i < per_batch; i++) CHK(hipMemCreate(&batch[b][i], chunk, &prop, 0));
    if (map_planes(sp, qp, batch[b], chunk, dev, b == 0, need_p)) return 1;
    for (int k = 0; k < 13; k++) CHK(hipMemset(qp.p[k], 0, (size_t)sp.X * sp.Y * kBytes[k]));
    if (time_mix(sp, qp, 15, &level[b])) return 1;
    if (unmap_planes(qp, chunk, need_p)) return 1;
    printf("  batch %2d: %.4f ms  %.0f GB/s\n", b, level[b], (double)sp.X * sp.Y * 108 / (level[b] * 1e-3) / 1e9);
    fflush(stdout);
  }
  std::vector<int> order(n_batches);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return level[a] < level[b]; });
  auto assemble = [&](const char *what, const int *which) -> int {
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    for (int i = 0; i < 4; i++) chunks.insert(chunks.end(), batch[which[i]].begin(), batch[which[i]].end());
    static bool first = true;
    if (map_planes(sf, qf, chunks, chunk, dev, first, need_f)) return 1;
    first = false;
    float ms = 0;
    if (time_mix(sf, qf, 15, &ms)) return 1;
    printf("  full grid from %s (batches %d %d %d %d: %.4f %.4f %.4f %.4f ms alone): %.4f ms  %.0f GB/s\n", what, which[0], which[1], which[2], which[3], level[which[0]], level[which[1]],
           level[which[2]], level[which[3]], ms, (double)sf.X * sf.Y * 108 / (ms * 1e-3) / 1e9);
    fflush(stdout);
    return unmap_planes(qf, chunk, need_f);
  };
  const int fast[4] = {order[0], order[1], order[2], order[3]};
  const int slow[4] = {order[n_batches - 1], order[n_batches - 2], order[n_batches - 3], order[n_batches - 4]};
  const int mixed[4] = {order[0], order[n_batches - 1], order[1], order[n_batches - 2]};
  for (int rep = 0; rep < 2; rep++) {
    if (assemble("the four FASTEST batches", fast)) return 1;
    if (assemble("the four SLOWEST batches", slow)) return 1;
    if (assemble("two fastest + two slowest", mixed)) return 1;
  }
  return 0;
}
