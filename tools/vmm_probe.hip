// Can the placement lottery be steered with the HIP virtual-memory API? (round-4 verdict, item 4)
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

int main(int argc, char **argv)
{
  Shape s;
  s.X = argc > 1 ? atoi(argv[1]) : 16384;
  s.Y = argc > 2 ? atoi(argv[2]) : 2048;
  const size_t chunk_mib = argc > 3 ? (size_t)atoi(argv[3]) : 64;
  const int tries = argc > 4 ? atoi(argv[4]) : 10;
  s.seg = 52;
  s.n_strips = s.X / 64;
  s.segs_per_band = (s.Y / 8 + s.seg - 1) / s.seg;
  const size_t n = (size_t)s.X * s.Y;
  const int reps = 15;
  printf("# %d x %d, wet stream skeleton (7 in + 6 out planes, 108 B/cell = %.2f GB per launch); median of %d launches after 12 warm-up launches\n", s.X, s.Y, n * 108 / 1e9, reps);

  // ---- (a) one hipMalloc per plane, re-rolled ----
  printf("## (a) hipMalloc per plane, %d fresh sets (earlier sets stay allocated, as in wx_tune_placement)\n", tries);
  std::vector<void *> keep;
  for (int t = 0; t < tries; t++) {
    Planes q;
    for (int k = 0; k < 13; k++) {
      void *p = nullptr;
      CHK(hipMalloc(&p, n * kBytes[k]));
      CHK(hipMemset(p, 0, n * kBytes[k]));
      q.p[k] = (char *)p;
      keep.push_back(p);
    }
    float ms = 0;
    if (time_mix(s, q, reps, &ms)) return 1;
    printf("  set %2d: %.4f ms  %.0f GB/s\n", t, ms, n * 108 / (ms * 1e-3) / 1e9);
    fflush(stdout);
  }
  for (void *p : keep) hipFree(p);
  keep.clear();
  CHK(hipDeviceSynchronize());

  // ---- (b) ONE pool of physical chunks, the assignment permuted ----
  int dev = 0;
  CHK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t chunk = chunk_mib << 20;
  chunk = (chunk + gran - 1) / gran * gran;
  size_t need[13], off[14], total_chunks = 0;
  off[0] = 0;
  for (int k = 0; k < 13; k++) {
    need[k] = (n * kBytes[k] + chunk - 1) / chunk + 1; // (+1: room for a skew of the plane's start inside its first chunk, section (c))
    off[k + 1] = off[k] + need[k];
    total_chunks += need[k];
  }
  const size_t pool_n = total_chunks + total_chunks / 4; // 1.25x
  printf("## (b) one pool of %zu physical chunks of %zu MiB (granularity %zu KiB; the planes need %zu), chunk -> plane assignment permuted\n", pool_n, chunk >> 20, gran >> 10,
         total_chunks);
  std::vector<hipMemGenericAllocationHandle_t> pool(pool_n);
  for (size_t i = 0; i < pool_n; i++) CHK(hipMemCreate(&pool[i], chunk, &prop, 0));
  Planes q;
  for (int k = 0; k < 13; k++) {
    void *va = nullptr;
    CHK(hipMemAddressReserve(&va, need[k] * chunk, 0, nullptr, 0));
    q.p[k] = (char *)va;
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::mt19937 rng(12345);
  std::vector<size_t> perm(pool_n);
  bool mapped = false;
  auto remap = [&](const std::vector<size_t> &assign) -> int { // assign[j] = pool chunk that backs slot j of the concatenated planes
    if (mapped)
      for (int k = 0; k < 13; k++) CHK(hipMemUnmap(q.p[k], need[k] * chunk));
    for (int k = 0; k < 13; k++) {
      for (size_t c = 0; c < need[k]; c++) CHK(hipMemMap(q.p[k] + c * chunk, chunk, 0, pool[assign[off[k] + c]], 0));
      CHK(hipMemSetAccess(q.p[k], need[k] * chunk, &acc, 1));
    }
    mapped = true;
    return 0;
  };
  for (int t = 0; t < tries + 3; t++) {
    std::iota(perm.begin(), perm.end(), 0);
    const char *what = "identity (chunks in creation order)";
    if (t == 1) {
      std::reverse(perm.begin(), perm.begin() + total_chunks);
      what = "reversed";
    } else if (t == 2) { // slot j of the concatenation takes chunk (j * 13 mod total): neighbouring pieces of a plane far apart in creation order
      for (size_t j = 0; j < total_chunks; j++) perm[j] = (j * 13) % total_chunks;
      bool ok = std::gcd((size_t)13, total_chunks) == 1;
      if (!ok) std::iota(perm.begin(), perm.end(), 0);
      what = ok ? "stride-13 interleave" : "identity again (13 divides the chunk count)";
    } else if (t >= 3) {
      std::shuffle(perm.begin(), perm.end(), rng); // (draws from the whole pool: the spare quarter takes part)
      what = "random permutation of the pool";
    }
    if (remap(perm)) return 1;
    if (t == 0)
      for (int k = 0; k < 13; k++) CHK(hipMemset(q.p[k], 0, n * kBytes[k]));
    float ms = 0;
    if (time_mix(s, q, reps, &ms)) return 1;
    printf("  assignment %2d: %.4f ms  %.0f GB/s   %s\n", t, ms, n * 108 / (ms * 1e-3) / 1e9, what);
    fflush(stdout);
  }
  // ---- (c) the same pool, identity assignment, plane k starting k * skew bytes into its first chunk: every chunk is chunk-aligned in
  //      physical memory, so the address bits below the chunk size are exactly what the skew says ----
  printf("## (c) identity assignment, plane k starts (k * skew) mod %zu MiB into its mapping\n", chunk >> 20);
  std::iota(perm.begin(), perm.end(), 0);
  if (remap(perm)) return 1;
  const size_t skews[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1048576, 4194304, 1048576 + 4096 + 256, 3293184, 2097152 + 65536, 8388608 + 262144 + 1024};
  for (int rep = 0; rep < 2; rep++)
    for (size_t sk : skews) {
      Planes qs;
      for (int k = 0; k < 13; k++) qs.p[k] = q.p[k] + ((size_t)k * sk) % chunk;
      float ms = 0;
      if (time_mix(s, qs, reps, &ms)) return 1;
      printf("  skew %9zu B: %.4f ms  %.0f GB/s\n", sk, ms, n * 108 / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
  for (int k = 0; k < 13; k++) {
    hipMemUnmap(q.p[k], need[k] * chunk);
    hipMemAddressFree(q.p[k], need[k] * chunk);
  }
  for (auto h : pool) hipMemRelease(h);
  return 0;
}
