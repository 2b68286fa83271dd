// wx_tile.h -- what the tiled dry kernel (wx_dry.h), the row-marching kernels and the host code share: the 64 x 16 tile order,
// fp32 LDS planes, the planar layout of the light texture and the result record of the out-of-line exact advection.
// (Round 1's two fused LDS-tiled kernels for the wet iteration lived here; the single row-marching kernel of wx_wet.h replaced them,
// and the per-pass kernels of wx_kernels.h stay as the independent cross-check -- history in DESIGN.md section 5.)
#pragma once
#include "wx_cells.h"

#include <cstdlib>
#include <algorithm>
// Tuning switches are environment variables ONLY in builds with -DWX_DEBUG (make debug: variants/libwxsim_debug.so; the experiment
// scripts under tools/ load that one through WXSIM_LIB). The shipped library reads no environment variable: it has one launch shape,
// and what a host may choose goes through wx_set_option.
#ifdef WX_DEBUG
inline const char *wx_tune_env(const char *name) { return std::getenv(name); }
#else
inline const char *wx_tune_env(const char *) { return nullptr; }
#endif

namespace wx {

constexpr int TX = 64, TY = 16;

// The light texture of the single-kernel path is stored as three planes: sunlight (x), net heating (y) and the two IR
// fluxes (zw). The boundary stage needs x and y only (8 instead of 16 B/cell); lighting reads x at its four filter taps
// and z / w of one row each, and writes all four channels.
template <typename F, typename F2> struct LightPlanesT {
  F *x, *y;
  F2 *zw;
};
using LightPlanes = LightPlanesT<float, float2>;
using LightPlanesC = LightPlanesT<const float, const float2>;

__device__ __forceinline__ size_t fidx(int x, int y, int X) { return (size_t)y * X + x; }
// wrap for i in [-n, 2n): tile halos of grids at least as large as the halo
__device__ __forceinline__ int wrapfast(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// XCD-aware tile order. The dispatcher places workgroup `id` on XCD id % 8 (observed, MI355X_MICROARCH.md); every XCD has
// its own L2. With the natural order horizontally adjacent tiles land on different XCDs and the 128-byte lines of
// their shared halo columns are fetched once per XCD. Here XCD k walks the column block [k*GX/8, (k+1)*GX/8) row band by
// row band, so that left/right (and, one band later, upper/lower) neighbours hit the same L2. Placement only affects
// speed, never results. MEASURED (16384x2048): FETCH_SIZE of kernel B drops from 1.18x to 1.05x of the algorithmic
// bytes, but the kernel gets 2 % SLOWER (the re-fetched halo lines were served by the Infinity Cache, and the
// column-block order concentrates each XCD on fewer HBM channels), so the natural order stays the default.
#ifndef WX_XCD_TILES
#define WX_XCD_TILES 0
#endif
#ifndef WX_GRID2D
#define WX_GRID2D 1
#endif
__device__ __forceinline__ void tile_of_block(int GX, int &bx, int &by)
{
#if WX_GRID2D
  bx = blockIdx.x;
  by = blockIdx.y;
#else
  const int id = blockIdx.x;
  if (WX_XCD_TILES && (GX & 7) == 0) {
    const int k = id & 7, j = id >> 3, w = GX >> 3;
    by = j / w;
    bx = k * w + (j - by * w);
  } else {
    by = id / GX;
    bx = id - by * GX;
  }
#endif
}
inline dim3 tile_grid(int X, int Y)
{
#if WX_GRID2D
  return dim3((X + 63) / 64, (Y + 15) / 16);
#else
  return dim3(((X + 63) / 64) * ((Y + 15) / 16));
#endif
}
__device__ __forceinline__ int tiles_x(int X) { return (X + 63) / 64; }

// fp32 plane set of a float4 field
template <int H, int W> struct Planes4 {
  float x[H][W], y[H][W], z[H][W], w[H][W];
  __device__ __forceinline__ void put(int r, int c, float4 v)
  {
    x[r][c] = v.x;
    y[r][c] = v.y;
    z[r][c] = v.z;
    w[r][c] = v.w;
  }
  __device__ __forceinline__ float4 get(int r, int c) const { return make_float4(x[r][c], y[r][c], z[r][c], w[r][c]); }
};

#ifndef WX_REACH
#define WX_REACH 1
#endif
namespace fb_ {
// advection is evaluated on x,y in [-1,0] (pressure needs the left and lower neighbour); its 7-point velocity
// stencil reaches 1 further, and so does the back-traced bilinear footprint as long as |v| < 1 cell/iteration
// (REACH = 1; the shaders document velocities as "-1.0 to 1.0", common.glsl:40-41): inputs on [-2,+1].
// Cells with a longer back-trace take the exact out-of-line path. REACH = 2 stages [-3,+2] and covers |v| < 2.
constexpr int REACH = WX_REACH;
constexpr int HL = 1 + REACH, HR = REACH, HD = 1 + REACH, HU = REACH;
constexpr int IW = TX + HL + HR, IH = TY + HD + HU;
constexpr int AW = TX + 1, AH = TY + 1;             // advection results on [-1,0]
constexpr float VMAX = REACH == 1 ? 0.9f : 1.9f;    // back-traces shorter than this stay inside the staged tile
struct SmemOut { // advection output needed by neighbours (aliases the input tiles after a barrier)
  float vx[AH][AW], vy[AH][AW], T[AH][AW];
  char4 w[AH][AW + 1];
};
} // namespace fb_

struct AdvOut {
  float4 b, w;
  char4 wl;
};

// ---- split iterations of a slab (wx_step_overlap) as ONE launch ----
// The iteration before a halo exchange has to finish its EDGE strips first (the neighbours wait for their columns), the iteration after
// it may read its ghost columns only once the exchange has written them. Rounds 2-4 cut such an iteration into two launch groups on two
// streams joined by events (+8...15 % per iteration on the metric's slab: profiles/r04_slab_protocol_cost.txt). Round 5: one launch over
// all strips; the DISPATCH ORDER puts the edge strips first (mode 1) or last (mode 2), and the hand-offs are device-side words:
//   arrive:  every edge wave adds 1 once its outputs are visible device-wide (agent-scope release); a one-wave gate kernel on the comm
//            stream polls the word and lets the pack kernel start while the interior strips still march;
//   epoch:   the comm stream bumps it behind the unpack; edge waves (dispatched last) poll it before their first load, then one
//            agent-scope acquire. A handful of polling waves cannot fill the chip, so the exchange they wait for always finds slots.
#ifndef WX_SPIN_LIMIT
#define WX_SPIN_LIMIT 2000000u // polls of ~2 us each
#endif
struct StripOrder {
  int mode;              // 0: plain launch (strip ranges as given); 1: edge strips first in dispatch order; 2: edge strips last
  int nl, nr0;           // the edge strips: [0, nl) and [nr0, n_strips_all)
  unsigned *arrive;      // NULL: nobody waits for the edge strips (else: word [0] of the handle's sync words; [2] = "a poll gave up")
  const unsigned *epoch; // NULL: the ghost columns are valid already (else: word [1])
  unsigned epoch_want;
  int edge_list;         // wet kernel: edge waves append their exact-path cells to the second list (consumed on the comm stream)
  int prio;              // s_setprio level of the edge waves (0: none): all waves of a slab's launch are resident at once, so the DISPATCH order
                         // alone does not make the edge strips finish first -- the SIMD's issue priority does
  int nofence;           // (timing experiments only: no release fence in front of the arrival -- WRONG results)
};
// position `sloc`-th of the n_part strips of one part (edge / interior) of the strip range [a, b) -> strip; returns false past the end
struct StripPick {
  int strip;
  bool is_edge;
};
// j: workgroup index within the XCD's share, wpb waves per workgroup; [a, b): the strips this XCD works on; n_seg segments each.
// Order: all (segment, group) pairs of the first part, segment-major, then those of the second part.
__device__ __forceinline__ bool strip_order_pick(const StripOrder &o, int a, int b, int n_seg, int wpb, int wave, int j, int &seg, StripPick &out)
{
  const int eL = max(0, min(b, o.nl) - a), r0 = max(a, o.nr0), eR = max(0, b - r0);
  const int ne = eL + eR, ni = (b - a) - ne;
  const int ge = (ne + wpb - 1) / wpb, gi = (ni + wpb - 1) / wpb;
  const bool edges_first = o.mode == 1;
  const int g1 = edges_first ? ge : gi;
  const bool first = j < g1 * n_seg;
  const int jj = first ? j : j - g1 * n_seg;
  const bool is_edge = first == edges_first;
  const int g = is_edge ? ge : gi, n = is_edge ? ne : ni;
  if (jj >= g * n_seg) return false;
  seg = jj / g;
  const int sloc = (jj - seg * g) * wpb + wave;
  if (sloc >= n) return false;
  out.is_edge = is_edge;
  out.strip = is_edge ? (sloc < eL ? a + sloc : r0 + (sloc - eL)) : a + eL + sloc;
  return true;
}
// workgroups per segment-set an XCD needs for the strips [a, b) under an order (host side: grid size)
inline int strip_order_groups(const StripOrder &o, int a, int b, int wpb)
{
  const int eL = std::max(0, std::min(b, o.nl) - a), r0 = std::max(a, o.nr0), eR = std::max(0, b - r0);
  const int ne = eL + eR, ni = (b - a) - ne;
  return (ne + wpb - 1) / wpb + (ni + wpb - 1) / wpb;
}
__device__ __forceinline__ void strip_order_prio(int prio)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio >= 3) __builtin_amdgcn_s_setprio(3);
#endif
}
__device__ __forceinline__ void strip_order_wait(const StripOrder &o)
{
#if defined(__HIP_DEVICE_COMPILE__)
  // ONE relaxed poll loop -> ONE agent-scope acquire -> plain loads (MI355X_MICROARCH.md, inter-workgroup visibility). Every spin is
  // bounded (~ seconds): a hand-off that never comes -- a protocol bug, a dead peer -- ends in an error flag the next blocking call
  // reports, not in a hung GPU
  unsigned spins = 0;
  while ((int)(__hip_atomic_load(o.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - o.epoch_want) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (++spins > WX_SPIN_LIMIT) {
      __hip_atomic_store(const_cast<unsigned *>(o.epoch) + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (2: an edge strip's epoch poll)
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
// every lane of the wave has issued its last store
__device__ __forceinline__ void strip_order_arrive(const StripOrder &o, int lane, bool stored)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if (stored && !o.nofence) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // writes the XCD's dirty lines back: the pack kernel may run on any XCD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the compiler may drop the wait behind buffer_wbl2: restated where it cannot)
  }
  if (lane == 0) __hip_atomic_fetch_add(o.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// ---- slabs exact at any speed: the fastest |vx| of an exchange period ----
// One iteration's dependency cone is 5 + (1 + floor|vx|) columns per side: the advection pass back-traces `fragCoord - vel` for ANY vel
// (advectionShader.frag:85-99, no clamp) and interpolates from the two columns around the foot point. The marching kernels therefore keep
// the largest |vx| they produce (post-boundary velocities in the wet kernel, the velocity pass's output in the dry one: what the
// back-trace uses) -- one v_max per row step and lane, one wave reduction + at most two atomics per wave -- so that the hosts can size
// the next exchange period (cone = 6 + floor(bound), wx_comm.h) and so that a period whose assumed bound was exceeded is REPORTED.
// Only the flow NEAR THE SLAB EDGES matters: the front of invalid ghost columns moves inwards from the edge of the local array, and a
// jet further inside needs many iterations to get there (it enters the watched zone -- three halo widths from either edge -- at least
// 2 * halo / |vx| iterations before it reaches the ghost columns: several exchange periods). A wave whose strip lies outside the zone
// reports nothing (zone_l / zone_r in strips; whole-domain handles watch everything).
struct VxTrack {
  int *max_bits;   // float bits of the largest |vx| since the last roll (only values >= 0.5 are recorded: below that the cone is 6 anyway)
  int *violation;  // float bits of a |vx| that reached `limit` (0: none)
  float limit;     // |vx| the current period's ghost columns allow (cone - 5); <= 0: whole-domain handle, nothing to check
  int zone_l, zone_r; // strips [0, zone_l) and [zone_r, n) are watched (zone_l >= zone_r: all of them)
  float limit_in;  // the strips in between consume no ghost column -- until |vx| spans the two halo widths that separate them from the ghost
                   // columns: a jet that fast (2 * halo - 8 cells / iteration: a state that has blown up) is a violation too (0: not checked)
};
__device__ __forceinline__ void vx_track_commit(const VxTrack &t, float lane_max, int lane, int strip = 0)
{
#if defined(__HIP_DEVICE_COMPILE__)
  float m = lane_max;
  if (t.zone_l < t.zone_r && strip >= t.zone_l && strip < t.zone_r) { // (wave-uniform) an interior strip
    if (!(t.limit_in > 0.0f)) return;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0 && t.violation != nullptr && m >= t.limit_in) atomicMax(t.violation, __float_as_int(m));
    return;
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0 && t.max_bits != nullptr && m >= 0.5f) {
    atomicMax(t.max_bits, __float_as_int(m)); // (bit patterns of positive floats order like ints)
    if (t.limit > 0.0f && m >= t.limit) atomicMax(t.violation, __float_as_int(m));
  }
#endif
}
// the same for kernels that do not track while they run (tiled dry kernel, per-pass kernels) and after uploads: max |vx| of a base texture
// (columns [0, col_l) and [col_r, X) of every row; col_l >= col_r: all columns)
__global__ void k_vx_scan(int X, int Y, int col_l, int col_r, const float4 *__restrict__ base, VxTrack t)
{
  float m = 0.0f, m_in = 0.0f;
  const size_t n = (size_t)X * Y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % (size_t)X);
    const float v = fabsf(base[i].x);
    if (col_l >= col_r || x < col_l || x >= col_r)
      m = fmaxf(m, v);
    else
      m_in = fmaxf(m_in, v);
  }
  t.zone_l = t.zone_r = 0;
  vx_track_commit(t, m, threadIdx.x & 63);
  if (t.limit_in > 0.0f) { // the columns in between: only against the interior limit
    t.zone_l = 0;
    t.zone_r = 1;
    vx_track_commit(t, m_in, threadIdx.x & 63, 0);
  }
}

// comm stream: wait until the edge strips of the launch that carries `want` arrivals (cumulative) are done
__global__ void k_strip_gate(unsigned *__restrict__ arrive, unsigned want)
{
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned spins = 0;
  while ((int)(__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (++spins > WX_SPIN_LIMIT) {
      __hip_atomic_store(arrive + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (1: the gate's arrival poll)
      __hip_atomic_store(arrive + 3, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
// comm stream, behind the unpack: the ghost columns are written
__global__ void k_strip_epoch(unsigned *__restrict__ epoch, unsigned value)
{
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(epoch, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// light texture: interleaved RGBA32F (the reference's layout: per-pass / single-kernel paths, readback, halo buffers of
// those paths) <-> the three planes of the marching kernel
__global__ void k_light_to_planes(size_t n, const float4 *__restrict__ src, LightPlanes dst)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 l = src[i];
    dst.x[i] = l.x;
    dst.y[i] = l.y;
    dst.zw[i] = make_float2(l.z, l.w);
  }
}
__global__ void k_light_from_planes(size_t n, LightPlanesC src, float4 *__restrict__ dst)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float2 zw = src.zw[i];
    dst[i] = make_float4(src.x[i], src.y[i], zw.x, zw.y);
  }
}

} // namespace wx
