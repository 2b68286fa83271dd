// wx_march.h -- the dry-air iteration (velocity -> advection -> pressure, BASELINE configs[1]) as a row-MARCHING
// kernel: one wavefront owns a 64-column strip and walks up the rows of its segment.
//
//   lane l <-> column c0 - 2 + l;  step r (new input row r):  load row r+1 (prefetch)  |  velocity(r-1)  |
//   advection(r-2)  |  pressure(r-2) -> store row r-2
//
// * vertical neighbours are the wave's own previous rows: kept in registers (pressure) or in a 4-row LDS ring that
//   is PRIVATE to the wave (advection's data-dependent taps) -- no vertical halo is ever re-loaded or re-computed;
// * horizontal neighbours come from the ring (lane +- 1) or a wavefront shuffle (pressure's left neighbour);
// * of the 64 lanes 60 produce output (velocity needs lane+1, advection lane+-1, pressure lane-1): 6.7 % redundancy
//   instead of 33 % (tile + halo) in wx_dry.h, and no workgroup barriers: the only synchronisation is the wave's own
//   in-order LDS queue;
// * the next row's global loads are issued before the current row's arithmetic (software prefetch).
// HBM traffic: base 16 R + 16 W, wall 4 R + 4 W per cell. Same per-cell arithmetic (wx_cells.h): bit-identical results.
// Only the water-free state (NO_WATER, see wx_dry.h) marches; anything else uses the tiled kernel.
#pragma once
#include "wx_cells.h"
#include "wx_dry.h"
#include "wx_wet.h"
#include <vector> // ld_row / st_row (scalar-base addressing), wave_from_left (DPP shift)

#ifndef WX_MARCH_BANDS
#define WX_MARCH_BANDS 1
#endif
#ifndef WX_MARCH_BAND_SEG
#define WX_MARCH_BAND_SEG 24 // (32768x4096, interleaved: 24-row band segments + tail 0.941-0.943 ms, 32-row 0.953, equal 32-row segments 0.954)
#endif
namespace wx {

#ifndef WX_MARCH_MAXSEG
#define WX_MARCH_MAXSEG 32 // upper bound of the rows one wave marches (3 warm-up rows per segment are redundant work; measured at 32768x4096: 32 rows 143.7, 64 rows 140.6, 128 rows 135.7 Gcell-steps/s)
#endif
#ifndef WX_MARCH_UNI_MEM
#define WX_MARCH_UNI_MEM 1 // measured: SGPR spills 12 -> 0, 0.31 -> 0.29 ms at 16384x2048
#endif
#ifndef WX_MARCH_XCD
#define WX_MARCH_XCD 1
#endif
#ifndef WX_MARCH_MINWAVES
#define WX_MARCH_MINWAVES 8
#endif
constexpr int MOUT = 60;           // output columns per wave (lanes 2..61)

// Ring of the last four rows. Slot 4 mirrors slot 0, so the two rows of a bilinear footprint (slots s, s+1 with
// s = row & 3) are always adjacent in memory: a footprint costs one address computation, its taps are constant offsets
// from it. The two edge lanes (never output lanes) READ as if they were lanes 1 / 62, so lane +- 1 needs no clamp.
// 5 x 64 x 20 B = 6400 B = five 1280-byte LDS granules: 25 waves per CU fit, the register budget allows 24.
constexpr int MRW = 64;
struct MarchRing {
  float vx[5][MRW], vy[5][MRW], P[5][MRW], T[5][MRW];
  char4 w[5][MRW];
  template <typename A, typename V> __device__ __forceinline__ static void put(A &plane, int slot, int lane, V v)
  {
    plane[slot][lane] = v;
    if (slot == 0) plane[4][lane] = v;
  }
};

struct MDryAcc {
  const MarchRing &rg;
  int lane1, yc; // lane1 = the lane this one reads as (edge lanes: 1 / 62); yc: unwrapped row counter of the own cell (ring slot = row & 3)
  __device__ __forceinline__ float4 base(int dx, int dy) const
  {
    const int s = (yc + dy) & 3, l = lane1 + dx;
    return make_float4(rg.vx[s][l], rg.vy[s][l], rg.P[s][l], rg.T[s][l]);
  }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return rg.w[(yc + dy) & 3][lane1 + dx]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return base(dx, dy); }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return wall(dx, dy); }
  __device__ __forceinline__ float4 water_off(int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
};

// bilinear footprint in the ring: one address, four constant-offset taps (the upper row of slot 3 is the mirror row 4)
struct MDryFp {
  const MarchRing &rg;
  int s, l;
  __device__ __forceinline__ float4 base(int i, int j) const { return make_float4(rg.vx[s + j][l + i], rg.vy[s + j][l + i], rg.P[s + j][l + i], rg.T[s + j][l + i]); }
  __device__ __forceinline__ char4 wall(int i, int j) const { return rg.w[s + j][l + i]; }
  __device__ __forceinline__ float4 water(int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
__device__ __forceinline__ MDryFp make_fp(const MDryAcc &a, int dx0, int dy0) { return MDryFp{a.rg, (a.yc + dy0) & 3, a.lane1 + dx0}; }

// WRITE_WALL = false: the host has established that advection cannot change the wall texture in this launch (no brush,
// no airplane crash, no negative vegetation left to clamp -- the only three ways, advectionShader.frag:189-227, 229-457),
// so the pass-through store is dropped and the wall buffers are not swapped: 36 B/cell, SURVEY's A_dry.
#ifndef WX_MARCH_FENCE
#define WX_MARCH_FENCE 0 // 1: wavefront-scope fence instead of __syncthreads() between the stages (the workgroup is one wave)
#endif
__device__ __forceinline__ void march_fence()
{
#if WX_MARCH_FENCE
  wave_fence();
#else
  __syncthreads();
#endif
}
#ifndef WX_MARCH_AIR
#define WX_MARCH_AIR 1 // wave-uniform free-air instantiation of the advection stage (rows without wall cells)
#endif
#ifndef WX_MARCH_UNROLL
#define WX_MARCH_UNROLL 1 // row steps per loop iteration. 2: 1.08 instead of 0.95 ms at 32768x4096 (more registers, fewer waves per SIMD), unlike k_march_wet
#endif
// QUIET: no brush input, no airplane event in this iteration (see advection_cell)
template <bool WRITE_DISP, bool WRITE_WALL, bool QUIET>
__global__ __launch_bounds__(64, WX_MARCH_MINWAVES) void k_march_dry(Geo g, Uni u_arg, const FullCtx *__restrict__ ctx, DryIn in, DryOut out, int n_strips, int seg_rows,
                                                                     int n_full, int n_half, int band_h, int n_seg, int strip_lo, int split_at, int strip_lo2, StripOrder order, VxTrack vx
#ifdef WX_MARCH_TIMING
                                                                     , unsigned long long *cycles
#endif
)
{
#ifdef WX_MARCH_TIMING
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  __shared__ MarchRing rg;
#if WX_MARCH_UNI_MEM
  // uniforms and the per-row profiles through the constant address space: scalar loads the compiler may re-issue anywhere
  // (through a generic pointer every load behind the first store of the kernel turns into a vector load, and waiting for it
  // means waiting for the row prefetch issued just before). The dry passes do not use iterNum.
  CUni &u = as_constant(ctx->u);
#else
  const Uni &u = u_arg;
#endif
  const CFloatP initial_T = as_constant(ctx->initial_T), snd_T = as_constant(ctx->snd_T), snd_W = as_constant(ctx->snd_W), snd_Vel = as_constant(ctx->snd_Vel);
  const int X = g.X, Y = g.Y;
  const int lane = threadIdx.x;
#if WX_MARCH_XCD
  // XCD-aware placement: workgroup id lands on XCD id % 8 (MI355X_MICROARCH.md). The (segment, strip) items are numbered
  // segment-major and XCD k takes the contiguous range [k*T/8, (k+1)*T/8): neighbouring strips run on the same XCD, so the
  // 128-byte lines they share (2 halo columns each side; 60-column strips are not line aligned) are fetched into ONE L2
  // instead of two. Measured FETCH_SIZE: 1.22x -> 1.00x of the bytes the strips need; 32768x4096: 1.39 -> 1.11 ms.
  const int total = n_strips * n_seg, k = blockIdx.x & 7, j = blockIdx.x >> 3;
  int seg, strip;
  bool is_edge = false; // (a split iteration's edge strip: waits for the ghost columns / reports when it is done)
  if (order.mode == 0 || order.mode == 3) {
    const int first = (int)(((long long)k * total) >> 3), count = (int)(((long long)(k + 1) * total) >> 3) - first;
    if (j >= count) return;
    const int item = first + j;
    seg = item / n_strips;
    const int sidx = item - seg * n_strips;
    strip = sidx < split_at ? strip_lo + sidx : strip_lo2 + (sidx - split_at); // (two strip ranges in one launch: the edges of a slab)
    if (order.mode == 3) is_edge = strip < order.nl || strip >= order.nr0; // (a shape without row bands: the hand-offs without the dispatch order)
  } else {
    // One launch over ALL strips of a slab (row bands: n_seg is a multiple of 8 and XCD k owns the segments [k * n_seg / 8, (k + 1) * n_seg / 8)),
    // the edge strips of all its segments first (or last) in the XCD's dispatch order (StripOrder, wx_tile.h)
    const int spx = n_seg >> 3;
    StripPick pk;
    if (!strip_order_pick(order, 0, n_strips, spx, 1, 0, j, seg, pk)) return;
    seg += k * spx;
    strip = pk.strip;
    is_edge = pk.is_edge;
  }
#else
  const int sidx = blockIdx.x % n_strips, seg = blockIdx.x / n_strips;
  const int strip = sidx < split_at ? strip_lo + sidx : strip_lo2 + (sidx - split_at);
#endif
  const int c_out = strip * MOUT + lane - 2;         // output column of this lane (may be >= X in the last strip)
  const int col = wrapmod(c_out, X);                 // column this lane loads / computes
  const bool lane_out = lane >= 2 && lane <= 61 && c_out < X;
  // band_h > 0: the rows are cut into eight bands (one per XCD with the contiguous item ranges above), each band into n_full segments
  // of seg_rows rows followed by n_half of seg_rows / 2 and n_half of seg_rows / 4 -- short segments at the END of every XCD's work
  // shorten the drain phase of the launch (cf. wet_launch_shape)
  int y_lo, y_hi;
  if (band_h > 0) {
    const int nb = n_full + 2 * n_half, bnd = seg / nb, sl = seg - bnd * nb;
    const int t1 = sl - n_full, t2 = t1 - n_half, h2 = seg_rows >> 1, h4 = seg_rows >> 2, y0 = bnd * band_h;
    y_lo = y0 + (t1 < 0 ? sl * seg_rows : (t2 < 0 ? n_full * seg_rows + t1 * h2 : n_full * seg_rows + n_half * h2 + t2 * h4));
    y_hi = min(y_lo + (t1 < 0 ? seg_rows : (t2 < 0 ? h2 : h4)), y0 + band_h);
    if (y_lo >= y_hi) {
      if (is_edge && order.arrive != nullptr) strip_order_arrive(order, lane, false); // (an empty segment still counts as an edge item that is done)
      return;
    }
  } else {
    y_lo = seg * seg_rows;
    y_hi = min(y_lo + seg_rows, Y);
  }
  if (is_edge && order.epoch != nullptr) strip_order_wait(order); // the ghost columns this strip reads are being written by the exchange
  if (is_edge) strip_order_prio(order.prio);
  const unsigned lo4 = (unsigned)col * 4u, lo16 = (unsigned)col * 16u;                      // byte offsets of the loaded column
  const unsigned so4 = lane_out ? (unsigned)c_out * 4u : 0u, so16 = so4 * 4u;              // ... of the stored column

  const int lr = lane < 1 ? 1 : (lane > 62 ? 62 : lane); // ring column the advection of this lane reads around
  const int lright = lane < 63 ? lane + 1 : 63;          // right neighbour for the velocity stage
  // registers: newest two input rows, last advection row
  float4 b_new, b_prev = make_float4(0.f, 0.f, 0.f, 0.f);
  int w_new; // the wall texel stays one raw dword until it is used (unpacking it next to the load would wait for the load)
  char4 w_prev = make_char4(0, 0, 0, 0);
  const int *__restrict__ wall_raw = reinterpret_cast<const int *>(in.wall);
  float adv_vy_prev = 0.f, adv_T_prev = 0.f;
  char4 adv_w_prev = make_char4(0, 0, 0, 0);
  float vx_seen = 0.f;              // largest |vx| the velocity pass produced in this wave's rows (VxTrack)
  int big1 = 0, big2 = 0, big3 = 0; // "some |v| >= 0.9" of velocity rows r-1, r-2, r-3
  int nw1 = 0, nw2 = 0, nw3 = 0;    // "no wall cell in the row" of input rows r-1, r-2, r-3 (the wall texture does not change in this kernel)
  // the output row of the previous step, stored at the top of this one (right behind the prefetch): the single vmcnt wait of a
  // step then covers a load and a store that have both had a whole step to complete (gfx9: one in-order counter for both)
  float4 st_p = make_float4(0.f, 0.f, 0.f, 0.f), st_ab = st_p;
  char4 st_w = make_char4(0, 0, 0, 0);
  bool st_valid = false;

  // prefetch of the first row
  int r = y_lo - 2;
  int yw_p1 = wrapmod(r + 1, Y), yw_m2 = wrapmod(r - 2, Y); // wrapped rows r+1 and r-2, advanced by one per step
  {
    const size_t e = (size_t)wrapmod(r, Y) * X;
    b_new = ld_row_v(in.base + e, lo16);
    w_new = ld_row_v(wall_raw + e, lo4);
  }
  auto step = [&]() __attribute__((always_inline)) {
    const int rc = r + 8; // non-negative ring counter
    const float4 b_cur = b_new;
    int w_raw = w_new;
    asm volatile("" : "+v"(w_raw)); // keeps the byte unpacking on this side of the prefetch
    const char4 w_cur = make_char4((signed char)(w_raw & 0xff), (signed char)((w_raw >> 8) & 0xff), (signed char)((w_raw >> 16) & 0xff), (signed char)(w_raw >> 24));
#if WX_MARCH_AIR
    const int nw0 = __all(w_cur.y != 0);
#endif
    // software prefetch: next row's loads are in flight while this row is processed
    if (r < y_hi + 1) {
      const size_t e = (size_t)yw_p1 * X;
      b_new = ld_row_v(in.base + e, lo16);
      w_new = ld_row_v(wall_raw + e, lo4);
    }
    if (st_valid && lane_out) { // row r-3
      const size_t e = (size_t)(r - 3) * X;
      st_row_v(out.base + e, so16, st_p);
      if (WRITE_WALL) st_row_v(out.wall + e, so4, st_w);
      if (WRITE_DISP) st_row_v(out.base_disp + e, so16, st_ab);
    }
    st_valid = false;
    // row r: P, T and wall enter the ring (velocity leaves them unchanged)
    MarchRing::put(rg.P, rc & 3, lane, b_cur.z);
    MarchRing::put(rg.T, rc & 3, lane, b_cur.w);
    MarchRing::put(rg.w, rc & 3, lane, w_cur);
    march_fence(); // one wave per workgroup: orders the wave's LDS traffic, no cross-wave wait

    if (r >= y_lo - 1) { // velocity of row r-1: P of the right neighbour from the ring, P above = this row
      const int s1 = (rc - 1) & 3;
      const float Pr = rg.P[s1][lright];
      float4 v = velocity_cell(u, b_prev, Pr, b_cur.z, w_prev.y);
      if (lane == 63) v.x = v.y = 0.0f; // has no right neighbour in the ring; no output lane ever reads this velocity
      MarchRing::put(rg.vx, s1, lane, v.x);
      MarchRing::put(rg.vy, s1, lane, v.y);
      vx_seen = fmaxf(vx_seen, fabsf(v.x));
      big1 = __any(fmaxf(fabsf(v.x), fabsf(v.y)) >= 0.9f); // any back-trace of this row that may leave the 3x3 cells?
    }
    march_fence();

    if (r >= y_lo + 1) { // advection of row y = r-2 (velocity rows r-3 .. r-1 are in the ring)
      const int y = yw_m2;
      float4 ab, aw;
      char4 awl;
      {
        const int yc = rc - 2, l1 = lr;
        bool fast = true;
        if (big1 | big2 | big3) { // wave-uniform: some velocity of rows y-1 .. y+1 is large -> per-lane test
          const float m = fmaxf(fmaxf(fmaxf(fabsf(rg.vx[yc & 3][l1]), fabsf(rg.vx[yc & 3][l1 - 1])),
                                      fmaxf(fabsf(rg.vx[(yc + 1) & 3][l1]), fabsf(rg.vx[(yc + 1) & 3][l1 - 1]))),
                                fmaxf(fmaxf(fabsf(rg.vy[yc & 3][l1]), fabsf(rg.vy[(yc - 1) & 3][l1])),
                                      fmaxf(fabsf(rg.vy[yc & 3][l1 + 1]), fabsf(rg.vy[(yc - 1) & 3][l1 + 1]))));
          fast = m < 0.9f;
        }
        if (fast) {
          MDryAcc a{rg, l1, yc};
#if WX_MARCH_AIR
          if (nw1 & nw2 & nw3) // (wave-uniform) no wall cell in the three rows the footprints reach: plain interpolation, no wall branch
            advection_cell<true, true, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
          else
#endif
            advection_cell<true, false, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
        } else { // exact out-of-line path (velocity recomputed from global memory)
          const AdvOut o = advection_cell_dry_global(ctx, in, false, col, y);
          ab = o.b;
          aw = o.w;
          awl = o.wl;
        }
      }
      if (r >= y_lo + 2) { // pressure of row y: left neighbour by a whole-wave shift, row below from registers
        const float vx_l = wave_from_left(ab.x);
        st_p = pressure_cell(ab, vx_l, adv_vy_prev, adv_T_prev, adv_w_prev.x, adv_w_prev.y);
        st_w = awl;
        st_ab = ab;
        st_valid = true;
      }
      adv_vy_prev = ab.y;
      adv_T_prev = ab.w;
      adv_w_prev = awl;
    }
    b_prev = b_cur;
    w_prev = w_cur;
    big3 = big2;
    big2 = big1;
#if WX_MARCH_AIR
    nw3 = nw2;
    nw2 = nw1;
    nw1 = nw0;
#endif
    yw_p1 = yw_p1 + 1 == Y ? 0 : yw_p1 + 1;
    yw_m2 = yw_m2 + 1 == Y ? 0 : yw_m2 + 1;
  };
#if WX_MARCH_UNROLL >= 2
  // several row steps per loop iteration: the values carried from step to step (prefetched row, previous rows, deferred stores)
  // change registers between the copies instead of being moved (cf. WX_WET_UNROLL2)
  for (; r <= y_hi + 1;) {
    step();
    r++;
#pragma unroll
    for (int k = 1; k < WX_MARCH_UNROLL; k++) {
      if (r > y_hi + 1) break;
      step();
      r++;
    }
  }
#else
  for (; r <= y_hi + 1; r++) step();
#endif
  if (st_valid && lane_out) { // the last row
    const size_t e = (size_t)(y_hi - 1) * X;
    st_row_v(out.base + e, so16, st_p);
    if (WRITE_WALL) st_row_v(out.wall + e, so4, st_w);
    if (WRITE_DISP) st_row_v(out.base_disp + e, so16, st_ab);
  }
  vx_track_commit(vx, vx_seen, lane, strip);
  if (is_edge && order.arrive != nullptr) strip_order_arrive(order, lane, true); // the halo exchange may pack this strip's columns
#ifdef WX_MARCH_TIMING
  if (lane == 0) {
    cycles[2 * ((size_t)seg * n_strips + strip)] = t_begin;
    cycles[2 * ((size_t)seg * n_strips + strip) + 1] = __builtin_readcyclecounter();
  }
#endif
}

// Segment height: the grid is cut so that the number of waves is just under a whole multiple of what the device holds
// at once (CUs x resident waves), i.e. every "round" of waves is full and the last one ends together.
inline int march_capacity()
{
  static int capacity = 0;
  if (!capacity) {
    int dev = 0, ncu = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_march_dry<false, false, true>, 64, 0) != hipSuccess || ncu <= 0 || nb <= 0)
      capacity = 256 * 4 * WX_MARCH_MINWAVES;
    else
      capacity = ncu * nb;
    if (wx_tune_env("WX_MARCH_DEBUG")) fprintf(stderr, "[wx_march] CUs=%d blocks/CU=%d capacity=%d\n", ncu, nb, capacity);
  }
  return capacity;
}
inline int march_seg_rows(int n_strips, int Y)
{
  const int capacity = march_capacity();
  if (const char *e = wx_tune_env("WX_MARCH_SEG")) return atoi(e) < Y ? atoi(e) : Y;
  int maxseg = WX_MARCH_MAXSEG;
  if (const char *e = wx_tune_env("WX_MARCH_MAXSEG")) maxseg = atoi(e) > 0 ? atoi(e) : maxseg;
  for (int k = 1; k < 64; k++) {
    const int nseg = (int)((long long)k * capacity / n_strips);
    if (nseg < 1) continue;
    const int rows = (Y + nseg - 1) / nseg;
    if (rows <= maxseg) return rows < 8 ? (Y < 8 ? Y : 8) : rows;
  }
  return maxseg < Y ? maxseg : Y;
}

inline int march_dry_strips(const Geo &g) { return (g.X + MOUT - 1) / MOUT; }

// A launch covers the strips [strip_lo, strip_lo + strip_count) (strip_count < 0: all of them): the whole width normally, the edge
// strips and the interior separately where a slab overlaps its halo exchange with compute (cf. launch_march_wet). The row
// segmentation is the one of the whole width, so that every strip is cut the same way whichever launch computes it.
// order (split iterations of a slab, StripOrder): the launch covers all strips, the edge strips first / last in dispatch order where the
// shape has row bands (else in the plain order: the device-side hand-offs work either way). Returns the number of edge items of the launch
// (= the arrivals a gate kernel has to wait for), 0 without an order.
inline int launch_march_dry(const Geo &g, const Uni &u, const FullCtx *ctx, const DryIn &in, const DryOut &out, bool write_disp, bool write_wall,
                             hipStream_t stream, int strip_lo = 0, int strip_count = -1, int strip_lo2 = 0, int strip_count2 = 0, const StripOrder *order = nullptr,
                             const VxTrack *vx = nullptr)
{
  const int n_strips_all = march_dry_strips(g), n_first = strip_count < 0 ? n_strips_all : strip_count;
  const int n_strips = n_first + (strip_count2 > 0 ? strip_count2 : 0);
  if (n_strips <= 0) return 0;
  int seg_rows = march_seg_rows(n_strips_all, g.Y);
  int n_seg = (g.Y + seg_rows - 1) / seg_rows, n_full = n_seg, n_half = 0, band_h = 0;
  if (WX_MARCH_XCD && WX_MARCH_BANDS && !wx_tune_env("WX_MARCH_NOTAIL") && !wx_tune_env("WX_MARCH_SEG") && g.Y % 8 == 0) {
    int R = WX_MARCH_BAND_SEG;
    const int bh = g.Y / 8;
    // (Round 5, measured: on grids whose 24-row band segments do not fill the chip -- 4096 x 1024, BASELINE configs[1], is 69 strips x 56
    // segments = 3 864 waves for 5 120 slots -- shorter segments that would fill it are SLOWER: 24 rows 30.3 us per iteration, 20 / 16 / 12
    // rows 30.7-31.0, 8 rows 31.6, 32 rows 36.9 (profiles/r05_c1_segment_sweep.txt; one handle, interleaved, at rest and moving). The
    // 3 warm-up rows per segment cost more than the idle slots; 24 stays.)
    if (const char *e = wx_tune_env("WX_MARCH_BAND_SEG")) R = atoi(e) >= 4 ? atoi(e) : R;
    const int tail = R / 2 + R / 4;
    if (bh >= 3 * R) { // tall enough for at least two full segments and the tail per band
      band_h = bh;
      seg_rows = R;
      n_half = 1;
      n_full = (bh - tail + R - 1) / R;
      n_seg = 8 * (n_full + 2 * n_half);
    }
  }
  StripOrder ord{};
  if (order && order->mode != 0) {
    ord = *order;
    if (!(band_h > 0 && WX_MARCH_XCD)) ord.mode = 3;
  }
  const int edge_items = ord.mode != 0 ? (ord.nl + (n_strips_all - ord.nr0)) * n_seg : 0;
  VxTrack vt = vx ? *vx : VxTrack{nullptr, nullptr, 0.0f, 0, 0};
  if (vx && vx->zone_l < vx->zone_r) { // the watched zone arrives in COLUMNS: this kernel's strips are MOUT columns wide
    vt.zone_l = (vx->zone_l + MOUT - 1) / MOUT;
    vt.zone_r = vx->zone_r / MOUT;
  }
  // (an ordered launch groups edge and interior strips separately: with one wave per workgroup the count is the same)
  const dim3 grid(WX_MARCH_XCD ? 8 * ((n_strips * n_seg + 7) / 8) : n_strips * n_seg);
  static bool dbg = wx_tune_env("WX_MARCH_DEBUG") != nullptr;
  if (dbg) {
    fprintf(stderr, "[wx_march] strips=%d seg_rows=%d segs=%d (bands of %d rows: %d full + 2 x %d short each) waves=%d\n", n_strips, seg_rows, n_seg, band_h, n_full, n_half, n_strips * n_seg);
    dbg = false;
  }
#ifdef WX_MARCH_TIMING
  static unsigned long long *cyc = nullptr;
  static int calls = 0;
  if (!cyc && hipMalloc((void **)&cyc, 16 * (size_t)n_strips_all * n_seg) != hipSuccess) return 0;
#define WX_LAUNCH_M(D, W, Q) hipLaunchKernelGGL((k_march_dry<D, W, Q>), grid, dim3(64), 0, stream, g, u, ctx, in, out, n_strips, seg_rows, n_full, n_half, band_h, n_seg, strip_lo, n_first, strip_lo2, ord, vt, cyc)
#else
#define WX_LAUNCH_M(D, W, Q) hipLaunchKernelGGL((k_march_dry<D, W, Q>), grid, dim3(64), 0, stream, g, u, ctx, in, out, n_strips, seg_rows, n_full, n_half, band_h, n_seg, strip_lo, n_first, strip_lo2, ord, vt)
#endif
  const bool quiet = !(u.userInputType >= 1) && !(u.airplaneValues[3] < 0.0f || u.airplaneValues[3] > 0.9f);
#define WX_LAUNCH_MQ(D, W) \
  do { if (quiet) WX_LAUNCH_M(D, W, true); else WX_LAUNCH_M(D, W, false); } while (0)
  if (write_disp) {
    if (write_wall) WX_LAUNCH_MQ(true, true); else WX_LAUNCH_MQ(true, false);
  } else {
    if (write_wall) WX_LAUNCH_MQ(false, true); else WX_LAUNCH_MQ(false, false);
  }
#undef WX_LAUNCH_MQ
#undef WX_LAUNCH_M
#ifdef WX_MARCH_TIMING
  if (++calls == 40) { // per segment: start / end relative to the first wave of its XCD band, duration (shader clock cycles)
    hipStreamSynchronize(stream);
    std::vector<unsigned long long> c(2 * (size_t)n_strips * n_seg);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    const int per = band_h > 0 ? n_seg / 8 : n_seg;
    for (int sg = 0; sg < n_seg; sg++) {
      if (band_h > 0 && sg / per != 0 && sg / per != 7) continue; // (first and last XCD only)
      unsigned long long t0 = ~0ull;
      for (int q = (sg / per) * per; q < (sg / per + 1) * per; q++)
        for (int k = 0; k < n_strips; k++) t0 = c[2 * ((size_t)q * n_strips + k)] < t0 ? c[2 * ((size_t)q * n_strips + k)] : t0;
      double st = 0, en = 0, du = 0, dmax = 0;
      for (int k = 0; k < n_strips; k++) {
        const size_t i = 2 * ((size_t)sg * n_strips + k);
        st += (double)(c[i] - t0); en += (double)(c[i + 1] - t0); du += (double)(c[i + 1] - c[i]);
        dmax = (double)(c[i + 1] - c[i]) > dmax ? (double)(c[i + 1] - c[i]) : dmax;
      }
      fprintf(stderr, "  dry seg %3d: start %8.0f end %8.0f dur avg %7.0f max %7.0f\n", sg, st / n_strips, en / n_strips, du / n_strips, dmax);
    }
  }
#endif
  return edge_items;
}

} // namespace wx
