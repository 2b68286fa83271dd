// wx_march2.h -- TWO iterations of the dry-air stencil (velocity -> advection -> pressure, twice) in ONE march (round 5 prototype;
// the round-4 verdict's item 3: "the only structural lever left is fewer bytes per iteration").
//
//   lane l <-> column c0 - 4 + l;  step r (new input row r):
//     load row r+1 (prefetch) | velocity_1(r-1) | advection_1(r-2), pressure_1(r-2) = the SECOND iteration's input row r-2, which never
//     leaves the wave | velocity_2(r-3) | advection_2(r-4), pressure_2(r-4) -> store row r-4
//
// HBM sees base 16 R + 16 W and wall 4 R per cell and PAIR of iterations: 18 B per cell-step instead of 36. The price: the cone of the
// chained stencils doubles (56 of 64 lanes produce output instead of 60; 7 warm-up steps per segment instead of 3), a second ring, and
// one wave now does twice the arithmetic per row step. Same cell functions on the same operands as k_march_dry (wx_cells.h): the result
// is bit-identical to two launches of it. A back-trace of 0.9 cells or more in the SECOND iteration has no exact path inside the march
// (its inputs exist in no texture): like the wet kernel, the wave leaves a placeholder and RECORDS the cell (Dry2Fix); k_dry2_fix behind
// the pair rebuilds the neighbourhood of every recorded cell from the pair's untouched inputs -- iteration 1 on an 11 x 11 patch, stage
// by stage through LDS, then iteration 2 for the three output cells the recorded one feeds -- one wavefront per entry (round 6; round 5
// repeated the WHOLE GRID twice for a single such cell: 2.4 x per iteration). Only a back-trace of three cells and more in the second
// iteration (or a list overflow) still repeats the pair with the one-iteration kernel. Only the water-free, wall-constant, brush-free
// state marches in pairs.
#pragma once
#include "wx_march.h"

namespace wx {

constexpr int M2OUT = 56, M2LO = 4; // output lanes 4 .. 59

// rings of the last four rows of both iterations (slot = row & 3) and of the last eight wall rows (slot = row & 7): 10 240 B per
// wave = eight 1 280-byte LDS granules -> 16 waves per CU
struct March2Ring {
  float vx[2][4][MRW], vy[2][4][MRW], P[2][4][MRW], T[2][4][MRW];
  char4 w[8][MRW];
};
static_assert(sizeof(March2Ring) == 10240, "two-iteration ring");

// accessor of iteration K (0 / 1): own cell = (lane1, row yc); rows yc-1 .. yc+1 are in the ring
template <int K> struct M2Acc {
  const March2Ring &rg;
  int lane1, yc;
  __device__ __forceinline__ float4 base(int dx, int dy) const
  {
    const int s = (yc + dy) & 3, l = lane1 + dx;
    return make_float4(rg.vx[K][s][l], rg.vy[K][s][l], rg.P[K][s][l], rg.T[K][s][l]);
  }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return rg.w[(yc + dy) & 7][lane1 + dx]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return base(dx, dy); }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return wall(dx, dy); }
  __device__ __forceinline__ float4 water_off(int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
// bilinear footprint: the two rows' slots once per footprint, the column taps constant offsets
template <int K> struct M2Fp {
  const March2Ring &rg;
  int s0, s1, w0, w1, l;
  __device__ __forceinline__ float4 base(int i, int j) const
  {
    const int s = j ? s1 : s0;
    return make_float4(rg.vx[K][s][l + i], rg.vy[K][s][l + i], rg.P[K][s][l + i], rg.T[K][s][l + i]);
  }
  __device__ __forceinline__ char4 wall(int i, int j) const { return rg.w[j ? w1 : w0][l + i]; }
  __device__ __forceinline__ float4 water(int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <int K> __device__ __forceinline__ M2Fp<K> make_fp(const M2Acc<K> &a, int dx0, int dy0)
{
  const int y = a.yc + dy0;
  return M2Fp<K>{a.rg, y & 3, (y + 1) & 3, y & 7, (y + 1) & 7, a.lane1 + dx0};
}

#ifndef WX_MARCH2_MINWAVES
#define WX_MARCH2_MINWAVES 4
#endif
// The second iteration's exact path. ctl: {entries, arrival ticket of the fix pass, what the host's hint word was last told, epoch of the
// pair that has to be repeated whole, pairs repeated whole, the pair whose list the hint word was last told about, cells recomputed (64 bits), two grid barriers (counter, release word)}; the fix
// pass leaves the first two at 0.
struct Dry2Fix {
  int *ctl;
  int2 *cells;         // second-iteration advection cells (column, row) with a back-trace of 0.9 cells or more, recorded by the wave that owns them
  int cap;
  int epoch;           // of this pair (counts from 1)
  int *hint;           // host-visible word (pinned, mapped; may be NULL): the length of the last list, read (stale) by the host to size the next fix launch
  const int *hint_host;
};
enum { D2_COUNT = 0, D2_TICKET = 1, D2_TOLD = 2, D2_REDO_EPOCH = 3, D2_N_REDO = 4, D2_TOLD_EPOCH = 5, D2_FIXED = 6, D2_BAR0 = 8, D2_BAR1 = 10, D2_WORDS = 12 };
// the tiles (8 of the strip's columns x 8 rows counted from the segment's first) of the lanes in `badmask`, each once per wave
// (rec: (tile row << 8) | mask of the tiles of that row already recorded; returns the new value)
__device__ __forceinline__ int dry2_record_tiles(unsigned long long badmask, int tr, int y_lo, int strip, int lane, int rec, const Dry2Fix &fix)
{
  int rec_tr = rec >> 8;
  unsigned rec_mask = (unsigned)rec & 0xffu;
  unsigned gm = 0;
  for (int gq = 0; gq < M2OUT / 8; gq++) gm |= ((badmask >> (M2LO + 8 * gq)) & 0xffull) ? (1u << gq) : 0u;
  if (tr != rec_tr) rec_mask = 0;
  const unsigned fresh = gm & ~rec_mask;
  rec_tr = tr;
  rec_mask |= gm;
  if (fresh && lane == 0) {
    int at = atomicAdd(fix.ctl + D2_COUNT, __popc(fresh));
    for (int gq = 0; gq < M2OUT / 8; gq++)
      if ((fresh >> gq) & 1u) {
        if (at < fix.cap) fix.cells[at] = make_int2(strip * M2OUT + 8 * gq, y_lo + 8 * tr);
        at++;
      }
  }
  return (rec_tr << 8) | (int)rec_mask;
}
// WRITE_DISP: also store the post-advection base of the SECOND iteration (baseTexture_1 of the last iteration of a frame: display side)
// TAINT: how a FIRST-iteration back-trace of 0.9 cells or more is handled -- see the loop. Same results either way; the host picks.
template <bool QUIET, bool WRITE_DISP, bool TAINT>
__global__ __launch_bounds__(64, WX_MARCH2_MINWAVES) void k_march_dry2(Geo g, const FullCtx *__restrict__ ctx, DryIn in, DryOut out, int n_strips, int seg_rows, int n_full, int n_half,
                                                                        int band_h, int n_seg, VxTrack vx, Dry2Fix fix)
{
  __shared__ March2Ring rg;
  CUni &u = as_constant(ctx->u);
  const CFloatP initial_T = as_constant(ctx->initial_T), snd_T = as_constant(ctx->snd_T), snd_W = as_constant(ctx->snd_W), snd_Vel = as_constant(ctx->snd_Vel);
  const int X = g.X, Y = g.Y;
  const int lane = threadIdx.x;
  // (segment, strip) items segment-major, XCD k takes the contiguous range [k * T / 8, (k + 1) * T / 8): see k_march_dry
  const int total = n_strips * n_seg, k = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int first = (int)(((long long)k * total) >> 3), count = (int)(((long long)(k + 1) * total) >> 3) - first;
  if (j >= count) return;
  const int item = first + j, seg = item / n_strips, strip = item - seg * n_strips;
  const int c_out = strip * M2OUT + lane - M2LO;
  const int col = wrapmod(c_out, X);
  const bool lane_out = lane >= M2LO && lane < M2LO + M2OUT && c_out < X;
  int y_lo, y_hi;
  if (band_h > 0) {
    const int nb = n_full + 2 * n_half, bnd = seg / nb, sl = seg - bnd * nb;
    const int t1 = sl - n_full, t2 = t1 - n_half, h2 = seg_rows >> 1, h4 = seg_rows >> 2, y0 = bnd * band_h;
    y_lo = y0 + (t1 < 0 ? sl * seg_rows : (t2 < 0 ? n_full * seg_rows + t1 * h2 : n_full * seg_rows + n_half * h2 + t2 * h4));
    y_hi = min(y_lo + (t1 < 0 ? seg_rows : (t2 < 0 ? h2 : h4)), y0 + band_h);
    if (y_lo >= y_hi) return;
  } else {
    y_lo = seg * seg_rows;
    y_hi = min(y_lo + seg_rows, Y);
  }
  const unsigned lo4 = (unsigned)col * 4u, lo16 = (unsigned)col * 16u;
  const unsigned so16 = lane_out ? (unsigned)c_out * 16u : 0u;
  const int lr = lane < 1 ? 1 : (lane > 62 ? 62 : lane); // ring column the advection stages read around (edge lanes never feed an output)
  const int lright = lane < 63 ? lane + 1 : 63;

  // carried registers
  float4 b_new, b_prev = make_float4(0.f, 0.f, 0.f, 0.f);     // input rows r+1 (prefetched) / r-1
  int w_new;
  int w_prev_y = 0;                                           // wall distance byte of row r-1
  float a1_vy = 0.f, a1_T = 0.f;                              // advection_1 output of the row below the one pressure_1 finishes
  char4 a1_w = make_char4(0, 0, 0, 0);
  float4 b2_prev = make_float4(0.f, 0.f, 0.f, 0.f);           // second iteration's input row r-3
  float a2_vy = 0.f, a2_T = 0.f;
  char4 a2_w = make_char4(0, 0, 0, 0);
  float4 st_p = make_float4(0.f, 0.f, 0.f, 0.f), st_ab = st_p;
  bool st_valid = false;
  float vx_seen = 0.f;
  unsigned h_big1 = 0, h_big2 = 0; // bit k: "some |v| >= 0.9" in the velocity row k steps back (bit 0 = the row this step produced)
  unsigned h_nw = 0;               // bit k: input row r-k holds no wall cell
  unsigned h_bad1 = 0;             // bit k: some lane of the first iteration's advection row k steps back was tainted (see there)
  int rec = -256;                  // (tile row of this segment << 8) | the 7-bit mask of its tiles that are on the exact-path list already

  const int *__restrict__ wall_raw = reinterpret_cast<const int *>(in.wall);
  int r = y_lo - 4;
  int yw_p1 = wrapmod(r + 1, Y), yw_m2 = wrapmod(r - 2, Y), yw_m4 = wrapmod(r - 4, Y);
  {
    const size_t e = (size_t)wrapmod(r, Y) * X;
    b_new = ld_row_v(in.base + e, lo16);
    w_new = ld_row_v(wall_raw + e, lo4);
  }
  for (; r <= y_hi + 3; r++) {
    const int rc = r + 16; // non-negative row counter: ring slots are rc & 3 / rc & 7
    const float4 b_cur = b_new;
    int w_raw = w_new;
    asm volatile("" : "+v"(w_raw));
    const char4 w_cur = unpack_wall(w_raw);
    h_nw = (h_nw << 1) | (__all(w_cur.y != 0) ? 1u : 0u);
    if (r < y_hi + 3) { // prefetch
      const size_t e = (size_t)yw_p1 * X;
      b_new = ld_row_v(in.base + e, lo16);
      w_new = ld_row_v(wall_raw + e, lo4);
    }
    if (st_valid && lane_out) { // row r-5, finished by the previous step
      st_row_v(out.base + (size_t)(r - 5) * X, so16, st_p);
      if (WRITE_DISP) st_row_v(out.base_disp + (size_t)(r - 5) * X, so16, st_ab);
    }
    st_valid = false;
    // ---- iteration 1 ----
    rg.P[0][rc & 3][lane] = b_cur.z;
    rg.T[0][rc & 3][lane] = b_cur.w;
    rg.w[rc & 7][lane] = w_cur;
    march_fence();
    h_big1 <<= 1;
    if (r >= y_lo - 3) { // velocity_1 of row r-1
      const int s1 = (rc - 1) & 3;
      float4 v = velocity_cell(u, b_prev, rg.P[0][s1][lright], b_cur.z, w_prev_y);
      if (lane == 63) v.x = v.y = 0.0f;
      rg.vx[0][s1][lane] = v.x;
      rg.vy[0][s1][lane] = v.y;
      vx_seen = fmaxf(vx_seen, fabsf(v.x));
      h_big1 |= __any(fmaxf(fabsf(v.x), fabsf(v.y)) >= 0.9f) ? 1u : 0u;
    }
    march_fence();
    float4 b2 = make_float4(0.f, 0.f, 0.f, 0.f); // pressure_1 of row r-2 = the second iteration's input row
    h_big2 <<= 1;
    if (r >= y_lo - 1) { // advection_1 of row r-2
      const int y = yw_m2, yc = rc - 2;
      float4 ab, aw;
      char4 awl;
      bool fast = true;
      if (TAINT) h_bad1 <<= 1;
      if (h_big1 & 7u) {
        if (!TAINT) {
          const float m = fmaxf(fmaxf(fmaxf(fabsf(rg.vx[0][yc & 3][lr]), fabsf(rg.vx[0][yc & 3][lr - 1])), fmaxf(fabsf(rg.vx[0][(yc + 1) & 3][lr]), fabsf(rg.vx[0][(yc + 1) & 3][lr - 1]))),
                                fmaxf(fmaxf(fabsf(rg.vy[0][yc & 3][lr]), fabsf(rg.vy[0][(yc - 1) & 3][lr])), fmaxf(fabsf(rg.vy[0][yc & 3][lr + 1]), fabsf(rg.vy[0][(yc - 1) & 3][lr + 1]))));
          fast = m < 0.9f;
        } else {
          const float q0 = rg.vx[0][yc & 3][lr], q1 = rg.vx[0][yc & 3][lr - 1], q2 = rg.vx[0][(yc + 1) & 3][lr], q3 = rg.vx[0][(yc + 1) & 3][lr - 1];
          const float q4 = rg.vy[0][yc & 3][lr], q5 = rg.vy[0][(yc - 1) & 3][lr], q6 = rg.vy[0][yc & 3][lr + 1], q7 = rg.vy[0][(yc - 1) & 3][lr + 1];
          const float m = fmaxf(fmaxf(fmaxf(fabsf(q0), fabsf(q1)), fmaxf(fabsf(q2), fabsf(q3))), fmaxf(fmaxf(fabsf(q4), fabsf(q5)), fmaxf(fabsf(q6), fabsf(q7))));
          const float sn = ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7)); // (fmaxf drops a NaN operand; the sum keeps it)
          fast = m < 0.9f && sn == sn;
          h_bad1 |= __any(!fast) ? 1u : 0u;
        }
      }
      const M2Acc<0> a{rg, lr, yc};
      if (fast) {
        if ((h_nw & 14u) == 14u) // no wall cell in input rows r-1 .. r-3
          advection_cell<true, true, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
        else
          advection_cell<true, false, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
      } else if (TAINT) {
        // A back-trace of 0.9 cells or more leaves the ring. The one-iteration kernel takes its out-of-line path through global memory here
        // (and so does the plain instantiation below) -- a serial chase per lane that makes a wave crossing a vortex the straggler of its
        // launch (+20 % on the pair kernel with a few thousand such cells). The TAINT instantiation, which the host launches while the
        // exact-path lists are not empty, taints instead: the cell's post-advection texel is NaN, every second-iteration value that depends
        // on it becomes NaN by arithmetic alone (+, -, x of the cell functions; the places where a NaN could be dropped -- fmaxf in the
        // speed tests -- test for it), and every second-iteration cell that is fast OR tainted puts its tile on the exact-path list:
        // k_dry2_fix recomputes BOTH iterations for those from the pair's inputs. Nothing tainted is ever kept.
        ab = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
        awl = a.wall(0, 0); // (the wall texture is constant in a pair: the caller checked)
      } else { // the first iteration's inputs are in global memory: the exact out-of-line path of k_march_dry
        const AdvOut o = advection_cell_dry_global(ctx, in, false, col, y);
        ab = o.b;
        awl = o.wl;
      }
      if (r >= y_lo) {
        b2 = pressure_cell(ab, wave_from_left(ab.x), a1_vy, a1_T, a1_w.x, a1_w.y);
        rg.P[1][yc & 3][lane] = b2.z;
        rg.T[1][yc & 3][lane] = b2.w;
      }
      a1_vy = ab.y;
      a1_T = ab.w;
      a1_w = awl;
      // ---- iteration 2: velocity_2 of row r-3 (P right of it from the ring: put by the previous step; P above = b2.z) ----
      if (r >= y_lo + 1) {
        const int s3 = (rc - 3) & 3;
        float4 v = velocity_cell(u, b2_prev, rg.P[1][s3][lright], b2.z, rg.w[(rc - 3) & 7][lane].y);
        if (lane == 63) v.x = v.y = 0.0f;
        rg.vx[1][s3][lane] = v.x;
        rg.vy[1][s3][lane] = v.y;
        vx_seen = fmaxf(vx_seen, fabsf(v.x));
        h_big2 |= __any(lane >= 2 && lane <= 61 && fmaxf(fabsf(v.x), fabsf(v.y)) >= 0.9f) ? 1u : 0u; // (fmaxf drops a NaN: tainted rows are covered by h_bad1 below)
      }
    }
    march_fence();
    if (r >= y_lo + 3) { // advection_2 of row r-4
      const int y = yw_m4, yc = rc - 4;
      float4 ab, aw;
      char4 awl;
      bool fast2 = true;
      if ((h_big2 & 7u) | (TAINT ? (h_bad1 & 0xffu) : 0u)) { // (some velocity of rows y-1 .. y+1 is large, or a first-iteration cell of the last rows was tainted)
        if (!TAINT) {
          const float m = fmaxf(fmaxf(fmaxf(fabsf(rg.vx[1][yc & 3][lr]), fabsf(rg.vx[1][yc & 3][lr - 1])), fmaxf(fabsf(rg.vx[1][(yc + 1) & 3][lr]), fabsf(rg.vx[1][(yc + 1) & 3][lr - 1]))),
                                fmaxf(fmaxf(fabsf(rg.vy[1][yc & 3][lr]), fabsf(rg.vy[1][(yc - 1) & 3][lr])), fmaxf(fabsf(rg.vy[1][yc & 3][lr + 1]), fabsf(rg.vy[1][(yc - 1) & 3][lr + 1]))));
          fast2 = m < 0.9f;
        } else {
          const float q0 = rg.vx[1][yc & 3][lr], q1 = rg.vx[1][yc & 3][lr - 1], q2 = rg.vx[1][(yc + 1) & 3][lr], q3 = rg.vx[1][(yc + 1) & 3][lr - 1];
          const float q4 = rg.vy[1][yc & 3][lr], q5 = rg.vy[1][(yc - 1) & 3][lr], q6 = rg.vy[1][yc & 3][lr + 1], q7 = rg.vy[1][(yc - 1) & 3][lr + 1];
          const float m = fmaxf(fmaxf(fmaxf(fabsf(q0), fabsf(q1)), fmaxf(fabsf(q2), fabsf(q3))), fmaxf(fmaxf(fabsf(q4), fabsf(q5)), fmaxf(fabsf(q6), fabsf(q7))));
          const float sn = ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7));
          fast2 = m < 0.9f && sn == sn;
        }
        // No exact path here for a fast cell (the inputs of this iteration exist in no texture): it keeps a placeholder, and the wave that
        // OWNS it as an output cell records the 8 x 8 TILE it lies in (8 of the strip's columns x 8 rows counted from the segment's first):
        // k_dry2_fix recomputes the 9 x 9 output cells such a tile's cells feed (their own, the right neighbour's pressure: vx of the left
        // cell, the upper neighbour's: vy, T, wall of the lower cell), whichever waves own those. Rare path. The TAINT instantiation
        // records a tile once per wave; the plain one -- which runs only until the host has heard of the first entries -- lane by lane
        // (duplicates are harmless: recomputing an output is idempotent), in the very shape of round 5's code: anything else in this branch
        // cost the clean flow 2-3 % through the compiler's layout of the loop around it.
        if (!TAINT) {
          if (!fast2 && lane_out && r - 4 >= y_lo) {
            const int at = atomicAdd(fix.ctl + D2_COUNT, 1);
            if (at < fix.cap) fix.cells[at] = make_int2(strip * M2OUT + ((lane - M2LO) & ~7), y_lo + ((r - 4 - y_lo) & ~7));
          }
        } else {
          const unsigned long long badmask = __ballot(!fast2 && lane_out && r - 4 >= y_lo);
          if (badmask) rec = dry2_record_tiles(badmask, (r - 4 - y_lo) >> 3, y_lo, strip, lane, rec, fix);
        }
      }
      const M2Acc<1> a{rg, lr, yc};
      if (fast2) {
        if ((h_nw & 56u) == 56u) // no wall cell in rows r-3 .. r-5
          advection_cell<true, true, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
        else
          advection_cell<true, false, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
      } else { // placeholder (the post-velocity texel)
        ab = a.base(0, 0);
        awl = a.wall(0, 0);
      }
      if (TAINT && (h_bad1 & 0xffu)) { // (wave-uniform: a first-iteration cell of the last rows was tainted) nothing to keep of a cell whose footprints
        // touched a tainted texel -- a NaN among its four post-advection values: its tile goes to the list like a fast cell's
        const float chk = (ab.x + ab.y) + (ab.z + ab.w);
        const unsigned long long badmask = __ballot(chk != chk && lane_out && r - 4 >= y_lo);
        if (badmask) rec = dry2_record_tiles(badmask, (r - 4 - y_lo) >> 3, y_lo, strip, lane, rec, fix);
      }
      if (r >= y_lo + 4) {
        st_p = pressure_cell(ab, wave_from_left(ab.x), a2_vy, a2_T, a2_w.x, a2_w.y);
        st_ab = ab;
        st_valid = true;
      }
      a2_vy = ab.y;
      a2_T = ab.w;
      a2_w = awl;
    }
    b_prev = b_cur;
    w_prev_y = w_cur.y;
    b2_prev = b2;
    yw_p1 = yw_p1 + 1 == Y ? 0 : yw_p1 + 1;
    yw_m2 = yw_m2 + 1 == Y ? 0 : yw_m2 + 1;
    yw_m4 = yw_m4 + 1 == Y ? 0 : yw_m4 + 1;
  }
  if (st_valid && lane_out) {
    st_row_v(out.base + (size_t)(y_hi - 1) * X, so16, st_p);
    if (WRITE_DISP) st_row_v(out.base_disp + (size_t)(y_hi - 1) * X, so16, st_ab);
  }
  vx_track_commit(vx, vx_seen, lane, strip);
}

// ---- k_dry2_fix: the second iteration's exact path -- one wavefront per recorded TILE ----
// Entry (X0, Y0): origin of an 8 x 8 tile holding cells whose advection_2 had a back-trace of 0.9 cells or more. Such a cell (x, y) feeds
// the OUTPUT cells (x, y), (x + 1, y) [pressure: vx of the left cell] and (x, y + 1) [pressure: vy, T, wall of the lower cell]: the tile's
// cells feed the 9 x 9 outputs [0, 8]^2 (offsets from the origin), which need advection_2 on [-1, 8]^2. The wave rebuilds what those read
// from the pair's untouched INPUTS, stage by stage through LDS, the way the passes follow each other -- one 25 x 25 frame [-8, 16]:
//   base_0 / wall      [-8, 16]   loaded once
//   velocity_1         [-8, 15]   velocity_cell
//   advection_1        [-5, 12]   advection_cell on the stage (back-traces shorter than three cells stay inside; the others take the
//                                 one-iteration kernels' exact path from global memory, advection_cell_dry_global: any length)
//   pressure_1         [-4, 12]   = the second iteration's input
//   velocity_2         [-4, 11]
//   advection_2        [-1, 8]    advection_cell on the stage: covers back-traces shorter than three cells
//   pressure_2         [0, 8]     stored (+ the display field)
// Same cell functions on the same operands as the marching loop: the same values (for the cells of the tile that were NOT fast, too:
// recomputing an output is idempotent, and the order of the entries does not matter). A second-iteration footprint that leaves the
// stage (|v| >= 3: the state is blowing up) raises the epoch word, and the whole pair is repeated, as it is when the list overflowed.
// (Round 6 began with one wavefront per CELL: a vortex ring of a thousand fast cells is forty tiles, and a tile costs three cells.)
constexpr int F2W = 25, F2N = F2W * F2W, F2C = 8; // stage width, cells, stage coordinate of the tile's origin
constexpr int F2T = 8;                            // tile edge
struct Dry2FixStage {
  float b0x[F2N], b0y[F2N], b0P[F2N], b0T[F2N]; // (b0x / b0y hold velocity_2 once velocity_1 is made)
  char4 w[F2N];
  float v1x[F2N], v1y[F2N];                     // (... and pressure_1's P / T once advection_1 is made)
  float a1x[F2N], a1y[F2N], a1P[F2N], a1T[F2N];
  char4 a1w[F2N];
};
// post-velocity base of one iteration on the stage: (vx, vy) valid on [lo, hi]^2, P / T / wall wherever those are
struct Dry2StageAcc {
  const float *vx, *vy, *P, *T;
  const char4 *w;
  int cx, cy, lo, hi;
  bool *bad; // set when a texel outside [lo, hi]^2 is asked for
  __device__ __forceinline__ int at(int dx, int dy) const
  {
    const int x = cx + dx, y = cy + dy;
    if (x < lo || x > hi || y < lo || y > hi) {
      *bad = true;
      return cy * F2W + cx;
    }
    return y * F2W + x;
  }
  __device__ __forceinline__ float4 base(int dx, int dy) const
  {
    const int i = at(dx, dy);
    return make_float4(vx[i], vy[i], P[i], T[i]);
  }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return w[at(dx, dy)]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return base(dx, dy); }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return wall(dx, dy); }
  __device__ __forceinline__ float4 water_off(int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
};

// the entries [first, n) in steps of `stride`, one wavefront each; returns true if one of them left the stage (the whole pair has to be repeated)
constexpr int F2A = F2T + 2; // advection_2 is made on an (F2T + 2)^2 block
struct Dry2FixOut {
  float x[F2A * F2A], y[F2A * F2A], P[F2A * F2A], T[F2A * F2A];
  char4 w[F2A * F2A];
};
template <bool QUIET, bool WRITE_DISP>
__device__ __forceinline__ bool dry2_fix_entries(const Geo &g, const FullCtx *__restrict__ ctx, const DryIn &in, const DryOut &out, const int2 *__restrict__ cells, int first, int stride,
                                                 int n, Dry2FixStage &st, Dry2FixOut &o, int lane, const VxTrack &vx)
{
  const int X = g.X, Y = g.Y;
  float *const b1P = st.v1x, *const b1T = st.v1y, *const v2x = st.b0x, *const v2y = st.b0y; // (the planes that take over dead ones)
  bool whole = false;
  for (int e = first; e < n; e += stride) {
    const int2 c = cells[e];
    for (int i = lane; i < F2N; i += 64) { // the pair's inputs
      const int sy = i / F2W, sx = i - sy * F2W;
      const size_t gi = fidx(wrapmod(c.x - F2C + sx, X), wrapmod(c.y - F2C + sy, Y), X);
      const float4 b = in.base[gi];
      st.b0x[i] = b.x;
      st.b0y[i] = b.y;
      st.b0P[i] = b.z;
      st.b0T[i] = b.w;
      st.w[i] = in.wall[gi];
    }
    wave_fence();
    for (int k = lane; k < 24 * 24; k += 64) { // velocity_1 on [0, 23]^2
      const int sy = k / 24, sx = k - sy * 24, i = sy * F2W + sx;
      const float4 v = velocity_cell(ctx->u, make_float4(st.b0x[i], st.b0y[i], st.b0P[i], st.b0T[i]), st.b0P[i + 1], st.b0P[i + F2W], st.w[i].y);
      st.v1x[i] = v.x;
      st.v1y[i] = v.y;
    }
    wave_fence();
    for (int k = lane; k < 18 * 18; k += 64) { // advection_1 on [3, 20]^2
      const int sy = 3 + k / 18, sx = 3 + k % 18, i = sy * F2W + sx;
      const int gx = wrapmod(c.x - F2C + sx, X), gy = wrapmod(c.y - F2C + sy, Y);
      bool left = false;
      const Dry2StageAcc a{st.v1x, st.v1y, st.b0P, st.b0T, st.w, sx, sy, 0, 23, &left};
      float4 ab, aw;
      char4 awl;
      advection_cell<true, false, false, QUIET>(ctx->u, g, ctx->initial_T, ctx->snd_T, ctx->snd_W, ctx->snd_Vel, gx, gy, a, ab, aw, awl);
      if (left) { // a footprint outside the stage: the exact path of the one-iteration kernels
        const AdvOut ao = advection_cell_dry_global(ctx, in, false, gx, gy);
        ab = ao.b;
        awl = ao.wl;
      }
      st.a1x[i] = ab.x;
      st.a1y[i] = ab.y;
      st.a1P[i] = ab.z;
      st.a1T[i] = ab.w;
      st.a1w[i] = awl;
    }
    wave_fence(); // (velocity_1 is dead from here on: its planes take pressure_1)
    for (int k = lane; k < 17 * 17; k += 64) { // pressure_1 on [4, 20]^2: the second iteration's input
      const int sy = 4 + k / 17, sx = 4 + k % 17, i = sy * F2W + sx;
      const float4 p = pressure_cell(make_float4(st.a1x[i], st.a1y[i], st.a1P[i], st.a1T[i]), st.a1x[i - 1], st.a1y[i - F2W], st.a1T[i - F2W], st.a1w[i - F2W].x,
                                     st.a1w[i - F2W].y);
      b1P[i] = p.z;
      b1T[i] = p.w;
    }
    wave_fence();
    for (int k = lane; k < 16 * 16; k += 64) { // velocity_2 on [4, 19]^2 (into the planes of base_0's velocities, dead since velocity_1)
      const int sy = 4 + k / 16, sx = 4 + k % 16, i = sy * F2W + sx;
      const float4 v = velocity_cell(ctx->u, make_float4(st.a1x[i], st.a1y[i], b1P[i], b1T[i]), b1P[i + 1], b1P[i + F2W], st.w[i].y);
      v2x[i] = v.x;
      v2y[i] = v.y;
    }
    wave_fence();
    // (slabs size their exchange periods by the largest |vx| they produce: the marching loop could not see the second iteration's velocity
    // where its input was tainted -- fmaxf drops NaN -- so the tile's own 64 cells report theirs here)
    vx_track_commit(vx, fabsf(v2x[(F2C + (lane >> 3)) * F2W + F2C + (lane & 7)]), lane, c.x / M2OUT);
    bool left2 = false;
    for (int k = lane; k < F2A * F2A; k += 64) { // advection_2 on [7, 16]^2
      const int oy = k / F2A, ox = k - oy * F2A, sy = F2C - 1 + oy, sx = F2C - 1 + ox;
      const Dry2StageAcc a{v2x, v2y, b1P, b1T, st.w, sx, sy, 4, 19, &left2};
      float4 ab, aw;
      char4 awl;
      advection_cell<true, false, false, QUIET>(ctx->u, g, ctx->initial_T, ctx->snd_T, ctx->snd_W, ctx->snd_Vel, wrapmod(c.x - F2C + sx, X), wrapmod(c.y - F2C + sy, Y), a, ab, aw, awl);
      o.x[k] = ab.x;
      o.y[k] = ab.y;
      o.P[k] = ab.z;
      o.T[k] = ab.w;
      o.w[k] = awl;
    }
    wave_fence();
    if (__any(left2)) {
      whole = true;
    } else {
      for (int k = lane; k < (F2T + 1) * (F2T + 1); k += 64) { // pressure_2 of the 9 x 9 outputs: (1 + i, 1 + j) of the advection_2 block
        const int oy = k / (F2T + 1), ox = k - oy * (F2T + 1), q = (oy + 1) * F2A + (ox + 1);
        const float4 ab = make_float4(o.x[q], o.y[q], o.P[q], o.T[q]);
        const float4 p = pressure_cell(ab, o.x[q - 1], o.y[q - F2A], o.T[q - F2A], o.w[q - F2A].x, o.w[q - F2A].y);
        const size_t gi = fidx(wrapmod(c.x + ox, X), wrapmod(c.y + oy, Y), X);
        out.base[gi] = p;
        if (WRITE_DISP) out.base_disp[gi] = ab;
      }
    }
    wave_fence(); // the stage is rewritten by the next entry
  }
  return whole;
}
// the list is empty again for the next pair (the caller: the LAST workgroup to get here -- every workgroup has read the count by then)
__device__ __forceinline__ void dry2_list_reset(const Dry2Fix &fix, int total, int n)
{
  int *ctl = fix.ctl;
  ctl[D2_TICKET] = 0;
  ctl[D2_TOLD] = total;
  ctl[D2_TOLD_EPOCH] = fix.epoch; // (k_dry2_post of the same pair must not take the word back: it finds the list empty because it was just consumed)
  atomicAdd(reinterpret_cast<unsigned long long *>(ctl + D2_FIXED), (unsigned long long)n * ((F2T + 1) * (F2T + 1))); // (output cells recomputed: 9 x 9 per tile)
  __hip_atomic_store(ctl + D2_COUNT, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fix.hint) __hip_atomic_store(fix.hint, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The WIDE fix pass: launched in front of k_dry2_post only while the host's (stale) hint word says that the lists hold entries -- up to
// 4096 workgroups of one wave. (The occupancy bound is k_march_dry's: the kernels share the out-of-line exact path advection_cell_dry_global,
// and the compiler gives a shared callee the LOOSEST register budget among its callers -- an unbounded one here would double the
// registers of every marching kernel.)
template <bool QUIET, bool WRITE_DISP>
__global__ __launch_bounds__(64, WX_MARCH_MINWAVES) void k_dry2_fix(Geo g, const FullCtx *__restrict__ ctx, DryIn in, DryOut out, Dry2Fix fix, VxTrack vx)
{
  __shared__ Dry2FixStage st;
  __shared__ Dry2FixOut o;
  int *ctl = fix.ctl;
  const int total = ctl[D2_COUNT], lane = threadIdx.x;
  if (total == 0) return; // (k_dry2_post behind this launch sets the hint word back)
  const int n = total <= fix.cap ? total : 0; // (an overflowed list: the whole pair is repeated, nothing to do here)
  const bool whole = dry2_fix_entries<QUIET, WRITE_DISP>(g, ctx, in, out, fix.cells, blockIdx.x, gridDim.x, n, st, o, lane, vx) || total > fix.cap;
  if (whole && lane == 0 && atomicMax(ctl + D2_REDO_EPOCH, fix.epoch) < fix.epoch) atomicAdd(ctl + D2_N_REDO, 1);
  if (lane == 0 && atomicAdd(ctl + D2_TICKET, 1) == (int)gridDim.x - 1) dry2_list_reset(fix, total, n);
}

// ---- the repeat of a WHOLE pair (a second-iteration back-trace of three cells or more, or an overflowed list): one iteration of the ----
// ---- one-iteration stencil over the pair's own (segment, strip) items, grid-stride ----
// The same cell functions as k_march_dry (its exact out-of-line path included: THIS iteration's inputs are in global memory), so the repeated
// pair equals two launches of k_march_dry bit for bit. Not tuned: it runs when the state is blowing up.
template <bool QUIET, bool WRITE_DISP>
__device__ __forceinline__ void march_dry_redo_items(const Geo &g, const FullCtx *__restrict__ ctx, const DryIn &in, const DryOut &out, int n_strips, int seg_rows, int n_full, int n_half,
                                                     int band_h, int n_seg, const VxTrack &vx, March2Ring &rg, int nwg)
{
  CUni &u = as_constant(ctx->u);
  const CFloatP initial_T = as_constant(ctx->initial_T), snd_T = as_constant(ctx->snd_T), snd_W = as_constant(ctx->snd_W), snd_Vel = as_constant(ctx->snd_Vel);
  const int X = g.X, Y = g.Y, lane = threadIdx.x;
  const int total = n_strips * n_seg;
  const int lr = lane < 1 ? 1 : (lane > 62 ? 62 : lane), lright = lane < 63 ? lane + 1 : 63;
  const int *__restrict__ wall_raw = reinterpret_cast<const int *>(in.wall);
  for (int item = blockIdx.x; item < total; item += nwg) {
    float vx_seen = 0.f;
    const int seg = item / n_strips, strip = item - seg * n_strips;
    const int c_out = strip * M2OUT + lane - M2LO, col = wrapmod(c_out, X);
    const bool lane_out = lane >= M2LO && lane < M2LO + M2OUT && c_out < X;
    int y_lo, y_hi;
    if (band_h > 0) {
      const int nb = n_full + 2 * n_half, bnd = seg / nb, sl = seg - bnd * nb;
      const int t1 = sl - n_full, t2 = t1 - n_half, h2 = seg_rows >> 1, h4 = seg_rows >> 2, y0 = bnd * band_h;
      y_lo = y0 + (t1 < 0 ? sl * seg_rows : (t2 < 0 ? n_full * seg_rows + t1 * h2 : n_full * seg_rows + n_half * h2 + t2 * h4));
      y_hi = min(y_lo + (t1 < 0 ? seg_rows : (t2 < 0 ? h2 : h4)), y0 + band_h);
      if (y_lo >= y_hi) continue;
    } else {
      y_lo = seg * seg_rows;
      y_hi = min(y_lo + seg_rows, Y);
    }
    const unsigned lo4 = (unsigned)col * 4u, lo16 = (unsigned)col * 16u, so16 = lane_out ? (unsigned)c_out * 16u : 0u;
    float4 b_prev = make_float4(0.f, 0.f, 0.f, 0.f);
    int w_prev_y = 0;
    float a_vy = 0.f, a_T = 0.f;
    char4 a_w = make_char4(0, 0, 0, 0);
    unsigned h_big = 0, h_nw = 0;
    march_fence(); // (the ring is reused from the previous item)
    for (int r = y_lo - 2; r <= y_hi + 1; r++) {
      const int rc = r + 16;
      const size_t e = (size_t)wrapmod(r, Y) * X;
      const float4 b_cur = ld_row_v(in.base + e, lo16);
      const char4 w_cur = unpack_wall(ld_row_v(wall_raw + e, lo4));
      h_nw = (h_nw << 1) | (__all(w_cur.y != 0) ? 1u : 0u);
      rg.P[0][rc & 3][lane] = b_cur.z;
      rg.T[0][rc & 3][lane] = b_cur.w;
      rg.w[rc & 7][lane] = w_cur;
      march_fence();
      h_big <<= 1;
      if (r >= y_lo - 1) {
        const int s1 = (rc - 1) & 3;
        float4 v = velocity_cell(u, b_prev, rg.P[0][s1][lright], b_cur.z, w_prev_y);
        if (lane == 63) v.x = v.y = 0.0f;
        rg.vx[0][s1][lane] = v.x;
        rg.vy[0][s1][lane] = v.y;
        vx_seen = fmaxf(vx_seen, fabsf(v.x));
        h_big |= __any(fmaxf(fabsf(v.x), fabsf(v.y)) >= 0.9f) ? 1u : 0u;
      }
      march_fence();
      if (r >= y_lo + 1) {
        const int y = wrapmod(r - 2, Y), yc = rc - 2;
        float4 ab, aw;
        char4 awl;
        bool fast = true;
        if (h_big & 7u) {
          const float m = fmaxf(fmaxf(fmaxf(fabsf(rg.vx[0][yc & 3][lr]), fabsf(rg.vx[0][yc & 3][lr - 1])), fmaxf(fabsf(rg.vx[0][(yc + 1) & 3][lr]), fabsf(rg.vx[0][(yc + 1) & 3][lr - 1]))),
                                fmaxf(fmaxf(fabsf(rg.vy[0][yc & 3][lr]), fabsf(rg.vy[0][(yc - 1) & 3][lr])), fmaxf(fabsf(rg.vy[0][yc & 3][lr + 1]), fabsf(rg.vy[0][(yc - 1) & 3][lr + 1]))));
          fast = m < 0.9f;
        }
        if (fast) {
          const M2Acc<0> a{rg, lr, yc};
          if ((h_nw & 14u) == 14u)
            advection_cell<true, true, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
          else
            advection_cell<true, false, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, y, a, ab, aw, awl);
        } else {
          const AdvOut o = advection_cell_dry_global(ctx, in, false, col, y);
          ab = o.b;
          aw = o.w;
          awl = o.wl;
        }
        if (r >= y_lo + 2) {
          const float4 p = pressure_cell(ab, wave_from_left(ab.x), a_vy, a_T, a_w.x, a_w.y);
          if (lane_out) {
            st_row_v(out.base + (size_t)(r - 2) * X, so16, p);
            if (WRITE_DISP) st_row_v(out.base_disp + (size_t)(r - 2) * X, so16, ab);
          }
        }
        a_vy = ab.y;
        a_T = ab.w;
        a_w = awl;
      }
      b_prev = b_cur;
      w_prev_y = w_cur.y;
    }
    vx_track_commit(vx, vx_seen, lane, strip); // (per item: the watched zone of a slab is in strips, as in k_march_dry2)
  }
}

// a barrier of the whole (co-resident) grid, reusable without a reset between launches: the last workgroup to arrive clears the counter
// and publishes the launch's own number; the others poll for it (bounded: a grid that cannot be resident traps instead of hanging)
__device__ __forceinline__ void dry2_grid_barrier(int *cnt, int *rel, int epoch, int lane, int nwg)
{
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (lane == 0) {
    if (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rel, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > 40000000u) __builtin_trap();
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  wave_fence();
#endif
}

// ---- k_dry2_post: everything that may have to follow a pair, in ONE launch (round 6: the fix pass and the two predicated repeat launches ----
// ---- were three dependent launches of ~5 us each behind every pair -- a quarter of a 4096 x 1024 pair) ----
// A small co-resident grid (D2_POST_GRID workgroups of one wave). Quiet pair -- list empty, epoch word not this pair's --: one scalar load
// per workgroup, and out. Entries on the list (the host's hint word was stale: a flow that has just produced its first fast cells; with a
// non-zero hint the wide k_dry2_fix runs in front and leaves the list empty): fixed here, grid-stride. The pair has to be repeated whole
// (decided by either fix pass): iteration 1 from the pair's inputs into the scratch buffer -> grid barrier -> iteration 2 into the pair's
// output (+ display field).
// (128: the barrier needs every workgroup of the launch resident at once, and up to eight slab handles may share one device in the group
// tests -- 8 x 128 workgroups of 32 KB of LDS fit the chip's 1 280 slots together, so no two such launches can starve each other)
#ifndef WX_POST_GRID
#define WX_POST_GRID 512
#endif
constexpr int D2_POST_GRID = WX_POST_GRID, D2_POST_ACTIVE = 128;
template <bool QUIET, bool WRITE_DISP>
__global__ __launch_bounds__(64, WX_MARCH_MINWAVES) void k_dry2_post(Geo g, const FullCtx *__restrict__ ctx, DryIn in, DryOut out, Dry2Fix fix, float4 *__restrict__ scratch, int n_strips,
                                                                      int seg_rows, int n_full, int n_half, int band_h, int n_seg, VxTrack vx)
{
  __shared__ union Mem {
    Dry2FixStage st;
    March2Ring rg;
    __device__ Mem() {}
  } m;
  __shared__ Dry2FixOut o;
  int *ctl = fix.ctl;
  const int lane = threadIdx.x;
  const int total = ctl[D2_COUNT];
  bool redo = ctl[D2_REDO_EPOCH] == fix.epoch;
  if (total == 0 && !redo) { // the usual case (the host's hint word is set back once)
    if (blockIdx.x == 0 && lane == 0 && fix.hint && ctl[D2_TOLD] != 0 && ctl[D2_TOLD_EPOCH] != fix.epoch) {
      ctl[D2_TOLD] = 0;
      __hip_atomic_store(fix.hint, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  // (only the first D2_POST_ACTIVE workgroups work: the barriers need every participant resident at once; the rest of the grid is there
  // for the quiet case only, see launch_march_dry2)
  const int nwg = min((int)gridDim.x, D2_POST_ACTIVE);
  if ((int)blockIdx.x >= nwg) return;
  if (total > 0) {
    const int n = total <= fix.cap ? total : 0;
    const bool whole = dry2_fix_entries<QUIET, WRITE_DISP>(g, ctx, in, out, fix.cells, blockIdx.x, nwg, n, m.st, o, lane, vx) || total > fix.cap;
    if (whole && lane == 0 && atomicMax(ctl + D2_REDO_EPOCH, fix.epoch) < fix.epoch) atomicAdd(ctl + D2_N_REDO, 1);
    dry2_grid_barrier(ctl + D2_BAR0, ctl + D2_BAR0 + 1, fix.epoch, lane, nwg); // every workgroup has read the count and raised what it had to raise
    if (blockIdx.x == 0 && lane == 0) dry2_list_reset(fix, total, n);
    redo = __hip_atomic_load(ctl + D2_REDO_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fix.epoch;
  }
  if (!redo) return;
  DryIn in2 = in;
  in2.base = scratch;
  DryOut out1 = out;
  out1.base = scratch;
  march_dry_redo_items<QUIET, false>(g, ctx, in, out1, n_strips, seg_rows, n_full, n_half, band_h, n_seg, vx, m.rg, nwg);
  dry2_grid_barrier(ctl + D2_BAR1, ctl + D2_BAR1 + 1, fix.epoch, lane, nwg);
  march_dry_redo_items<QUIET, WRITE_DISP>(g, ctx, in2, out, n_strips, seg_rows, n_full, n_half, band_h, n_seg, vx, m.rg, nwg);
}

// Unit segment height of a row band: 3/16 of the band -- five full segments + 1/2 + 1/4 per band. One handle each, interleaved,
// ms per iteration (profiles/r05_dry_pairs.txt): 32768 x 4096 (512-row bands) 48 rows 0.694, 64 0.688, 96 0.674, 128 0.690 (one iteration
// per launch 0.947); 16384 x 2048 (256-row bands) 24 rows 0.191, 32 0.188, 48 0.181, 64 0.189 (0.250); 4096 x 1024 (128-row bands) 16 rows
// 0.035, 24 0.030, 32 0.036 (0.031-0.037). Taller segments re-run fewer warm-up steps (7 per segment), shorter ones fill the chip.
#ifndef WX_MARCH2_BAND_SEG
#define WX_MARCH2_BAND_SEG 0 // 0: 3/16 of the band height; else the unit segment height in rows
#endif
inline int march2_strips(const Geo &g) { return (g.X + M2OUT - 1) / M2OUT; }
// (whole width only: pairs do not take part in split iterations)
// fix: the second iteration's exact-path list + control words; scratch: a base-sized buffer for the intermediate state of a pair repeated whole
inline void launch_march_dry2(const Geo &g, const Uni &u, const FullCtx *ctx, const DryIn &in, const DryOut &out, bool write_disp, hipStream_t stream, const VxTrack *vx,
                              const Dry2Fix &fix, float4 *scratch)
{
  const int n_strips = march2_strips(g);
  int R = WX_MARCH2_BAND_SEG;
  if (R <= 0) {
    R = std::max(8, std::min(128, ((3 * (g.Y / 8)) / 16) & ~3));
    // narrow slabs: where segments of that height leave more than half of the chip's wave slots (16 per CU) empty, the height that fills
    // them once -- 2048 + 2 x 42 columns x 2048 rows (39 strips): 48 rows 0.0438, 24 rows 0.0351 ms per iteration; the north star's slab
    // (75 strips x 4096 rows) and 4096 x 1024 keep their 96 / 24 rows (profiles/r05_slab_segment_specs.txt)
    const long long fill = (long long)n_strips * g.Y / (16LL * 256LL);
    if (g.Y % 8 == 0 && 2 * fill <= R) R = std::max(8, (int)((fill + 7) / 8) * 8);
  }
  if (const char *e = wx_tune_env("WX_MARCH2_BAND_SEG")) R = atoi(e) >= 8 ? atoi(e) : R;
  int seg_rows = std::min(R, g.Y), n_seg = (g.Y + seg_rows - 1) / seg_rows, n_full = n_seg, n_half = 0, band_h = 0;
  if (g.Y % 8 == 0 && g.Y / 8 >= 3 * R) {
    const int bh = g.Y / 8, tail = R / 2 + R / 4;
    band_h = bh;
    n_half = 1;
    n_full = (bh - tail + R - 1) / R;
    n_seg = 8 * (n_full + 2 * n_half);
  }
  VxTrack vt = vx ? *vx : VxTrack{nullptr, nullptr, 0.0f, 0, 0};
  if (vx && vx->zone_l < vx->zone_r) {
    vt.zone_l = (vx->zone_l + M2OUT - 1) / M2OUT;
    vt.zone_r = vx->zone_r / M2OUT;
  }
  const dim3 grid(8 * ((n_strips * n_seg + 7) / 8));
  const bool quiet = !(u.userInputType >= 1) && !(u.airplaneValues[3] < 0.0f || u.airplaneValues[3] > 0.9f);
  // The host's (stale) hint word -- the length of the last exact-path list it has heard of -- picks the instantiation: clean flows run the plain
  // one (first-iteration fast cells, should one appear, take the inline path through global memory), flows with fast cells the TAINT one
  // (no inline path: those cells go to the list with what depends on them). Identical results; an A/B switch in the debug build.
  const int last = fix.hint_host ? *(volatile const int *)fix.hint_host : -1;
  bool taint = last > 0;
  if (const char *e = wx_tune_env("WX_MARCH2_TAINT")) taint = atoi(e) != 0;
#define WX_LAUNCH_M2(Q, D, T) hipLaunchKernelGGL((k_march_dry2<Q, D, T>), grid, dim3(64), 0, stream, g, ctx, in, out, n_strips, seg_rows, n_full, n_half, band_h, n_seg, vt, fix)
#define WX_LAUNCH_M2T(Q, D) \
  do { if (taint) WX_LAUNCH_M2(Q, D, true); else WX_LAUNCH_M2(Q, D, false); } while (0)
  if (quiet) {
    if (write_disp) WX_LAUNCH_M2T(true, true); else WX_LAUNCH_M2T(true, false);
  } else {
    if (write_disp) WX_LAUNCH_M2T(false, true); else WX_LAUNCH_M2T(false, false);
  }
#undef WX_LAUNCH_M2T
#undef WX_LAUNCH_M2
  if (wx_tune_env("WX_MARCH2_NOREDO")) return; // (timing experiments)
  // While the host's (stale) hint word says that the lists hold entries: the wide fix pass -- one wavefront per recorded cell, room for four
  // times the last list, at most 4096 workgroups of one wave (the list is walked grid-stride: any size is correct).
  if (last != 0) {
    const dim3 fgrid(last < 0 ? 1024 : std::min(4096, std::max(64, 4 * last)));
#define WX_LAUNCH_F(Q, D) hipLaunchKernelGGL((k_dry2_fix<Q, D>), fgrid, dim3(64), 0, stream, g, ctx, in, out, fix, vt)
    if (quiet) {
      if (write_disp) WX_LAUNCH_F(true, true); else WX_LAUNCH_F(true, false);
    } else {
      if (write_disp) WX_LAUNCH_F(false, true); else WX_LAUNCH_F(false, false);
    }
#undef WX_LAUNCH_F
  }
  // ... and ONE launch for everything else that may have to follow: entries the hint did not announce, the repeat of the whole pair
  // (inputs -> scratch -> the pair's output buffer + the display field of the second iteration)
#define WX_LAUNCH_P(Q, D) hipLaunchKernelGGL((k_dry2_post<Q, D>), dim3(D2_POST_GRID), dim3(64), 0, stream, g, ctx, in, out, fix, scratch, n_strips, seg_rows, n_full, n_half, band_h, n_seg, vt)
  if (quiet) {
    if (write_disp) WX_LAUNCH_P(true, true); else WX_LAUNCH_P(true, false);
  } else {
    if (write_disp) WX_LAUNCH_P(false, true); else WX_LAUNCH_P(false, false);
  }
#undef WX_LAUNCH_P
}

} // namespace wx
