// wx_dry.h -- BASELINE configs[1]: the dry-air iteration (pass_mask = velocity | advection | pressure) as ONE
// LDS-tiled kernel: the "fused pressure + velocity + advection stencil" of the north star.
//
//   velocity (app.js:5832-5839) -> advection (5881-5890) -> pressure (5893-5900) on a TX x TY tile:
//   pressure on the tile  <- advection output on [-1,0]  <- velocity output on [-2,+1]  <- base_0 on [-2,+2] (P at +1)
//
// HBM traffic per cell-iteration: WATER = false (water texture identically zero outside walls, the config-2
// state): read base 16 + wall 4, write base 16 + wall 4 = 40 B (algorithmic A_dry = 36 B: the wall write is a
// pass-through). WATER = true: + water 16 R + 16 W. Same per-cell arithmetic (wx_cells.h) as every other kernel
// set: bit-identical results. Buffers are pointer-swapped by the host after each launch.
#pragma once
#include "wx_cells.h"
#include "wx_tile.h"
#include "wx_wet.h" // FullCtx

namespace wx {

#ifndef WX_NTD
#define WX_NTD 512
#endif
#ifndef WX_D_MINWAVES
#define WX_D_MINWAVES 6
#endif
constexpr int NTD = WX_NTD;

namespace fd {
constexpr int Q = fb_::REACH;                 // advection reach (1: |v| < 0.9 stays in the tile)
constexpr int L = 1 + Q;                      // velocity output / wall / water on [-(1+Q), +Q]
constexpr int VW = TX + 2 * Q + 1, VH = TY + 2 * Q + 1;
constexpr int B0W = VW + 1, B0H = VH + 1;     // base_0 one further on the high side (P of the right / upper neighbour)
static_assert(VW == fb_::IW && VH == fb_::IH, "advection input tile");
template <bool WATER> struct Smem {
  Planes4<B0H, B0W> b;
  char4 w[VH][VW + 1];
  float qx[WATER ? VH : 1][WATER ? VW : 1], qy[WATER ? VH : 1][WATER ? VW : 1], qz[WATER ? VH : 1][WATER ? VW : 1],
    qw[WATER ? VH : 1][WATER ? VW : 1];
};
} // namespace fd

struct DryIn {
  const float4 *base;
  const char4 *wall;
  const float4 *water; // only read when WATER
};
struct DryOut {
  float4 *base;
  char4 *wall;
  float4 *water;     // only written when WATER
  float4 *base_disp; // optional post-advection base
};

template <bool WATER> struct LDryAcc {
  const fd::Smem<WATER> &sm;
  int lx, ly;
  __device__ __forceinline__ float4 base(int dx, int dy) const { return sm.b.get(ly + dy, lx + dx); }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return sm.w[ly + dy][lx + dx]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return sm.b.get(ly + dy, lx + dx); }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return sm.w[ly + dy][lx + dx]; }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const
  {
    if (WATER) return make_float4(sm.qx[ly + dy][lx + dx], sm.qy[ly + dy][lx + dx], sm.qz[ly + dy][lx + dx], sm.qw[ly + dy][lx + dx]);
    // dry state: 0 in air; wall cells only read their own texel and overwrite .x with the sentinel (:403-409)
    return make_float4(0.f, 0.f, 0.f, 0.f);
  }
};

// exact path for back-traces that leave the tile: velocity output recomputed from global memory
struct GDryAcc {
  const Uni &u;
  DryIn in;
  bool has_water;
  int X, Y, x, y;
  __device__ __forceinline__ float4 vel_at(int dx, int dy) const
  {
    const int xx = wrapmod(x + dx, X), yy = wrapmod(y + dy, Y);
    const int xr = xx + 1 == X ? 0 : xx + 1, yu = yy + 1 == Y ? 0 : yy + 1;
    return velocity_cell(u, in.base[fidx(xx, yy, X)], in.base[fidx(xr, yy, X)].z, in.base[fidx(xx, yu, X)].z, in.wall[fidx(xx, yy, X)].y);
  }
  __device__ __forceinline__ float4 base(int dx, int dy) const { return vel_at(dx, dy); }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return vel_at(dx, dy); }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return in.wall[fidx(wrapmod(x + dx, X), wrapmod(y + dy, Y), X)]; }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return wall(dx, dy); }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const
  {
    const size_t gi = fidx(wrapmod(x + dx, X), wrapmod(y + dy, Y), X);
    if (has_water) return in.water[gi];
    const char4 wl = in.wall[gi]; // trivial water texture: 0 in air, the wall marker (advectionShader.frag:403-409) in walls
    return make_float4(wl.y == 0 ? (wl.x == WALLTYPE_WATER ? 1002.0f : 1001.0f) : 0.0f, 0.f, 0.f, 0.f);
  }
};

__device__ __noinline__ AdvOut advection_cell_dry_global(const FullCtx *__restrict__ c, DryIn in, bool has_water, int x, int y)
{
  GDryAcc a{c->u, in, has_water, c->g.X, c->g.Y, x, y};
  AdvOut o;
  advection_cell(c->u, c->g, c->initial_T, c->snd_T, c->snd_W, c->snd_Vel, x, y, a, o.b, o.w, o.wl);
  return o;
}

template <bool WATER>
__device__ __forceinline__ void advect_dry_cell(const Uni &u, const Geo &g, const FullCtx *ctx, const fd::Smem<WATER> &sm, const DryIn &in, int x,
                                                int y, int lx, int ly, float4 &b, float4 &w, char4 &wl)
{
  const float m = fmaxf(fmaxf(fmaxf(fabsf(sm.b.x[ly][lx]), fabsf(sm.b.x[ly][lx - 1])), fmaxf(fabsf(sm.b.x[ly + 1][lx]), fabsf(sm.b.x[ly + 1][lx - 1]))),
                        fmaxf(fmaxf(fabsf(sm.b.y[ly][lx]), fabsf(sm.b.y[ly - 1][lx])), fmaxf(fabsf(sm.b.y[ly][lx + 1]), fabsf(sm.b.y[ly - 1][lx + 1]))));
#ifdef WX_ABL_NOADV
  b = sm.b.get(ly, lx);
  w = make_float4(0.f, 0.f, 0.f, 0.f);
  wl = sm.w[ly][lx];
  return;
#endif
  if (m < fb_::VMAX) {
    LDryAcc<WATER> a{sm, lx, ly};
    advection_cell<!WATER>(u, g, ctx->initial_T, ctx->snd_T, ctx->snd_W, ctx->snd_Vel, x, y, a, b, w, wl);
  } else {
    const AdvOut o = advection_cell_dry_global(ctx, in, WATER, x, y);
    b = o.b;
    w = o.w;
    wl = o.wl;
  }
}

template <bool WATER, bool WRITE_DISP>
__global__ __launch_bounds__(NTD, WX_D_MINWAVES) void k_fused_dry(Geo g, Uni u, const FullCtx *__restrict__ ctx, DryIn in, DryOut out)
{
  using namespace fd;
  __shared__ union {
    Smem<WATER> in;
    fb_::SmemOut out;
  } sm;
  const int X = g.X, Y = g.Y;
  const int tid = threadIdx.x;
  int tbx, tby;
  tile_of_block(tiles_x(X), tbx, tby);
  const int tx0 = tbx * TX, ty0 = tby * TY;
  const bool small = (X < TX + 8) || (Y < TY + 8);
#define WX_WRAPX(v) (small ? wrapmod((v), X) : wrapfast((v), X))
#define WX_WRAPY(v) (small ? wrapmod((v), Y) : wrapfast((v), Y))

  // ---- stage 0: base_0 on [-L, +Q+1], wall_0 (and water) on [-L, +Q] ----
  for (int i = tid; i < B0W * B0H; i += NTD) {
    const int ly = i / B0W, lx = i - ly * B0W;
    sm.in.b.put(ly, lx, in.base[fidx(WX_WRAPX(tx0 + lx - L), WX_WRAPY(ty0 + ly - L), X)]);
  }
  for (int i = tid; i < VW * VH; i += NTD) {
    const int ly = i / VW, lx = i - ly * VW;
    const size_t gi = fidx(WX_WRAPX(tx0 + lx - L), WX_WRAPY(ty0 + ly - L), X);
    sm.in.w[ly][lx] = in.wall[gi];
    if (WATER) {
      const float4 q = in.water[gi];
      sm.in.qx[ly][lx] = q.x;
      sm.in.qy[ly][lx] = q.y;
      sm.in.qz[ly][lx] = q.z;
      sm.in.qw[ly][lx] = q.w;
    }
  }
  __syncthreads();

  // ---- stage 1: velocity on [-L, +Q]^2 in place ----
  for (int i = tid; i < VW * VH; i += NTD) {
    const int ly = i / VW, lx = i - ly * VW;
    const float4 b = velocity_cell(u, sm.in.b.get(ly, lx), sm.in.b.z[ly][lx + 1], sm.in.b.z[ly + 1][lx], sm.in.w[ly][lx].y);
    sm.in.b.x[ly][lx] = b.x;
    sm.in.b.y[ly][lx] = b.y;
  }
  __syncthreads();

  // ---- stage 2: advection on [-1,0]^2, results in registers ----
  constexpr int RPT = TY / (NTD / TX);
  const int cx = tid & (TX - 1);
  float4 breg[RPT], wreg[RPT];
  char4 wlreg[RPT];
  float4 eb = make_float4(0.f, 0.f, 0.f, 0.f);
  char4 ewl = make_char4(0, 0, 0, 0);
  const bool extra = tid < TX + TY + 1;
  const int ecx = (tid < TX) ? tid : -1;
  const int ecy = (tid < TX) ? -1 : tid - TX - 1;
  {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int cy = (tid / TX) + k * (NTD / TX);
      advect_dry_cell<WATER>(u, g, ctx, sm.in, in, WX_WRAPX(tx0 + cx), WX_WRAPY(ty0 + cy), cx + L, cy + L, breg[k], wreg[k], wlreg[k]);
    }
    if (extra) {
      float4 w;
      advect_dry_cell<WATER>(u, g, ctx, sm.in, in, WX_WRAPX(tx0 + ecx), WX_WRAPY(ty0 + ecy), ecx + L, ecy + L, eb, w, ewl);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NTD / TX);
    sm.out.vx[cy + 1][cx + 1] = breg[k].x;
    sm.out.vy[cy + 1][cx + 1] = breg[k].y;
    sm.out.T[cy + 1][cx + 1] = breg[k].w;
    sm.out.w[cy + 1][cx + 1] = wlreg[k];
  }
  if (extra) {
    sm.out.vx[ecy + 1][ecx + 1] = eb.x;
    sm.out.vy[ecy + 1][ecx + 1] = eb.y;
    sm.out.T[ecy + 1][ecx + 1] = eb.w;
    sm.out.w[ecy + 1][ecx + 1] = ewl;
  }
  __syncthreads();

  // ---- stage 3: pressure on the tile ----
  const int x = tx0 + cx;
  if (x >= X) return;
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NTD / TX);
    const int y = ty0 + cy;
    if (y >= Y) break;
    const size_t gi = fidx(x, y, X);
    const float4 b = breg[k];
    const char4 wd = sm.out.w[cy][cx + 1];
    out.base[gi] = pressure_cell(b, sm.out.vx[cy + 1][cx], sm.out.vy[cy][cx + 1], sm.out.T[cy][cx + 1], wd.x, wd.y);
    if (WRITE_DISP) out.base_disp[gi] = b;
    out.wall[gi] = wlreg[k];
    if (WATER) out.water[gi] = wreg[k];
  }
#undef WX_WRAPX
#undef WX_WRAPY
}

inline void launch_fused_dry(const Geo &g, const Uni &u, const FullCtx *ctx, const DryIn &in, const DryOut &out, bool water, bool write_disp, hipStream_t stream)
{
  const dim3 grid = tile_grid(g.X, g.Y);
  if (water) {
    if (write_disp)
      hipLaunchKernelGGL((k_fused_dry<true, true>), grid, dim3(NTD), 0, stream, g, u, ctx, in, out);
    else
      hipLaunchKernelGGL((k_fused_dry<true, false>), grid, dim3(NTD), 0, stream, g, u, ctx, in, out);
  } else {
    if (write_disp)
      hipLaunchKernelGGL((k_fused_dry<false, true>), grid, dim3(NTD), 0, stream, g, u, ctx, in, out);
    else
      hipLaunchKernelGGL((k_fused_dry<false, false>), grid, dim3(NTD), 0, stream, g, u, ctx, in, out);
  }
}

} // namespace wx
