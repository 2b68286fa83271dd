// wx_full.h -- the WHOLE iteration (reference draws 1-7, app.js:5832-5930) as ONE LDS-tiled kernel (gfx950).
//
// velocity -> curl -> vorticity -> boundary -> advection -> pressure -> lighting for a TX x TY tile, with every
// intermediate texture (velocity output, curl, vortForce, post-boundary base/water/wall, advection output)
// living only in LDS. HBM traffic per cell-iteration: read base 16 + wall 4 + water 16 + light_0 16 +
// light_src 16, write base 16 + wall 4 + water 16 + light 16  = 68 R + 52 W = 120 B (vs 196 B for the two-kernel
// form and ~380 B for the reference's pass structure); optional outputs (post-advection base, post-boundary
// water, curl) are written only when a consumer can see them (last iteration of a wx_step call / particles).
//
// Stencil cone (tile = [0,TX) x [0,TY)):
//   pressure/lighting on the tile  <- advection on [-1,0]  <- boundary output on [-3,+2]
//   boundary <- velocity on cross+-1 = [-4,+3], vortForce on [-4,+2] <- curl on [-5,+3] <- velocity on [-5,+4]
//   <- base_0 on [-5,+5] (P at +1), wall_0 on [-5,+4].
// The per-cell arithmetic is the shared wx_cells.h code: results are bit-identical to the other kernel sets.
//
// A tile may not overwrite what neighbouring tiles still read as halo, so the kernel reads one buffer set and
// writes another; the host swaps the pointers after every launch (wxsim.hip: iterate_full).
#pragma once
#include "wx_cells.h"
#include "wx_fused.h"

namespace wx {

#ifndef WX_NTF
#define WX_NTF 512
#endif
#ifndef WX_F_MINWAVES
#define WX_F_MINWAVES 4
#endif
constexpr int NTF = WX_NTF;

namespace ff {
// a region [lo,+hi] spans cells lo .. T-1+hi: T - lo + hi cells per axis. With Q = fb_::REACH (2 below):
constexpr int Q = fb_::REACH;
constexpr int BL = 1 + Q;                           // boundary output on [-(1+Q), +Q]   ([-3,+2])  == advection input tile
constexpr int VL = 2 + Q;                           // vortForce on [-(2+Q), +Q]         ([-4,+2])
constexpr int R = 3 + Q;                            // curl on [-(3+Q), +(Q+1)]          ([-5,+3]); velocity / wall_0 on
                                                    // [-(3+Q), +(Q+2)] ([-5,+4]); base_0 on [-(3+Q), +(Q+3)] ([-5,+5])
constexpr int B0W = TX + 2 * Q + 6, B0H = TY + 2 * Q + 6;
constexpr int W0W = TX + 2 * Q + 5, W0H = TY + 2 * Q + 5;
constexpr int CW = TX + 2 * Q + 4, CH = TY + 2 * Q + 4;
constexpr int VW = TX + 2 * Q + 2, VH = TY + 2 * Q + 2;
constexpr int BW = TX + 2 * Q + 1, BH = TY + 2 * Q + 1;
constexpr int NB = (BW * BH + NTF - 1) / NTF;       // boundary cells per thread
constexpr int RPT = TY / (NTF / TX);                // tile rows per thread
static_assert(BW == fb_::IW && BH == fb_::IH, "advection input tile");
struct Phase1 {
  Planes4<B0H, B0W> b;
  char4 w[W0H][W0W + 1];
  float c[CH][CW];
  float vx[VH][VW], vy[VH][VW];
};
} // namespace ff

struct FullIn {
  const float4 *base;   // base_0: post-pressure state of the previous iteration
  const char4 *wall;    // wall_0
  const float4 *water;  // water_1: post-advection water of the previous iteration
  const float4 *light0; // lightTexture_0 (what boundaryShader samples)
  const float4 *light_src;
  const float4 *fb;     // precipitation feedback / deposition, or NULL when known to be zero
  const float2 *dep;
};
struct FullOut {
  float4 *base;      // post-pressure
  char4 *wall;
  float4 *water;     // post-advection
  float4 *light;
  float4 *base_disp; // optional: post-advection base (baseTexture_1)
  float4 *water0;    // optional: post-boundary water (waterTexture_0)
  float *curl;       // optional
};

// ---- boundary accessor on the phase-1 tiles; (cx,cy) in [-BL, TX+Q) x [-BL, TY+Q) ----
struct FBoundaryAcc {
  const ff::Phase1 &sm;
  const FullIn &in;
  float4 w00;
  int X, Y, x, y, cx, cy;
  __device__ __forceinline__ float4 base(int dx, int dy) const { return sm.b.get(cy + ff::R + dy, cx + ff::R + dx); }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return sm.w[cy + ff::R + dy][cx + ff::R + dx]; }
  __device__ __forceinline__ float2 vort(int dx, int dy) const { return make_float2(sm.vx[cy + ff::VL + dy][cx + ff::VL + dx], sm.vy[cy + ff::VL + dy][cx + ff::VL + dx]); }
  __device__ __forceinline__ float4 water(int dx, int dy) const
  {
    if (dx == 0 && dy == 0) return w00;
    return in.water[fidx(wrapi(x + dx, X), wrapi(y + dy, Y), X)];
  }
  __device__ __forceinline__ float4 light(int dy) const
  {
    int yy = y + dy;
    yy = yy < 0 ? 0 : (yy > Y - 1 ? Y - 1 : yy);
    return in.light0[fidx(x, yy, X)];
  }
  __device__ __forceinline__ float light_y0() const { return light(0).y; }
  __device__ __forceinline__ float light_x0() const { return light(0).x; }
  __device__ __forceinline__ float2 light_xy_up() const { const float4 l = light(1); return make_float2(l.x, l.y); }
  __device__ __forceinline__ bool has_fb() const { return in.fb != nullptr; }
  __device__ __forceinline__ float4 fb() const { return in.fb ? in.fb[fidx(x, y, X)] : make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ float2 dep() const { return in.dep ? in.dep[fidx(x, y, X)] : make_float2(0.f, 0.f); }
};

// ---- exact out-of-tile fallback: the post-boundary texel of an ARBITRARY cell recomputed from global memory
//      (velocity, curl and vortForce evaluated on the fly). Only reached by cells whose back-trace is longer than
//      VMAX cells per iteration; out of line to keep the main kernel small. ----
struct FullCtx { // static per wx_set_params; the per-launch items (buffer pointers, iterNum) travel as arguments
  Geo g;
  Uni u;
  const float *initial_T, *snd_T, *snd_W, *snd_Vel;
};
struct SlowArgs {
  const FullCtx *ctx;
  FullIn in;
  float iterNum;
};

struct GRecomputeAcc {
  const Uni &u_;
  const FullIn &in_;
  int X, Y, x, y;
  __device__ __forceinline__ int wx_(int dx) const { return wrapmod(x + dx, X); }
  __device__ __forceinline__ int wy_(int dy) const { return wrapmod(y + dy, Y); }
  __device__ __forceinline__ float4 vel_at(int xx, int yy) const
  { // velocity pass output at (xx,yy) (already wrapped)
    const int xr = xx + 1 == X ? 0 : xx + 1, yu = yy + 1 == Y ? 0 : yy + 1;
    return velocity_cell(u_, in_.base[fidx(xx, yy, X)], in_.base[fidx(xr, yy, X)].z, in_.base[fidx(xx, yu, X)].z, in_.wall[fidx(xx, yy, X)].y);
  }
  __device__ __forceinline__ float curl_at(int xx, int yy) const
  {
    const int xr = xx + 1 == X ? 0 : xx + 1, yu = yy + 1 == Y ? 0 : yy + 1;
    const float4 v = vel_at(xx, yy);
    return curl_cell(v.x, v.y, vel_at(xr, yy).y, vel_at(xx, yu).x);
  }
  __device__ __forceinline__ float4 base(int dx, int dy) const { return vel_at(wx_(dx), wy_(dy)); }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return in_.wall[fidx(wx_(dx), wy_(dy), X)]; }
  __device__ __forceinline__ float4 water(int dx, int dy) const { return in_.water[fidx(wx_(dx), wy_(dy), X)]; }
  __device__ __forceinline__ float2 vort(int dx, int dy) const
  {
    const int xx = wx_(dx), yy = wy_(dy);
    const int xl = xx == 0 ? X - 1 : xx - 1, xr = xx + 1 == X ? 0 : xx + 1, yd = yy == 0 ? Y - 1 : yy - 1, yu = yy + 1 == Y ? 0 : yy + 1;
    return vorticity_cell(curl_at(xx, yy), curl_at(xl, yy), curl_at(xr, yy), curl_at(xx, yd), curl_at(xx, yu));
  }
  __device__ __forceinline__ float4 light(int dy) const
  {
    int yy = y + dy;
    yy = yy < 0 ? 0 : (yy > Y - 1 ? Y - 1 : yy);
    return in_.light0[fidx(x, yy, X)];
  }
  __device__ __forceinline__ float light_y0() const { return light(0).y; }
  __device__ __forceinline__ float light_x0() const { return light(0).x; }
  __device__ __forceinline__ float2 light_xy_up() const { const float4 l = light(1); return make_float2(l.x, l.y); }
  __device__ __forceinline__ bool has_fb() const { return in_.fb != nullptr; }
  __device__ __forceinline__ float4 fb() const { return in_.fb ? in_.fb[fidx(x, y, X)] : make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ float2 dep() const { return in_.dep ? in_.dep[fidx(x, y, X)] : make_float2(0.f, 0.f); }
};

struct BOut {
  float4 b, w;
  char4 wl;
};
__device__ __noinline__ BOut boundary_texel_global(const SlowArgs *__restrict__ sa, int x, int y)
{
  const FullCtx *c = sa->ctx;
  Uni u = c->u;
  u.iterNum = sa->iterNum;
  u.iterI = (int)sa->iterNum;
  GRecomputeAcc a{u, sa->in, c->g.X, c->g.Y, x, y};
  BOut o;
  boundary_cell(u, u.iterNum, u.iterI, c->g, c->initial_T, x, y, a, o.b, o.w, o.wl);
  return o;
}

struct GAdvectAccFull {
  const SlowArgs *sa;
  int X, Y, x, y;
  __device__ __forceinline__ BOut at(int dx, int dy) const { return boundary_texel_global(sa, wrapmod(x + dx, X), wrapmod(y + dy, Y)); }
  __device__ __forceinline__ float4 base(int dx, int dy) const { return at(dx, dy).b; }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return at(dx, dy).wl; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return at(dx, dy).b; }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const { return at(dx, dy).w; }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return at(dx, dy).wl; }
};
__device__ __noinline__ AdvOut advection_cell_recompute(const FullCtx *__restrict__ c, FullIn in, float iterNum, int x, int y)
{
  const SlowArgs sa{c, in, iterNum};
  GAdvectAccFull a{&sa, c->g.X, c->g.Y, x, y};
  AdvOut o;
  advection_cell(c->u, c->g, c->initial_T, c->snd_T, c->snd_W, c->snd_Vel, x, y, a, o.b, o.w, o.wl);
  return o;
}

__device__ __forceinline__ void advect_full_cell(const Uni &u, const Geo &g, const float *initial_T, const float *snd_T, const float *snd_W,
                                                 const float *snd_Vel, const fb_::SmemIn &sm, const FullCtx *ctx, const FullIn &in, int x, int y,
                                                 int lx, int ly, float4 &b, float4 &w, char4 &wl)
{
  using namespace fb_;
  const float m = fmaxf(fmaxf(fmaxf(fabsf(sm.b.x[ly][lx]), fabsf(sm.b.x[ly][lx - 1])), fmaxf(fabsf(sm.b.x[ly + 1][lx]), fabsf(sm.b.x[ly + 1][lx - 1]))),
                        fmaxf(fmaxf(fabsf(sm.b.y[ly][lx]), fabsf(sm.b.y[ly - 1][lx])), fmaxf(fabsf(sm.b.y[ly][lx + 1]), fabsf(sm.b.y[ly - 1][lx + 1]))));
  if (m < VMAX) {
    LAdvectAcc a{sm, lx, ly};
    advection_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, x, y, a, b, w, wl);
  } else {
    const AdvOut o = advection_cell_recompute(ctx, in, u.iterNum, x, y);
    b = o.b;
    w = o.w;
    wl = o.wl;
  }
}

// Uniforms and geometry are NOT kernel arguments here: 60+ scalar uniforms kept live across seven stages overflow
// the SGPR file (the first version spilled ~1600 SGPRs through v_readlane). They live in device memory (FullCtx)
// and every stage re-reads the few it needs after its barrier.
#define WX_STAGE_UNI()                 \
  Uni u = ctx->u;                      \
  u.iterNum = iterNum;                 \
  u.iterI = (int)iterNum;

template <bool OPT_OUT>
__global__ __launch_bounds__(NTF, WX_F_MINWAVES) void k_fused_full(const FullCtx *ctx, float iterNum, FullIn in, FullOut out)
{
  using namespace ff;
  __shared__ union {
    Phase1 p1;        // base_0 / wall_0 tiles, curl, vortForce
    fb_::SmemIn p2;   // post-boundary base / water / wall on [-3,+2]
    fb_::SmemOut p3;  // advection output on [-1,0]
  } sm;
  const int X = ctx->g.X, Y = ctx->g.Y;
  const int tid = threadIdx.x;
  int tbx, tby;
  tile_of_block(tiles_x(X), tbx, tby);
  const int tx0 = tbx * TX, ty0 = tby * TY;
  const bool small = (X < TX + 16) || (Y < TY + 16);
#define WX_WRAPX(v) (small ? wrapmod((v), X) : wrapfast((v), X))
#define WX_WRAPY(v) (small ? wrapmod((v), Y) : wrapfast((v), Y))

  // ---- stage 0: base_0 on [-5,+5], wall_0 on [-5,+4] ----
  for (int i = tid; i < B0W * B0H; i += NTF) {
    const int ly = i / B0W, lx = i - ly * B0W;
    sm.p1.b.put(ly, lx, in.base[fidx(WX_WRAPX(tx0 + lx - R), WX_WRAPY(ty0 + ly - R), X)]);
  }
  for (int i = tid; i < W0W * W0H; i += NTF) {
    const int ly = i / W0W, lx = i - ly * W0W;
    sm.p1.w[ly][lx] = in.wall[fidx(WX_WRAPX(tx0 + lx - R), WX_WRAPY(ty0 + ly - R), X)];
  }
  // own-cell inputs of the boundary stage: issue the loads now, consume after stage 3
  float4 bw00[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const int i = tid + k * NTF;
    if (i < BW * BH) {
      const int ry = i / BW, rx = i - ry * BW;
      bw00[k] = in.water[fidx(WX_WRAPX(tx0 + rx - BL), WX_WRAPY(ty0 + ry - BL), X)];
    }
  }
  __syncthreads();

  // ---- stage 1: velocity on [-5,+4]^2 in place ----
  {
    Uni u;
    u.vel_keep = ctx->u.vel_keep;
    u.wind_add = ctx->u.wind_add;
    for (int i = tid; i < W0W * W0H; i += NTF) {
      const int ly = i / W0W, lx = i - ly * W0W;
      const float4 b = velocity_cell(u, sm.p1.b.get(ly, lx), sm.p1.b.z[ly][lx + 1], sm.p1.b.z[ly + 1][lx], sm.p1.w[ly][lx].y);
      sm.p1.b.x[ly][lx] = b.x;
      sm.p1.b.y[ly][lx] = b.y;
    }
  }
  __syncthreads();

  // ---- stage 2: curl on [-5,+3]^2 ----
  for (int i = tid; i < CW * CH; i += NTF) {
    const int ly = i / CW, lx = i - ly * CW;
    sm.p1.c[ly][lx] = curl_cell(sm.p1.b.x[ly][lx], sm.p1.b.y[ly][lx], sm.p1.b.y[ly][lx + 1], sm.p1.b.x[ly + 1][lx]);
  }
  __syncthreads();

  // ---- stage 3: vortForce on [-4,+2]^2 ----
  for (int i = tid; i < VW * VH; i += NTF) {
    const int ly = i / VW, lx = i - ly * VW;
    const float2 v = vorticity_cell(sm.p1.c[ly + 1][lx + 1], sm.p1.c[ly + 1][lx], sm.p1.c[ly + 1][lx + 2], sm.p1.c[ly][lx + 1], sm.p1.c[ly + 2][lx + 1]);
    sm.p1.vx[ly][lx] = v.x;
    sm.p1.vy[ly][lx] = v.y;
  }
  __syncthreads();

  // ---- stage 4: boundary on [-3,+2]^2, results in registers (the phase-1 tiles are still being read) ----
  float4 bb[NB], bq[NB];
  char4 bwl[NB];
  {
    WX_STAGE_UNI();
    const Geo g = ctx->g;
    const float *initial_T = ctx->initial_T;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      const int i = tid + k * NTF;
      if (i < BW * BH) {
        const int ry = i / BW, rx = i - ry * BW;
        const int cx = rx - BL, cy = ry - BL;
        const int x = WX_WRAPX(tx0 + cx), y = WX_WRAPY(ty0 + cy);
        FBoundaryAcc a{sm.p1, in, bw00[k], X, Y, x, y, cx, cy};
        boundary_cell(u, u.iterNum, u.iterI, g, initial_T, x, y, a, bb[k], bq[k], bwl[k]);
        if (OPT_OUT) {
          if (cx >= 0 && cx < TX && cy >= 0 && cy < TY && tx0 + cx < X && ty0 + cy < Y) {
            const size_t gi = fidx(tx0 + cx, ty0 + cy, X);
            out.water0[gi] = bq[k];
            out.curl[gi] = sm.p1.c[cy + R][cx + R];
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const int i = tid + k * NTF;
    if (i < BW * BH) {
      const int ry = i / BW, rx = i - ry * BW;
      sm.p2.b.put(ry, rx, bb[k]);
      sm.p2.q.put(ry, rx, bq[k]);
      sm.p2.w[ry][rx] = bwl[k];
    }
  }
  __syncthreads();

  // ---- stage 5: advection on [-1,0]^2, results in registers ----
  const int cx = tid & (TX - 1);
  float4 breg[RPT], wreg[RPT];
  char4 wlreg[RPT];
  float4 eb = make_float4(0.f, 0.f, 0.f, 0.f);
  char4 ewl = make_char4(0, 0, 0, 0);
  const bool extra = tid < TX + TY + 1;
  const int ecx = (tid < TX) ? tid : -1;
  const int ecy = (tid < TX) ? -1 : tid - TX - 1;
  {
    WX_STAGE_UNI();
    const Geo g = ctx->g;
    const float *initial_T = ctx->initial_T, *snd_T = ctx->snd_T, *snd_W = ctx->snd_W, *snd_Vel = ctx->snd_Vel;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int cy = (tid / TX) + k * (NTF / TX);
      advect_full_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, sm.p2, ctx, in, WX_WRAPX(tx0 + cx), WX_WRAPY(ty0 + cy), cx + fb_::HL, cy + fb_::HD,
                       breg[k], wreg[k], wlreg[k]);
    }
    if (extra) {
      float4 w;
      advect_full_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, sm.p2, ctx, in, WX_WRAPX(tx0 + ecx), WX_WRAPY(ty0 + ecy), ecx + fb_::HL, ecy + fb_::HD,
                       eb, w, ewl);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NTF / TX);
    sm.p3.vx[cy + 1][cx + 1] = breg[k].x;
    sm.p3.vy[cy + 1][cx + 1] = breg[k].y;
    sm.p3.T[cy + 1][cx + 1] = breg[k].w;
    sm.p3.w[cy + 1][cx + 1] = wlreg[k];
  }
  if (extra) {
    sm.p3.vx[ecy + 1][ecx + 1] = eb.x;
    sm.p3.vy[ecy + 1][ecx + 1] = eb.y;
    sm.p3.T[ecy + 1][ecx + 1] = eb.w;
    sm.p3.w[ecy + 1][ecx + 1] = ewl;
  }
  __syncthreads();

  // ---- stage 6: pressure + lighting on the tile ----
  const int x = tx0 + cx;
  if (x >= X) return;
  {
    WX_STAGE_UNI();
    const Geo g = ctx->g;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int cy = (tid / TX) + k * (NTF / TX);
      const int y = ty0 + cy;
      if (y >= Y) break;
      const size_t gi = fidx(x, y, X);
      const float4 b = breg[k];
      const char4 wd = sm.p3.w[cy][cx + 1];
      out.base[gi] = pressure_cell(b, sm.p3.vx[cy + 1][cx], sm.p3.vy[cy][cx + 1], sm.p3.T[cy][cx + 1], wd.x, wd.y);
      if (OPT_OUT) out.base_disp[gi] = b;
      out.water[gi] = wreg[k];
      out.wall[gi] = wlreg[k];
      LLightAcc la{sm.p3, in.light_src, wreg[k], wlreg[k], b.w, X, x, cx, cy};
      out.light[gi] = lighting_cell(u, g, x, y, la);
    }
  }
#undef WX_WRAPX
#undef WX_WRAPY
}

inline void launch_fused_full(const Geo &g, float iterNum, const FullIn &in, const FullCtx *ctx, const FullOut &out, bool opt_out,
                              hipStream_t stream)
{
  const dim3 grid = tile_grid(g.X, g.Y);
  if (opt_out)
    hipLaunchKernelGGL(k_fused_full<true>, grid, dim3(NTF), 0, stream, ctx, iterNum, in, out);
  else
    hipLaunchKernelGGL(k_fused_full<false>, grid, dim3(NTF), 0, stream, ctx, iterNum, in, out);
}

} // namespace wx
