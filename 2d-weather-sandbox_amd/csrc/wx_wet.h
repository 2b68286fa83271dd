// wx_wet.h -- the WHOLE wet iteration (reference draws 1-7, app.js:5832-5930: velocity -> curl -> vorticity -> boundary ->
// advection -> pressure -> lighting) as ONE row-MARCHING kernel (gfx950). One wavefront owns a 64-column strip (a workgroup
// = four neighbouring strips, no barrier between them) and walks up the rows of its segment; per step (newest input row r) it
//
//   loads row r+1 (prefetch)  |  velocity(r-1)  |  curl(r-2)  |  vortForce(r-3)  |  boundary(r-3)  |  advection(r-4)  |
//   pressure(r-4) + lighting(r-4) -> stores row r-4
//
// * every intermediate texture of the reference (velocity output, curl, vortForce, post-boundary base / water / wall,
//   advection output) lives in registers or in a wave-PRIVATE LDS ring of the last six rows; nothing but the iteration's
//   inputs and outputs touches HBM: base 16 R + 16 W, water 16 R + 16 W, wall 4 R + 4 W, light 16 R + 16 W = 104 B/cell
//   (round 1's two-kernel form: 184 B/cell; reference pass structure: ~380 B/cell);
// * vertical neighbours are the wave's own earlier rows (ring or carried registers), horizontal neighbours of values that
//   were just computed come from whole-wave DPP shifts (v_mov_b32_dpp wave_shr / wave_shl: one VALU instruction, no LDS
//   round trip), data-dependent taps (advection's back-trace, the sun ray) from the ring;
// * the boundary stage overwrites its input rows IN PLACE (the only later readers of the pre-boundary values are the same
//   lane one row up: carried in registers), so six ring rows serve both the pre- and the post-boundary state;
// * no vertical halo is re-loaded or re-computed, there are no workgroup barriers (the wave's LDS queue is in order;
//   wavefront-scope fences keep the compiler from reordering across stages); 56 of the 64 lanes produce output
//   (cone of the seven chained stencils: 4 columns per side for |v| < 0.9 cell/iteration);
// * cells whose back-trace is longer than that (|v| >= 0.9 cell/iteration: the core of a strong updraft) are only RECORDED by the
//   marching loop: the (up to three) OUTPUT cells each of them feeds -- its own, the right neighbour's pressure, the upper
//   neighbour's pressure / lighting -- are appended to a device list, and a second small kernel (k_wet_fix, one wavefront per list
//   entry, spread over the whole chip) recomputes them exactly and overwrites the placeholders: 64 lanes rebuild an 8 x 8 patch
//   of post-boundary texels around the cell from the iteration's inputs, three lanes advect from that patch. The hot loop contains
//   no call and no scratch access, and a storm core costs microseconds (round 2: the recording wave redid such cells serially from
//   global memory after its loop, ~1 ms PER CELL -- one fast cell tripled the iteration time);
// * global loads are issued one step ahead of their use (software prefetch), each with the lag its consumer has (base / wall
//   row r+1, water and light row r-1, feedback row r-2), and the stores of a row are issued at the top of the NEXT step, right
//   behind the prefetch: the only vmcnt wait of the common path (top of the step) then covers operations that have had a
//   whole step to complete -- gfx9 counts loads and stores in one in-order counter.
// * launch shape (wet_launch_shape): the rows are cut into eight bands, one per XCD, each band into a few full segments followed
//   by segments of 1/2, 1/4 (and 1/8) the height -- short segments at the END of the dispatch order shorten the drain phase of
//   the launch; grids lower than 512 rows use column blocks per XCD instead.
// Same per-cell arithmetic as the per-pass kernels (wx_cells.h): bit-identical results.
#pragma once
#include "wx_cells.h"
#include "wx_tile.h"
#include <cstddef>
#include <type_traits>
#include <cstdlib>
#include <algorithm>

namespace wx {

#ifndef WX_WET_MINWAVES
#define WX_WET_MINWAVES 4
#endif
#ifndef WX_WET_UNI_COPY
#define WX_WET_UNI_COPY 0
#endif
#ifndef WX_WET_WPB
#define WX_WET_WPB 4 // wavefronts per workgroup: independent strips (no barrier between them); the dispatcher spreads the waves of
                     // ONE workgroup evenly over the four SIMDs of a CU, which it does not guarantee for single-wave workgroups
#endif
#ifndef WX_ABL_FORCE_AIR
#define WX_ABL_FORCE_AIR 0 // (instruction-budget builds: every row takes the free-air instantiations; wrong near terrain)
#endif
#ifndef WX_WET_ARGS_MEM
#define WX_WET_ARGS_MEM 1
#endif
#ifndef WX_WET_BANDS
#define WX_WET_BANDS 1
#endif
#ifndef WX_WET_FB_COND
#define WX_WET_FB_COND 0 // 1: load feedback rows only where a tile holds feedback (measured: 1.012-1.021 vs 0.974-0.977 ms with the always-issued loads from a row of zeros)
#endif
#ifndef WX_WET_TAIL
#define WX_WET_TAIL 1
#endif
#ifndef WX_WET_ALPHA_DEFAULT
#define WX_WET_ALPHA_DEFAULT 1.0 // cost of a row below air_from_row relative to a free-air row when the segment borders are placed
#endif
#ifndef WX_WET_PRIO_ROTATE
#define WX_WET_PRIO_ROTATE 0
#endif
#ifndef WX_WET_SKIP_LOADS
#define WX_WET_SKIP_LOADS 1 // 1: no water / light loads in the first two warm-up steps; light_0.x only near walls. 2: only the latter. 0: neither
#endif
#ifndef WX_WET_NT_STORES
#define WX_WET_NT_STORES 0
#endif
#ifndef WX_WET_ZW0
#define WX_WET_ZW0 1 // wave-uniform skip of the precipitation-visual / smoke interpolations where those channels are zero (advection_cell NO_ZW)
#endif
#ifndef WX_WET_DEP_NEAR
#define WX_WET_DEP_NEAR 1 // deposition rows are only loaded where a surface wall cell can read them
#endif
#ifndef WX_WET_PRIO_MEM
#define WX_WET_PRIO_MEM 0 // s_setprio level while a step issues its prefetch and its deferred stores (0: none)
#endif
#ifndef WX_WET_UNROLL2
#define WX_WET_UNROLL2 1 // two row steps per loop iteration (measured -1.2 .. -1.6 % at 16384x2048: fewer register moves for the carried values)
#endif
#ifndef WX_WET_AIR
#define WX_WET_AIR 1 // wave-uniform free-air instantiations of the boundary / advection / lighting stages
#endif

struct FullCtx { // static per wx_set_params; the per-launch items (buffer pointers, iterNum) travel as arguments
  Geo g;
  Uni u;
  const float *initial_T, *snd_T, *snd_W, *snd_Vel;
};

struct WetIn {
  const float4 *base;   // base_0: post-pressure state of the previous iteration
  const char4 *wall;    // wall_0
  const float4 *water;  // water_1: post-advection water of the previous iteration
  LightPlanesC l0;      // lightTexture_0 (what boundaryShader samples)
  LightPlanesC lsrc;    // source of this iteration's lighting pass (light_0 or light_1)
  const float3 *fb;     // precipitation feedback (three channels) / deposition, or NULL when known to be zero
  const float2 *dep;
  const unsigned char *fb_zero; // per 64x16 tile t: [2t] feedback all zero there, [2t + 1] deposition all zero there (may be NULL)
  const float4 *zero_row;       // one row (X texels) of zeros: what rows of all-zero tiles are "loaded" from
  int fb_txn;
};
struct WetOut {
  float4 *base;      // post-pressure
  char4 *wall;
  float4 *water;     // post-advection
  LightPlanes light;
  float *p_disp;     // optional (OPT_OUT): post-advection PRESSURE -- all of baseTexture_1 (what the display samples) that the post-pressure base texture and
                     // t_disp do not already hold: pressure_cell changes P everywhere and T directly above land only (k_base_disp_assemble makes the RGBA texels on demand)
  float4 *water0;    // optional (OPT_OUT): post-boundary water (waterTexture_0); NULL = not stored (made on demand)
  float *curl;       // optional (OPT_OUT)
  float *t_disp;     // optional (runtime): post-advection temperature where the pressure pass changes it, for the droplets and for baseTexture_1 on demand
#ifdef WX_WET_TIMING
  unsigned long long *cycles; // (tuning builds) per wave: s_memtime at start / end
#endif
};

// Output cells left to the exact path (see k_wet_fix): appended by the marching kernel, consumed by the fix kernel of the same
// launch group. count keeps counting past cap (the host reports the overflow as an error: the state has blown up).
struct WetFixList {
  int *count;
  int2 *cells;
  int cap;
  const int *hint_host; // (host address of the same word)
  int *fastest; // bit pattern of the largest |velocity component| [cells / iteration] that sent a cell to this list (wx_fastest_velocity)
  int *hint; // host-visible word (pinned, mapped): the fix pass leaves the length of the list it consumed here; the host looks at it
             // (stale, never waited for) to size the NEXT fix launches -- a launch over the whole chip costs 10 us to find an empty list
};

// ---- exact out-of-line path: the post-boundary texel of an ARBITRARY cell recomputed from global memory (velocity, curl
//      and vortForce evaluated on the fly). Only used for output cells fed by a back-trace longer than 0.9 cells. ----
struct WetSlowArgs {
  const FullCtx *ctx;
  WetIn in;
  float iterNum;
};
struct GWetRecomputeAcc {
  const Uni &u_;
  const WetIn &in_;
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
  __device__ __forceinline__ float light_y0() const { return in_.l0.y[fidx(x, y, X)]; }
  __device__ __forceinline__ float light_x0() const { return in_.l0.x[fidx(x, y, X)]; }
  __device__ __forceinline__ float2 light_xy_up() const
  {
    const size_t i = fidx(x, y + 1 > Y - 1 ? Y - 1 : y + 1, X); // light textures: CLAMP_TO_EDGE in T
    return make_float2(in_.l0.x[i], in_.l0.y[i]);
  }
  __device__ __forceinline__ bool has_fb() const { return in_.fb != nullptr; }
  __device__ __forceinline__ float4 fb() const
  {
    if (!in_.fb) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float3 v = in_.fb[fidx(x, y, X)];
    return make_float4(v.x, v.y, v.z, 0.f);
  }
  __device__ __forceinline__ float2 dep() const { return in_.dep ? in_.dep[fidx(x, y, X)] : make_float2(0.f, 0.f); }
};
struct BOut {
  float4 b, w;
  char4 wl;
};
__device__ __noinline__ BOut wet_boundary_texel_global(const WetSlowArgs *__restrict__ sa, int x, int y)
{
  const FullCtx *c = sa->ctx;
  GWetRecomputeAcc a{c->u, sa->in, c->g.X, c->g.Y, x, y};
  BOut o;
  boundary_cell(c->u, sa->iterNum, (int)sa->iterNum, c->g, c->initial_T, x, y, a, o.b, o.w, o.wl);
  return o;
}
struct GWetAdvectAcc {
  const WetSlowArgs *sa;
  int X, Y, x, y;
  __device__ __forceinline__ BOut at(int dx, int dy) const { return wet_boundary_texel_global(sa, wrapmod(x + dx, X), wrapmod(y + dy, Y)); }
  __device__ __forceinline__ float4 base(int dx, int dy) const { return at(dx, dy).b; }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return at(dx, dy).wl; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return at(dx, dy).b; }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const { return at(dx, dy).w; }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return at(dx, dy).wl; }
};
__device__ __noinline__ AdvOut wet_advection_cell_recompute(const WetSlowArgs *__restrict__ sa, int x, int y)
{
  const FullCtx *c = sa->ctx;
  GWetAdvectAcc a{sa, c->g.X, c->g.Y, x, y};
  AdvOut o;
  advection_cell(c->u, c->g, c->initial_T, c->snd_T, c->snd_W, c->snd_Vel, x, y, a, o.b, o.w, o.wl);
  return o;
}
struct GWetLightAcc {
  LightPlanesC l;
  float4 water_;
  char4 wall_;
  float T0, Td;
  int X, x;
  __device__ __forceinline__ float T(int dy) const { return dy == 0 ? T0 : Td; }
  __device__ __forceinline__ float4 water() const { return water_; }
  __device__ __forceinline__ char4 wall() const { return wall_; }
  __device__ __forceinline__ float sun_at(int dx, int j) const { return l.x[fidx(wrapmod(x + dx, X), j, X)]; }
  __device__ __forceinline__ float ir_down_at(int j) const { return l.zw[fidx(x, j, X)].x; }
  __device__ __forceinline__ float ir_up_at(int j) const { return l.zw[fidx(x, j, X)].y; }
};
// every output of the iteration at cell (x, y) -- post-pressure base, post-advection water / wall, light -- from the
// iteration's inputs alone, any back-trace length
__device__ __noinline__ void wet_output_cell_exact(const FullCtx *__restrict__ c, const WetIn *__restrict__ in, const WetOut *__restrict__ out, float iterNum,
                                                   int opt_out, int x, int y)
{
  const WetSlowArgs sa{c, *in, iterNum};
  const int X = c->g.X, Y = c->g.Y;
  const AdvOut o0 = wet_advection_cell_recompute(&sa, x, y);
  const AdvOut oL = wet_advection_cell_recompute(&sa, x == 0 ? X - 1 : x - 1, y);
  const AdvOut oD = wet_advection_cell_recompute(&sa, x, y == 0 ? Y - 1 : y - 1);
  const size_t gi = fidx(x, y, X);
  out->base[gi] = pressure_cell(o0.b, oL.b.x, oD.b.y, oD.b.w, oD.wl.x, oD.wl.y);
  out->water[gi] = o0.w;
  out->wall[gi] = o0.wl;
  GWetLightAcc la{in->lsrc, o0.w, o0.wl, o0.b.w, oD.b.w, X, x};
  const float4 l = lighting_cell(c->u, c->g, x, y, la);
  out->light.x[gi] = l.x;
  out->light.y[gi] = l.y;
  out->light.zw[gi] = make_float2(l.z, l.w);
  if (opt_out) out->p_disp[gi] = o0.b.z;
  if (out->t_disp) out->t_disp[gi] = o0.b.w;
}

// ---- the wave-private ring ----
constexpr int WOUT = 56;          // output columns per wave: lanes 4 .. 59
constexpr int WLO = 4;            // first output lane
constexpr int WPAD = 1;           // never-written pad entry on each side of a ring row (lane -1 / lane 64 reads land there)
constexpr int WRW = 64 + 2 * WPAD;
constexpr int WQ = 3;             // rows r-5 .. r-3 of the post-boundary base / wall / water planes: slot = row mod 3 (round 4: rows
                                  // r-2 .. r are carried in registers until the boundary stage needs their horizontal neighbours --
                                  // the ring is what bounds the occupancy: 50 -> 35 plane-rows = 3 -> 4 waves per SIMD)
constexpr int WL = 4;             // rows r-5 .. r-2 of the source light planes: slot = row & 3
struct WetRing {
  float P[WQ][WRW], T[WQ][WRW], vx[WQ][WRW], vy[WQ][WRW];
  char4 wl[WQ][WRW];
  float qx[WQ][WRW], qy[WQ][WRW], qz[WQ][WRW], qw[WQ][WRW];
  float lx[WL][WRW], lw[WL][WRW]; // sunlight and upward IR of the source light texture
};
static_assert(sizeof(WetRing) == (5 * WQ + 4 * WQ + 2 * WL) * WRW * 4, "ring layout");

// whole-wave shifts: lane i <- lane i-1 / lane i+1; the edge lane keeps its own value
__device__ __forceinline__ float wave_from_left(float v)
{
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_from_right(float v)
{
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
// orders the wave's LDS traffic across stages for the compiler; the hardware executes one wave's LDS operations in order
__device__ __forceinline__ void wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ char4 unpack_wall(int w)
{
  return make_char4((signed char)(w & 0xff), (signed char)((w >> 8) & 0xff), (signed char)((w >> 16) & 0xff), (signed char)(w >> 24));
}

// boundary stage, row yb: own texel, the row below and the water / light_0 texels of the row above in registers, the base / wall
// texels of the row above and the horizontal wall neighbours in the ring
struct MWBoundaryAcc {
  const WetRing &rg;
  int li, o0;              // ring column; in-plane offset of ring row yb (its pre-boundary T and wall: the horizontal neighbours)
  float vxU, TU;           // row above: velocity-pass vx, T (carried registers)
  char4 wU;
  float4 b00, q00, qU;     // velocity output / pre-boundary water of the own cell; pre-boundary water of the cell above
  char4 w00, wD;
  float vxD, TD, qzD, qwD; // row below: velocity-pass vx, pre-boundary T, soil moisture / snow of the pre-boundary water
  float qzL, qwL, qzR, qwR; // left / right neighbour's soil moisture / snow (fetched only in the iterations that use them)
  float2 vf;
  float vfLy, vfDx;
  float l0x, l0y, l0xU, l0yU; // light_0 sunlight / net heating at the own cell and the cell above (row clamped)
  float4 fb_;
  float2 dep_;
  bool hasfb;
  __device__ __forceinline__ float4 base(int dx, int dy) const
  {
    if (dx == 0 && dy == 0) return b00;
    if (dy == 1) return make_float4(vxU, 0.f, 0.f, TU);
    if (dy == -1) return make_float4(vxD, 0.f, 0.f, TD);
    return make_float4(0.f, 0.f, 0.f, (&rg.T[0][0])[o0 + li + dx]); // (+-1, 0): temperature of the neighbouring sea cell
  }
  __device__ __forceinline__ float4 water(int dx, int dy) const
  {
    if (dx == 0 && dy == 0) return q00;
    if (dx == 0 && dy == -1) return make_float4(0.f, 0.f, qzD, qwD);
    if (dx == 0 && dy == 1) return qU;
    return dx < 0 ? make_float4(0.f, 0.f, qzL, qwL) : make_float4(0.f, 0.f, qzR, qwR);
  }
  __device__ __forceinline__ char4 wall(int dx, int dy) const
  {
    if (dx == 0 && dy == 0) return w00;
    if (dy == -1) return wD;
    if (dy == 1) return wU;
    return (&rg.wl[0][0])[o0 + li + dx];
  }
  __device__ __forceinline__ float2 vort(int dx, int dy) const
  {
    if (dx == 0 && dy == 0) return vf;
    if (dx == -1) return make_float2(0.f, vfLy); // the boundary pass uses .y of the left and .x of the lower neighbour only
    return make_float2(vfDx, 0.f);
  }
  __device__ __forceinline__ float light_y0() const { return l0y; }
  __device__ __forceinline__ float light_x0() const { return l0x; }
  __device__ __forceinline__ float2 light_xy_up() const { return make_float2(l0xU, l0yU); }
  __device__ __forceinline__ bool has_fb() const { return hasfb; }
  __device__ __forceinline__ float4 fb() const { return fb_; }
  __device__ __forceinline__ float2 dep() const { return dep_; }
};

// advection stage, row ya: post-boundary rows ya-1 .. ya+1 in the ring
struct MWAdvAcc {
  const WetRing &rg;
  int li;
  int ob[3]; // in-plane offsets of rows ya-1, ya, ya+1 (wave-uniform; the base / wall and the water planes share their slots)
  __device__ __forceinline__ float4 base(int dx, int dy) const
  {
    const int o = ob[dy + 1] + li + dx;
    return make_float4((&rg.vx[0][0])[o], (&rg.vy[0][0])[o], (&rg.P[0][0])[o], (&rg.T[0][0])[o]);
  }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return (&rg.wl[0][0])[ob[dy + 1] + li + dx]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return base(dx, dy); }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return wall(dx, dy); }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const
  {
    const int o = ob[dy + 1] + li + dx;
    return make_float4((&rg.qx[0][0])[o], (&rg.qy[0][0])[o], (&rg.qz[0][0])[o], (&rg.qw[0][0])[o]);
  }
};
// bilinear footprint (dx0, dy0 in {-1, 0}): the two rows are picked per lane from the three wave-uniform row offsets
struct MWFp {
  const WetRing &rg;
  int b0, b1, q0, q1; // offsets (row offset + column) of the lower / upper footprint row in the base and water planes
  __device__ __forceinline__ float4 base(int i, int j) const
  {
    const int o = (j ? b1 : b0) + i;
    return make_float4((&rg.vx[0][0])[o], (&rg.vy[0][0])[o], (&rg.P[0][0])[o], (&rg.T[0][0])[o]);
  }
  __device__ __forceinline__ char4 wall(int i, int j) const { return (&rg.wl[0][0])[(j ? b1 : b0) + i]; }
  __device__ __forceinline__ float4 water(int i, int j) const
  {
    const int o = (j ? q1 : q0) + i;
    return make_float4((&rg.qx[0][0])[o], (&rg.qy[0][0])[o], (&rg.qz[0][0])[o], (&rg.qw[0][0])[o]);
  }
};
__device__ __forceinline__ MWFp make_fp(const MWAdvAcc &a, int dx0, int dy0)
{
  const bool lo = dy0 < 0; // dy0 is -1 or 0 for every cell that takes this path
  const int c = a.li + dx0;
  const int o0 = (lo ? a.ob[0] : a.ob[1]) + c, o1 = (lo ? a.ob[1] : a.ob[2]) + c;
  return MWFp{a.rg, o0, o1, o0, o1};
}

// lighting stage, row y (an OUTPUT row: unwrapped == wrapped row index); ring row k of the light planes holds texture row clamp(k)
struct MWLightAcc {
  const WetRing &rg;
  int li;
  float T0, Tdown, lz_up;
  float4 water_;
  char4 wall_;
  __device__ __forceinline__ float T(int dy) const { return dy == 0 ? T0 : Tdown; } // dy in {0,-1}
  __device__ __forceinline__ float4 water() const { return water_; }
  __device__ __forceinline__ char4 wall() const { return wall_; }
  __device__ __forceinline__ float sun_at(int dx, int j) const { return rg.lx[(j + 8) & (WL - 1)][li + dx]; }
  __device__ __forceinline__ float ir_down_at(int) const { return lz_up; }                           // row min(y+1, Y-1): carried register
  __device__ __forceinline__ float ir_up_at(int j) const { return rg.lw[(j + 8) & (WL - 1)][li]; } // row max(y-1, 0)
};

// global access as (wave-uniform row pointer) + (per-lane 32-bit byte offset): the form the hardware addresses with a scalar base
// register pair and one offset VGPR -- no 64-bit vector arithmetic per access
// (the readfirstlane pair pins the row pointer to scalar registers and keeps the loop optimiser from turning the access into a
// per-lane 64-bit pointer induction variable)
__device__ __forceinline__ unsigned long long uniform_addr(const void *p)
{
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T ld_row(const T *row, unsigned &byte_off)
{
  typedef const __attribute__((address_space(1))) char *GBytes; // (an integer round trip would otherwise yield a generic "flat" pointer)
  typedef const __attribute__((address_space(1))) T *GPtr;
  // keeps the zero-extension next to the access (instruction selection folds it only within a block). The offset is taken by
  // reference and redefined IN PLACE: on a by-value copy every access paid a v_mov of the loop-invariant original (15 per row step)
  asm("" : "+v"(byte_off));
  return *(GPtr)((GBytes)uniform_addr(row) + byte_off);
}
template <int N> struct NativeVec;
template <> struct NativeVec<4> { typedef unsigned type; };
template <> struct NativeVec<8> { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <> struct NativeVec<16> { typedef unsigned type __attribute__((ext_vector_type(4))); };
template <class T> __device__ __forceinline__ void st_row(T *row, unsigned &byte_off, T v)
{
  typedef __attribute__((address_space(1))) char *GBytes;
  asm("" : "+v"(byte_off));
#if WX_WET_NT_STORES
  // streamed once, read again only by the next launch: keep the output rows from evicting the input lines that neighbouring
  // strips still share (halo columns) out of the 4 MB L2
  typedef typename NativeVec<sizeof(T)>::type NV;
  typedef __attribute__((address_space(1))) NV *GPtr;
  __builtin_nontemporal_store(__builtin_bit_cast(NV, v), (GPtr)((GBytes)uniform_addr(row) + byte_off));
#else
  typedef __attribute__((address_space(1))) T *GPtr;
  *(GPtr)((GBytes)uniform_addr(row) + byte_off) = v;
#endif
}
#else // host pass of the single-source compile: same meaning, never executed
template <class T> __device__ __forceinline__ T ld_row(const T *row, unsigned &byte_off) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(row) + byte_off); }
template <class T> __device__ __forceinline__ void st_row(T *row, unsigned &byte_off, T v) { *reinterpret_cast<T *>(reinterpret_cast<char *>(row) + byte_off) = v; }
#endif
// by-value forms (a private copy of the offset per access): what the dry marching kernel uses -- there the in-place form is 2 % slower
template <class T> __device__ __forceinline__ T ld_row_v(const T *row, unsigned byte_off) { return ld_row(row, byte_off); }
template <class T> __device__ __forceinline__ void st_row_v(T *row, unsigned byte_off, T v) { st_row(row, byte_off, v); }
__device__ __forceinline__ int ring_back(int s, int k, int n) // slot of the row k steps behind the one in slot s (ring of n)
{
  const int v = s - k;
  return v < 0 ? v + n : v;
}

// A launch covers the strips [strip_lo, strip_lo + n_strips) of the n_strips_all strips of the (slab of the) grid: the whole
// width normally; the edge strips and the interior separately where a slab overlaps its halo exchange with compute.
// OPT_OUT: also store what only display-side consumers see (curl, post-boundary water, post-advection base): last iteration of a
// wx_step call. HAS_FB: the precipitation feedback / deposition textures may be non-zero.
// fix: the list the output cells fed by a back-trace longer than 0.9 cells are appended to (k_wet_fix recomputes them).
// The rows are cut into segments of (possibly) different heights: start[s] .. start[s+1]. Rows near terrain take the general,
// branchy instantiations and cost about twice as much as free air, so the host makes the bottom segments shorter
// (wet_launch_shape) and all waves of a launch finish together.
constexpr int WMAXSEG = 128;
struct WetSegs {
  int n_seg;
  int bands;              // experimental (WX_WET_BANDS=1): XCD k takes the row band [k*Y/8, (k+1)*Y/8) of ALL strips; start[] is relative to it
  int start[WMAXSEG + 1];
};
// QUIET: no brush input and no airplane event in this iteration (the host looks at the uniforms): advection_cell without those sections.
template <bool OPT_OUT, bool HAS_FB, bool QUIET>
__global__ __launch_bounds__(64 * WX_WET_WPB, WX_WET_MINWAVES) void k_march_wet(const FullCtx *__restrict__ ctx, float iterNum, WetIn in_arg, WetOut out_arg,
                                                                    WetFixList fix_arg, WetFixList fix_edge_arg, int n_strips, int strip_lo,
                                                                    int n_strips_all, WetSegs segs, int split_at, int strip_lo2, StripOrder order_arg, VxTrack vx_arg)
{
  struct KArgs { // layout of the kernel-argument segment (the parameter list as a struct)
    const FullCtx *ctx;
    float iterNum;
    WetIn in;
    WetOut out;
    WetFixList fix[2]; // [1]: the edge strips' own list in a split iteration (StripOrder::edge_list)
    int n_strips, strip_lo, n_strips_all;
    WetSegs segs;
    int split_at, strip_lo2;
    StripOrder order;
    VxTrack vx;
  };
#if WX_WET_ARGS_MEM && defined(__HIP_DEVICE_COMPILE__)
  // The ~22 plane pointers are read from the kernel-argument segment (constant address space) where they are used: separate
  // two-dword scalar loads that the register allocator can re-issue, instead of 44 SGPRs preloaded in wide loads that it can only
  // spill -- and an SGPR spill / restore is a v_writelane / v_readlane, i.e. a VECTOR instruction in a VALU-bound loop.
  typedef const __attribute__((address_space(4))) char *KBytes;
  const KBytes ka_c = (KBytes)__builtin_amdgcn_kernarg_segment_ptr();
  const __attribute__((address_space(4))) WetIn &in = *(const __attribute__((address_space(4))) WetIn *)(ka_c + offsetof(KArgs, in));
  const __attribute__((address_space(4))) WetOut &out = *(const __attribute__((address_space(4))) WetOut *)(ka_c + offsetof(KArgs, out));
  // (the split-iteration order likewise: read where it is used -- prologue and epilogue --, nothing of it lives in the row loop)
  const __attribute__((address_space(4))) StripOrder &order_c = *(const __attribute__((address_space(4))) StripOrder *)(ka_c + offsetof(KArgs, order));
#define WX_ORDER() (StripOrder{order_c.mode, order_c.nl, order_c.nr0, order_c.arrive, order_c.epoch, order_c.epoch_want, order_c.edge_list, order_c.prio, order_c.nofence})
#else
  const WetIn &in = in_arg;
  const WetOut &out = out_arg;
#define WX_ORDER() (order_arg)
#endif
  __shared__ WetRing rings[WX_WET_WPB];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WetRing &rg = rings[wave];
  // Everything the wave reads from the context goes through the constant address space: scalar loads that the compiler may
  // issue (and re-issue) anywhere. Through a generic pointer every such load behind the kernel's first store becomes a VECTOR
  // load with a uniform address, and waiting for it means waiting for the row prefetch issued just before (one in-order counter).
#if WX_WET_UNI_COPY
  const Uni u = ctx->u; // read before the first store of the kernel: scalar loads, kept in SGPRs
#else
  CUni &u = as_constant(ctx->u);
#endif
  const Geo g = ctx->g;
  const CFloatP initial_T = as_constant(ctx->initial_T), snd_T = as_constant(ctx->snd_T), snd_W = as_constant(ctx->snd_W), snd_Vel = as_constant(ctx->snd_Vel);
  const int X = g.X, Y = g.Y;
  const int lane = threadIdx.x & 63, li = lane + WPAD;
  const int iterI = (int)iterNum;
  const bool smooth_iter = iterI % 100 == 0; // the only iterations in which the boundary pass reads its horizontal water neighbours
  // XCD-aware placement: workgroup id lands on XCD id % 8 (MI355X_MICROARCH.md), every XCD has its own L2. XCD k takes the
  // column block of strips [k*S/8, (k+1)*S/8) of EVERY segment: neighbouring strips (which share two 128-byte lines of halo
  // columns per field) hit the same L2, and every XCD gets the same mix of cheap free-air rows and expensive rows near terrain
  // (a contiguous range of segment-major items would hand all the terrain segments to XCD 0). Bottom segments first.
  // A workgroup = WX_WET_WPB neighbouring strips of one segment, one wavefront each.
  const int n_seg = segs.n_seg, k = blockIdx.x & 7, j = blockIdx.x >> 3;
  const bool bands = segs.bands != 0;
  const int sk0 = bands ? 0 : (k * n_strips) >> 3, nk = bands ? n_strips : (((k + 1) * n_strips) >> 3) - sk0, gk = (nk + WX_WET_WPB - 1) / WX_WET_WPB;
  int seg, strip;
  bool is_edge = false; // (wave-uniform) a split iteration's edge strip: waits for the ghost columns / reports when it is done
  bool sig_edge = false, edge_list = false;
  const int order_mode = WX_ORDER().mode;
  if (order_mode == 0 || order_mode == 4) {
    if (j >= gk * n_seg) return;
    seg = j / gk;
    const int sloc = (j - seg * gk) * WX_WET_WPB + wave;
    if (sloc >= nk) return;
    // (a launch may cover two strip ranges -- the left and the right edge strips of a slab: the first split_at strips start at strip_lo,
    // the others at strip_lo2)
    const int sidx = sk0 + sloc;
    strip = sidx < split_at ? strip_lo + sidx : strip_lo2 + (sidx - split_at);
    is_edge = order_mode == 4; // (the edge group of the two-launch protocol: every strip of the launch is an edge strip -- issue priority only)
  } else { // one launch over ALL strips, the edge strips first (or last) in dispatch order (StripOrder, wx_tile.h)
    const StripOrder order = WX_ORDER();
    StripPick pk;
    if (!strip_order_pick(order, sk0, sk0 + nk, n_seg, WX_WET_WPB, wave, j, seg, pk)) return;
    strip = pk.strip;
    is_edge = pk.is_edge;
    sig_edge = is_edge && order.arrive != nullptr;
    edge_list = is_edge && order.edge_list != 0;
  }
  const int item = ((bands ? k * n_seg : 0) + seg) * n_strips_all + strip;
  const int band_lo = bands ? (int)(((long long)k * g.Y) >> 3) : 0, band_hi = bands ? (int)(((long long)(k + 1) * g.Y) >> 3) : g.Y;
  const int c_out = strip * WOUT + lane - WLO; // output column of this lane (may be >= X in the last strip, < 0 in the first)
  const int col = wrapmod(c_out, X);           // column this lane loads / computes
  const bool lane_out = lane >= WLO && lane < WLO + WOUT && c_out < X;
  unsigned lo4 = (unsigned)col * 4u, lo8 = (unsigned)col * 8u, lo16 = (unsigned)col * 16u; // byte offsets of the loaded column
  unsigned lo12 = (unsigned)col * 12u; // (feedback texels)
  unsigned so4 = lane_out ? (unsigned)c_out * 4u : 0u, so8 = so4 * 2u, so16 = so4 * 4u;    // ... of the stored column
  const int y_lo = band_lo + segs.start[seg], y_hi = min(band_lo + segs.start[seg + 1], band_hi);
  if (y_lo >= y_hi) { // (an empty segment of a clipped band still counts as an edge item that is done)
    if (sig_edge) strip_order_arrive(WX_ORDER(), lane, false);
    return;
  }
  if (is_edge) { // the ghost columns this strip reads are being written by the exchange
    const StripOrder order = WX_ORDER();
    if (order.epoch != nullptr) strip_order_wait(order);
    strip_order_prio(order.prio);
  }
#define WX_WALL_RAW (reinterpret_cast<const int *>(in.wall))
  (void)item;

  // ---- registers carried from step to step ----
  float4 pf_b, pf_q = make_float4(0.f, 0.f, 0.f, 0.f);                // prefetched: base row r, water row r-2
  int pf_w;                                                          // wall row r (raw dword)
  float pf_lx = 0.f, pf_l0x = 0.f, pf_l0y = 0.f;                      // source sunlight, light_0 sunlight / net heating, row r-2
  float2 pf_lzw = make_float2(0.f, 0.f);                              // source IR fluxes row r-2
  float3 pf_fb = make_float3(0.f, 0.f, 0.f);                          // feedback / deposition row r-3 (HAS_FB)
  float2 pf_dep = make_float2(0.f, 0.f);
  bool fb_have = false, dep_have = false;                             // wave-uniform: the tile(s) of that row hold feedback / deposition
  unsigned short pf_flag = 0x0101;                                    // "feedback | deposition tile is all zero" flags (low | high byte) of the row prefetched next
  float4 b_prev = make_float4(0.f, 0.f, 0.f, 0.f);                    // base_0 row r-1
  int w_prev = 0;
  float4 q1 = make_float4(0.f, 0.f, 0.f, 0.f);                        // pre-boundary water row r-3
  float v1x = 0.f, v1y = 0.f;                                         // velocity row r-2
  float p2 = 0.f, t2 = 0.f;                                           // P, T of row r-2 (unchanged by the velocity pass)
  int w2 = 0;                                                         // wall row r-2 (raw dword)
  float v3x = 0.f, v3y = 0.f, p3 = 0.f, t3 = 0.f;                     // velocity output of row r-3: what the boundary stage starts from
  int w3 = 0;
  float c1 = 0.f, c2 = 0.f;                                           // curl rows r-3, r-4
  float vfDx = 0.f;                                                   // vortForce.x row r-4
  float TD = 0.f, vxD = 0.f, qzD = 0.f, qwD = 0.f;                    // pre-boundary values of row r-4 (see MWBoundaryAcc)
  char4 wD = make_char4(0, 0, 0, 0);
  float l0x1 = 0.f, l0y1 = 0.f, lz1 = 0.f;                            // light_0 sunlight / net heating, source IR_down of row r-3
  float adv_vy_prev = 0.f, adv_T_prev = 0.f;                          // advection output row r-5
  char4 adv_w_prev = make_char4(0, 0, 0, 0);
  // wave-uniform row flags as bit histories (bit 0 = the newest row, shifted up by one per step; one SGPR each instead of one per row:
  // the loop is short of them -- every spilled SGPR is a v_writelane / v_readlane, i.e. a vector instruction)
  unsigned h_big = 0;    // "some |v| >= 0.9" of post-boundary rows r-3, r-4, r-5
  unsigned h_nowall = 0; // "no wall cell" of the same rows
  unsigned h_zw0 = 0;    // "precipitation-visual and smoke channels of the water are all zero" of the same rows
  unsigned h_near = 15;  // "some cell at or next to a wall" of input rows r .. r-3
  float vx_seen = 0.f;   // largest |vx| among the post-boundary velocities of this wave's rows (VxTrack: slabs size their exchange period by it)
#define WX_H_SET(h, v) h = ((h) & ~1u) | ((v) ? 1u : 0u)
#define WX_H_ROT(h) h = ((h) << 1) | ((h) & 1u)
  // outputs of the previous step, stored at the top of this one
  float4 st_p = make_float4(0.f, 0.f, 0.f, 0.f), st_q = st_p, st_l = st_p, st_ab = st_p;
  char4 st_w = make_char4(0, 0, 0, 0);
  bool st_valid = false;
  bool st_td = false; // (wave-uniform) the stored row holds a cell directly above a land surface cell: its post-advection T differs from the post-pressure one

#ifdef WX_WET_TIMING
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  int r = y_lo - 4;
  { // prefetch of the first row
    const size_t e = (size_t)wrapmod(r, Y) * X;
    pf_b = ld_row(in.base + e, lo16);
    pf_w = ld_row(WX_WALL_RAW + e, lo4);
  }
  int sq3 = (r - 3 + 12) % WQ; // ring slot of row r-3 (base / wall / water planes)
  // wrapped (REPEAT) row indices of rows r+1, r-1 .. r-4, advanced by one per step (a general modulo costs ~20 scalar instructions)
  int yw_p1 = wrapmod(r + 1, Y), yw_m1 = wrapmod(r - 1, Y), yw_m2 = wrapmod(r - 2, Y), yw_m3 = wrapmod(r - 3, Y), yw_m4 = wrapmod(r - 4, Y);
#if WX_WET_PRIO_ROTATE
  // wave slot within the SIMD (HW_ID.WAVE_ID): the waves that share a SIMD have different ones
  const int prio_phase = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
#endif
  int t = 0;
  // WARM: one of the first eight steps of the segment, in which the stages come alive one after the other (t >= ...); the steady-state
  // instantiation has none of those wave-uniform branches -- every one of them is a control-flow merge at which the carried values of
  // both paths meet, i.e. a bundle of v_mov copies per step (a third of the loop's vector instructions were v_mov_b32)
#define WX_T_GE(n) (!WARM || t >= (n))
  auto step = [&](auto warm_tag) __attribute__((always_inline)) {
    constexpr bool WARM = decltype(warm_tag)::value;
#if WX_WET_PRIO_ROTATE
    { // the SIMD issues the OLDEST ready wave first: without this the first-dispatched waves run ~30 % faster than the last ones
      const int pr = (t + prio_phase) % 3;
      if (pr == 0) __builtin_amdgcn_s_setprio(0);
      else if (pr == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(2);
    }
#endif
    const float4 b_cur = pf_b, q_up = pf_q;       // base row r, water row r-2
    int w_raw = pf_w;
    const float lx_cur = pf_lx, l0x_cur = pf_l0x, l0y_cur = pf_l0y; // light row r-2
    const float2 lzw_cur = pf_lzw;
    const float4 fb_cur = make_float4(pf_fb.x, pf_fb.y, pf_fb.z, 0.f); // feedback row r-3
    const float2 dep_cur = pf_dep;
    const bool fb_cur_have = fb_have || dep_have;
    asm volatile("" : "+v"(w_raw)); // keeps the byte unpacking on this side of the prefetch
    const char4 w_cur = unpack_wall(w_raw);
#ifdef WX_STAGE_MARKS
    asm volatile("; @@ring");
#endif
    // ---- row r-3 enters the ring with its pre-boundary T and wall (what the boundary stage reads of its horizontal neighbours; the
    //      slot was row r-6's, last read by the previous step's advection); light row r-2 ----
    // Ahead of the prefetch: behind it (and behind the deferred stores) the compiler puts an s_waitcnt vmcnt in front of these LDS
    // writes that waits for the loads just issued -- a memory latency per step (5 % of the feedback instantiation,
    // profiles/r02_particles_ring_first.txt; the other one shows the same wait as soon as the stores lose their address copies).
    auto ring_in = [&]() {
      const int o = sq3 * WRW + li;
      (&rg.T[0][0])[o] = t3;
      (&rg.wl[0][0])[o] = unpack_wall(w3);
      rg.lx[(r - 2 + 8) & (WL - 1)][li] = lx_cur;
      rg.lw[(r - 2 + 8) & (WL - 1)][li] = lzw_cur.y;
    };
    ring_in();
#ifdef WX_STAGE_MARKS
    asm volatile("; @@prefetch");
#endif
#if WX_WET_PRIO_MEM
    __builtin_amdgcn_s_setprio(WX_WET_PRIO_MEM); // the step's loads and stores go out ahead of the other waves' arithmetic
#endif
    // ---- software prefetch: the loads of the next step are in flight while this one computes ----
    if (r < y_hi + 3) {
      {
        const size_t e = (size_t)yw_p1 * X;
        pf_b = ld_row(in.base + e, lo16);
        pf_w = ld_row(WX_WALL_RAW + e, lo4);
      }
      if (WX_WET_SKIP_LOADS != 1 || WX_T_GE(2)) { // (the first warm-up steps of a segment only feed velocity / curl: no water, no light yet)
        const int rl = r - 1;
        const size_t ew = (size_t)yw_m1 * X;
        pf_q = ld_row(in.water + ew, lo16);
        // light textures clamp in y (sun ray / IR taps of the lighting pass) ...
        const size_t el = (size_t)(rl < 0 ? 0 : (rl > Y - 1 ? Y - 1 : rl)) * X;
        pf_lx = ld_row(in.lsrc.x + el, lo4);
        pf_lzw = ld_row(in.lsrc.zw + el, lo8);
        // ... while the boundary pass samples light_0 at its own (wrapped) row and at the row above it, clamped
        pf_l0y = ld_row(in.l0.y + ew, lo4);
        // light_0's sunlight is only read by cells next to a wall and by surface wall cells, of rows r-1 and r-2: skipped while
        // none of the wall rows loaded so far (r-3 .. r) has such a cell (in even iterations the load coincides with pf_lx anyway)
        if (WX_WET_SKIP_LOADS) WX_H_SET(h_near, __any(w_cur.y <= 1));
        if (!WX_WET_SKIP_LOADS || (h_near & 15u)) pf_l0x = ld_row(in.l0.x + ew, lo4);
      }
      if (HAS_FB) {
        // does any of the (up to three) 64x16 tiles under this strip hold feedback in row r-2? The flag byte was loaded one step
        // ago (pf_flag), so the vote costs no wait of its own
        fb_have = __any((pf_flag & 0xffu) == 0);
        dep_have = __any((pf_flag >> 8) == 0); // (only droplets that reach the ground deposit: few tiles)
#if defined(WX_ABL_FB_NOFLAG) || defined(WX_ABL_FB_NOLOAD)
        fb_have = dep_have = false; // (timing experiments only: wrong results)
#endif
#ifdef WX_ABL_FB_NOLOAD
        pf_fb = make_float3(0.f, 0.f, 0.f);
        pf_dep = make_float2(0.f, 0.f);
#else
        {
          // always the same two loads -- from the textures' row, or from a row of zeros (L2 resident) where the tiles are known
          // to be zero: a conditional load would make the number of loads per step, which the waits are built on, vary
          const size_t e = (size_t)yw_m2 * X;
#if WX_WET_FB_COND
          if (fb_have) {
            pf_fb = ld_row(in.fb + e, lo12);
            pf_dep = ld_row((dep_have && (!WX_WET_DEP_NEAR || (h_near & 4u))) ? in.dep + e : reinterpret_cast<const float2 *>(in.zero_row), lo8);
          } else {
            pf_fb = make_float3(0.f, 0.f, 0.f);
            pf_dep = make_float2(0.f, 0.f);
          }
#else
          pf_fb = ld_row(fb_have ? in.fb + e : reinterpret_cast<const float3 *>(in.zero_row), lo12);
          // (the deposition texture is only read by surface wall cells, boundaryShader.frag:390-475: rows without a cell at or next to
          // a wall take it from the row of zeros too)
          pf_dep = ld_row((dep_have && (!WX_WET_DEP_NEAR || (h_near & 4u))) ? in.dep + e : reinterpret_cast<const float2 *>(in.zero_row), lo8);
#endif
        }
#endif
#if !defined(WX_ABL_FB_NOFLAG) && !defined(WX_ABL_FB_NOLOAD)
        pf_flag = in.fb_zero != nullptr ? reinterpret_cast<const unsigned short *>(in.fb_zero)[(yw_m1 >> 4) * in.fb_txn + (col >> 6)] : (unsigned short)0; // row r-1, voted on next step
#endif
      }
    }
#ifdef WX_STAGE_MARKS
    asm volatile("; @@stores");
#endif
    // ---- the stores of the previous step's row (r-5), issued behind the prefetch ----
    if (st_valid && lane_out) {
      const size_t e = (size_t)(r - 5) * X;
      st_row(out.base + e, so16, st_p);
      st_row(out.water + e, so16, st_q);
      st_row(out.wall + e, so4, st_w);
      st_row(out.light.x + e, so4, st_l.x);
      st_row(out.light.y + e, so4, st_l.y);
      st_row(out.light.zw + e, so8, make_float2(st_l.z, st_l.w));
      if (OPT_OUT) st_row(out.p_disp + e, so4, st_ab.z);
#ifndef WX_ABL_NO_TDISP
      // post-advection temperature for the droplets: only rows in which the pressure pass changed it (k_precipitation's precip_T makes
      // the same test per texel and reads the post-pressure T everywhere else)
      if (out.t_disp && st_td) st_row(out.t_disp + e, so4, st_ab.w);
#endif
    }
    st_valid = false;
#if WX_WET_PRIO_MEM
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef WX_STAGE_MARKS
    asm volatile("; @@velocity");
#endif
    // ---- velocity of row r-1 ----
    float v0x = 0.f, v0y = 0.f;
    if (WX_T_GE(1)) {
      const float4 v = velocity_cell(u, b_prev, wave_from_right(b_prev.z), b_cur.z, unpack_wall(w_prev).y);
      v0x = v.x;
      v0y = v.y;
    }
#ifdef WX_STAGE_MARKS
    asm volatile("; @@curlvort");
#endif
    // ---- curl of row r-2, vortForce of row r-3 (registers + wave shifts only) ----
    float c0 = 0.f;
    if (WX_T_GE(2)) {
      c0 = curl_cell(v1x, v1y, wave_from_right(v1y), v0x);
      if (OPT_OUT) {
        const int yc = r - 2;
        if (lane_out && yc >= y_lo && yc < y_hi) st_row(out.curl + (size_t)yc * X, so4, c0);
      }
    }
    float2 vf = make_float2(0.f, 0.f);
    if (WX_T_GE(4)) vf = vorticity_cell(c1, wave_from_left(c1), wave_from_right(c1), c2, c0);
    const float vfLy = wave_from_left(vf.y);
    float qzL = 0.f, qwL = 0.f, qzR = 0.f, qwR = 0.f;
    if (smooth_iter) { // (wave-uniform) soil moisture / snow smoothing between surface cells: the neighbours' water texels
      qzL = wave_from_left(q1.z);
      qwL = wave_from_left(q1.w);
      qzR = wave_from_right(q1.z);
      qwR = wave_from_right(q1.w);
    }
    wave_fence();

#ifdef WX_STAGE_MARKS
    asm volatile("; @@boundary");
#endif
    // ---- boundary of row yb = r-3, written back in place ----
    if (WX_T_GE(4)) {
      const int ob0 = sq3 * WRW;
      const float4 b00 = make_float4(v3x, v3y, p3, t3);
      const char4 w00 = unpack_wall(w3);
      if (WX_T_GE(5)) {
        const int yb = yw_m3;
        const bool top = yb + 1 > Y - 1; // light_0 is CLAMP_TO_EDGE in y: the row "above" the top row is the top row itself
        MWBoundaryAcc a{rg, li, ob0, v1x, t2, unpack_wall(w2), b00, q1, q_up, w00, wD, vxD, TD, qzD, qwD, qzL, qwL, qzR, qwR, vf, vfLy, vfDx,
                        l0x1, l0y1, top ? l0x1 : l0x_cur, top ? l0y1 : l0y_cur, fb_cur, dep_cur, HAS_FB && fb_cur_have};
        float4 bb, bq;
        char4 bwl;
#ifdef WX_ABL_NOBOUNDARY // (ablation builds for the per-stage instruction budget; not bit-exact)
        bb = a.base(0, 0);
        bq = a.water(0, 0);
        bwl = a.wall(0, 0);
        bb.x += vf.x + vfLy + vfDx;
#else
        // free air (no wall within one cell, terrain at least 8 rows below) in every lane that feeds something: the
        // branch-free instantiation. Most rows of most strips; the general one handles everything else.
        if (WX_ABL_FORCE_AIR || (WX_WET_AIR && __all(lane < 2 || lane > 60 || air_cell(w00, a.wall(-1, 0), wD, a.wall(1, 0), a.wall(0, 1)))))
          boundary_cell<true>(u, iterNum, iterI, g, initial_T, col, yb, a, bb, bq, bwl);
        else
          boundary_cell<false>(u, iterNum, iterI, g, initial_T, col, yb, a, bb, bq, bwl);
#endif
        wave_fence(); // every lane has read its neighbours' pre-boundary values
        (&rg.vx[0][0])[ob0 + li] = bb.x;
        (&rg.vy[0][0])[ob0 + li] = bb.y;
        (&rg.P[0][0])[ob0 + li] = bb.z;
        (&rg.T[0][0])[ob0 + li] = bb.w;
        (&rg.wl[0][0])[ob0 + li] = bwl;
        const int oq = sq3 * WRW + li;
        (&rg.qx[0][0])[oq] = bq.x;
        (&rg.qy[0][0])[oq] = bq.y;
        (&rg.qz[0][0])[oq] = bq.z;
        (&rg.qw[0][0])[oq] = bq.w;
        vx_seen = fmaxf(vx_seen, fabsf(bb.x));
        // back-traces of this row that may leave the 3x3 cells? (lanes 2 .. 60 feed an advection that is used)
        WX_H_SET(h_big, __any(lane >= 2 && lane <= 60 && !(fmaxf(fabsf(bb.x), fabsf(bb.y)) < 0.9f)));
        WX_H_SET(h_nowall, __all(lane < 2 || lane > 60 || bwl.y != 0)); // no wall cell in this post-boundary row (as far as advection reads it)
        WX_H_SET(h_zw0, __all(bq.z == 0.0f && bq.w == 0.0f));          // no rain / snow / smoke anywhere in it
        if (OPT_OUT) {
          const int yo = r - 3;
          // (NULL: the host makes waterTexture_0 on demand -- only saves read it, wxsim.hip materialize_water0)
          if (out.water0 && lane_out && yo >= y_lo && yo < y_hi) st_row(out.water0 + (size_t)yo * X, so16, bq);
        }
      }
      // the pre-boundary values of this row are what the row above reads as its lower neighbour
      TD = b00.w;
      vxD = b00.x;
      wD = w00;
      qzD = q1.z;
      qwD = q1.w;
    }
    wave_fence();

#ifdef WX_STAGE_MARKS
    asm volatile("; @@advection");
#endif
    // ---- advection of row ya = r-4 ----
    if (WX_T_GE(7)) {
      const int ya = yw_m4;
      float4 ab, aw;
      char4 awl;
      MWAdvAcc a{rg, li, {ring_back(sq3, 2, WQ) * WRW, ring_back(sq3, 1, WQ) * WRW, sq3 * WRW}};
      bool fast = true;
      if (h_big & 7u) { // wave-uniform: some velocity of rows ya-1 .. ya+1 is large -> per-lane test of the eight that matter
        const float *vxp = &rg.vx[0][0], *vyp = &rg.vy[0][0];
        const int o0 = a.ob[1] + li, om = a.ob[0] + li, op = a.ob[2] + li;
        const float m = fmaxf(fmaxf(fmaxf(fabsf(vxp[o0]), fabsf(vxp[o0 - 1])), fmaxf(fabsf(vxp[op]), fabsf(vxp[op - 1]))),
                              fmaxf(fmaxf(fabsf(vyp[o0]), fabsf(vyp[om])), fmaxf(fabsf(vyp[o0 + 1]), fabsf(vyp[om + 1]))));
        fast = m < 0.9f || lane < 3 || lane > 59; // (lanes outside 3 .. 59 feed nothing)
        if (!fast) {
          // This cell (column c_out, unwrapped row yu) keeps a placeholder; the OUTPUT cells it feeds that this wave owns -- its own,
          // the right neighbour's (pressure: vx of the left cell), the upper neighbour's (pressure / lighting: vy, T, wall of the lower
          // cell) -- go to the fix list. Rare path: one returning atomic per such lane.
          const int yu = r - 4; // == y_lo - 1 + (t - 7)
          const bool row_mine = yu >= y_lo, up_mine = yu + 1 < y_hi;
          const bool oA = lane_out && row_mine, oB = lane + 1 >= WLO && lane + 1 < WLO + WOUT && c_out + 1 < X && row_mine, oC = lane_out && up_mine;
          int n_add = (int)oA + (int)oB + (int)oC;
#ifdef WX_ABL_NOFIX // (timing-only ablation builds produce garbage velocities: keep them from flooding the exact path)
          n_add = 0;
#endif
          // (the list: read from the kernel-argument segment HERE, in the rare branch -- nothing of it is live in the loop; the edge strips
          // of a split iteration have a list of their own, consumed on the comm stream before the halo is packed)
#if WX_WET_ARGS_MEM && defined(__HIP_DEVICE_COMPILE__)
          const __attribute__((address_space(4))) WetFixList &fix =
              *(const __attribute__((address_space(4))) WetFixList *)(ka_c + offsetof(KArgs, fix) + (edge_list ? sizeof(WetFixList) : 0));
#else
          const WetFixList &fix = edge_list ? fix_edge_arg : fix_arg;
#endif
          if (fix.fastest) atomicMax(fix.fastest, __float_as_int(m)); // (m >= 0.9 or NaN: the bit patterns of positive floats order like ints)
          if (n_add) {
            int at = atomicAdd(fix.count, n_add);
            if (at + n_add <= fix.cap) {
              if (oA) fix.cells[at++] = make_int2(c_out, yu);
              if (oB) fix.cells[at++] = make_int2(c_out + 1, yu);
              if (oC) fix.cells[at++] = make_int2(c_out, yu + 1);
            }
          }
        }
      }
#ifdef WX_ABL_NOADV
      fast = false;
#endif
      if (fast) {
        if (WX_ABL_FORCE_AIR || (WX_WET_AIR && (h_nowall & 7u) == 7u)) { // (wave-uniform) plain instead of wall-aware interpolation, no wall branch
          if (WX_WET_ZW0 && (h_zw0 & 7u) == 7u) // ... and nothing to interpolate in the precipitation-visual / smoke channels
          {
#ifdef WX_STAGE_MARKS
            asm volatile("; @@advair");
#endif
            advection_cell<false, true, true, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, ya, a, ab, aw, awl);
#ifdef WX_STAGE_MARKS
            asm volatile("; @@advairend");
#endif
          }
          else
            advection_cell<false, true, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, ya, a, ab, aw, awl);
        } else {
          advection_cell<false, false, false, QUIET>(u, g, initial_T, snd_T, snd_W, snd_Vel, col, ya, a, ab, aw, awl);
        }
      } else { // placeholder (the post-boundary texel): this cell and the two it feeds are recomputed after the loop
        ab = a.base(0, 0);
        aw = a.water_off(0, 0);
        awl = a.wall(0, 0);
      }
#ifdef WX_STAGE_MARKS
    asm volatile("; @@presslight");
#endif
      // ---- pressure + lighting of row ya: kept in registers, stored at the top of the next step ----
      const float vx_l = wave_from_left(ab.x);
      if (WX_T_GE(8)) {
        st_p = pressure_cell(ab, vx_l, adv_vy_prev, adv_T_prev, adv_w_prev.x, adv_w_prev.y);
        if (out.t_disp) st_td = __any(adv_w_prev.y == 0 && adv_w_prev.x == 1); // (pressure_cell's condition, any lane of the row)
        MWLightAcc la{rg, li, ab.w, adv_T_prev, lz1, aw, awl};
#ifdef WX_ABL_NOLIGHT
        st_l = make_float4(la.sun_at(0, r - 4), la.ir_up_at(r - 5), lz1, ab.w);
#else
        if (WX_ABL_FORCE_AIR || (WX_WET_AIR && __all(lane < WLO || lane >= WLO + WOUT || (awl.y != 0 && awl.z != 1))))
          st_l = lighting_cell<true>(u, g, col, r - 4, la);
        else
          st_l = lighting_cell<false>(u, g, col, r - 4, la);
#endif
        st_q = aw;
        st_w = awl;
        st_ab = ab;
        st_valid = true;
      }
      adv_vy_prev = ab.y;
      adv_T_prev = ab.w;
      adv_w_prev = awl;
    }
#ifdef WX_STAGE_MARKS
    asm volatile("; @@rotate");
#endif
    // ---- rotate the carried rows ----
    v3x = v1x;
    v3y = v1y;
    p3 = p2;
    t3 = t2;
    w3 = w2;
    p2 = b_prev.z;
    t2 = b_prev.w;
    w2 = w_prev;
    b_prev = b_cur;
    w_prev = w_raw;
    q1 = q_up;
    v1x = v0x;
    v1y = v0y;
    c2 = c1;
    c1 = c0;
    vfDx = vf.x;
    l0x1 = l0x_cur;
    l0y1 = l0y_cur;
    lz1 = lzw_cur.x;
    WX_H_ROT(h_big);
    WX_H_ROT(h_nowall);
    WX_H_ROT(h_zw0);
    WX_H_ROT(h_near);
    sq3 = sq3 + 1 == WQ ? 0 : sq3 + 1;
    yw_m4 = yw_m3;
    yw_m3 = yw_m2;
    yw_m2 = yw_m1;
    yw_m1 = yw_m1 + 1 == Y ? 0 : yw_m1 + 1;
    yw_p1 = yw_p1 + 1 == Y ? 0 : yw_p1 + 1;
  };
  // (a segment has at least one row: at least nine steps)
  for (; t < 8; r++, t++) step(std::true_type{});
  if (WX_WET_UNROLL2 && !HAS_FB && !OPT_OUT) { // (doubling the other instantiations re-measured at four waves per SIMD: neutral, profiles/r04_unroll_variants_four_waves.txt)
    // two steps per loop iteration: the values carried from step to step (prefetched rows, the previous rows' registers, the deferred
    // stores) change registers between the two copies instead of being moved: -1.4 % without feedback loads; WITH them (particles on)
    // the doubled loop is 4-6 % slower, and the display-writing one (every tenth iteration) loses 2-8 %: only the plain instantiation
    // is doubled (profiles/r03_unroll_variants.txt)
    for (; r <= y_hi + 3;) {
      step(std::false_type{});
      r++;
      if (r > y_hi + 3) break;
      step(std::false_type{});
      r++;
#if WX_WET_UNROLL2 >= 3
      if (r > y_hi + 3) break;
      step(std::false_type{});
      r++;
#endif
    }
  } else {
    for (; r <= y_hi + 3; r++) step(std::false_type{});
  }
#undef WX_T_GE
  // ---- the last row ----
  if (st_valid && lane_out) {
    const size_t e = (size_t)(y_hi - 1) * X;
    st_row(out.base + e, so16, st_p);
    st_row(out.water + e, so16, st_q);
    st_row(out.wall + e, so4, st_w);
    st_row(out.light.x + e, so4, st_l.x);
    st_row(out.light.y + e, so4, st_l.y);
    st_row(out.light.zw + e, so8, make_float2(st_l.z, st_l.w));
    if (OPT_OUT) st_row(out.p_disp + e, so4, st_ab.z);
    if (out.t_disp && st_td) st_row(out.t_disp + e, so4, st_ab.w);
  }
  {
#if WX_WET_ARGS_MEM && defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) VxTrack &vc = *(const __attribute__((address_space(4))) VxTrack *)(ka_c + offsetof(KArgs, vx));
    vx_track_commit(VxTrack{vc.max_bits, vc.violation, vc.limit, vc.zone_l, vc.zone_r, vc.limit_in}, vx_seen, lane, strip);
#else
    vx_track_commit(vx_arg, vx_seen, lane, strip);
#endif
  }
  if (sig_edge) strip_order_arrive(WX_ORDER(), lane, true); // the halo exchange may pack this strip's columns
#undef WX_ORDER
#ifdef WX_WET_TIMING
  if (lane == 0) {
    out.cycles[2 * (size_t)item] = t_begin;
    out.cycles[2 * (size_t)item + 1] = __builtin_readcyclecounter();
  }
#endif
}

#undef WX_WALL_RAW
#undef WX_H_SET
#undef WX_H_ROT

// ---- k_wet_fix: the output cells on the fix list, recomputed exactly -- one wavefront per cell ----
// Output (X0, Y0) = pressure + lighting of the advected cells (X0, Y0), (X0 - 1, Y0) [vx of the left neighbour] and (X0, Y0 - 1)
// [vy, T, wall of the lower one]. The 64 lanes rebuild the 8 x 8 patch [X0 - 4, X0 + 3] x [Y0 - 4, Y0 + 3] of POST-BOUNDARY texels from
// the iteration's inputs (velocity, curl and vortForce evaluated on the fly, wet_boundary_texel_global), three lanes advect the three
// cells from that patch with the very same advection_cell -- it covers every back-trace shorter than two cells -- and lane 0
// finishes and stores the cell. Should a footprint leave the patch (|v| >= 2: the state has blown up) the fully general path from
// global memory (wet_output_cell_exact) takes over. Recomputing an output is idempotent (always from the iteration's inputs), so
// duplicates on the list and the order of the entries do not matter.
constexpr int WPATCH = 8, WPATCH_C = 4;
struct WetPatch {
  float vx[WPATCH * WPATCH], vy[WPATCH * WPATCH], P[WPATCH * WPATCH], T[WPATCH * WPATCH];
  float qx[WPATCH * WPATCH], qy[WPATCH * WPATCH], qz[WPATCH * WPATCH], qw[WPATCH * WPATCH];
  char4 wl[WPATCH * WPATCH];
};
struct WetPatchAcc {
  const WetPatch &p;
  int cx, cy; // patch coordinates of the own cell
  bool *bad;  // set when a texel outside the patch is asked for
  __device__ __forceinline__ int at(int dx, int dy) const
  {
    const int x = cx + dx, y = cy + dy;
    if (x < 0 || x >= WPATCH || y < 0 || y >= WPATCH) {
      *bad = true;
      return 0;
    }
    return y * WPATCH + x;
  }
  __device__ __forceinline__ float4 base(int dx, int dy) const
  {
    const int i = at(dx, dy);
    return make_float4(p.vx[i], p.vy[i], p.P[i], p.T[i]);
  }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return p.wl[at(dx, dy)]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return base(dx, dy); }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return wall(dx, dy); }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const
  {
    const int i = at(dx, dy);
    return make_float4(p.qx[i], p.qy[i], p.qz[i], p.qw[i]);
  }
};

// The patch of post-boundary texels is built COOPERATIVELY, stage by stage through LDS, the way the reference's passes follow each
// other (round 3, second version; the first let every lane rebuild its texel recursively from global memory -- velocity, curl and
// vortForce of the same cells up to 45 times per lane -- and took ~33 us per list entry):
//   base / wall      13 x 13  [-6, 6]   loaded once
//   velocity output  12 x 12  [-6, 5]   velocity_cell
//   curl             11 x 11  [-6, 4]   curl_cell
//   vortForce         9 x 9   [-5, 3]   vorticity_cell
//   boundary          8 x 8   [-4, 3]   boundary_cell on an accessor that reads the stages (water, light_0, feedback: global)
// Same cell functions, same operands: the same values as the marching loop computes.
constexpr int WSB = 13, WSV = 12, WSC = 11, WSF = 9; // stage widths; all stages start at offset -6 but vortForce (-5) and the patch (-4)
struct WetFixStage {
  float bx[WSB * WSB], by[WSB * WSB], bP[WSB * WSB], bT[WSB * WSB];
  char4 bw[WSB * WSB];
  float vx[WSV * WSV], vy[WSV * WSV];
  float cu[WSC * WSC];
  float fx[WSF * WSF], fy[WSF * WSF];
};
struct WetStageBoundaryAcc { // the interface of GWetRecomputeAcc, served from the stages; (px, py) = patch coordinates of the own texel
  const WetFixStage &st;
  const Uni &u_;
  const WetIn &in_;
  int X, Y, x, y; // wrapped global coordinates of the own texel
  int px, py;
  __device__ __forceinline__ float4 base(int dx, int dy) const
  { // the velocity pass's output: vx, vy from the velocity stage, P and T unchanged (offset -6 -> patch offset -4: +2)
    const int iv = (py + 2 + dy) * WSV + (px + 2 + dx), ib = (py + 2 + dy) * WSB + (px + 2 + dx);
    return make_float4(st.vx[iv], st.vy[iv], st.bP[ib], st.bT[ib]);
  }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return st.bw[(py + 2 + dy) * WSB + (px + 2 + dx)]; }
  __device__ __forceinline__ float4 water(int dx, int dy) const { return in_.water[fidx(wrapmod(x + dx, X), wrapmod(y + dy, Y), X)]; }
  __device__ __forceinline__ float2 vort(int dx, int dy) const
  { // vortForce stage starts at -5: patch offset -4 -> +1
    const int i = (py + 1 + dy) * WSF + (px + 1 + dx);
    return make_float2(st.fx[i], st.fy[i]);
  }
  __device__ __forceinline__ float light_y0() const { return in_.l0.y[fidx(x, y, X)]; }
  __device__ __forceinline__ float light_x0() const { return in_.l0.x[fidx(x, y, X)]; }
  __device__ __forceinline__ float2 light_xy_up() const
  {
    const size_t i = fidx(x, y + 1 > Y - 1 ? Y - 1 : y + 1, X); // light textures: CLAMP_TO_EDGE in T
    return make_float2(in_.l0.x[i], in_.l0.y[i]);
  }
  __device__ __forceinline__ bool has_fb() const { return in_.fb != nullptr; }
  __device__ __forceinline__ float4 fb() const
  {
    if (!in_.fb) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float3 v = in_.fb[fidx(x, y, X)];
    return make_float4(v.x, v.y, v.z, 0.f);
  }
  __device__ __forceinline__ float2 dep() const { return in_.dep ? in_.dep[fidx(x, y, X)] : make_float2(0.f, 0.f); }
};
// the 8 x 8 patch of post-boundary texels around (cx, cy), all 64 lanes of the wave
__device__ __noinline__ void wet_fix_build_patch(const FullCtx *__restrict__ c, const WetIn *__restrict__ in, float iterNum, WetFixStage *__restrict__ stp,
                                                 WetPatch *__restrict__ ptp, int cx, int cy, int lane)
{
  WetFixStage &st = *stp;
  WetPatch &pt = *ptp;
  const int X = c->g.X, Y = c->g.Y;
  for (int i = lane; i < WSB * WSB; i += 64) { // base_0 and wall_0 of the 13 x 13 neighbourhood
    const int sy = i / WSB, sx = i - sy * WSB;
    const size_t gi = fidx(wrapmod(cx - 6 + sx, X), wrapmod(cy - 6 + sy, Y), X);
    const float4 b = in->base[gi];
    st.bx[i] = b.x;
    st.by[i] = b.y;
    st.bP[i] = b.z;
    st.bT[i] = b.w;
    st.bw[i] = in->wall[gi];
  }
  wave_fence();
  for (int i = lane; i < WSV * WSV; i += 64) { // velocity pass
    const int sy = i / WSV, sx = i - sy * WSV, ib = sy * WSB + sx;
    const float4 v = velocity_cell(c->u, make_float4(st.bx[ib], st.by[ib], st.bP[ib], st.bT[ib]), st.bP[ib + 1], st.bP[ib + WSB], st.bw[ib].y);
    st.vx[i] = v.x;
    st.vy[i] = v.y;
  }
  wave_fence();
  for (int i = lane; i < WSC * WSC; i += 64) { // curl
    const int sy = i / WSC, sx = i - sy * WSC, iv = sy * WSV + sx;
    st.cu[i] = curl_cell(st.vx[iv], st.vy[iv], st.vy[iv + 1], st.vx[iv + WSV]);
  }
  wave_fence();
  for (int i = lane; i < WSF * WSF; i += 64) { // vortForce of [-5, 3]: curl index +1
    const int sy = i / WSF, sx = i - sy * WSF, ic = (sy + 1) * WSC + (sx + 1);
    const float2 f = vorticity_cell(st.cu[ic], st.cu[ic - 1], st.cu[ic + 1], st.cu[ic - WSC], st.cu[ic + WSC]);
    st.fx[i] = f.x;
    st.fy[i] = f.y;
  }
  wave_fence();
  { // boundary pass, one patch texel per lane
    const int px = lane & (WPATCH - 1), py = lane >> 3;
    const int gx = wrapmod(cx - WPATCH_C + px, X), gy = wrapmod(cy - WPATCH_C + py, Y);
    const WetStageBoundaryAcc a{st, c->u, *in, X, Y, gx, gy, px, py};
    float4 bb, bq;
    char4 bwl;
    boundary_cell(c->u, iterNum, (int)iterNum, c->g, c->initial_T, gx, gy, a, bb, bq, bwl);
    pt.vx[lane] = bb.x;
    pt.vy[lane] = bb.y;
    pt.P[lane] = bb.z;
    pt.T[lane] = bb.w;
    pt.qx[lane] = bq.x;
    pt.qy[lane] = bq.y;
    pt.qz[lane] = bq.z;
    pt.qw[lane] = bq.w;
    pt.wl[lane] = bwl;
  }
  wave_fence();
}

template <bool OPT_OUT>
__global__ __launch_bounds__(256) void k_wet_fix(const FullCtx *__restrict__ ctx, float iterNum, WetIn in, WetOut out, int *__restrict__ count,
                                                 const int2 *__restrict__ cells, int cap, int *__restrict__ overflow, int *__restrict__ hint)
{
  __shared__ WetPatch patches[4];
  __shared__ WetFixStage stages[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WetPatch &pt = patches[wave];
  // An OVERFLOWED list is not consumed at all: an appender whose 1-3 entries straddle cap writes none of them (k_march_wet, `at + n_add <=
  // fix.cap`), so up to two slots below cap may hold whatever the allocation held before -- coordinates nobody checked (found by
  // tools/fuzz_parity.py: a memory access fault a few cases after a handle whose list had overflowed). The overflow is reported and the results
  // since are invalid either way (wx_step's next blocking call fails with WX_E_STATE).
  const int total = *count, n = total <= cap ? total : 0;
  if (total == 0) {
    // The usual case, and a launch that is pure latency on a small grid (5 us of a 20 us iteration at 100 x 100): nothing to recompute and
    // nothing to reset -- one load, and out. count[2] remembers what the host's hint word was last told: it is set back once.
    if (blockIdx.x == 0 && threadIdx.x == 0 && hint && count[2] != 0) {
      count[2] = 0;
      __hip_atomic_store(hint, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (total > cap && blockIdx.x == 0 && threadIdx.x == 0) *overflow = total;
  const int X = ctx->g.X, Y = ctx->g.Y;
  // (entries whose footprints leave the patch fall back to wet_output_cell_exact, which builds its own argument block)
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const int2 c = cells[i];
    wet_fix_build_patch(ctx, &in, iterNum, &stages[wave], &pt, c.x, c.y, lane);
    AdvOut o;
    o.b = o.w = make_float4(0.f, 0.f, 0.f, 0.f);
    o.wl = make_char4(0, 0, 0, 0);
    bool bad = false;
    if (lane < 3) { // lane 0: the cell itself, lane 1: its left neighbour, lane 2: the cell below
      const int ox = lane == 1 ? -1 : 0, oy = lane == 2 ? -1 : 0;
      const WetPatchAcc a{pt, WPATCH_C + ox, WPATCH_C + oy, &bad};
      advection_cell(ctx->u, ctx->g, ctx->initial_T, ctx->snd_T, ctx->snd_W, ctx->snd_Vel, wrapmod(c.x + ox, X), wrapmod(c.y + oy, Y), a, o.b, o.w, o.wl);
    }
    const bool any_bad = __any(bad);
    const float vx_l = __shfl(o.b.x, 1), vy_d = __shfl(o.b.y, 2), T_d = __shfl(o.b.w, 2);
    const int wl_d = __shfl(*reinterpret_cast<const int *>(&o.wl), 2);
    if (lane == 0) {
      if (any_bad) {
        wet_output_cell_exact(ctx, &in, &out, iterNum, OPT_OUT, c.x, c.y);
      } else {
        const char4 wD = unpack_wall(wl_d);
        const size_t gi = fidx(c.x, c.y, X);
        out.base[gi] = pressure_cell(o.b, vx_l, vy_d, T_d, wD.x, wD.y);
        out.water[gi] = o.w;
        out.wall[gi] = o.wl;
        GWetLightAcc la{in.lsrc, o.w, o.wl, o.b.w, T_d, X, c.x};
        const float4 l = lighting_cell(ctx->u, ctx->g, c.x, c.y, la);
        out.light.x[gi] = l.x;
        out.light.y[gi] = l.y;
        out.light.zw[gi] = make_float2(l.z, l.w);
        if (OPT_OUT) out.p_disp[gi] = o.b.z;
        if (out.t_disp) out.t_disp[gi] = o.b.w;
      }
    }
    wave_fence(); // the patch is rewritten by the next entry
  }
  // the list is empty again for the next launch group: reset by the LAST workgroup to get here (every workgroup has read the count
  // by then) -- count[1] is the arrival ticket -- which saves a memset in the stream per iteration
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(count + 1, 1) == (int)gridDim.x - 1) {
    count[1] = 0;
    count[2] = total;
    __hip_atomic_store(count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hint) __hip_atomic_store(hint, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Segmentation. The grid is cut into about WX_WET_ROUNDS times as many waves as the device holds at once (the hardware deals
// workgroups to CUs as earlier ones retire; bottom segments are dispatched first), and rows below `air_from_row` -- the lowest
// row above which every cell is free air, -1 if unknown -- count WX_WET_ALPHA times a free-air row when the segment borders
// are placed. Measured at 16384x2048 (gpurun_out/r2j..r2l): 1 round 29-34, 2 rounds 36-37, 4 rounds 38-40, 6 rounds 38-40
// Gcell-steps/s -- many short waves beat few long ones although each segment re-runs 8 warm-up rows; weighting terrain rows
// (alpha 2-2.5) is worth +15 % with one round and nothing from two rounds on.
// Why (tuning build -DWX_WET_TIMING: s_memtime per wave, profiles/r02_wet_wave_timing.txt): the kernel is VALU-bound at 3 waves per
// SIMD (524 VALU instructions per row step x 4 cycles x 3 waves = the ~6800 cycles a wave takes per row), and the SIMD issues its
// OLDEST ready wave first. With one round of 9 long segments the first-dispatched waves run at 6000 cycles / row and the last at 8100;
// when the old ones retire nothing refills their slots and the young ones finish at 1-2 waves per SIMD, which cannot hide their own
// memory latency: 77 % of the SIMD-time is used, against 89 % with four rounds, and that costs more than the warm-up rows saved
// (0.87-0.90 ms against 0.835-0.845). Rotating s_setprio per step (WX_WET_PRIO_ROTATE=1) equalises the rates but marches the waves
// of a SIMD in lock-step into their memory waits (-5 %); giving older segments more rows (WX_WET_SKEW) shifts the rates with it.
// Halo columns are not the problem: an s_barrier per row step, which keeps the four strips of a workgroup on the same rows, lowers
// FETCH_SIZE by 2 % and costs 6 % of time -- the neighbouring strips' lines already hit in L2; what the counters show above the
// algorithmic bytes (reads 1.15x) is the 8 warm-up rows per segment.
#ifndef WX_WET_ROUNDS
#define WX_WET_ROUNDS 4
#endif
struct WetLaunch {
  int n_strips;
  WetSegs segs;
};
inline int wet_capacity()
{
  static int capacity = 0;
  if (!capacity) {
    int dev = 0, ncu = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_march_wet<false, false, true>, 64 * WX_WET_WPB, 0) != hipSuccess || ncu <= 0 || nb <= 0) {
      capacity = 256 * 4 * WX_WET_MINWAVES;
    } else {
      // the occupancy API can answer one workgroup per CU too many (MI355X_MICROARCH.md, "Residency"): bound it by what the kernel's own
      // register and LDS footprint admit -- 512 registers per lane and SIMD in granules of 8, 160 KiB of LDS per CU; a workgroup is
      // WX_WET_WPB waves, one per SIMD
      hipFuncAttributes fa;
      int by_regs = nb, by_lds = nb;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_march_wet<false, false, true>)) == hipSuccess) {
        if (fa.numRegs > 0) by_regs = (512 / (((fa.numRegs + 7) / 8) * 8)) * 4 / WX_WET_WPB;
        if (fa.sharedSizeBytes > 0) by_lds = (int)(163840 / fa.sharedSizeBytes);
        if (wx_tune_env("WX_MARCH_DEBUG")) fprintf(stderr, "[wx_wet] numRegs=%d lds=%zu: blocks/CU by registers %d, by LDS %d, occupancy API %d\n", fa.numRegs, fa.sharedSizeBytes, by_regs, by_lds, nb);
      }
      nb = std::max(1, std::min(nb, std::min(by_regs, by_lds)));
      capacity = ncu * nb * WX_WET_WPB;
    }
    if (wx_tune_env("WX_MARCH_DEBUG")) fprintf(stderr, "[wx_wet] CUs=%d blocks/CU=%d capacity=%d\n", ncu, nb, capacity);
  }
  return capacity;
}
inline double wet_alpha()
{
  double alpha = WX_WET_ALPHA_DEFAULT;
  if (const char *e = wx_tune_env("WX_WET_ALPHA")) alpha = atof(e) >= 1.0 ? atof(e) : alpha;
  return alpha;
}
// bands_mode (WX_OPT_ROW_BANDS): 0 = column blocks, 1 = row bands on grids at least 512 rows high (default), 2 = row bands wherever
// a band has two rows (tests: the shape of wide slabs on small grids)
inline WetLaunch wet_launch_shape(const Geo &g, int air_from_row, int bands_mode = 1)
{
  WetLaunch w;
  // Grids at least 512 rows high: XCD k takes the row band [k*Y/8, (k+1)*Y/8) of ALL strips instead of a column block of every
  // segment -- each XCD then streams one contiguous eighth of every plane: -3..-8 % kernel time at 16384x2048 and 32768x4096,
  // -5 % on a 4192x4096 slab, -9 % on a 2144x2048 one (profiles/r02_wet_tail_shape.txt).
  // Round 5, the reference's own sizes (tools/ref_sizes_bands.py, profiles/r05_ref_sizes.txt): NARROW grids below 512 rows gain from the
  // bands too -- 2500 x 300 (the reference's default: 45 strips) 45.3 -> 36.9 us per iteration, 5000 x 400 65.4 -> 60.0 -- while wide low
  // ones lose (16000 x 300: 120.5 -> 142.6): bands also for grids of fewer than 128 strips that are at least 128 rows high.
  // Round 6 (tools/ref_sizes_minrows.py, profiles/r06_ref_sizes_minrows.txt): the border re-measured on 17 sizes -- 7200 x 256 / x 448
  // (129 strips) -15 / -8 % with bands, 8000 x 300 ... 500 (143 strips) -5 % (x 350: +2.5 %), 10000 x 480 (179) +9 %, 12000 and 16000
  // columns even or worse: fewer than 146 strips. The minimum segment height below is at its optimum for every reference size.
  const int strips_all = (g.X + WOUT - 1) / WOUT;
  bool bands = bands_mode >= 2 ? g.Y >= 16 : (bands_mode != 0 && WX_WET_BANDS && (g.Y >= 8 * 64 || (strips_all < 146 && g.Y >= 128)));
  if (const char *e = wx_tune_env("WX_WET_BANDS")) bands = atoi(e) >= 2 ? g.Y >= 16 : (atoi(e) != 0 && g.Y >= 8 * 64); // (2: tests force it on small grids)
  const int Y = bands ? (g.Y + 7) / 8 : g.Y; // (bands: the shape of ONE band; the kernel clips it to the band's own height)
  w.segs.bands = bands ? 1 : 0;
  w.n_strips = (g.X + WOUT - 1) / WOUT;
  int rounds = WX_WET_ROUNDS, minrows = 24; // (narrow slabs: 2144x2048 runs 17 % faster with 30-row unit segments + tail than with equal 32-row ones)
  const double alpha = wet_alpha();
  if (const char *e = wx_tune_env("WX_WET_ROUNDS")) rounds = atoi(e) > 0 ? atoi(e) : rounds;
  // Grids that cannot fill the chip with 24-row segments (the reference's own sizes, 100 x 100 ... 2048 x 512): a wave's row steps
  // are serial, so parallelism is worth more than the 8 redundant warm-up rows of a segment -- rows per segment that fill the chip
  // once, at least 4 (2 while there is less than one wave per CU). 100x100: 64 -> 21 us per iteration, 1024x512: 87 -> 39,
  // 2048x512: 86 -> 51; from 4096x1024 on nothing changes (profiles/r04_small_grid_segments.txt).
  {
    const long long cap = wet_capacity(), per_row_set = (long long)w.n_strips * (bands ? 8 : 1);
    auto waves_at = [&](int m) { return per_row_set * ((Y + m - 1) / m); };
    if (waves_at(minrows) < cap) {
      const int fill = (int)(per_row_set * Y / cap), floor_rows = waves_at(4) <= 256 ? 2 : 4;
      minrows = std::max(floor_rows, std::min(minrows, fill));
    }
  }
  if (const char *e = wx_tune_env("WX_WET_MINROWS")) minrows = atoi(e) > 0 ? atoi(e) : minrows;
  // workgroups per segment: 8 XCD column blocks x ceil(strips of the block / waves per workgroup); the device holds capacity / WPB
  const int wg_per_seg = bands ? 8 * ((w.n_strips + WX_WET_WPB - 1) / WX_WET_WPB) : 8 * (((w.n_strips + 7) / 8 + WX_WET_WPB - 1) / WX_WET_WPB);
  int n_seg = (int)((long long)rounds * (wet_capacity() / WX_WET_WPB) / wg_per_seg);
  double skew = 0.0; // > 0: earlier-dispatched (lower) segments get more rows: the SIMD issues its OLDEST ready wave first
  if (const char *e = wx_tune_env("WX_WET_SKEW")) skew = atof(e);
  if (const char *e = wx_tune_env("WX_WET_SEG")) n_seg = atoi(e) > 0 ? (Y + atoi(e) - 1) / atoi(e) : n_seg;
  n_seg = n_seg < 1 ? 1 : (n_seg > WMAXSEG ? WMAXSEG : n_seg);
  if (n_seg > (Y + minrows - 1) / minrows) n_seg = (Y + minrows - 1) / minrows; // (8 warm-up rows per segment are redundant work)
  const int A = (air_from_row < 0 || air_from_row > Y) ? 0 : air_from_row;           // unknown: uniform segments
  const double total = alpha * A + (Y - A);
  // weights of the segments in dispatch order (bottom first): 1 + skew/2 .. 1 - skew/2, or an explicit list
  // WX_WET_SPEC="29x1,5x0.5,5x0.25" (count x weight, ...; the counts give the number of segments): short segments LAST shorten the
  // drain phase of the launch, in which finished waves are not replaced
  double wt[WMAXSEG];
  for (int sg = 0; sg < n_seg; sg++) wt[sg] = 1.0 + skew * (0.5 - (n_seg > 1 ? (double)sg / (n_seg - 1) : 0.5));
  if (WX_WET_TAIL && skew == 0.0 && !wx_tune_env("WX_WET_SEG") && !wx_tune_env("WX_WET_NOTAIL")) {
    // default shape: (rounds - 1) rounds of full segments, then about half a round each of segments of weight 1/2, 1/4 and 1/8.
    // The launch ends with a drain phase in which finished waves are not replaced; short segments at the end of the dispatch order
    // make it short: -5 % kernel time against equal segments (interleaved A/B, profiles/r02_wet_tail_shape.txt)
    const double per_round = (double)(wet_capacity() / WX_WET_WPB) / wg_per_seg;
    // (a band segment is a large part of a round already: one segment per tail level there)
    const int c = bands ? 1 : ((int)(per_round / 2.0 + 0.5) > 1 ? (int)(per_round / 2.0 + 0.5) : 1);
    const double rr_list[6] = {(double)rounds, 3.0, 2.0, 1.75, 1.5, 1.25};
    for (int ri = 0; ri < 6; ri++) { // (fewer rounds if a unit segment would fall below `minrows` rows; else equal segments)
      const double rr = rr_list[ri];
      if (rr > rounds) continue;
      int n_full = (int)((rr - 1.0) * per_round + 0.1); // (16384x2048: 3.9 -> 3 full segments of 68 rows per band: 0.770-0.772 against 0.789-0.826 ms with 4 of 54, two boxes)
      if (bands) { // unit segments of about 64 rows
        const int by_rows = (int)(Y / 64.0 + 0.5) - 1;
        n_full = n_full > by_rows ? n_full : by_rows;
        // ... and not below ~52 while the launch still holds well over one round of waves: at four waves per SIMD the rounds rule
        // alone asks for 5 x 44 rows + tail at 16384 x 2048; 4 x 52 + 1/2 + 1/4 + 1/8 is 1.3 % faster on one placement
        // (profiles/r04_ring_diet.txt). Narrow slabs stay with the short segments that fill the chip.
        const int by_unit = (int)(Y / 52.0 - 0.875 + 0.5);
        // (... "well over": three and a half rounds. The 8192-column slab of a 2-GPU run -- 148 strips, two rounds of 52-row units -- runs
        // 6.6 % faster with the ten 25-row units + tail the rounds rule gives it: profiles/r05_slab_segment_specs.txt)
        if (n_full > by_unit && by_unit >= 1 && (long long)w.n_strips * 8 * (by_unit + 3) >= 7LL * wet_capacity() / 2) n_full = by_unit;
      }
      const int n8 = bands ? (n_full >= 4 ? 1 : 0) : c - 1; // segments of weight 1/8
      const int n = n_full + 2 * c + n8;
      const double units = n_full + c * 0.5 + c * 0.25 + n8 * 0.125;
      if (n_full >= 1 && n <= WMAXSEG && Y / units >= minrows) {
        n_seg = n;
        for (int sg = 0; sg < n; sg++) wt[sg] = sg < n_full ? 1.0 : (sg < n_full + c ? 0.5 : (sg < n_full + 2 * c ? 0.25 : 0.125));
        break;
      }
    }
  }
  if (const char *e = wx_tune_env("WX_WET_SPEC")) {
    int n = 0;
    for (const char *q = e; *q && n < WMAXSEG;) {
      char *end = nullptr;
      const long cnt = strtol(q, &end, 10);
      if (end == q || *end != 'x') break;
      const double wv = strtod(end + 1, &end);
      for (long k = 0; k < cnt && n < WMAXSEG; k++) wt[n++] = wv > 0.0 ? wv : 1.0;
      q = *end == ',' ? end + 1 : end;
      if (*q == 0) break;
    }
    if (n > 0 && n <= (Y + 7) / 8) n_seg = n;
  }
  // border s at the row where the accumulated cost reaches the share of segments 0 .. s-1
  w.segs.start[0] = 0;
  double wsum = 0.0, wacc = 0.0;
  for (int sg = 0; sg < n_seg; sg++) wsum += wt[sg];
  for (int sg = 1; sg < n_seg; sg++) {
    wacc += wt[sg - 1];
    const double c = total * wacc / wsum;
    int y = c <= alpha * A ? (int)(c / alpha + 0.5) : A + (int)(c - alpha * A + 0.5);
    if (y <= w.segs.start[sg - 1]) y = w.segs.start[sg - 1] + 1;
    w.segs.start[sg] = y < Y ? y : Y;
  }
  w.segs.start[n_seg] = Y;
  while (n_seg > 1 && w.segs.start[n_seg - 1] >= Y) n_seg--; // (tiny grids: drop empty segments)
  w.segs.start[n_seg] = Y;
  w.segs.n_seg = n_seg;
  return w;
}

// the same shape with every segment cut in two (segments of fewer than 12 rows stay): for a launch of a few strips that runs NEXT TO a
// launch that fills the chip -- the edge strips of a slab beside its interior -- and should be done first: half the row steps per wave
inline WetLaunch wet_shape_halved(const WetLaunch &w)
{
  WetLaunch h = w;
  int n = 0;
  for (int sg = 0; sg < w.segs.n_seg; sg++) n += w.segs.start[sg + 1] - w.segs.start[sg] >= 12 ? 2 : 1;
  if (n > WMAXSEG) return w; // (the table is full: leave the shape as it is)
  n = 0;
  for (int sg = 0; sg < w.segs.n_seg; sg++) {
    const int a = w.segs.start[sg], b = w.segs.start[sg + 1];
    h.segs.start[n++] = a;
    if (b - a >= 12) h.segs.start[n++] = a + (b - a) / 2;
  }
  h.segs.start[n] = w.segs.start[w.segs.n_seg];
  h.segs.n_seg = n;
  return h;
}

// order (split iterations of a slab, StripOrder): returns the number of edge items of the launch (= the arrivals a gate kernel waits for)
inline int launch_march_wet(const WetLaunch &w, float iterNum, const FullCtx *ctx, const WetIn &in, const WetOut &out, const WetFixList &fix,
                             bool opt_out, bool quiet, hipStream_t stream, int strip_lo = 0, int strip_count = -1, int strip_lo2 = 0, int strip_count2 = 0,
                             const StripOrder *order = nullptr, const WetFixList *fix_edge = nullptr, const VxTrack *vx = nullptr)
{
  // (strip_count2 > 0: a second strip range in the same launch -- the two edges of a slab)
  const int ns1 = strip_count < 0 ? w.n_strips : strip_count, ns = ns1 + (strip_count2 > 0 ? strip_count2 : 0);
  if (ns <= 0) return 0;
  StripOrder ord{};
  if (order && order->mode != 0) ord = *order; // (an ordered launch covers the whole width: strip_lo 0, all strips)
  // 8 XCDs x (workgroups of the largest column block) x segments; surplus workgroups / waves exit at once
  int groups = ((w.segs.bands ? ns : (ns + 7) / 8) + WX_WET_WPB - 1) / WX_WET_WPB;
  if (ord.mode != 0 && ord.mode != 4) { // edge and interior strips are grouped into workgroups separately: up to one more group per XCD
    groups = 0;
    for (int k = 0; k < 8; k++) {
      const int a = w.segs.bands ? 0 : (k * ns) >> 3, b = w.segs.bands ? ns : ((k + 1) * ns) >> 3;
      groups = std::max(groups, strip_order_groups(ord, a, b, WX_WET_WPB));
    }
  }
  const dim3 grid(8 * groups * w.segs.n_seg);
  static bool dbg = wx_tune_env("WX_MARCH_DEBUG") != nullptr;
  if (dbg) {
    fprintf(stderr, "[wx_wet] strips=%d segs=%d waves=%d first/last segment rows=%d/%d\n", w.n_strips, w.segs.n_seg, w.n_strips * w.segs.n_seg,
            w.segs.start[1] - w.segs.start[0], w.segs.start[w.segs.n_seg] - w.segs.start[w.segs.n_seg - 1]);
    dbg = false;
  }
  const bool has_fb = in.fb != nullptr;
  const WetFixList fe = fix_edge ? *fix_edge : fix;
  VxTrack vt = vx ? *vx : VxTrack{nullptr, nullptr, 0.0f, 0, 0};
  if (vx && vx->zone_l < vx->zone_r) { // the watched zone arrives in COLUMNS: this kernel's strips are WOUT columns wide
    vt.zone_l = (vx->zone_l + WOUT - 1) / WOUT;
    vt.zone_r = vx->zone_r / WOUT;
  }
#define WX_LAUNCH_W(O, F, Q) \
  hipLaunchKernelGGL((k_march_wet<O, F, Q>), grid, dim3(64 * WX_WET_WPB), 0, stream, ctx, iterNum, in, out, fix, fe, ns, strip_lo, w.n_strips, w.segs, ns1, strip_lo2, ord, vt)
#define WX_LAUNCH_WQ(O, F) \
  do { if (quiet) WX_LAUNCH_W(O, F, true); else WX_LAUNCH_W(O, F, false); } while (0)
  if (opt_out) {
    if (has_fb) WX_LAUNCH_WQ(true, true); else WX_LAUNCH_WQ(true, false);
  } else {
    if (has_fb) WX_LAUNCH_WQ(false, true); else WX_LAUNCH_WQ(false, false);
  }
#undef WX_LAUNCH_WQ
#undef WX_LAUNCH_W
  return (ord.mode != 0 && ord.mode != 4) ? (ord.nl + (w.n_strips - ord.nr0)) * w.segs.n_seg * (w.segs.bands ? 8 : 1) : 0;
}

// the fix pass of a launch group: every list entry recomputed exactly, spread over the chip (exits at once while the list is empty)
inline void launch_wet_fix(float iterNum, const FullCtx *ctx, const WetIn &in, const WetOut &out, const WetFixList &fix, int *overflow, bool opt_out,
                           hipStream_t stream, int wgs = 0)
{
  // the whole chip (512 workgroups = one wavefront per list entry up to 2048 entries) while the last list the host has heard of held
  // entries, a corner of it while the lists are empty; any grid is correct (grid-stride loop)
  const int last = fix.hint ? *(volatile const int *)fix.hint_host : -1; // (-1: no host word -- the whole chip every time)
  // four entries per workgroup with room for four times the last list (its length varies from iteration to iteration: room for twice
  // cost the particle flow 6 us, 21 -> 27), 32 .. 512 workgroups (wgs: the edge group of a slab -- a few strips -- asks for 64); the
  // list is walked grid-stride, so any size is correct
  int want = last < 0 ? 512 : (last > 0 ? std::min(512, std::max(32, last)) : 32);
  if (const char *e = wx_tune_env("WX_FIX_WGS")) want = atoi(e) > 0 ? atoi(e) : want; // (tuning)
  const dim3 grid(wgs > 0 ? wgs : want), block(256);
  if (opt_out)
    hipLaunchKernelGGL((k_wet_fix<true>), grid, block, 0, stream, ctx, iterNum, in, out, fix.count, fix.cells, fix.cap, overflow, fix.hint);
  else
    hipLaunchKernelGGL((k_wet_fix<false>), grid, block, 0, stream, ctx, iterNum, in, out, fix.count, fix.cells, fix.cap, overflow, fix.hint);
}

// lowest row above which every cell is free air in the sense of air_cell(): 1 + the highest row holding a wall cell, a cell next
// to one, or a cell less than 9 rows above one (block-wise maximum into *out, which the caller zeroes)
__global__ void k_air_from_row(int X, int Y, const char4 *__restrict__ wall, int *__restrict__ out)
{
  int best = 0;
  const size_t n = (size_t)X * Y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const char4 w = wall[i];
    if (w.y <= 1 || w.z <= 9) {
      const int row = (int)(i / X) + 1;
      // (the top row always sees the wall row 0 through the y-wrap: it is not terrain, and one row does not matter to the cost model)
      if (row < Y) best = row > best ? row : best;
    }
  }
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(out, best);
}

} // namespace wx
