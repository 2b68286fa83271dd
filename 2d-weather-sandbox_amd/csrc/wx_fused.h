// wx_fused.h -- the iteration as TWO fused, LDS-tiled kernels (gfx950).
//
//   kernel A = velocity + curl + vorticity + boundary      (reference draws 1-4, app.js:5832-5878)
//   kernel B = advection + pressure + lighting             (reference draws 5-7, app.js:5881-5930)
//
// Each workgroup (256 threads = 4 wavefronts) owns a TX x TY = 64 x 16 tile. The chained stencils are
// evaluated on shrinking halo regions held in LDS, so intermediate textures (velocity output, curl,
// vortForce, advection output) never travel to HBM:
//   HBM traffic / cell-iteration:  A: 52 R + 36 W   B: 52 R + 52 W (+4 W curl, +16 W post-advection base only when
//   a consumer can see them: last iteration of a wx_step call / particles)  = 192 B vs ~380 B for the reference's
//   pass structure.
// The per-cell arithmetic is the same wx_cells.h code as the per-pass kernels: results are bit-identical.
//
// LDS layout: one fp32 PLANE per channel (structure of arrays, odd row stride). A wavefront reading one
// channel of 64 neighbouring cells touches 64 consecutive banks -- conflict free -- whereas a single-channel
// read out of an array of float4 is a 4-way bank conflict (measured: 56 % of the LDS cycles of the first
// version). Unused channels of an accessor call are dead code and cost nothing.
//
// Ping-pong in fused mode (a tile may not overwrite what neighbouring tiles still read as halo):
//   A: base[0], wall[0], water[1], light[0]  ->  base[1] (post-boundary), water[0], wall[1], curl
//   B: base[1], water[0], wall[1], light[src] -> base[0] (post-pressure), wall[0], water[1], light[dst],
//                                                base[2] (post-advection = the reference's baseTexture_1)
#pragma once
#include "wx_cells.h"

namespace wx {

constexpr int TX = 64, TY = 16;
#ifndef WX_NTA
#define WX_NTA 512
#endif
#ifndef WX_NTB
#define WX_NTB 512
#endif
#ifndef WX_B_MINWAVES
#define WX_B_MINWAVES 6
#endif
constexpr int NTA = WX_NTA; // kernel A: 47 VGPRs -> two 16-wave workgroups per CU = 8 waves/SIMD
constexpr int NTB = WX_NTB; // kernel B: three 8-wave workgroups per CU (52 KB LDS each) = 6 waves/SIMD, <= 80 VGPRs
constexpr bool kHaveFused = true;

// The light texture of the two-kernel path is stored as three planes: sunlight (x), net heating (y) and the two IR
// fluxes (zw). The boundary stage needs x and y only (8 instead of 16 B/cell); lighting reads x at its four filter taps
// and z / w of one row each, and writes all four channels. Both kernels are at the HBM ceiling, so bytes are time.
template <typename F, typename F2> struct LightPlanesT {
  F *x, *y;
  F2 *zw;
};
using LightPlanes = LightPlanesT<float, float2>;
using LightPlanesC = LightPlanesT<const float, const float2>;

struct FusedAIn {
  const float4 *base;
  const char4 *wall;
  const float4 *water;
  LightPlanesC light; // light_0 (boundaryShader samples lightTexture_0 only)
  const float4 *fb;
  const float2 *dep;
  // per 64x16 tile (the splat kernels use the same tiling): the feedback AND deposition textures are known to be all zero
  // there, so the tile neither loads them (24 B/cell) nor evaluates their terms (x +- 0, x / 1: same values)
  const unsigned char *fb_zero;
  int fb_txn;
};
// Copy-on-write of kernel A's water / wall outputs: away from terrain and without particle feedback the boundary pass
// returns both bit-identical to its inputs. Kernel A then leaves the 64x16 tile unwritten and sets clean[tile] = 1, and
// kernel B reads such tiles from A's INPUT buffers (water_alt / wall_alt). 20 of kernel A's 88 B/cell disappear for
// every clean tile; results are unchanged by construction (the skipped stores would have written the same bits).
struct FusedBIn {
  const float4 *base, *water;
  const char4 *wall;
  LightPlanesC light;      // the source light texture of this iteration's lighting pass
  const float4 *water_alt; // what kernel A read: valid wherever clean[tile] != 0
  const char4 *wall_alt;
  const unsigned int *clean; // per tile (one dword: read with scalar loads), row pitch txn; nullptr = every tile was written
  int txn;
  __device__ __forceinline__ bool is_clean(int gx, int gy) const { return clean && clean[(gy >> 4) * txn + (gx >> 6)]; }
  __device__ __forceinline__ float4 water_at(int gx, int gy, size_t gi) const { return (is_clean(gx, gy) ? water_alt : water)[gi]; }
  __device__ __forceinline__ char4 wall_at(int gx, int gy, size_t gi) const { return (is_clean(gx, gy) ? wall_alt : wall)[gi]; }
};
static_assert(TX == 64 && TY == 16, "is_clean() hard-codes the tile shape");

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

// Row-wise tile fill: wave w takes the rows w, w + NT/64, ... of a W x H region whose lower-left cell is (x_lo, y_lo);
// lane l takes the columns l and l + 64. The wrapped global column is computed once per lane and the row part of every
// address is wave-uniform (scalar), so an element costs ~4 vector instructions instead of ~18 for the flat
// "i -> (i / W, i % W) -> wrap -> address" form; the row loop is straight-line code (see the copy-on-write note).
template <int NT, int W, int H, bool SMALL, class F>
__device__ __forceinline__ void fill_rows(int tid, int x_lo, int y_lo, int X, int Y, F &&f) // f(ly, lx, global index)
{
  static_assert(W > 64 && W <= 128, "one full 64-column chunk plus a narrow remainder");
  constexpr int NW = NT / 64, ROUNDS = (H + NW - 1) / NW, E = W - 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int gx = SMALL ? wrapmod(x_lo + lane, X) : wrapfast(x_lo + lane, X);
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    // constant trip count and no branch: rows past the end are clamped (a few waves re-load row H-1 and store the same
    // values again), so that all loads of a thread are issued back to back
    const int ly = min(wave + r * NW, H - 1);
    const int gy = SMALL ? wrapmod(y_lo + ly, Y) : wrapfast(y_lo + ly, Y);
    f(ly, lane, fidx(gx, gy, X));
  }
  // the E remaining columns: one element per thread
  static_assert(E * H <= NT, "remainder fits one pass");
  if (tid < E * H) {
    const int ly = tid / E, lx = 64 + tid - ly * E;
    const int ex = SMALL ? wrapmod(x_lo + lx, X) : wrapfast(x_lo + lx, X);
    const int ey = SMALL ? wrapmod(y_lo + ly, Y) : wrapfast(y_lo + ly, Y);
    f(ly, lx, fidx(ex, ey, X));
  }
}

// The same decomposition for an in-LDS stage over a W x H region: f(ly, lx), every element exactly once (the velocity
// stage updates in place).
template <int NT, int W, int H, class F> __device__ __forceinline__ void for_rows(int tid, F &&f)
{
  static_assert(W > 64 && W <= 128, "one full 64-column chunk plus a narrow remainder");
  constexpr int NW = NT / 64, ROUNDS = (H + NW - 1) / NW, E = W - 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
  for (int r = 0; r < ROUNDS; r++)
    if (wave + r * NW < H) f(wave + r * NW, lane); // wave-uniform condition
  static_assert(E * H <= NT, "remainder fits one pass");
  if (tid < E * H) {
    const int ly = tid / E;
    f(ly, 64 + tid - ly * E);
  }
}

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

// ================================================================================================
// kernel A
// ================================================================================================
namespace fa {
constexpr int HL = 2, HR = 3, HD = 2, HU = 3; // halo of the base_0 tile: velocity is needed on [-2,+2]^2 and reads P at +1
constexpr int BW = TX + HL + HR, BH = TY + HD + HU; // 69 x 21
constexpr int WW = TX + 4, WH = TY + 4;             // wall tile, halo 2
constexpr int CW = TX + 3, CH = TY + 3;             // curl on x,y in [-2,+1]
constexpr int VW = TX + 1, VH = TY + 1;             // vortForce on x,y in [-1,0]
struct Smem {
  Planes4<BH, BW> b;
  char4 w[WH][WW + 1];
  float c[CH][CW];
  float vx[VH][VW], vy[VH][VW];
};
} // namespace fa

struct LBoundaryAcc {
  const fa::Smem &sm;
  FusedAIn in;
  float4 w00;
  int X, Y, x, y, cx, cy;
  bool tile_fb; // wave-uniform: this tile has particle feedback
  __device__ __forceinline__ float4 base(int dx, int dy) const { return sm.b.get(cy + fa::HD + dy, cx + fa::HL + dx); }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return sm.w[cy + 2 + dy][cx + 2 + dx]; }
  __device__ __forceinline__ float2 vort(int dx, int dy) const { return make_float2(sm.vx[cy + 1 + dy][cx + 1 + dx], sm.vy[cy + 1 + dy][cx + 1 + dx]); }
  __device__ __forceinline__ float4 water(int dx, int dy) const
  {
    if (dx == 0 && dy == 0) return w00;
    return in.water[fidx(wrapi(x + dx, X), wrapi(y + dy, Y), X)]; // near-surface / wall cells only
  }
  __device__ __forceinline__ float4 light(int dy) const
  {
    int yy = y + dy;
    yy = yy < 0 ? 0 : (yy > Y - 1 ? Y - 1 : yy);
    const size_t i = fidx(x, yy, X);
    return make_float4(in.light.x[i], in.light.y[i], 0.0f, 0.0f); // the boundary pass reads sunlight and net heating only
  }
  __device__ __forceinline__ float light_y0() const { return light(0).y; }
  __device__ __forceinline__ float light_x0() const { return light(0).x; }
  __device__ __forceinline__ float2 light_xy_up() const { const float4 l = light(1); return make_float2(l.x, l.y); }
  __device__ __forceinline__ bool has_fb() const { return tile_fb; }
  __device__ __forceinline__ float4 fb() const { return tile_fb ? in.fb[fidx(x, y, X)] : make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ float2 dep() const { return tile_fb ? in.dep[fidx(x, y, X)] : make_float2(0.f, 0.f); }
};

// WRITE_CURL: the curl texture has only display-side consumers (app.js:6081-6219), so it is stored only by the last
// iteration of a wx_step call.
// SMALL: tiny grids whose tile + halo may wrap more than once (general modulo instead of one compare; a template
// parameter because a uniform branch inside the fill loops keeps the loads of successive elements from overlapping).
template <bool WRITE_CURL, bool SMALL>
__global__ __launch_bounds__(NTA) void k_fused_a(Geo g, Uni u, const float *__restrict__ initial_T, FusedAIn in, float4 *__restrict__ base_out,
                                                float4 *__restrict__ water_out, char4 *__restrict__ wall_out, float *__restrict__ curl_out,
                                                unsigned int *__restrict__ clean_out)
{
  using namespace fa;
  __shared__ Smem sm;
  const int X = g.X, Y = g.Y;
  const int tid = threadIdx.x;
  int tbx, tby;
  tile_of_block(tiles_x(X), tbx, tby);
  const int tx0 = tbx * TX, ty0 = tby * TY;

  // ---- stage 0: base_0 and wall_0 tiles with halo (REPEAT wrap on both axes) ----
  fill_rows<NTA, BW, BH, SMALL>(tid, tx0 - HL, ty0 - HD, X, Y, [&](int ly, int lx, size_t gi) { sm.b.put(ly, lx, in.base[gi]); });
  fill_rows<NTA, WW, WH, SMALL>(tid, tx0 - 2, ty0 - 2, X, Y, [&](int ly, int lx, size_t gi) { sm.w[ly][lx] = in.wall[gi]; });
  __syncthreads();

  // ---- stage 1: velocity on [-2,+2]^2, in place (writes vx, vy; neighbours are only read for P) ----
  for_rows<NTA, WW, WH>(tid, [&](int ly, int lx) {
    const float4 b = velocity_cell(u, sm.b.get(ly, lx), sm.b.z[ly][lx + 1], sm.b.z[ly + 1][lx], sm.w[ly][lx].y);
    sm.b.x[ly][lx] = b.x;
    sm.b.y[ly][lx] = b.y;
  });
  __syncthreads();

  // ---- stage 2: curl on [-2,+1]^2 ----
  for_rows<NTA, CW, CH>(tid, [&](int ly, int lx) { sm.c[ly][lx] = curl_cell(sm.b.x[ly][lx], sm.b.y[ly][lx], sm.b.y[ly][lx + 1], sm.b.x[ly + 1][lx]); });
  __syncthreads();

  // ---- stage 3: vortForce on [-1,0]^2 ----
  for_rows<NTA, VW, VH>(tid, [&](int ly, int lx) {
    const float2 v = vorticity_cell(sm.c[ly + 1][lx + 1], sm.c[ly + 1][lx], sm.c[ly + 1][lx + 2], sm.c[ly][lx + 1], sm.c[ly + 2][lx + 1]);
    sm.vx[ly][lx] = v.x;
    sm.vy[ly][lx] = v.y;
  });
  __syncthreads();

  // ---- stage 4: boundary on the tile ----
  const bool tile_fb = in.fb != nullptr && !(in.fb_zero != nullptr && in.fb_zero[tby * in.fb_txn + tbx] != 0);
  const int cx = tid & (TX - 1);
  const int x = tx0 + cx;
  constexpr int RPT = TY / (NTA / TX);
  float4 wv[RPT];
  char4 wlv[RPT];
  bool same = true; // water and wall outputs of this thread's cells are bit-identical to the inputs
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NTA / TX);
    const int y = ty0 + cy;
    if (x >= X || y >= Y) continue;
    const size_t gi = fidx(x, y, X);
    LBoundaryAcc a{sm, in, in.water[gi], X, Y, x, y, cx, cy, tile_fb};
    float4 b, w;
    char4 wl;
#ifdef WX_ABL_NOBOUNDARY
    b = a.base(0, 0);
    w = a.w00;
    wl = a.wall(0, 0);
    b.x += a.vort(0, 0).x + a.vort(-1, 0).y + a.vort(0, -1).x;
#else
    boundary_cell(u, u.iterNum, u.iterI, g, initial_T, x, y, a, b, w, wl);
#endif
    base_out[gi] = b;
    if (WRITE_CURL) curl_out[gi] = sm.c[cy + 2][cx + 2];
    wv[k] = w;
    wlv[k] = wl;
    const char4 w0 = a.wall(0, 0);
    same = same && __float_as_int(w.x) == __float_as_int(a.w00.x) && __float_as_int(w.y) == __float_as_int(a.w00.y) &&
           __float_as_int(w.z) == __float_as_int(a.w00.z) && __float_as_int(w.w) == __float_as_int(a.w00.w) && wl.x == w0.x && wl.y == w0.y &&
           wl.z == w0.z && wl.w == w0.w;
  }
  const int tile_clean = clean_out ? __syncthreads_and(same) : 0;
  if (!tile_clean) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int y = ty0 + (tid / TX) + k * (NTA / TX);
      if (x >= X || y >= Y) continue;
      water_out[fidx(x, y, X)] = wv[k];
      wall_out[fidx(x, y, X)] = wlv[k];
    }
  }
  if (clean_out && tid == 0) clean_out[tby * tiles_x(X) + tbx] = (unsigned int)tile_clean;
}

// ================================================================================================
// kernel B
// ================================================================================================
#ifndef WX_B_UNI_MEM
#define WX_B_UNI_MEM 1 // measured: SGPR spills 38 -> 0, kernel B 0.697 -> 0.66 ms
#endif
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
struct SmemIn {
  Planes4<IH, IW> b, q; // post-boundary base, water
  char4 w[IH][IW + 1];
};
struct SmemOut { // advection output needed by neighbours (aliases SmemIn after a barrier)
  float vx[AH][AW], vy[AH][AW], T[AH][AW];
  char4 w[AH][AW + 1];
};
} // namespace fb_

// LDS-only accessor: valid when every back-trace of the cell is shorter than VMAX
struct LAdvectAcc {
  const fb_::SmemIn &sm;
  int lx, ly; // position of the own cell inside the input tiles
  __device__ __forceinline__ float4 base(int dx, int dy) const { return sm.b.get(ly + dy, lx + dx); }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return sm.w[ly + dy][lx + dx]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return sm.b.get(ly + dy, lx + dx); }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const { return sm.q.get(ly + dy, lx + dx); }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return sm.w[ly + dy][lx + dx]; }
};

// global-memory accessor for the rare cells whose back-trace leaves the tile
struct GAdvectAccB {
  FusedBIn in;
  int X, Y, x, y;
  __device__ __forceinline__ size_t at_off(int dx, int dy) const { return fidx(wrapmod(x + dx, X), wrapmod(y + dy, Y), X); }
  __device__ __forceinline__ float4 base(int dx, int dy) const { return in.base[at_off(dx, dy)]; }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return wall_off(dx, dy); }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return in.base[at_off(dx, dy)]; }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const
  {
    const int gx = wrapmod(x + dx, X), gy = wrapmod(y + dy, Y);
    return in.water_at(gx, gy, fidx(gx, gy, X));
  }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const
  {
    const int gx = wrapmod(x + dx, X), gy = wrapmod(y + dy, Y);
    return in.wall_at(gx, gy, fidx(gx, gy, X));
  }
};

// Context of the slow path, kept in device memory so that the out-of-line call passes pointers to global
// memory instead of forcing the kernel arguments through scratch. (advection does not use iterNum.)
struct SlowCtx {
  Geo g;
  Uni u;
  const float *initial_T, *snd_T, *snd_W, *snd_Vel;
  FusedBIn in;
};
struct AdvOut {
  float4 b, w;
  char4 wl;
};

__device__ __noinline__ AdvOut advection_cell_global(const SlowCtx *__restrict__ c, int x, int y)
{
  GAdvectAccB a{c->in, c->g.X, c->g.Y, x, y};
  AdvOut o;
  advection_cell(c->u, c->g, c->initial_T, c->snd_T, c->snd_W, c->snd_Vel, x, y, a, o.b, o.w, o.wl);
  return o;
}

// one region cell of stage 1
__device__ __forceinline__ void advect_tile_cell(const Uni &u, const Geo &g, const float *initial_T, const float *snd_T, const float *snd_W,
                                                 const float *snd_Vel, const fb_::SmemIn &sm, const SlowCtx *ctx, int x, int y, int lx, int ly,
                                                 float4 &b, float4 &w, char4 &wl)
{
  using namespace fb_;
  // the three staggered back-trace velocities are combinations of these eight (advectionShader.frag:85-89)
  const float m = fmaxf(fmaxf(fmaxf(fabsf(sm.b.x[ly][lx]), fabsf(sm.b.x[ly][lx - 1])), fmaxf(fabsf(sm.b.x[ly + 1][lx]), fabsf(sm.b.x[ly + 1][lx - 1]))),
                        fmaxf(fmaxf(fabsf(sm.b.y[ly][lx]), fabsf(sm.b.y[ly - 1][lx])), fmaxf(fabsf(sm.b.y[ly][lx + 1]), fabsf(sm.b.y[ly - 1][lx + 1]))));
#ifdef WX_ABL_NOADV
  b = sm.b.get(ly, lx);
  w = sm.q.get(ly, lx);
  wl = sm.w[ly][lx];
  return;
#endif
  if (m < VMAX) {
    LAdvectAcc a{sm, lx, ly};
    advection_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, x, y, a, b, w, wl);
  } else { // rare (|v| >= 1.9 cells/iteration, or NaN): back-trace may leave the staged tile
    const AdvOut o = advection_cell_global(ctx, x, y);
    b = o.b;
    w = o.w;
    wl = o.wl;
  }
}

struct LLightAcc {
  const fb_::SmemOut &so;
  const float4 *light_;
  float4 water_;
  char4 wall_;
  float T0;
  int X, x, cx, cy;
  __device__ __forceinline__ float T(int dy) const { return dy == 0 ? T0 : so.T[cy][cx + 1]; } // dy in {0,-1}
  __device__ __forceinline__ float4 water() const { return water_; }
  __device__ __forceinline__ char4 wall() const { return wall_; }
  __device__ __forceinline__ float4 light_at(int dx, int j) const { return light_[fidx(wrapfast(x + dx, X), j, X)]; }
  __device__ __forceinline__ float sun_at(int dx, int j) const { return light_at(dx, j).x; }
  __device__ __forceinline__ float ir_down_at(int j) const { return light_at(0, j).z; }
  __device__ __forceinline__ float ir_up_at(int j) const { return light_at(0, j).w; }
};
// the same over the planar light texture of the two-kernel path
struct LLightAccP {
  const fb_::SmemOut &so;
  LightPlanesC light_;
  float4 water_;
  char4 wall_;
  float T0;
  int X, x, cx, cy;
  __device__ __forceinline__ float T(int dy) const { return dy == 0 ? T0 : so.T[cy][cx + 1]; } // dy in {0,-1}
  __device__ __forceinline__ float4 water() const { return water_; }
  __device__ __forceinline__ char4 wall() const { return wall_; }
  __device__ __forceinline__ float sun_at(int dx, int j) const { return light_.x[fidx(wrapfast(x + dx, X), j, X)]; }
  __device__ __forceinline__ float ir_down_at(int j) const { return light_.zw[fidx(x, j, X)].x; }
  __device__ __forceinline__ float ir_up_at(int j) const { return light_.zw[fidx(x, j, X)].y; }
};

template <bool WRITE_DISP, bool SMALL>
__global__ __launch_bounds__(NTB, WX_B_MINWAVES) void k_fused_b(Geo g, Uni u_arg, const float *__restrict__ initial_T, const float *__restrict__ snd_T,
                                                const float *__restrict__ snd_W, const float *__restrict__ snd_Vel, FusedBIn in,
                                                const SlowCtx *__restrict__ ctx, float4 *__restrict__ base_out, float4 *__restrict__ base_disp, float4 *__restrict__ water_out,
                                                char4 *__restrict__ wall_out, LightPlanes light_out, float *__restrict__ t_disp)
{
  using namespace fb_;
  __shared__ union {
    SmemIn in;
    SmemOut out;
  } sm;
#if WX_B_UNI_MEM
  const Uni &u = ctx->u; // uniforms re-loaded from device memory on demand instead of living in (spilled) SGPRs; iterNum is not used by B's passes
#else
  const Uni &u = u_arg;
#endif
  const int X = g.X, Y = g.Y;
  const int tid = threadIdx.x;
  int tbx, tby;
  tile_of_block(tiles_x(X), tbx, tby);
  const int tx0 = tbx * TX, ty0 = tby * TY;
  constexpr bool small = SMALL;
  // copy-on-write flags of the 3x3 tiles this workgroup reads from: wave-uniform addresses -> scalar loads. (The host
  // passes clean == nullptr for grids so small that the halo wraps more than once.)
  unsigned int cflag = 0; // bit t: tile (t % 3 - 1, t / 3 - 1) was left unwritten by kernel A
  if (in.clean) {
    const int gtx = tiles_x(X), gty = (Y + TY - 1) / TY;
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int nx = wrapfast(tbx + t % 3 - 1, gtx), ny = wrapfast(tby + t / 3 - 1, gty);
      cflag |= (in.clean[ny * in.txn + nx] != 0u ? 1u : 0u) << t;
    }
  }


  // ---- stage 0: post-boundary base / water / wall tiles with halo ----
  // (flat index form: with three arrays per element the row-wise fill_rows() of kernel A measured 4 % slower here)
  for (int i = tid; i < IW * IH; i += NTB) {
    const int ly = i / IW, lx = i - ly * IW;
    const int gx = SMALL ? wrapmod(tx0 + lx - HL, X) : wrapfast(tx0 + lx - HL, X);
    const int gy = SMALL ? wrapmod(ty0 + ly - HD, Y) : wrapfast(ty0 + ly - HD, Y);
    const size_t gi = fidx(gx, gy, X);
    sm.in.b.put(ly, lx, in.base[gi]);
    // tiles kernel A left unwritten are read from A's inputs (pure arithmetic on the 9-bit mask: the fill loop must stay
    // branch-free for the loads of several elements to overlap)
    const bool cl = ((cflag >> (((ly >= HD) + (ly >= HD + TY)) * 3 + (lx >= HL) + (lx >= HL + TX))) & 1u) != 0u;
    sm.in.q.put(ly, lx, (cl ? in.water_alt : in.water)[gi]);
    sm.in.w[ly][lx] = (cl ? in.wall_alt : in.wall)[gi];
  }
  __syncthreads();

  // ---- stage 1: advection on [-1,0]^2, results in registers ----
  const int cx = tid & (TX - 1);
  constexpr int RPT = TY / (NTB / TX); // tile rows per thread
  float4 breg[RPT], wreg[RPT];
  char4 wlreg[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NTB / TX);
    const int x = small ? wrapmod(tx0 + cx, X) : wrapfast(tx0 + cx, X), y = small ? wrapmod(ty0 + cy, Y) : wrapfast(ty0 + cy, Y);
    advect_tile_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, sm.in, ctx, x, y, cx + HL, cy + HD, breg[k], wreg[k], wlreg[k]);
  }
  // left column (cx = -1, cy = -1..TY-1) and bottom row (cy = -1, cx = 0..TX-1): TX + TY + 1 extra cells
  float4 eb = make_float4(0.f, 0.f, 0.f, 0.f);
  char4 ewl = make_char4(0, 0, 0, 0);
  const bool extra = tid < TX + TY + 1;
  const int ecx = (tid < TX) ? tid : -1;
  const int ecy = (tid < TX) ? -1 : tid - TX - 1;
  if (extra) {
    const int x = small ? wrapmod(tx0 + ecx, X) : wrapfast(tx0 + ecx, X), y = small ? wrapmod(ty0 + ecy, Y) : wrapfast(ty0 + ecy, Y);
    float4 w;
    advect_tile_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, sm.in, ctx, x, y, ecx + HL, ecy + HD, eb, w, ewl);
  }
  __syncthreads(); // every thread is done reading the input tiles: their LDS is reused for the outputs

#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NTB / TX);
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

  // ---- stage 2: pressure + lighting on the tile ----
  const int x = tx0 + cx;
  if (x >= X) return;
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NTB / TX);
    const int y = ty0 + cy;
    if (y >= Y) break;
    const size_t gi = fidx(x, y, X);
    const float4 b = breg[k];
    const char4 wd = sm.out.w[cy][cx + 1];
    base_out[gi] = pressure_cell(b, sm.out.vx[cy + 1][cx], sm.out.vy[cy][cx + 1], sm.out.T[cy][cx + 1], wd.x, wd.y);
    if (WRITE_DISP) base_disp[gi] = b;
    if (t_disp) t_disp[gi] = b.w; // post-advection temperature for the droplets (wave-uniform pointer test)
    water_out[gi] = wreg[k];
    wall_out[gi] = wlreg[k];
#ifdef WX_ABL_NOLIGHT
    const float4 l = make_float4(in.light.x[gi], in.light.y[gi], in.light.zw[gi].x, in.light.zw[gi].y);
#else
    LLightAccP la{sm.out, in.light, wreg[k], wlreg[k], b.w, X, x, cx, cy};
    const float4 l = lighting_cell(u, g, x, y, la);
#endif
    light_out.x[gi] = l.x;
    light_out.y[gi] = l.y;
    light_out.zw[gi] = make_float2(l.z, l.w);
  }
}

// light texture: interleaved RGBA32F (the reference's layout: per-pass / single-kernel paths, readback, halo buffers of
// those paths) <-> the three planes of the two-kernel path
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

// fills the tiles of `dst` that kernel A left unwritten (clean != 0) from A's input buffer
__global__ __launch_bounds__(256) void k_cow_resolve(int X, int Y, const unsigned int *__restrict__ clean, const float4 *__restrict__ src,
                                                     float4 *__restrict__ dst)
{
  if (!clean[blockIdx.y * gridDim.x + blockIdx.x]) return;
  const int x = blockIdx.x * TX + (threadIdx.x & 63);
  if (x >= X) return;
  for (int cy = threadIdx.x >> 6; cy < TY; cy += 4) {
    const int y = blockIdx.y * TY + cy;
    if (y < Y) dst[fidx(x, y, X)] = src[fidx(x, y, X)];
  }
}

inline void launch_fused_a(const Geo &g, const Uni &u, const float *initial_T, const FusedAIn &in, float4 *base_out, float4 *water_out,
                           char4 *wall_out, float *curl_out, bool write_curl, unsigned int *clean_out, hipStream_t stream)
{
  const dim3 grid = tile_grid(g.X, g.Y);
  const bool small = (g.X < TX + 8) || (g.Y < TY + 8);
#define WX_LAUNCH_A(C, S) hipLaunchKernelGGL((k_fused_a<C, S>), grid, dim3(NTA), 0, stream, g, u, initial_T, in, base_out, water_out, wall_out, curl_out, clean_out)
  if (write_curl) {
    if (small) WX_LAUNCH_A(true, true); else WX_LAUNCH_A(true, false);
  } else {
    if (small) WX_LAUNCH_A(false, true); else WX_LAUNCH_A(false, false);
  }
#undef WX_LAUNCH_A
}

inline void launch_fused_b(const Geo &g, const Uni &u, const float *initial_T, const float *snd_T, const float *snd_W, const float *snd_Vel,
                           const FusedBIn &in, const SlowCtx *ctx, float4 *base_out, float4 *base_disp, float4 *water_out, char4 *wall_out,
                           const LightPlanes &light_out, bool write_disp, float *t_disp, hipStream_t stream)
{
  const dim3 grid = tile_grid(g.X, g.Y);
  const bool small = (g.X < TX + 8) || (g.Y < TY + 8);
#define WX_LAUNCH_B(D, S)                                                                                                                  \
  hipLaunchKernelGGL((k_fused_b<D, S>), grid, dim3(NTB), 0, stream, g, u, initial_T, snd_T, snd_W, snd_Vel, in, ctx, base_out, base_disp, \
                     water_out, wall_out, light_out, t_disp)
  if (write_disp) {
    if (small) WX_LAUNCH_B(true, true); else WX_LAUNCH_B(true, false);
  } else {
    if (small) WX_LAUNCH_B(false, true); else WX_LAUNCH_B(false, false);
  }
#undef WX_LAUNCH_B
}

} // namespace wx
