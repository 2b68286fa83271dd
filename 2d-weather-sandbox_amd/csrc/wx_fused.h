// wx_fused.h -- the iteration as TWO fused, LDS-tiled kernels (gfx950).
//
//   kernel A = velocity + curl + vorticity + boundary      (reference draws 1-4, app.js:5832-5878)
//   kernel B = advection + pressure + lighting             (reference draws 5-7, app.js:5881-5930)
//
// Each workgroup (256 threads = 4 wavefronts) owns a TX x TY = 64 x 16 tile. The chained stencils are
// evaluated on shrinking halo regions held in LDS, so intermediate textures (velocity output, curl,
// vortForce, advection output) never travel to HBM:
//   HBM traffic / cell-iteration:  A: 52 R + 40 W   B: 52 R + 52 W (+16 W when the post-advection base is
//   requested)  ~= 196 B vs ~380 B for the reference's pass structure.
// The per-cell arithmetic is the same wx_cells.h code as the per-pass kernels: results are bit-identical.
//
// Ping-pong in fused mode (a tile may not overwrite what neighbouring tiles still read as halo):
//   A: base[0], wall[0], water[1], light[0]  ->  base[1] (post-boundary), water[0], wall[1], curl
//   B: base[1], water[0], wall[1], light[src] -> base[0] (post-pressure), wall[0], water[1], light[dst],
//                                                base[2] (post-advection = the reference's baseTexture_1)
#pragma once
#include "wx_cells.h"

namespace wx {

constexpr int TX = 64, TY = 16, NT = 256;
constexpr bool kHaveFused = true;

struct FusedAIn {
  const float4 *base;
  const char4 *wall;
  const float4 *water, *light, *fb;
  const float2 *dep;
};
struct FusedBIn {
  const float4 *base, *water;
  const char4 *wall;
  const float4 *light;
};

__device__ __forceinline__ size_t fidx(int x, int y, int X) { return (size_t)y * X + x; }

// ================================================================================================
// kernel A
// ================================================================================================
namespace fa {
constexpr int HL = 2, HR = 3, HD = 2, HU = 3; // halo of the base_0 tile: velocity is needed on [-2,+2]^2 and reads P at +1
constexpr int BW = TX + HL + HR, BH = TY + HD + HU;
constexpr int WW = TX + 4, WH = TY + 4;       // wall tile, halo 2
constexpr int CW = TX + 3, CH = TY + 3;       // curl on x,y in [-2,+1]
constexpr int VW = TX + 1, VH = TY + 1;       // vortForce on x,y in [-1,0]
} // namespace fa

struct LBoundaryAcc {
  const float4 (*sb)[fa::BW];
  const char4 (*sw)[fa::WW];
  const float2 (*sv)[fa::VW];
  FusedAIn in;
  float4 w00;
  int X, Y, x, y, cx, cy;
  __device__ __forceinline__ float4 base(int dx, int dy) const { return sb[cy + fa::HD + dy][cx + fa::HL + dx]; }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return sw[cy + 2 + dy][cx + 2 + dx]; }
  __device__ __forceinline__ float2 vort(int dx, int dy) const { return sv[cy + 1 + dy][cx + 1 + dx]; }
  __device__ __forceinline__ float4 water(int dx, int dy) const
  {
    if (dx == 0 && dy == 0) return w00;
    return in.water[fidx(wrapi(x + dx, X), wrapi(y + dy, Y), X)]; // near-surface / wall cells only
  }
  __device__ __forceinline__ float4 light(int dy) const
  {
    int yy = y + dy;
    yy = yy < 0 ? 0 : (yy > Y - 1 ? Y - 1 : yy);
    return in.light[fidx(x, yy, X)];
  }
  __device__ __forceinline__ float4 fb() const { return in.fb ? in.fb[fidx(x, y, X)] : make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ float2 dep() const { return in.dep ? in.dep[fidx(x, y, X)] : make_float2(0.f, 0.f); }
};

__global__ __launch_bounds__(NT) void k_fused_a(Geo g, Uni u, const float *__restrict__ initial_T, FusedAIn in, float4 *__restrict__ base_out,
                                                float4 *__restrict__ water_out, char4 *__restrict__ wall_out, float *__restrict__ curl_out)
{
  using namespace fa;
  __shared__ float4 sb[BH][BW];
  __shared__ char4 sw[WH][WW];
  __shared__ float sc[CH][CW];
  __shared__ float2 sv[VH][VW];
  const int X = g.X, Y = g.Y;
  const int tid = threadIdx.x;
  const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY;

  // ---- stage 0: base_0 and wall_0 tiles with halo (REPEAT wrap on both axes) ----
  for (int i = tid; i < BW * BH; i += NT) {
    const int ly = i / BW, lx = i - ly * BW;
    sb[ly][lx] = in.base[fidx(wrapmod(tx0 + lx - HL, X), wrapmod(ty0 + ly - HD, Y), X)];
  }
  for (int i = tid; i < WW * WH; i += NT) {
    const int ly = i / WW, lx = i - ly * WW;
    sw[ly][lx] = in.wall[fidx(wrapmod(tx0 + lx - 2, X), wrapmod(ty0 + ly - 2, Y), X)];
  }
  __syncthreads();

  // ---- stage 1: velocity on [-2,+2]^2, in place (writes .xy, neighbours are only read for .z) ----
  for (int i = tid; i < WW * WH; i += NT) {
    const int ly = i / WW, lx = i - ly * WW;
    const float4 b = velocity_cell(u, sb[ly][lx], sb[ly][lx + 1].z, sb[ly + 1][lx].z, sw[ly][lx].y);
    sb[ly][lx].x = b.x;
    sb[ly][lx].y = b.y;
  }
  __syncthreads();

  // ---- stage 2: curl on [-2,+1]^2 ----
  for (int i = tid; i < CW * CH; i += NT) {
    const int ly = i / CW, lx = i - ly * CW;
    const float4 c = sb[ly][lx];
    sc[ly][lx] = curl_cell(c.x, c.y, sb[ly][lx + 1].y, sb[ly + 1][lx].x);
  }
  __syncthreads();

  // ---- stage 3: vortForce on [-1,0]^2 ----
  for (int i = tid; i < VW * VH; i += NT) {
    const int ly = i / VW, lx = i - ly * VW;
    sv[ly][lx] = vorticity_cell(sc[ly + 1][lx + 1], sc[ly + 1][lx], sc[ly + 1][lx + 2], sc[ly][lx + 1], sc[ly + 2][lx + 1]);
  }
  __syncthreads();

  // ---- stage 4: boundary on the tile ----
  const int cx = tid & (TX - 1);
  const int x = tx0 + cx;
  if (x >= X) return;
#pragma unroll
  for (int k = 0; k < TY / (NT / TX); k++) {
    const int cy = (tid / TX) + k * (NT / TX);
    const int y = ty0 + cy;
    if (y >= Y) break;
    const size_t gi = fidx(x, y, X);
    LBoundaryAcc a{sb, sw, sv, in, in.water[gi], X, Y, x, y, cx, cy};
    float4 b, w;
    char4 wl;
    boundary_cell(u, g, initial_T, x, y, a, b, w, wl);
    base_out[gi] = b;
    water_out[gi] = w;
    wall_out[gi] = wl;
    curl_out[gi] = sc[cy + 2][cx + 2];
  }
}

// ================================================================================================
// kernel B
// ================================================================================================
namespace fb_ {
// advection is evaluated on x,y in [-1,0] (pressure needs the left and lower neighbour); its 7-point velocity
// stencil reaches 1 further and the back-traced bilinear footprint (|v| < 1) one more: inputs on [-3,+2].
constexpr int HL = 3, HR = 2, HD = 3, HU = 2;
constexpr int IW = TX + HL + HR, IH = TY + HD + HU; // 69 x 21
constexpr int AW = TX + 1, AH = TY + 1;             // advection results on [-1,0]
} // namespace fb_

struct LAdvectAcc {
  const float4 (*sb)[fb_::IW];
  const float4 (*sq)[fb_::IW];
  const char4 (*sw)[fb_::IW];
  FusedBIn in;
  int X, Y, x, y, lx, ly; // lx, ly: position of the own cell inside the input tiles
  __device__ __forceinline__ float4 base(int dx, int dy) const { return sb[ly + dy][lx + dx]; }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return sw[ly + dy][lx + dx]; }
  __device__ __forceinline__ bool inside(int dx, int dy) const
  {
    return (unsigned)(lx + dx) < (unsigned)fb_::IW && (unsigned)(ly + dy) < (unsigned)fb_::IH;
  }
  __device__ __forceinline__ size_t gat(int dx, int dy) const { return fidx(wrapmod(x + dx, X), wrapmod(y + dy, Y), X); }
  // data-dependent footprint: LDS when it lies in the staged tile (|v| < 1 always does), global memory otherwise
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return inside(dx, dy) ? sb[ly + dy][lx + dx] : in.base[gat(dx, dy)]; }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const { return inside(dx, dy) ? sq[ly + dy][lx + dx] : in.water[gat(dx, dy)]; }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return inside(dx, dy) ? sw[ly + dy][lx + dx] : in.wall[gat(dx, dy)]; }
};

struct LLightAcc {
  const float4 (*sa)[fb_::AW]; // advection output base on [-1,0]
  const float4 *light_;
  float4 water_;
  char4 wall_;
  int X, x, cx, cy;
  __device__ __forceinline__ float T(int dy) const { return sa[cy + 1 + dy][cx + 1].w; } // dy in {0,-1}
  __device__ __forceinline__ float4 water() const { return water_; }
  __device__ __forceinline__ char4 wall() const { return wall_; }
  __device__ __forceinline__ float4 light_at(int dx, int j) const { return light_[fidx(wrapmod(x + dx, X), j, X)]; }
};

template <bool WRITE_DISP>
__global__ __launch_bounds__(NT) void k_fused_b(Geo g, Uni u, const float *__restrict__ initial_T, const float *__restrict__ snd_T,
                                                const float *__restrict__ snd_W, const float *__restrict__ snd_Vel, FusedBIn in,
                                                float4 *__restrict__ base_out, float4 *__restrict__ base_disp, float4 *__restrict__ water_out,
                                                char4 *__restrict__ wall_out, float4 *__restrict__ light_out)
{
  using namespace fb_;
  __shared__ float4 sb[IH][IW];
  __shared__ float4 sq[IH][IW];
  __shared__ char4 sw[IH][IW];
  __shared__ float4 sa[AH][AW]; // advection output: base
  __shared__ char4 saw[AH][AW]; // advection output: wall
  const int X = g.X, Y = g.Y;
  const int tid = threadIdx.x;
  const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY;

  // ---- stage 0: post-boundary base / water / wall tiles with halo ----
  for (int i = tid; i < IW * IH; i += NT) {
    const int ly = i / IW, lx = i - ly * IW;
    const size_t gi = fidx(wrapmod(tx0 + lx - HL, X), wrapmod(ty0 + ly - HD, Y), X);
    sb[ly][lx] = in.base[gi];
    sq[ly][lx] = in.water[gi];
    sw[ly][lx] = in.wall[gi];
  }
  __syncthreads();

  // ---- stage 1: advection on [-1,0]^2; own cells keep their water in registers ----
  const int cx = tid & (TX - 1);
  constexpr int RPT = TY / (NT / TX); // rows per thread
  float4 wreg[RPT];
  char4 wlreg[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NT / TX);
    const int x = wrapmod(tx0 + cx, X), y = wrapmod(ty0 + cy, Y);
    LAdvectAcc a{sb, sq, sw, in, X, Y, x, y, cx + HL, cy + HD};
    float4 b;
    advection_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, x, y, a, b, wreg[k], wlreg[k]);
    sa[cy + 1][cx + 1] = b;
    saw[cy + 1][cx + 1] = wlreg[k];
  }
  // left column (cx = -1, cy = -1..TY-1) and bottom row (cy = -1, cx = 0..TX-1): TX + TY + 1 extra cells
  if (tid < TX + TY + 1) {
    const int ecx = (tid < TX) ? tid : -1;
    const int ecy = (tid < TX) ? -1 : tid - TX - 1;
    const int x = wrapmod(tx0 + ecx, X), y = wrapmod(ty0 + ecy, Y);
    LAdvectAcc a{sb, sq, sw, in, X, Y, x, y, ecx + HL, ecy + HD};
    float4 b, w;
    char4 wl;
    advection_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, x, y, a, b, w, wl);
    sa[ecy + 1][ecx + 1] = b;
    saw[ecy + 1][ecx + 1] = wl;
  }
  __syncthreads();

  // ---- stage 2: pressure + lighting on the tile ----
  const int x = tx0 + cx;
  if (x >= X) return;
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int cy = (tid / TX) + k * (NT / TX);
    const int y = ty0 + cy;
    if (y >= Y) break;
    const size_t gi = fidx(x, y, X);
    const float4 b = sa[cy + 1][cx + 1];
    const float4 bd = sa[cy][cx + 1];
    const char4 wd = saw[cy][cx + 1];
    base_out[gi] = pressure_cell(b, sa[cy + 1][cx].x, bd.y, bd.w, wd.x, wd.y);
    if (WRITE_DISP) base_disp[gi] = b;
    water_out[gi] = wreg[k];
    wall_out[gi] = wlreg[k];
    LLightAcc la{sa, in.light, wreg[k], wlreg[k], X, x, cx, cy};
    light_out[gi] = lighting_cell(u, g, x, y, la);
  }
}

inline void launch_fused_a(const Geo &g, const Uni &u, const float *initial_T, const FusedAIn &in, float4 *base_out, float4 *water_out,
                           char4 *wall_out, float *curl_out, hipStream_t stream)
{
  const dim3 grid((g.X + TX - 1) / TX, (g.Y + TY - 1) / TY);
  hipLaunchKernelGGL(k_fused_a, grid, dim3(NT), 0, stream, g, u, initial_T, in, base_out, water_out, wall_out, curl_out);
}

inline void launch_fused_b(const Geo &g, const Uni &u, const float *initial_T, const float *snd_T, const float *snd_W, const float *snd_Vel,
                           const FusedBIn &in, float4 *base_out, float4 *base_disp, float4 *water_out, char4 *wall_out, float4 *light_out,
                           bool write_disp, hipStream_t stream)
{
  const dim3 grid((g.X + TX - 1) / TX, (g.Y + TY - 1) / TY);
  if (write_disp)
    hipLaunchKernelGGL(k_fused_b<true>, grid, dim3(NT), 0, stream, g, u, initial_T, snd_T, snd_W, snd_Vel, in, base_out, base_disp, water_out,
                       wall_out, light_out);
  else
    hipLaunchKernelGGL(k_fused_b<false>, grid, dim3(NT), 0, stream, g, u, initial_T, snd_T, snd_W, snd_Vel, in, base_out, base_disp, water_out,
                       wall_out, light_out);
}

} // namespace wx
