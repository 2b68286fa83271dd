// wx_tile.h -- what the tiled dry kernel (wx_dry.h), the row-marching kernels and the host code share: the 64 x 16 tile order,
// fp32 LDS planes, the planar layout of the light texture and the result record of the out-of-line exact advection.
// (Round 1's two fused LDS-tiled kernels for the wet iteration lived here; the single row-marching kernel of wx_wet.h replaced them,
// and the per-pass kernels of wx_kernels.h stay as the independent cross-check -- history in DESIGN.md section 5.)
#pragma once
#include "wx_cells.h"

#include <cstdlib>
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
