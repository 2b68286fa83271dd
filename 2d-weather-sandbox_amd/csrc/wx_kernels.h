// wx_kernels.h -- per-pass HIP kernels (one launch per reference draw call) and utility kernels.
//
// These are the parity baseline: each kernel is one reference pass with one thread per cell, coalesced
// float4 / char4 traffic and neighbour reads served by L1/L2. The row-marching kernels
// (wx_wet.h, wx_march.h) compute exactly the same per-cell arithmetic (wx_cells.h) with a fraction of the HBM traffic.
#pragma once
#include "wx_cells.h"
#include <hip/hip_fp16.h>

namespace wx {

constexpr int BX = 64; // one wavefront per row segment: 64 lanes x 16 B = 1 KiB coalesced
constexpr int BY = 4;

struct GridPtrs {
  const float4 *base, *water;
  const char4 *wall;
  const float2 *vort;
  const float4 *light;
  const float3 *fb; // (mass, heat, vapour): the reference's fourth channel is never written except at its mailbox texel (1,0)
  const float2 *dep;
};

__device__ __forceinline__ size_t cidx(int x, int y, int X) { return (size_t)y * X + x; }

// ---- velocity: base_0, wall_0 -> base_1, wall_1 (app.js:5832-5839) ----
__global__ __launch_bounds__(BX *BY) void k_velocity(Geo g, Uni u, const float4 *__restrict__ base_in, const char4 *__restrict__ wall_in,
                                                      float4 *__restrict__ base_out, char4 *__restrict__ wall_out)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= g.X || y >= g.Y) return;
  const int xr = (x + 1 == g.X) ? 0 : x + 1, yu = (y + 1 == g.Y) ? 0 : y + 1;
  const float4 b = base_in[cidx(x, y, g.X)];
  const char4 w = wall_in[cidx(x, y, g.X)];
  const float Pr = base_in[cidx(xr, y, g.X)].z, Pu = base_in[cidx(x, yu, g.X)].z;
  base_out[cidx(x, y, g.X)] = velocity_cell(u, b, Pr, Pu, w.y);
  wall_out[cidx(x, y, g.X)] = w;
}

// ---- curl (app.js:5842-5847) ----
__global__ __launch_bounds__(BX *BY) void k_curl(Geo g, const float4 *__restrict__ base_in, float *__restrict__ curl_out)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= g.X || y >= g.Y) return;
  const int xr = (x + 1 == g.X) ? 0 : x + 1, yu = (y + 1 == g.Y) ? 0 : y + 1;
  const float4 c = base_in[cidx(x, y, g.X)];
  curl_out[cidx(x, y, g.X)] = curl_cell(c.x, c.y, base_in[cidx(xr, y, g.X)].y, base_in[cidx(x, yu, g.X)].x);
}

// ---- vorticity (app.js:5850-5855) ----
__global__ __launch_bounds__(BX *BY) void k_vorticity(Geo g, const float *__restrict__ curl_in, float2 *__restrict__ vort_out)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= g.X || y >= g.Y) return;
  const int xr = (x + 1 == g.X) ? 0 : x + 1, yu = (y + 1 == g.Y) ? 0 : y + 1;
  const int xl = (x == 0) ? g.X - 1 : x - 1, yd = (y == 0) ? g.Y - 1 : y - 1;
  vort_out[cidx(x, y, g.X)] = vorticity_cell(curl_in[cidx(x, y, g.X)], curl_in[cidx(xl, y, g.X)], curl_in[cidx(xr, y, g.X)],
                                             curl_in[cidx(x, yd, g.X)], curl_in[cidx(x, yu, g.X)]);
}

// ---- boundary (app.js:5858-5878) ----
struct GBoundaryAcc {
  GridPtrs p;
  int X, Y, x, y;
  __device__ __forceinline__ size_t at(int dx, int dy) const { return cidx(wrapi(x + dx, X), wrapi(y + dy, Y), X); }
  __device__ __forceinline__ float4 base(int dx, int dy) const { return p.base[at(dx, dy)]; }
  __device__ __forceinline__ float4 water(int dx, int dy) const { return p.water[at(dx, dy)]; }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return p.wall[at(dx, dy)]; }
  __device__ __forceinline__ float2 vort(int dx, int dy) const { return p.vort[at(dx, dy)]; }
  __device__ __forceinline__ float4 light(int dy) const
  {
    int yy = y + dy;
    yy = yy < 0 ? 0 : (yy > Y - 1 ? Y - 1 : yy);
    return p.light[cidx(x, yy, X)];
  }
  __device__ __forceinline__ float light_y0() const { return light(0).y; }
  __device__ __forceinline__ float light_x0() const { return light(0).x; }
  __device__ __forceinline__ float2 light_xy_up() const { const float4 l = light(1); return make_float2(l.x, l.y); }
  __device__ __forceinline__ bool has_fb() const { return p.fb != nullptr; }
  __device__ __forceinline__ float4 fb() const
  {
    if (!p.fb) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float3 v = p.fb[cidx(x, y, X)];
    return make_float4(v.x, v.y, v.z, 0.f);
  }
  __device__ __forceinline__ float2 dep() const { return p.dep ? p.dep[cidx(x, y, X)] : make_float2(0.f, 0.f); }
};

__global__ __launch_bounds__(BX *BY) void k_boundary(Geo g, Uni u, const float *__restrict__ initial_T, GridPtrs in, float4 *__restrict__ base_out,
                                                      float4 *__restrict__ water_out, char4 *__restrict__ wall_out)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= g.X || y >= g.Y) return;
  GBoundaryAcc a{in, g.X, g.Y, x, y};
  float4 b, w;
  char4 wl;
  boundary_cell(u, u.iterNum, u.iterI, g, initial_T, x, y, a, b, w, wl);
  const size_t i = cidx(x, y, g.X);
  base_out[i] = b;
  water_out[i] = w;
  wall_out[i] = wl;
}

// ---- advection (app.js:5881-5890) ----
struct GAdvectAcc {
  GridPtrs p;
  int X, Y, x, y;
  __device__ __forceinline__ size_t at(int dx, int dy) const { return cidx(wrapi(x + dx, X), wrapi(y + dy, Y), X); }
  __device__ __forceinline__ size_t at_off(int dx, int dy) const { return cidx(wrapmod(x + dx, X), wrapmod(y + dy, Y), X); }
  __device__ __forceinline__ float4 base(int dx, int dy) const { return p.base[at(dx, dy)]; }
  __device__ __forceinline__ char4 wall(int dx, int dy) const { return p.wall[at(dx, dy)]; }
  __device__ __forceinline__ float4 base_off(int dx, int dy) const { return p.base[at_off(dx, dy)]; }
  __device__ __forceinline__ float4 water_off(int dx, int dy) const { return p.water[at_off(dx, dy)]; }
  __device__ __forceinline__ char4 wall_off(int dx, int dy) const { return p.wall[at_off(dx, dy)]; }
};

__global__ __launch_bounds__(BX *BY) void k_advection(Geo g, Uni u, const float *__restrict__ initial_T, const float *__restrict__ snd_T,
                                                       const float *__restrict__ snd_W, const float *__restrict__ snd_Vel, GridPtrs in,
                                                       float4 *__restrict__ base_out, float4 *__restrict__ water_out, char4 *__restrict__ wall_out)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= g.X || y >= g.Y) return;
  GAdvectAcc a{in, g.X, g.Y, x, y};
  float4 b, w;
  char4 wl;
  advection_cell(u, g, initial_T, snd_T, snd_W, snd_Vel, x, y, a, b, w, wl);
  const size_t i = cidx(x, y, g.X);
  base_out[i] = b;
  water_out[i] = w;
  wall_out[i] = wl;
}

// ---- pressure (app.js:5893-5900) ----
__global__ __launch_bounds__(BX *BY) void k_pressure(Geo g, const float4 *__restrict__ base_in, const char4 *__restrict__ wall_in,
                                                      float4 *__restrict__ base_out, char4 *__restrict__ wall_out)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= g.X || y >= g.Y) return;
  const int xl = (x == 0) ? g.X - 1 : x - 1, yd = (y == 0) ? g.Y - 1 : y - 1;
  const float4 b = base_in[cidx(x, y, g.X)];
  const float4 bd = base_in[cidx(x, yd, g.X)];
  const char4 wd = wall_in[cidx(x, yd, g.X)];
  base_out[cidx(x, y, g.X)] = pressure_cell(b, base_in[cidx(xl, y, g.X)].x, bd.y, bd.w, wd.x, wd.y);
  wall_out[cidx(x, y, g.X)] = wall_in[cidx(x, y, g.X)];
}

// ---- lighting (app.js:5903-5930) ----
struct GLightAcc {
  const float4 *base_, *water_;
  const char4 *wall_;
  const float4 *light_;
  int X, Y, x, y;
  __device__ __forceinline__ float T(int dy) const { return base_[cidx(x, wrapi(y + dy, Y), X)].w; }
  __device__ __forceinline__ float4 water() const { return water_[cidx(x, y, X)]; }
  __device__ __forceinline__ char4 wall() const { return wall_[cidx(x, y, X)]; }
  __device__ __forceinline__ float4 light_at(int dx, int j) const { return light_[cidx(wrapmod(x + dx, X), j, X)]; }
  __device__ __forceinline__ float sun_at(int dx, int j) const { return light_at(dx, j).x; }
  __device__ __forceinline__ float ir_down_at(int j) const { return light_at(0, j).z; }
  __device__ __forceinline__ float ir_up_at(int j) const { return light_at(0, j).w; }
};

__global__ __launch_bounds__(BX *BY) void k_lighting(Geo g, Uni u, const float4 *__restrict__ base_in, const float4 *__restrict__ water_in,
                                                      const char4 *__restrict__ wall_in, const float4 *__restrict__ light_in,
                                                      float4 *__restrict__ light_out)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= g.X || y >= g.Y) return;
  GLightAcc a{base_in, water_in, wall_in, light_in, g.X, g.Y, x, y};
  light_out[cidx(x, y, g.X)] = lighting_cell(u, g, x, y, a);
}

// ---- emittedLight: the lighting pass's second render target (lightingShader.frag:15, 60-78, 98-101, 143-166; RGBA16F) ----
// Only the display shaders read it, so the iteration kernels do not write it: it is recomputed on demand for the rectangle a
// consumer asks for, from what the last lighting pass read -- its source light texture (still intact: the pass wrote the other
// one), the post-advection water (= current) and the wall texture (pressure passes it through unchanged).
struct half4 {
  __half x, y, z, w;
};
template <bool PLANAR> struct GEmitAcc {
  const float4 *water_;
  const char4 *wall_;
  const float4 *light_;
  const float *sun_;
  int X, Y, x, y;
  __device__ __forceinline__ float T(int) const { return 0.0f; } // (feeds only the light output, which is discarded)
  __device__ __forceinline__ float4 water() const { return water_[cidx(x, y, X)]; }
  __device__ __forceinline__ char4 wall() const { return wall_[cidx(x, y, X)]; }
  __device__ __forceinline__ float sun_at(int dx, int j) const
  {
    const size_t i = cidx(wrapmod(x + dx, X), j, X);
    return PLANAR ? sun_[i] : light_[i].x;
  }
  __device__ __forceinline__ float ir_down_at(int) const { return 0.0f; }
  __device__ __forceinline__ float ir_up_at(int) const { return 0.0f; }
};

template <bool PLANAR>
__global__ __launch_bounds__(BX *BY) void k_emitted(Geo g, Uni u, int x0, int y0, int w, int h, const float4 *__restrict__ water_in,
                                                     const char4 *__restrict__ wall_in, const float4 *__restrict__ light_in,
                                                     const float *__restrict__ sun_in, half4 *__restrict__ out)
{
  const int x = x0 + blockIdx.x * BX + threadIdx.x, y = y0 + blockIdx.y * BY + threadIdx.y;
  if (x >= x0 + w || y >= y0 + h) return;
  GEmitAcc<PLANAR> a{water_in, wall_in, light_in, sun_in, g.X, g.Y, x, y};
  float4 e;
  lighting_cell<false, true>(u, g, x, y, a, &e);
  out[cidx(x, y, g.X)] = half4{__float2half_rn(e.x), __float2half_rn(e.y), __float2half_rn(e.z), __float2half_rn(e.w)};
}

// ---- device-side terrain generator (wx_setup_terrain): the 1-D part of setupShader.frag:26-61, 72, 74 ----
// rand / noise of the setup shader in double, operation for operation what weather_sandbox_amd.synth and host/sim_host.js evaluate on
// the host (the library is built with -ffp-contract=off: no fused multiply-adds, as on the host). One thread per local column; g = its global column.
__device__ inline double tn_rand(double n) // setupShader.frag:26
{
  const double v = sin(n) * 43758.5453123;
  return v - floor(v);
}
__device__ inline double tn_noise(double p) // setupShader.frag:28-33
{
  const double fl = floor(p), fc = p - fl;
  const double a = tn_rand(fl) * (1.0 - fc), b = tn_rand(fl + 1.0) * fc;
  return a + b - 0.5;
}
__global__ void k_terrain_columns(int Xl, int Xg, int col0, int Y, double seed, double height_mult, int snap, double sim_height, int *__restrict__ wall_rows,
                                  unsigned int *__restrict__ sea, double *__restrict__ veg_noise, float *__restrict__ snow)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Xl) return;
  int g = (col0 + x) % Xg;
  if (g < 0) g += Xg;
  const int gs = snap > 1 ? g / snap * snap : g; // terrain constant over `snap` columns
  double h = 0.0;
  if (height_mult < 0.05) { // all sea (setupShader.frag:44-46)
    h = 0.0;
  } else if (height_mult < 0.10) { // all land
    h = 0.005;
  } else { // hills / mountains: octaves of value noise (setupShader.frag:52-59)
    const double var = ((double)gs + 0.5) * 0.001;
    for (double i = 2.0; i < 1000.0; i *= 1.5) {
      const double r = tn_rand(seed + i) * 10.0;
      const double arg = var * i;
      h += tn_noise(arg + r) * 0.5 / i;
    }
    h *= height_mult;
  }
  // wall where texCoord.y < texelSize.y or texCoord.y < height: rows below max(1, ceil(h * Y - 0.5)), even thickness, 8 rows of air left
  long long nrows = (long long)ceil(h * (double)Y - 0.5);
  if (nrows < 1) nrows = 1;
  if (snap > 1) nrows = (nrows + snap - 1) / snap * snap;
  if (nrows > Y - 8) nrows = Y - 8;
  wall_rows[x] = (int)nrows;
  sea[x] = h < 1.0 / (double)Y ? 1u : 0u; // setupShader.frag:65
  const double fx = (double)g + 0.5;
  const double r0 = tn_rand(seed) * 10.0;
  veg_noise[x] = tn_noise(fx * 0.01 + r0) * 150.0;                                            // setupShader.frag:72
  const double sn = (h * sim_height - 2000.0) * 100.0 / 3000.0;                               // setupShader.frag:74: map_rangeC(height_m, 2000, 5000, 0, 100)
  snow[x] = (float)fmin(fmax(sn, 0.0), 100.0);
}

// ---- device-side initialiser (wx_setup_columns): the 2-D part of setupShader.frag:63-89 ----
// Column x is wall below row wall_rows[x] (sea if sea[x], else land), air above; per-row sounding for the air cells.
__global__ __launch_bounds__(BX *BY) void k_setup_columns(int X, int Y, const int *__restrict__ wall_rows, const unsigned int *__restrict__ sea,
                                                           const double *__restrict__ veg_noise, const float *__restrict__ snow,
                                                           const float *__restrict__ T_air, const float *__restrict__ tot,
                                                           const float *__restrict__ cloud, float4 *__restrict__ base, float4 *__restrict__ water,
                                                           char4 *__restrict__ wall)
{
  const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
  if (x >= X || y >= Y) return;
  const int nrows = wall_rows[x];
  const bool is_wall = y < nrows, is_sea = is_wall && sea[x] != 0u, is_land = is_wall && !is_sea;
  float4 b = make_float4(0.f, 0.f, 0.f, is_wall ? (is_sea ? 25.0f + 273.15f : 1000.0f) : T_air[y]);
  float4 w = make_float4(is_wall ? (is_sea ? 1002.0f : 1001.0f) : tot[y], is_wall ? 0.0f : cloud[y], is_land ? 25.0f : (is_sea ? 100.0f : 0.0f),
                         is_land ? snow[x] : 0.0f);
  int veg = 0;
  if (is_land) { // setupShader.frag:72: 110 - fragCoord.y * 2 + noise * 150, evaluated in double like the host generator
    const double v = 110.0 - ((double)y + 0.5) * 2.0 + veg_noise[x];
    veg = (int)fmin(fmax(trunc(v), 0.0), 127.0);
  }
  const int vdist = y - nrows + 1;
  const int wdist = is_wall ? 0 : min(max(vdist, 1), 127);
  const size_t i = cidx(x, y, X);
  base[i] = b;
  water[i] = w;
  wall[i] = make_char4((signed char)(sea[x] != 0u ? 2 : 1), (signed char)wdist, (signed char)min(max(vdist, -127), 127), (signed char)veg);
}

// ---- pass-through copies for masked-off passes (pass_mask) ----
__global__ void k_copy16(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_copy4(const char4 *__restrict__ src, char4 *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// Particles: precipitationShader.vert:66-293 + point rasterisation with ONE,ONE blending
// (app.js:5940-5953). One thread per droplet; splats are fp32 atomic adds (order-nondeterministic,
// exactly like the reference's blend unit).
//
// Column slabs (no reference counterpart). The pool is PARTITIONED: an active droplet is tracked by the rank whose owned
// columns contain it (plus, as a ghost copy, by a neighbour while it is within `halo` columns of the common edge); every other rank
// only knows "active elsewhere" (remote flag) and skips it after reading one byte. Inactive droplets are static records -- their
// state does not change until they spawn -- that every rank holds; the spawn probe (precipitationShader.vert:82-84) hashes to a
// position anywhere in the domain, so every rank tests every inactive record against its own columns (8 bytes + two integer hashes
// per record and iteration) and acts on the ones that land where its grid is valid (owned columns + halo - 6j ghost columns in
// iteration j of an exchange period: droplets in the overlap are processed redundantly, and identically, by both neighbours, like
// ghost cells). Nothing is communicated inside a period. At the halo exchange:
//   * every rank packs the droplets whose active / inactive status FLIPPED while it processed them (flips: one bit per iteration of
//     the period) and that it had in its owned columns at some point: (index, flip history, final record) -- a few hundred per period --
//     and the ranks all-gather these events (slab.py). Per droplet the report with the earliest first flip wins, then the one that saw
//     the most flips, then the lower rank: a rank that spawned a droplet from a stale inactive record after another rank had already
//     spawned it (it could not know) loses and drops its phantom; among equal histories the rank that processed the droplet last
//     wins. Everybody takes the winners' records;
//   * then every rank demotes the active droplets that are no longer in its owned columns and sends the ones within `halo` columns of
//     an edge to that neighbour (same batch of send / recv as the grid halos), which keeps them as ghost copies.
// The pool is updated IN PLACE on slab handles (a droplet is only ever touched by its own thread). Sprites never wrap around the
// domain edge (the reference clips them at the framebuffer): if the edge ("seam") runs through the local array, anchors right of it
// are stored one column further right and the box sum never crosses it.
// ------------------------------------------------------------------------------------------------
struct SlabP {
  int lo, hi;         // local columns [lo, hi): sample positions this rank processes in this iteration
  int own_lo, own_hi; // owned local columns
  int seam;           // local column of global column 0 if it lies strictly inside the local array, else 0
  int stamp;          // iteration of the exchange period (0-based) + 1; 0 = whole-domain handle
  const unsigned char *remote; // per droplet: tracked by another rank
  unsigned short *flips;       // per droplet: bit j set = active / inactive status flipped in iteration j of this period (as seen here)
  unsigned char *owned_once;   // per droplet: low 4 bits = last iteration (+1) of this period in which this rank processed it; bit 7 =
                               // processed inside the OWNED columns at least once (this rank reports its flips)
  int exact;                   // WX_OPT_POOL_EXACT: the ranks exchange their status flips and lightning requests after EVERY iteration
                               // (wx_pool_events_pack / _apply per iteration), so a lightning request counts once -- from the rank that
                               // has the spawn position in its OWNED columns -- and is accepted or rejected on the sum over all ranks
};
#ifndef WX_ABL_PRECIP
#define WX_ABL_PRECIP 0 // (timing builds, wrong results) 2: no deposit atomics, 4: the gathers read a fixed texel, 8: the pool is not written back
#endif
struct DevState {
  float inactiveDroplets; // the `inactiveDroplets` uniform, refreshed on the device every 600 iterations
  float lightning[4];     // lightningDataTexture (1x1 RGBA32F)
  float px_count;         // this iteration's 1-px blends into texel (0,0): +1 per still-inactive droplet
  float px_light[4];      // this iteration's 1-px blends into texel (1,0): lightning requests
  int scratch_int;        // small device-side result slot (terrain scan of the marching wet kernel)
  int ghost_nontrivial;   // set by k_halo_unpack when a neighbour's ghost columns carry water (or a negative vegetation byte)
                          // although the host asserted a water-free domain (wx_slab_assert_water_free): reported by the next blocking call
  int fix_overflow;       // set by k_wet_fix when more output cells needed the exact path than its list holds (the entry count)
  int pool_overflow;      // set when an exchange buffer of the partitioned droplet pool received more entries than it holds
  int pool_seen_max;      // largest status-flip count any rank reported since the transport last looked (k_pool_check)
  int fastest_bits;       // float bits of the largest |velocity component| that took a cell to the exact path (marching wet kernel)
  int pool_retired;       // exact mode: droplets that ended the iteration inactive among this iteration's status flips (k_pool_events_apply)
  float mailbox_w;        // fourth channel of the feedback texture's texel (1,0) -- the only texel whose alpha the reference ever writes (the
                          // lightning request's fourth component): the texture itself is stored with three channels (lightning_update keeps it)
  int vx_max_bits;        // float bits of the largest |vx| the marching kernels produced since the last roll (VxTrack, wx_tile.h; >= 0.5 only)
  int cone_violation;     // float bits of a |vx| that reached the limit the current exchange period was sized for (slabs; 0: none)
};
// exchange time: this slab's measured maximum goes to `out` (one word; all-gathered / copied to the host by the transport), the
// accumulator starts over. One thread.
__global__ void k_vx_roll(DevState *st, int *__restrict__ out)
{
  out[0] = atomicExch(&st->vx_max_bits, 0);
}

// Splat accumulation: a 12x12 point sprite anchored at pixel (i0,j0) adds the same value to pixels
// i0..i0+11 x j0..j0+11. Instead of 144 x 5 atomics per droplet the value is added ONCE at the anchor of a padded
// accumulation grid and a 12x12 box sum (k_splat_box) produces the feedback / deposition textures.
// Anchor range: i0 = ceil(xw - 6.5) in [-6, X-6]  ->  column q = i0 + 6 in [0, X]; rows likewise.
struct SplatGrid {
  float3 *acc3;  // (mass, heat, vapor) deposits, pitch AP (12-byte anchors: the fourth channel of the reference's blend target is never used)
  float2 *acc2;  // (rain, snow) deposits
  unsigned char *dirty;   // per 64x16 tile t of the accumulation grid: [t] holds deposits; [T + t] holds (rain, snow) deposits too
                          // (T = TXn * TYn; only droplets that reach the ground deposit there: a small part of the tiles)
  unsigned char *fb_zero; // per 64x16 tile t of the textures: [2t] feedback known to be all zero, [2t + 1] deposition known to be
  int AP, AH;    // pitch (>= X+1) and rows (>= Y+1) of the accumulation grid
  int TXn, TYn;  // tiles per row / column
  // work lists of one iteration (k_splat_classify): work[8*par + {0,1,2,3,4}] = number of feedback tiles to box-sum, feedback tiles
  // to zero, accumulation tiles to clear, deposition tiles to box-sum, deposition tiles to zero (par = iteration parity: the kernel
  // that fills one set of counters resets the other); the lists follow at work[16 + k*TXn*TYn], k = 0 .. 4
  int *work;
};
constexpr int STX = 64, STY = 16; // splat tile

// Deterministic splat order (wx_set_option(WX_OPT_SPLAT_ORDER, 1); tests): instead of fp32 atomics in arrival order every droplet
// RECORDS its deposit -- key = anchor index (r * AP + q), DET_KEY_LIGHT for a lightning request, DET_KEY_NONE for nothing -- the
// records are radix-sorted by key (stable: equal keys stay in droplet-index order) and k_splat_runs adds each run sequentially,
// i.e. every anchor receives ((d_i0 + d_i1) + d_i2) ... with i0 < i1 < i2 the droplets that hit it. The box sum that follows is
// order-free already (index-anchored trees), so the feedback / deposition textures become a pure function of the droplet pool.
struct DetSplat {
  int *key;   // per droplet (NULL: atomics)
  float *val; // per droplet 5 floats: (mass, heat, vapor | rain, snow), or the four lightning-request channels
};
constexpr int DET_KEY_NONE = 0x7fffffff;
__device__ __forceinline__ int det_key_light(const SplatGrid &sg) { return sg.AP * sg.AH; }

// common.glsl:103-111
__device__ __forceinline__ uint32_t hash_u32(uint32_t x)
{
  x += (x << 10u);
  x ^= (x >> 6u);
  x += (x << 3u);
  x ^= (x >> 11u);
  x += (x << 15u);
  return x;
}
// common.glsl:126-137
__device__ __forceinline__ float random2d(float sx, float sy)
{
  uint32_t h = hash_u32(__float_as_uint(sx) + hash_u32(__float_as_uint(sy)));
  h &= 0x007FFFFFu;
  h |= 0x3F800000u;
  const float r2 = __uint_as_float(h);
  return r2 - 1.0f * floorf(r2 / 1.0f);
}
// pow(x, 1./3.) for x > 0 as a fixed sequence of exactly-rounded operations (bit-reproducible CPU/GPU):
// bit-level seed + 4 Newton steps.
__device__ __forceinline__ float det_cbrt(float x)
{
  float y = __uint_as_float(__float_as_uint(x) / 3u + 709921077u);
#pragma unroll
  for (int i = 0; i < 4; i++) y = (2.0f * y + x / (y * y)) / 3.0f;
  return y;
}

// initRainDrops() (app.js:4901-4913) on the device: every droplet starts inactive, its five fields are random seeds --
// (r, r, -10 + r, r, r) with r uniform in [0, 1). The reference draws them from Math.random(); here field c of droplet i is a pure
// function of (seed, i, c) through the shaders' own integer hash (common.glsl:103-111), 24 random bits each, so every slab of a
// decomposed domain generates the same pool and a numpy restatement reproduces it bit for bit (tests).
__global__ void k_init_droplets(int n, uint32_t seed, float *__restrict__ d0, float *__restrict__ d1)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int c = 0; c < 5; c++) {
    const uint32_t h = hash_u32(seed + hash_u32((uint32_t)i * 5u + (uint32_t)c));
    float r = (float)(h >> 8) * (1.0f / 16777216.0f);
    if (c == 2) r = -10.0f + r;
    d0[(size_t)i * 5 + c] = r;
    d1[(size_t)i * 5 + c] = r;
  }
}

// texel of the LOCAL array that holds global texture coordinate (u, v); the column may lie outside the local array
// (>= g.X) on slab handles -- callers check local_col() first
__device__ __forceinline__ int local_col(const Geo &g, float u_)
{
  int lc = wrapmod((int)floorf(u_ * (float)g.Xg), g.Xg) - g.xoff;
  return lc < 0 ? lc + g.Xg : lc;
}
__device__ __forceinline__ size_t texel(const Geo &g, float u_, float v_)
{
  const int iy = wrapmod((int)floorf(v_ * (float)g.Y), g.Y);
#if WX_ABL_PRECIP & 4
  return (cidx(local_col(g, u_), iy, g.X) & 255) + (size_t)(g.Y / 2) * g.X;
#endif
  return cidx(local_col(g, u_), iy, g.X);
}

__device__ __forceinline__ void atomic_add_f(float *p, float v)
{
  if (v != 0.0f) unsafeAtomicAdd(p, v);
}

// Post-advection temperature of texel t on the marching wet kernel's path (t_in != nullptr): the pressure pass changes T in one kind of
// cell only -- directly above a land surface cell (pressure_cell: wd_dist == 0 && wd_type == 1, the snow-melt hand-off) -- so the
// marching kernel stores the post-advection temperature (t_in) only for rows that hold such a cell, and everywhere else the
// post-pressure texture's T (base_in) IS the post-advection value. The test here is the one pressure_cell makes, on the same
// post-advection wall texel of the cell below (rows wrap): wherever it holds, the row's t_in was stored.
__device__ __forceinline__ float precip_T(const Geo &g, size_t t, const float4 *__restrict__ base_in, const float *__restrict__ t_in,
                                          const char4 *__restrict__ wall_in)
{
  if (!t_in) return base_in[t].w;
  const size_t row = (size_t)g.X, tb = t >= row ? t - row : t + (size_t)(g.Y - 1) * row;
  const char4 wb = wall_in[tb];
  return (wb.y == 0 && wb.x == 1) ? t_in[t] : base_in[t].w;
}

// baseTexture_1 (post-advection base: what the display samples, app.js:6081-6110) of the marching wet kernel's display iteration, rows
// [y0, y0 + rows): the pressure pass (pressure_cell, wx_cells.h) leaves vx and vy alone, changes P everywhere and T only directly above a land
// surface cell -- so the iteration stores its post-advection P (p_disp) and, in rows that hold such a cell, its post-advection T (t_disp:
// the plane the droplets read through precip_T), and the texel is (vx, vy of the post-pressure texture, p_disp, precip_T). Bit-identical
// to the texture the iteration stored whole until round 5 (tests/test_gpu_parity.py).
__global__ void k_base_disp_assemble(int X, int Y, int y0, int rows, const float4 *__restrict__ base_cur, const float *__restrict__ p_disp, const float *__restrict__ t_disp,
                                     const char4 *__restrict__ wall_cur, float4 *__restrict__ out)
{
  const size_t n = (size_t)rows * X, off = (size_t)y0 * X;
  Geo g{};
  g.X = X;
  g.Y = Y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t t = off + i;
    const float4 b = base_cur[t];
    out[t] = make_float4(b.x, b.y, p_disp[t], precip_T(g, t, base_cur, t_disp, wall_cur));
  }
}

// one droplet of the pool: transform-feedback update + its deposit. Returns true for a droplet that stays inactive (the reference
// blends +1 per such droplet into texel (0,0), precipitationShader.vert:158-159: counted by the caller).
__device__ __forceinline__ bool precip_droplet(const int i, const Geo &g, const Uni &u, int n_drops, const float *__restrict__ drops_in,
                                               const float4 *__restrict__ base_in, const float4 *__restrict__ water_in,
                                               DevState *__restrict__ st, float *__restrict__ drops_out, const SplatGrid &sg, const SlabP &sp,
                                               const float *__restrict__ t_in, const DetSplat &det, const char4 *__restrict__ wall_in)
{
  // t_in != nullptr (marching wet kernel): base_in is the POST-pressure base texture, whose velocity components equal the
  // post-advection ones the reference samples (pressure_cell only touches P and T); the post-advection temperature comes from
  // precip_T: t_in where the pressure pass changed it, base_in everywhere else.
  bool live = i < n_drops;
  if (live && det.key) det.key[i] = DET_KEY_NONE; // (overwritten below if this droplet deposits something)
  const int Y = g.Y;
  const float resX = (float)g.Xg, resY = (float)Y;
  const float initalMass = 0.15f;
  bool count_inactive = false, claim = false;

  float feedback[4] = {0.f, 0.f, 0.f, 0.f}, deposition[2] = {0.f, 0.f};
  float size = 1.0f, gposx = -2.0f, gposy = -2.0f; // default: clipped

  if (live && sp.stamp) { // slab handle: is the droplet mine to look at, and is its sample position (spawn probe / own position) in my valid columns?
    if (sp.remote[i]) {
      live = false; // active on another rank
    } else {
      const float dpx = drops_in[5 * (size_t)i], m0 = drops_in[5 * (size_t)i + 2];
      const float tcx = m0 < 0.0f ? random2d(m0, dpx + u.iterNum * 0.3754f) : dpx / 2.0f + 0.5f;
      const int lc = local_col(g, tcx);
      if (lc < sp.lo || lc >= sp.hi)
        live = false; // not where my grid is valid: a static inactive record, or a ghost copy the neighbour tracks (the pool is in place)
      else
        claim = lc >= sp.own_lo && lc < sp.own_hi;
    }
  }

  if (live) {
    const float dpx = drops_in[5 * (size_t)i], dpy = drops_in[5 * (size_t)i + 1];
    const float m0 = drops_in[5 * (size_t)i + 2], m1 = drops_in[5 * (size_t)i + 3];
    const float density = drops_in[5 * (size_t)i + 4];
    float newPosx = dpx, newPosy = dpy, newM0 = m0, newM1 = m1, newDensity = density;
    bool isActive = true, spawned = false, lightningSpawned = false;
    float tcx = 0.f, tcy = 0.f, realTemp = 0.f;
    float4 base = make_float4(0.f, 0.f, 0.f, 0.f), water = base;

    if (m0 < 0.0f) { // inactive :72-162
      tcx = random2d(m0, dpx + u.iterNum * 0.3754f);
      tcy = random2d(m1, dpx + u.iterNum * 0.073162f);
      const size_t t = texel(g, tcx, tcy);
      // A spawn probe lands on a random texel: every plane it touches is a cache line of its own from HBM, and those random lines are
      // what the kernel's time goes into. Cloud water first: a probe into cloud-free air (most of them) is over without the
      // temperature -- it cannot pass either threshold; the velocity is only fetched by a probe that spawns.
      water = water_in[t];
      const bool cloudy = water.y > fminf(u.aboveZeroThreshold, u.subZeroThreshold);
      if (cloudy) {
        if (t_in)
          base.w = precip_T(g, t, base_in, t_in, wall_in);
        else
          base = base_in[t];
      }
      realTemp = base.w - tcy * u.dryLapse;
      const float threshold = (realTemp > CtoK(0.0f)) ? u.aboveZeroThreshold : u.subZeroThreshold;
      if (cloudy && water.y > threshold && base.w < 500.0f) {
        const float spawnChance = ((water.y - threshold) / (st->inactiveDroplets + 10.0f)) * resX * resY * u.spawnChanceMult;
        const float c10 = water.y * 10.0f;
        const float pw = c10 * c10; // pow(x, 2.0)
        const float nrmRand = pw - floorf(pw);
        if (spawnChance > nrmRand) {
          spawned = true;
          if (t_in) {
            const float4 b = base_in[t];
            base.x = b.x;
            base.y = b.y;
            base.z = b.z;
          }
          newPosx = (tcx - 0.5f) * 2.0f;
          newPosy = (tcy - 0.5f) * 2.0f;
          if (realTemp < CtoK(0.0f)) {
            newM0 = 0.0f;
            newM1 = initalMass;
            feedback[1] += newM1 * u.meltingHeat;
            newDensity = u.snowDensity;
            const float cloudPlusPrecipDensity = water.y + water.z;
            const float lightningSpawnChance = fmaxf((cloudPlusPrecipDensity - 2.5f) * 0.0033f, 0.0f);
            if (st->lightning[2] < u.iterNum - 30.0f && random2d(base.w * 0.2324f, water.x * 7.7f) < lightningSpawnChance) {
              lightningSpawned = true;
              isActive = false;
              size = 1.0f;
              feedback[0] = tcx;
              feedback[1] = tcy;
              feedback[2] = u.iterNum;
              feedback[3] = clampf(cloudPlusPrecipDensity / 10.0f + (random2d(tcx, tcy) - 0.5f), 0.01f, 4.0f);
              gposx = -1.0f + g.texX * 3.0f;
              gposy = -1.0f + g.texY;
            }
          } else {
            newM0 = initalMass;
            newM1 = 0.0f;
            newDensity = 1.0f;
          }
          feedback[2] -= initalMass;
        }
      }
      if (spawned) {
        if (!lightningSpawned) {
          size = 1.0f;
          gposx = newPosx;
          gposy = newPosy;
        }
      } else {
        isActive = false;
        count_inactive = true; // feedback[MASS] = 1 into texel (0,0): wave-aggregated below
      }
    }

    if (isActive) { // :164-288
      if (!spawned) {
        tcx = dpx / 2.0f + 0.5f;
        tcy = dpy / 2.0f + 0.5f;
        const size_t t = texel(g, tcx, tcy);
        water = water_in[t];
        base = base_in[t];
        if (t_in) base.w = precip_T(g, t, base_in, t_in, wall_in);
        realTemp = base.w - tcy * u.dryLapse;
      }
      const float totalMass = newM0 + newM1;
      if (totalMass < 0.04f) {
        feedback[1] = -(totalMass * u.evapHeat);
        feedback[2] = totalMass;
        newM0 = -2.0f - dpx;
        newM1 = dpy;
      } else if (newPosy < -1.0f || water.x > 1000.0f) {
        const size_t tu = texel(g, tcx, tcy + g.texY);
        if (precip_T(g, tu, base_in, t_in, wall_in) > 500.0f) newPosy += g.texY * 1.0f;
        deposition[0] = newM0;
        deposition[1] = newM1;
        newM0 = -2.0f - dpx;
        newM1 = dpy;
      } else {
        const float surfaceArea = det_cbrt(totalMass);
        const float growthRate = fmaxf(map_range(realTemp, CtoK(0.0f), CtoK(-30.0f), u.growthRate0C, u.growthRate_30C), u.growthRate0C);
        float growth = water.y * growthRate * surfaceArea;
        if (realTemp < CtoK(0.0f) && water.y > 0.0f && density == 1.0f) growth += surfaceArea * water.z * 0.0030f;
        feedback[2] -= growth * 1.0f;
        if (realTemp < CtoK(0.0f)) {
          newM1 += growth;
          feedback[1] += growth * u.meltingHeat;
          const float freezing = fminf((CtoK(0.0f) - realTemp) * u.freezingRate * surfaceArea, newM0);
          newM0 -= freezing;
          newM1 += freezing;
          feedback[1] += freezing * u.meltingHeat;
        } else {
          newM0 += growth;
          const float melting = fminf((realTemp - CtoK(0.0f)) * u.meltingRate * surfaceArea, newM1);
          newM1 -= melting;
          newM0 += melting;
          feedback[1] -= melting * u.meltingHeat;
          newDensity = fminf(newDensity + (melting / totalMass) * 1.00f, 1.0f);
        }
        float dropletTemp = base.w - tcy * u.dryLapse;
        if (newM1 > 0.0f) dropletTemp = fminf(dropletTemp, CtoK(0.0f));
        const float evapAndSubli = fmaxf((maxWater(dropletTemp) - water.x) * surfaceArea * u.evapRate, 0.0f);
        const float evap = fminf(newM0, evapAndSubli);
        const float subli = fminf(newM1, evapAndSubli - evap);
        newM0 -= evap;
        newM1 -= subli;
        feedback[2] += evap;
        feedback[2] += subli;
        feedback[1] -= evap * u.evapHeat;
        feedback[1] -= subli * u.evapHeat;
        feedback[1] -= subli * u.meltingHeat;

        newPosx += base.x / resX * 2.0f;
        newPosy += base.y / resY * 2.0f;
        newPosy -= u.fallSpeed * newDensity * sqrtf(totalMass / surfaceArea);
        {
          const float t = newPosx + 1.0f;
          newPosx = (t - 2.0f * floorf(t / 2.0f)) - 1.0f;
        }
        feedback[0] = totalMass;
      }
      const float pntSize = 12.0f, pntSurface = 12.0f * 12.0f;
      feedback[0] /= pntSurface;
      feedback[1] /= pntSurface;
      feedback[2] /= pntSurface;
      deposition[0] /= pntSize;
      deposition[1] /= pntSize;
      size = pntSize;
      gposx = newPosx;
      gposy = newPosy;
    }

#if WX_ABL_PRECIP & 8
    if (newPosx == 12345.f) // (timing build: the pool is not written back)
#endif
    {
      drops_out[5 * (size_t)i + 0] = newPosx;
      drops_out[5 * (size_t)i + 1] = newPosy;
      drops_out[5 * (size_t)i + 2] = newM0;
      drops_out[5 * (size_t)i + 3] = newM1;
      drops_out[5 * (size_t)i + 4] = fmaxf(newDensity, 0.0f);
    }
    if (sp.stamp) { // what the exchange needs to know: did the status flip, and did I have the droplet in my owned columns
      // ... or was the record of an inactive droplet rewritten without a flip: a droplet that spawns inside a cell it cannot live in (a wall
      // cell that holds cloud water: water.x > 1000) retires in the same iteration with NEW seeds (-2 - x, y, :199-203), and the other ranks
      // must probe with those from now on (found by tools/fuzz_parity.py --mode group on grids a few dozen rows high, where the wall rows at
      // the top are a tenth of the domain). Its event ends inactive, so the 600-iteration count leaves it out as the reference does (:158).
      if ((m0 >= 0.0f) != (newM0 >= 0.0f) || (spawned && newM0 < 0.0f)) sp.flips[i] |= (unsigned short)(1u << (sp.stamp - 1));
      sp.owned_once[i] = (unsigned char)((sp.owned_once[i] & 0x80) | sp.stamp | (claim ? 0x80 : 0));
    }
  }

  if (!live || count_inactive) return count_inactive;

  // point sprite: clip test on the centre (precipitationShader.vert gl_Position / gl_PointSize)
  if (!(gposx >= -1.0f && gposx <= 1.0f && gposy >= -1.0f && gposy <= 1.0f)) return false;
  const float xw = (gposx + 1.0f) * 0.5f * resX, yw = (gposy + 1.0f) * 0.5f * resY;
  if (size <= 1.0f) {
    // the only 1-px sprite that reaches this point is a lightning request, drawn at pixel (1,0)
    // (precipitationShader.vert:135-139)
    if (sp.stamp && sp.exact && !claim) return false; // (exact slabs: the owner of the spawn position files the request, once)
    if (det.key) {
      det.key[i] = det_key_light(sg);
      for (int c = 0; c < 4; c++) det.val[5 * (size_t)i + c] = feedback[c];
      return false;
    }
    atomic_add_f(&st->px_light[0], feedback[0]);
    atomic_add_f(&st->px_light[1], feedback[1]);
    atomic_add_f(&st->px_light[2], feedback[2]);
    atomic_add_f(&st->px_light[3], feedback[3]);
    return false;
  }
  // 12x12 sprite: every pixel whose centre lies in [w - 6, w + 6); one deposit at the anchor pixel
  int q = (int)ceilf(xw - 6.0f - 0.5f) + 6;
  const int r = (int)ceilf(yw - 6.0f - 0.5f) + 6;
  // global anchor column -> local accumulation column (see SlabP::seam)
  if (sp.seam > 0 && xw < (float)g.xoff)
    q += sp.seam + 1; // low-x side of the domain edge: right of the seam, stored one column further right
  else
    q -= g.xoff;
  if (q < 0 || q >= sg.AP) return false; // the sprite lies outside this slab
  const size_t ai = (size_t)r * sg.AP + q;
  if (det.key) {
    det.key[i] = (int)ai;
    det.val[5 * (size_t)i + 0] = feedback[0];
    det.val[5 * (size_t)i + 1] = feedback[1];
    det.val[5 * (size_t)i + 2] = feedback[2];
    det.val[5 * (size_t)i + 3] = deposition[0];
    det.val[5 * (size_t)i + 4] = deposition[1];
    return false;
  }
#if WX_ABL_PRECIP & 2
  return false;
#endif
  float *f = reinterpret_cast<float *>(sg.acc3 + ai);
  atomic_add_f(f + 0, feedback[0]);
  atomic_add_f(f + 1, feedback[1]);
  atomic_add_f(f + 2, feedback[2]);
  const int tile = (r / STY) * sg.TXn + (q / STX);
  sg.dirty[tile] = 1;
  // (rain, snow): only a droplet that reaches the ground deposits -- adding the zeros of all the others would be two more atomics
  // per droplet and, worse, would make every tile under a cloud a deposition tile to box-sum, write and clear (x + 0 == x: same sums)
  if (deposition[0] != 0.0f || deposition[1] != 0.0f) {
    float *d = reinterpret_cast<float *>(sg.acc2 + ai);
    atomic_add_f(d + 0, deposition[0]);
    atomic_add_f(d + 1, deposition[1]);
    sg.dirty[sg.TXn * sg.TYn + tile] = 1;
  }
  return false;
}

// The pool is walked by a fixed number of workgroups in chunks of 256 droplets (grid-stride): one workgroup per chunk meant 16 384
// wavefronts of a few microseconds each at 1 M droplets, and the launch lived on ~1 500 resident waves of the 8 192 the chip holds
// (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES, profiles/r03_particles_precip_counters.txt) whatever its gathers and atomics cost.
__global__ __launch_bounds__(256) void k_precipitation(Geo g, Uni u, int n_drops, const float *__restrict__ drops_in,
                                                        const float4 *__restrict__ base_in, const float4 *__restrict__ water_in,
                                                        DevState *__restrict__ st, float *__restrict__ drops_out, SplatGrid sg, SlabP sp,
                                                        const float *__restrict__ t_in, DetSplat det, const char4 *__restrict__ wall_in)
{
  // t_in != nullptr (two-kernel path): base_in is the POST-pressure base texture, whose velocity components equal the
  // post-advection ones the reference samples (pressure_cell only touches P and T), and t_in holds the post-advection
  // temperature -- kernel B then stores 4 instead of 16 extra bytes per cell for the droplets.
  int count = 0; // still-inactive droplets seen by this wave (wave-uniform)
  for (int base_i = blockIdx.x * 256; base_i < n_drops; base_i += gridDim.x * 256) {
    const bool c = precip_droplet(base_i + (int)threadIdx.x, g, u, n_drops, drops_in, base_in, water_in, st, drops_out, sg, sp, t_in, det, wall_in);
    count += __popcll(__ballot(c));
  }
  // one atomic per workgroup (integers: exact in fp32 in any order)
  __shared__ int wave_count[4];
  if ((threadIdx.x & 63) == 0) wave_count[threadIdx.x >> 6] = count;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int c = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
    if (c) unsafeAtomicAdd(&st->px_count, (float)c);
  }
}

// deterministic splat order: one thread per sorted record; the first record of a run of equal keys adds the run in order
__global__ __launch_bounds__(256) void k_splat_runs(int n, const int *__restrict__ key_sorted, const int *__restrict__ idx_sorted,
                                                     const float *__restrict__ val, SplatGrid sg, DevState *__restrict__ st)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int k = key_sorted[t];
  if (k == DET_KEY_NONE || (t > 0 && key_sorted[t - 1] == k)) return;
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const bool light = k == det_key_light(sg);
  for (int u = t; u < n && key_sorted[u] == k; u++) {
    const float *v = val + 5 * (size_t)idx_sorted[u];
    for (int c = 0; c < (light ? 4 : 5); c++) a[c] += v[c];
  }
  if (light) {
    for (int c = 0; c < 4; c++) st->px_light[c] += a[c];
    return;
  }
  sg.acc3[k] = make_float3(a[0], a[1], a[2]);
  const int r = k / sg.AP, q = k - r * sg.AP, tile = (r / STY) * sg.TXn + (q / STX);
  sg.dirty[tile] = 1;
  if (a[3] != 0.0f || a[4] != 0.0f) { // (as in precip_droplet: a run that deposits no rain / snow leaves the zero that is there)
    sg.acc2[k] = make_float2(a[3], a[4]);
    sg.dirty[sg.TXn * sg.TYn + tile] = 1;
  }
}

// 12x12 box sum of the deposits -> precipitationFeedbackTexture (RGBA32F) + precipitationDepositionTexture (RG32F).
// out(i,j) = sum of acc(q,r) for q in [i-5, i+6], r in [j-5, j+6] (anchor q = i0+6 covers pixels i0..i0+11), columns first.
// One workgroup per 64x16 output tile; tiles whose 3x3 neighbourhood holds no deposits only (re)write zeros, and
// not even that when the texture tile is already known to be zero. Replaces the per-iteration clear of both
// textures (app.js:5933-5934) and the blend-unit splats.
// seam > 0 (slab handle whose local array contains the domain edge at a tile boundary): sums never cross it, anchors
// right of it sit one column further right; then the (0,0) / (1,0) mailbox texels are not injected either.
// Which tiles need work this iteration? One thread per tile: texture tiles whose 3x3 neighbourhood of accumulation tiles holds
// deposits are box-summed, texture tiles that still hold the previous iteration's feedback are zeroed, accumulation tiles with
// deposits are cleared afterwards. (One workgroup per tile with an early exit cost 0.2 ms at 16384 x 2048: 32768 workgroups, each
// two dependent global round trips long, three per CU at a time because of the LDS the box sum needs.)
// append t to a work list for the lanes that want it: one atomic per wavefront instead of one per lane (30 000 increments of one
// address were most of this kernel's 8 us). All lanes of the wave must call it (converged).
__device__ __forceinline__ void wave_append(int *counter, int *list, bool want, int t)
{
  const unsigned long long mask = __ballot(want);
  if (!mask) return;
  const int lane = threadIdx.x & 63, first = __ffsll((long long)mask) - 1;
  int base = 0;
  if (lane == first) base = atomicAdd(counter, __popcll(mask));
  base = __shfl(base, first);
  if (want) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = t;
}
__global__ __launch_bounds__(256) void k_splat_classify(int X, int Y, SplatGrid sg, int mailbox, int par)
{
  const int T = sg.TXn * sg.TYn, t = blockIdx.x * blockDim.x + threadIdx.x;
  int *cnt = sg.work + 8 * par;
  if (t == 0)
    for (int k = 0; k < 5; k++) sg.work[8 * (par ^ 1) + k] = 0;
  // (no early return: wave_append needs the whole wave)
  const bool tile = t < T;
  const int tc = tile ? t : 0, tby = tc / sg.TXn, tbx = tc - tby * sg.TXn;
  wave_append(&cnt[2], sg.work + 16 + 2 * T, tile && sg.dirty[tc] != 0, tc);
  const bool tex = tile && tbx * STX < X && tby * STY < Y; // (the accumulation grid is one anchor wider / higher than the texture)
  int a = 0, a2 = 0;
  if (tex)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int ax = tbx + dx, ay = tby + dy;
        if (ax >= 0 && ay >= 0 && ax < sg.TXn && ay < sg.TYn) {
          a |= sg.dirty[ay * sg.TXn + ax];
          a2 |= sg.dirty[T + ay * sg.TXn + ax];
        }
      }
  const bool corner = mailbox && tc == 0;
  wave_append(&cnt[0], sg.work + 16, tex && a != 0, tc);
  wave_append(&cnt[1], sg.work + 16 + T, tex && a == 0 && (!sg.fb_zero[2 * tc] || corner), tc);
  wave_append(&cnt[3], sg.work + 16 + 3 * T, tex && a2 != 0, tc);
  wave_append(&cnt[4], sg.work + 16 + 4 * T, tex && a2 == 0 && !sg.fb_zero[2 * tc + 1], tc);
}

// One (tile, texture) work item: KIND 0 = the feedback texture from acc3, KIND 1 = the deposition texture
// from acc2. Splitting a tile by texture brings the staging buffer from 41 to 25 KB (6 instead of 3 workgroups per CU: the kernel
// is latency-bound between its load, sum and store phases) and doubles the number of independent items.
template <int KIND>
__device__ __forceinline__ void splat_box_tile(float (*pl)[STY + 11][77], int X, int Y, const SplatGrid &sg, const DevState *__restrict__ st,
                                               float3 *__restrict__ fb, float2 *__restrict__ dep, int x0, int y0, int qmin, int qmax, int qshift, bool corner)
{
  constexpr int WW = STX + 11, WH = STY + 11, NCH = KIND == 0 ? 3 : 2;
  const int tid = threadIdx.x;
  // stage the deposit window (zero outside the accumulation grid). All loads of a thread are issued before the first LDS write:
  // as a rolled loop this was load -> wait -> write eight times in a row, i.e. eight memory latencies per tile.
  constexpr int NST = (WW * WH + 255) / 256;
  float3 v4[KIND == 0 ? NST : 1];
  float2 v2[KIND == 0 ? 1 : NST];
#pragma unroll
  for (int k = 0; k < NST; k++) {
    const int i = tid + 256 * k, ly = i / WW, lx = i - ly * WW;
    const int q = x0 - 5 + lx, r = y0 - 5 + ly;
    const bool inside = i < WW * WH && q >= qmin && r >= 0 && q <= qmax && r <= Y;
    if (KIND == 0) {
      v4[k] = make_float3(0.f, 0.f, 0.f);
      if (inside) v4[k] = sg.acc3[(size_t)r * sg.AP + q + qshift];
    } else {
      v2[k] = make_float2(0.f, 0.f);
      if (inside) v2[k] = sg.acc2[(size_t)r * sg.AP + q + qshift];
    }
  }
#pragma unroll
  for (int k = 0; k < NST; k++) {
    const int i = tid + 256 * k, ly = i / WW, lx = i - ly * WW;
    if (i < WW * WH) {
      if (KIND == 0) {
        pl[0][ly][lx] = v4[k].x;
        pl[1][ly][lx] = v4[k].y;
        pl[2][ly][lx] = v4[k].z;
      } else {
        pl[0][ly][lx] = v2[k].x;
        pl[1][ly][lx] = v2[k].y;
      }
    }
  }
  __syncthreads();
  // A 12-sum as a tree whose nodes depend on the absolute index only -- pair sums p[i] = a[i] + a[i+1], quads q[i] = p[i] + p[i+2],
  // s[i] = (q[i] + q[i+4]) + q[i+8] -- so that neighbouring outputs share their partial sums (4 additions per output instead of 11)
  // and the result does not depend on where the tile or slab boundaries lie.
  // vertical pass: one (channel, window column) per thread, the 16 sums written back in place
  float vres[STY];
  const bool vtask = tid < NCH * WW;
  const int vc = tid / WW, vx = tid - vc * WW;
  if (vtask) {
    float a[WH];
#pragma unroll
    for (int i = 0; i < WH; i++) a[i] = pl[vc][i][vx];
#pragma unroll
    for (int i = 0; i < WH - 1; i++) a[i] = a[i] + a[i + 1];
#pragma unroll
    for (int i = 0; i < WH - 3; i++) a[i] = a[i] + a[i + 2];
#pragma unroll
    for (int j = 0; j < STY; j++) vres[j] = (a[j] + a[j + 4]) + a[j + 8];
#pragma unroll
    for (int j = 0; j < STY; j++) pl[vc][j][vx] = vres[j]; // (the column is this thread's alone until the barrier)
  }
  __syncthreads();
  // horizontal pass: a run of 4 output cells of one row per thread, all channels of the texture
  const int run = (tid & 7) + 8 * ((tid >> 5) & 1), row = ((tid >> 3) & 3) + 4 * (tid >> 6);
  float res[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    float a[15];
#pragma unroll
    for (int k = 0; k < 15; k++) a[k] = pl[c][row][run * 4 + k];
#pragma unroll
    for (int k = 0; k < 14; k++) a[k] = a[k] + a[k + 1];
#pragma unroll
    for (int k = 0; k < 12; k++) a[k] = a[k] + a[k + 2];
#pragma unroll
    for (int m = 0; m < 4; m++) res[c][m] = (a[m] + a[m + 4]) + a[m + 8];
  }
  // The sums go back through the staging buffer so that a wavefront stores whole rows: written straight from the run-of-four layout,
  // every store instruction scattered 16-byte pieces at a 64-byte stride -- four times the write requests, and the kernel's stores
  // alone took as long as everything else in it (timing builds, profiles/r03_particles_ablation.txt).
  __syncthreads(); // every thread has read its window columns
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int m = 0; m < 4; m++) pl[c][row][run * 4 + m] = res[c][m];
  __syncthreads();
  const int cx = tid & 63, cyg = tid >> 6;
#pragma unroll
  for (int k = 0; k < STY / 4; k++) {
    const int ry = cyg + 4 * k, x = x0 + cx, y = y0 + ry;
    if (x < X && y < Y) {
      if (KIND == 0) {
        float3 v = make_float3(pl[0][ry][cx], pl[1][ry][cx], pl[2][ry][cx]);
        if (corner && y == 0 && x == 0) v.x += st->px_count;
        if (corner && y == 0 && x == 1) { // (the fourth component of the request: DevState::mailbox_w, lightning_update)
          v.x += st->px_light[0];
          v.y += st->px_light[1];
          v.z += st->px_light[2];
        }
        fb[(size_t)y * X + x] = v;
      } else {
        dep[(size_t)y * X + x] = make_float2(pl[0][ry][cx], pl[1][ry][cx]);
      }
    }
  }
  __syncthreads(); // (the next item re-uses the staging buffer)
}

__global__ __launch_bounds__(256, 4) void k_splat_box(int X, int Y, SplatGrid sg, const DevState *__restrict__ st, float3 *__restrict__ fb,
                                                   float2 *__restrict__ dep, int seam, int mailbox, int par)
{
  constexpr int WH = STY + 11, PW = 77; // (pitch 77: the 4 rows x 8 runs a half-wave reads in the horizontal pass hit 32 banks)
  __shared__ float pl[3][WH][PW];
  const int tid = threadIdx.x;
  const int T = sg.TXn * sg.TYn, n_box = sg.work[8 * par], n_zero = sg.work[8 * par + 1], n_box2 = sg.work[8 * par + 3], n_zero2 = sg.work[8 * par + 4];
  const int cx = tid & 63, cyg = tid >> 6;
  // items: (tile, texture) for the tiles to box-sum, then the tiles to zero, dealt out round-robin. The launch must not hold more
  // workgroups than the chip does at once (splat_box_grid): with 2048 of them on a chip that holds 1536, the 512 of the second round
  // started their twelve items when the first round had finished. (Handing the items out through an atomic counter instead costs more
  // than it balances: 24 000 returning atomics on one address serialise, 0.37 instead of 0.25 ms.)
  // items: feedback tiles to box-sum, deposition tiles to box-sum (few: only where droplets reach the ground), feedback tiles to
  // zero, deposition tiles to zero
  const int e0 = n_box, e1 = e0 + n_box2, e2 = e1 + n_zero, e3 = e2 + n_zero2;
  for (int wi = blockIdx.x; wi < e3; wi += gridDim.x) {
    const int kind = wi < e0 ? 0 : (wi < e1 ? 1 : (wi < e2 ? 2 : 3));
    const int tile = kind == 0 ? sg.work[16 + wi] : (kind == 1 ? sg.work[16 + 3 * T + (wi - e0)] : (kind == 2 ? sg.work[16 + T + (wi - e1)] : sg.work[16 + 4 * T + (wi - e2)]));
    const int tby = tile / sg.TXn, tbx = tile - tby * sg.TXn;
    const int x0 = tbx * STX, y0 = tby * STY;
    const bool corner = mailbox && (tbx == 0 && tby == 0);
    const bool right = seam > 0 && x0 >= seam;
    const int qmin = right ? seam : 0, qmax = (seam > 0 && !right) ? seam : X, qshift = right ? 1 : 0;
    if (kind == 2) { // the texture tile holds the feedback of an earlier iteration (or the mailbox texels): zero it
      for (int k = 0; k < STY / 4; k++) {
        const int x = x0 + cx, y = y0 + cyg + 4 * k;
        if (x < X && y < Y) {
          float3 v = make_float3(0.f, 0.f, 0.f);
          if (corner && y == 0 && x == 0) v.x = st->px_count;
          if (corner && y == 0 && x == 1) v = make_float3(st->px_light[0], st->px_light[1], st->px_light[2]);
          fb[(size_t)y * X + x] = v;
        }
      }
      if (tid == 0) sg.fb_zero[2 * tile] = corner ? 0 : 1;
    } else if (kind == 3) { // ... the deposition of an earlier iteration
      for (int k = 0; k < STY / 4; k++) {
        const int x = x0 + cx, y = y0 + cyg + 4 * k;
        if (x < X && y < Y) dep[(size_t)y * X + x] = make_float2(0.f, 0.f);
      }
      if (tid == 0) sg.fb_zero[2 * tile + 1] = 1;
    } else if (kind == 0) {
      splat_box_tile<0>(pl, X, Y, sg, st, fb, dep, x0, y0, qmin, qmax, qshift, corner);
      if (tid == 0) sg.fb_zero[2 * tile] = 0;
    } else {
      splat_box_tile<1>(pl, X, Y, sg, st, fb, dep, x0, y0, qmin, qmax, qshift, corner);
      if (tid == 0) sg.fb_zero[2 * tile + 1] = 0;
    }
  }
}

// workgroups of a k_splat_box launch: what the device holds at once
inline int splat_box_grid()
{
  static int grid = 0;
  if (!grid) {
    if (const char *e = wx_tune_env("WX_SPLAT_BOX_WGS")) grid = atoi(e); // (tuning)
  }
  if (grid <= 0) {
    int dev = 0, ncu = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_splat_box, 256, 0) != hipSuccess || ncu <= 0 || nb <= 0)
      grid = 1024;
    else {
      hipFuncAttributes fa; // (the occupancy API can answer one workgroup per CU too many: bound it by the LDS the kernel really uses)
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_splat_box)) == hipSuccess && fa.sharedSizeBytes > 0) nb = std::min(nb, (int)(160 * 1024 / fa.sharedSizeBytes));
      grid = ncu * std::max(nb, 1);
    }
  }
  return grid;
}

// zero the accumulation tiles that hold deposits (after k_splat_box consumed them)
struct LightningArgs { // (the one-thread lightningLocation pass rides in the clear kernel: one launch less per iteration)
  float iterNum;
  int refresh_inactive, mailbox, defer;
  const float3 *fb;
  DevState *st;
};
__device__ __forceinline__ void lightning_update(float iterNum, int refresh_inactive, const float3 *__restrict__ fb, DevState *st, int mailbox, int defer);
__global__ __launch_bounds__(256) void k_splat_clear(int X, int Y, SplatGrid sg, int par, LightningArgs la)
{
  if (la.st && blockIdx.x == 0 && threadIdx.x == 0) lightning_update(la.iterNum, la.refresh_inactive, la.fb, la.st, la.mailbox, la.defer);
  const int T = sg.TXn * sg.TYn, n = sg.work[8 * par + 2];
  for (int wi = blockIdx.x; wi < n; wi += gridDim.x) {
    const int tile = sg.work[16 + 2 * T + wi];
    const bool rain = sg.dirty[T + tile] != 0; // (uniform; the flag is reset below, behind the barrier)
    const int tby = tile / sg.TXn, tbx = tile - tby * sg.TXn;
    const int x0 = tbx * STX, y0 = tby * STY;
    for (int i = threadIdx.x; i < STX * STY; i += 256) {
      const int q = x0 + (i & 63), r = y0 + (i >> 6);
      if (q < sg.AP && r < sg.AH) {
        sg.acc3[(size_t)r * sg.AP + q] = make_float3(0.f, 0.f, 0.f);
        if (rain) sg.acc2[(size_t)r * sg.AP + q] = make_float2(0.f, 0.f);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) sg.dirty[tile] = sg.dirty[T + tile] = 0;
  }
}

// lightningLocationShader.frag:24-38 + the 600-iteration inactive count (app.js:5957-5966), one thread
// (slab handles: the mailbox texels are not part of the local feedback texture -- the request is taken from the
// accumulator directly, the inactive count is refreshed by wx_pool_edges_pack)
__device__ __forceinline__ void lightning_update(float iterNum, int refresh_inactive, const float3 *__restrict__ fb, DevState *st, int mailbox, int defer)
{
  if (defer) { // exact slabs: the request stays in px_light until the ranks have summed theirs (k_pool_exact_resolve)
    st->px_count = 0.f;
    return;
  }
  // texel (1,0); its fourth channel (0 + the request's, as the blend unit adds it to the cleared texture) lives in the state
  const float4 n = mailbox ? make_float4(fb[1].x, fb[1].y, fb[1].z, 0.0f + st->px_light[3]) : make_float4(st->px_light[0], st->px_light[1], st->px_light[2], st->px_light[3]);
  if (mailbox) st->mailbox_w = n.w;
  st->px_count = 0.f; // consumed by k_splat_box
  st->px_light[0] = st->px_light[1] = st->px_light[2] = st->px_light[3] = 0.f;
  if (refresh_inactive && mailbox) st->inactiveDroplets = fb[0].x;
  if (n.z < fmaxf(iterNum - 1.0f, 1.0f) || n.z > iterNum) return; // discard
  st->lightning[0] = n.x;
  st->lightning[1] = n.y;
  st->lightning[2] = n.z;
  st->lightning[3] = n.w;
}

// ---- slab particle pool exchange (see SlabP) ----
struct PoolEvent { // 32 bytes
  int gid;
  int key;      // smaller wins: first flip << 18 | (15 - number of flips) << 14 | (15 - last iteration processed) << 10 | rank
  float rec[5]; // final record of the reporting rank
  int pad;
};
struct PoolRec { // 24 bytes: an active droplet handed to a neighbour as ghost copy
  int gid;
  float rec[5];
};
// buffers start with a 16-byte header whose first int is the number of entries
constexpr int POOL_HDR = 16;

// exact mode: the first thread also files this rank's ITERATION RECORD -- an event with gid -1 carrying the lightning request it
// collected (px_light, then cleared) and the real deposit at the domain's texel (0,0) (what the 600-iteration count reads on top of
// the inactive count, app.js:5957-5966; NULL on ranks that do not own global column 0). The record has a FIXED place, entry 0 -- the
// host starts the entry counter at 1 in these modes (pool_events_pack_mode) --, so that its readers (k_pool_exact_resolve every
// iteration, k_pool_lightning_latest every period: one thread each, on the critical path) look at one location instead of scanning up
// to 64 K entries per rank for it, and so that an overflowing rank can never lose it
__global__ void k_pool_events_pack(int n, int rank, int cap, unsigned short *__restrict__ flips, unsigned char *__restrict__ owned_once,
                                   const float *__restrict__ drops, int *__restrict__ hdr, PoolEvent *__restrict__ ev, DevState *st, int exact,
                                   const float3 *__restrict__ fb00)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (exact && i == 0) { // exact: 1 = the iteration record (lightning REQUEST), 2 = the period record (this rank's lightning STATE)
    PoolEvent e;
    e.gid = -1;
    e.key = rank;
    for (int c = 0; c < 4; c++) e.rec[c] = exact == 1 ? st->px_light[c] : st->lightning[c];
    e.rec[4] = fb00 ? fb00->x : 0.f;
    e.pad = 0;
    ev[0] = e; // (entry 0 is reserved: the counter started at 1)
    if (exact == 1) st->px_light[0] = st->px_light[1] = st->px_light[2] = st->px_light[3] = 0.f;
  }
  if (i >= n) return;
  const unsigned f = flips[i], meta = owned_once[i];
  const bool mine = (meta & 0x80) != 0;
  flips[i] = 0;
  owned_once[i] = 0;
  if (!f || !mine) return;
  const int at = atomicAdd(hdr, 1);
  if (at >= cap) return; // (the host reports hdr[0] > cap as an error)
  PoolEvent e;
  e.gid = i;
  e.key = ((__ffs((int)f) - 1) << 18) | ((15 - __popc(f)) << 14) | ((15 - (int)(meta & 15)) << 10) | rank;
  for (int c = 0; c < 5; c++) e.rec[c] = drops[5 * (size_t)i + c];
  e.pad = 0;
  ev[at] = e;
}
// pass 1 over the gathered events of all ranks: per droplet the smallest key
__global__ void k_pool_events_best(int n_ranks, size_t stride_bytes, int cap, const char *__restrict__ bufs, int *__restrict__ best)
{
  const int r = blockIdx.y;
  const int *hdr = reinterpret_cast<const int *>(bufs + (size_t)r * stride_bytes);
  const PoolEvent *ev = reinterpret_cast<const PoolEvent *>(bufs + (size_t)r * stride_bytes + POOL_HDR);
  const int cnt = min(hdr[0], cap);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += gridDim.x * blockDim.x)
    if (ev[k].gid >= 0) atomicMin(&best[ev[k].gid], ev[k].key);
}
// pass 2: every rank but the reporter takes the winner's record (a rank that spawned a phantom loses it here). Whether an ACTIVE
// droplet is then tracked here -- as owner or as ghost copy -- is decided by its position in k_pool_edges_pack, which runs next
// exact mode (events every iteration, edges only once per period): a winner's ACTIVE record is kept as a ghost copy only where it lies
// inside this rank's local array; `retired` counts the winners that ended the iteration inactive (once per droplet: the winner's)
__global__ void k_pool_events_apply(int n_ranks, size_t stride_bytes, int cap, const char *__restrict__ bufs, int *__restrict__ best, int my_rank,
                                    float *__restrict__ drops, unsigned char *__restrict__ remote, Geo g, int exact, int *__restrict__ retired)
{
  const int r = blockIdx.y;
  const int *hdr = reinterpret_cast<const int *>(bufs + (size_t)r * stride_bytes);
  const PoolEvent *ev = reinterpret_cast<const PoolEvent *>(bufs + (size_t)r * stride_bytes + POOL_HDR);
  const int cnt = min(hdr[0], cap);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += gridDim.x * blockDim.x) {
    const PoolEvent e = ev[k];
    if (e.gid < 0 || best[e.gid] != e.key) continue;
    if (exact && retired && e.rec[2] < 0.0f) atomicAdd(retired, 1);
    if ((e.key & 1023) == my_rank) continue;
    for (int c = 0; c < 5; c++) drops[5 * (size_t)e.gid + c] = e.rec[c];
    unsigned char rem = 0;
    if (exact && e.rec[2] >= 0.0f) {
      const int lc = local_col(g, e.rec[0] / 2.0f + 0.5f);
      rem = (lc < 0 || lc >= g.X) ? 1 : 0;
    }
    remote[e.gid] = rem;
  }
}
// exact mode, after k_pool_events_apply: the iteration records of all ranks -> the lightning request of the whole domain, accepted or
// rejected exactly as lightning_update does (two requests of one iteration add up to a start time that is discarded), and -- in the
// iterations in which the reference refreshes it -- the `inactiveDroplets` uniform: the droplets that were inactive before AND after
// this iteration (every rank holds every inactive record: counted locally into px_count, minus this iteration's retirements) plus the
// real deposit at texel (0,0), in the order the mailbox texel is built (box sum + count)
__global__ void k_pool_exact_resolve(int n_ranks, size_t stride_bytes, int cap, const char *__restrict__ bufs, DevState *st, float iterNum, int refresh,
                                     const int *__restrict__ retired)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float n[4] = {0.f, 0.f, 0.f, 0.f}, fb00 = 0.f;
  for (int r = 0; r < n_ranks; r++) { // (rank order: a fixed summation order)
    const int *hdr = reinterpret_cast<const int *>(bufs + (size_t)r * stride_bytes);
    const PoolEvent *ev = reinterpret_cast<const PoolEvent *>(bufs + (size_t)r * stride_bytes + POOL_HDR);
    if (min(hdr[0], cap) >= 1 && ev[0].gid < 0) { // the iteration record: entry 0 (k_pool_events_pack)
      for (int c = 0; c < 4; c++) n[c] += ev[0].rec[c];
      fb00 += ev[0].rec[4];
    }
  }
  if (refresh) st->inactiveDroplets = fb00 + (st->px_count - (float)retired[0]);
  st->px_count = 0.f;
  if (n[2] < fmaxf(iterNum - 1.0f, 1.0f) || n[2] > iterNum) return; // discard
  for (int c = 0; c < 4; c++) st->lightning[c] = n[c];
}
// per-period protocol through the library's transport: every rank's lightning state travels as its period record; the latest strike
// wins (equal start times: the lower rank), as slab.py's reconcile_lightning does with two all-reduces
__global__ void k_pool_lightning_latest(int n_ranks, size_t stride_bytes, int cap, const char *__restrict__ bufs, DevState *st)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float best[4] = {0.f, 0.f, 0.f, 0.f};
  bool have = false;
  for (int r = 0; r < n_ranks; r++) {
    const int *hdr = reinterpret_cast<const int *>(bufs + (size_t)r * stride_bytes);
    const PoolEvent *ev = reinterpret_cast<const PoolEvent *>(bufs + (size_t)r * stride_bytes + POOL_HDR);
    if (min(hdr[0], cap) >= 1 && ev[0].gid < 0 && ev[0].rec[2] > 0.f && (!have || ev[0].rec[2] > best[2])) { // the period record: entry 0
      for (int c = 0; c < 4; c++) best[c] = ev[0].rec[c];
      have = true;
    }
  }
  if (have)
    for (int c = 0; c < 4; c++) st->lightning[c] = best[c];
}
// inactive records this rank holds (exact mode: all of them, current), into px_count
__global__ void k_pool_count_inactive(int n, const float *__restrict__ drops, const unsigned char *__restrict__ remote, DevState *st)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inactive = i < n && !remote[i] && drops[5 * (size_t)i + 2] < 0.0f;
  const unsigned long long m = __ballot(inactive);
  if (m != 0ull && (threadIdx.x & 63) == __ffsll((long long)m) - 1) unsafeAtomicAdd(&st->px_count, (float)__popcll(m));
}
__global__ void k_pool_events_reset(int n_ranks, size_t stride_bytes, int cap, const char *__restrict__ bufs, int *__restrict__ best)
{
  const int r = blockIdx.y;
  const int *hdr = reinterpret_cast<const int *>(bufs + (size_t)r * stride_bytes);
  const PoolEvent *ev = reinterpret_cast<const PoolEvent *>(bufs + (size_t)r * stride_bytes + POOL_HDR);
  const int cnt = min(hdr[0], cap);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += gridDim.x * blockDim.x)
    if (ev[k].gid >= 0) best[ev[k].gid] = 0x7fffffff;
}
// active droplets I hold: outside my owned columns -> the owner is elsewhere (remote); inside and within `halo` columns of an edge ->
// a ghost copy for that neighbour. count_inactive != 0: also count the inactive records into st->px_count (the 600-iteration refresh
// of the `inactiveDroplets` uniform, app.js:5957-5966: every rank holds every inactive record, so the count needs no collective).
__global__ void k_pool_edges_pack(Geo g, int n, int own_lo, int own_hi, int halo, int cap, float *__restrict__ drops, unsigned char *__restrict__ remote,
                                  int *__restrict__ hdrL, PoolRec *__restrict__ recL, int *__restrict__ hdrR, PoolRec *__restrict__ recR,
                                  DevState *st, int count_inactive)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool inactive = false;
  if (i < n && !remote[i]) {
    const float px = drops[5 * (size_t)i], m0 = drops[5 * (size_t)i + 2];
    inactive = m0 < 0.0f;
    if (!inactive) {
      const int lc = local_col(g, px / 2.0f + 0.5f);
      if (lc < own_lo || lc >= own_hi) {
        remote[i] = 1;
      } else {
        PoolRec r;
        r.gid = i;
        for (int c = 0; c < 5; c++) r.rec[c] = drops[5 * (size_t)i + c];
        if (hdrL && lc < own_lo + halo) {
          const int at = atomicAdd(hdrL, 1);
          if (at < cap) recL[at] = r;
        }
        if (hdrR && lc >= own_hi - halo) {
          const int at = atomicAdd(hdrR, 1);
          if (at < cap) recR[at] = r;
        }
      }
    }
  }
  if (count_inactive) {
    const unsigned long long m = __ballot(inactive);
    if (m != 0ull && (threadIdx.x & 63) == __ffsll((long long)m) - 1) unsafeAtomicAdd(&st->px_count, (float)__popcll(m));
  }
}
__global__ void k_pool_edges_apply(int cap, const int *__restrict__ hdr, const PoolRec *__restrict__ rec, float *__restrict__ drops,
                                   unsigned char *__restrict__ remote)
{
  const int cnt = min(hdr[0], cap);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += gridDim.x * blockDim.x) {
    const PoolRec r = rec[k];
    for (int c = 0; c < 5; c++) drops[5 * (size_t)r.gid + c] = r.rec[c];
    remote[r.gid] = 0;
  }
}
// readback aid: 0 = tracked elsewhere (the local record is stale), 1 = inactive (every rank's record), 2 = active in my owned columns
// (this rank's record is THE record), 3 = active ghost copy
__global__ void k_pool_flags(Geo g, int n, int own_lo, int own_hi, const float *__restrict__ drops, const unsigned char *__restrict__ remote,
                             unsigned char *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned char f = 0;
  if (!remote[i]) {
    const float px = drops[5 * (size_t)i], m0 = drops[5 * (size_t)i + 2];
    if (m0 < 0.0f) {
      f = 1;
    } else {
      const int lc = local_col(g, px / 2.0f + 0.5f);
      f = (lc >= own_lo && lc < own_hi) ? 2 : 3;
    }
  }
  out[i] = f;
}
// note != 0: also keep the largest entry count seen (what the library's transport sizes the next all-gathers by)
__global__ void k_pool_check(int n_bufs, size_t stride_bytes, int cap, const char *__restrict__ bufs, DevState *st, int note)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_bufs) {
    const int cnt = *reinterpret_cast<const int *>(bufs + (size_t)r * stride_bytes);
    if (cnt > cap) atomicMax(&st->pool_overflow, cnt);
    if (note) atomicMax(&st->pool_seen_max, cnt);
  }
}
__global__ void k_inactive_from_count(DevState *st)
{
  st->inactiveDroplets = st->px_count;
  st->px_count = 0.f;
}

// ------------------------------------------------------------------------------------------------
// halo pack / unpack of the state carried across iterations (base_0, wall_0, water_1, light_0, light_1)
// buffer layout: [base h*Y float4][water h*Y float4][light0 h*Y float4][light1 h*Y float4][wall h*Y char4]
// base_only (round 5): a slab of the agreed water-free dry stencil -- its iterations write nothing but the base texture, so the ghost
// columns of every other texture stay what the upload made them -- sends [base h*Y float4] alone: 16 of 68 bytes per cell
// ------------------------------------------------------------------------------------------------
struct HaloPtrs {
  float4 *base, *water, *light0, *light1;
  LightPlanes lp0, lp1; // used instead of light0 / light1 while the light texture is stored as planes (lp0.x != nullptr);
                        // the BUFFER always carries interleaved RGBA texels
  char4 *wall;
  float3 *fb;  // particle feedback / deposition textures: exchanged only on handles that carry particles (the BUFFER carries RGBA texels)
  float2 *dep; // buffer layout then: [4 x h*Y float4][fb h*Y float4][dep h*Y float2][wall h*Y char4]
};
// one or both sides per launch (blockIdx.y = slot): x_start / buffers of slot 0 and 1. Both sides in ONE launch matter where the small
// kernel runs next to a marching kernel that holds every wave slot: a second launch queues behind thousands of workgroups.
struct HaloBufs {
  int x_start[2];
  float4 *buf16[2];
  float2 *buf8[2];
  char4 *buf4[2];
};
__global__ void k_halo_pack(HaloPtrs f, int X, int Y, int h, HaloBufs hb, int base_only)
{
  const int n = h * Y, x_start = hb.x_start[blockIdx.y];
  float4 *buf16 = hb.buf16[blockIdx.y];
  float2 *buf8 = hb.buf8[blockIdx.y];
  char4 *buf4 = hb.buf4[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int y = i / h, c = i - y * h;
    const size_t s = cidx(x_start + c, y, X);
    buf16[i] = f.base[s];
    if (base_only) continue;
    buf16[n + i] = f.water[s];
    if (f.lp0.x) {
      const float2 a = f.lp0.zw[s], b = f.lp1.zw[s];
      buf16[2 * n + i] = make_float4(f.lp0.x[s], f.lp0.y[s], a.x, a.y);
      buf16[3 * n + i] = make_float4(f.lp1.x[s], f.lp1.y[s], b.x, b.y);
    } else {
      buf16[2 * n + i] = f.light0[s];
      buf16[3 * n + i] = f.light1[s];
    }
    buf4[i] = f.wall[s];
    if (f.fb) {
      const float3 v = f.fb[s];
      buf16[4 * n + i] = make_float4(v.x, v.y, v.z, 0.f);
      buf8[i] = f.dep[s];
    }
  }
}
__global__ void k_halo_unpack(HaloPtrs f, int X, int Y, int h, HaloBufs hb, int *ghost_nontrivial, int base_only)
{
  const int n = h * Y, x_start = hb.x_start[blockIdx.y];
  const float4 *buf16 = hb.buf16[blockIdx.y];
  const float2 *buf8 = hb.buf8[blockIdx.y];
  const char4 *buf4 = hb.buf4[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int y = i / h, c = i - y * h;
    const size_t s = cidx(x_start + c, y, X);
    f.base[s] = buf16[i];
    if (base_only) continue;
    const float4 q = buf16[n + i];
    f.water[s] = q;
    if (ghost_nontrivial) { // same test as wx_upload's: 0 in air, only the wall marker in wall cells, vegetation >= 0
      const char4 wl = buf4[i];
      const float x_expected = wl.y == 0 ? (wl.x == 2 ? 1002.0f : 1001.0f) : 0.0f;
      if (!(q.x == x_expected && q.y == 0.0f && q.z == 0.0f && q.w == 0.0f && wl.w >= 0)) *ghost_nontrivial = 1;
    }
    if (f.lp0.x) {
      const float4 a = buf16[2 * n + i], b = buf16[3 * n + i];
      f.lp0.x[s] = a.x;
      f.lp0.y[s] = a.y;
      f.lp0.zw[s] = make_float2(a.z, a.w);
      f.lp1.x[s] = b.x;
      f.lp1.y[s] = b.y;
      f.lp1.zw[s] = make_float2(b.z, b.w);
    } else {
      f.light0[s] = buf16[2 * n + i];
      f.light1[s] = buf16[3 * n + i];
    }
    f.wall[s] = buf4[i];
    if (f.fb) {
      const float4 v = buf16[4 * n + i];
      f.fb[s] = make_float3(v.x, v.y, v.z);
      f.dep[s] = buf8[i];
    }
  }
}

} // namespace wx
