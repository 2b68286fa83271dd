// wx_cells.h -- per-cell physics of the simulation iteration as HIP device functions (gfx950).
//
// One function per reference pass; each is written against an ACCESSOR (template parameter) that
// yields neighbour texels, so the same arithmetic serves the per-pass kernels (global memory) and the
// row-marching (wave-private LDS ring) and LDS-tiled kernels. Semantics: NEAREST + REPEAT sampling on both axes,
// light texture clamped in y, fp32 evaluated left-to-right with NO contraction (-ffp-contract=off),
// RGBA8I stores saturate, pow() with constant exponents is a fixed multiply chain / sqrt so that the
// result is bit-reproducible (GLSL leaves pow's rounding undefined).
//
// Reference files restated here (paths relative to the reference root):
//   shaders/vertex/simShader.vert:21-34        coordinates
//   shaders/fragment/velocityShader.frag:32-61
//   shaders/fragment/curlShader.frag:12-19
//   shaders/fragment/vorticityShader.frag:19-38
//   shaders/fragment/boundaryShader.frag:56-531
//   shaders/fragment/advectionShader.frag:65-457 (+ common.glsl:194-254 bilerp / bilerpWall)
//   shaders/fragment/pressureShader.frag:16-43
//   shaders/fragment/lightingShader.frag:38-170 (light output; reflectedLight is display-only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wx {

// channel enums: common.glsl:42-95
enum { VX = 0, VY = 1, PRESSURE = 2, TEMPERATURE = 3 };
enum { TYPE = 0, DISTANCE = 1, VERT_DISTANCE = 2, VEGETATION = 3 };
enum {
  WALLTYPE_INERT = 0,
  WALLTYPE_LAND = 1,
  WALLTYPE_WATER = 2,
  WALLTYPE_FIRE = 3,
  WALLTYPE_URBAN = 4,
  WALLTYPE_RUNWAY = 5,
  WALLTYPE_INDUSTRIAL = 6
};

// common.glsl:9-35
constexpr float lightHeatingConst = 0.000002f;
constexpr float waterHeatExchangeRate = 0.0002f;
constexpr float waterHeatCapacity = 50.0f;
constexpr float fullWhiteSnowHeight = 10.0f;
constexpr float snowMassToHeight = 0.05f;
constexpr float snowMeltRate = 0.000015f;
constexpr float maxWaterTemp = 40.0f;
constexpr float ALBEDO_SNOW = 0.85f;
constexpr float ALBEDO_SNOW_FOREST = 0.30f;
constexpr float ALBEDO_FOREST = 0.10f;
constexpr float ALBEDO_DRYSOIL = 0.30f;
constexpr float ALBEDO_WETSOIL = 0.15f;
constexpr float ALBEDO_URBAN = 0.08f;
constexpr float ALBEDO_INDUSTRIAL = 0.08f;
constexpr float ALBEDO_RUNWAY = 0.04f;
constexpr float ALBEDO_WATER = 0.05f;

// Geometry of the handle's (slab of the) grid.
struct Geo {
  int X, Y;     // local width (incl. ghost columns) and height
  int Xg, xoff; // global width; global x of local column 0 (already reduced to [0, Xg))
  float sx, sy; // quad-UV scale (1.0 when quad_scale == 0)
  float texX, texY;
};

// Uniforms (wx_params) plus host-derived constants. The per-cell functions below take them as `const UT &u`: UT is Uni, or Uni
// in the constant address space (CUni) -- loads through it are scalar loads the compiler may re-issue anywhere, so a long
// kernel neither holds 60 uniforms in SGPRs nor falls back to vector loads once it has stored something.
struct Uni {
  float dragMultiplier, wind;
  float vorticity, landEvaporation, waterEvaporation, dynamicWaterTemperature;
  float evapHeat, waterWeight, dryLapse;
  float meltingHeat, condensationRate, globalDrying, globalHeating, soundingForcing;
  float globalEffectsStartAlt, globalEffectsEndAlt, waterTemperature;
  float sunIntensity, greenhouseGases, waterGreenHouseEffect, IR_rate;
  float aboveZeroThreshold, subZeroThreshold, spawnChanceMult, snowDensity, fallSpeed;
  float growthRate0C, growthRate_30C, freezingRate, meltingRate, evapRate;
  float userInputValues[4];
  float userInputMove[2];
  int userInputType;
  int wrapHorizontally;
  float airplaneValues[4];
  // derived on the host once per wx_set_params (same fp32 expressions the shaders evaluate per fragment)
  float cos_a, sin_a, sin_ma;      // sin/cos of the uniform sunAngle
  float vel_keep;                  // 1. - dragMultiplier * 0.0002          (velocityShader.frag:52)
  float wind_add;                  // wind * 0.000001                       (velocityShader.frag:60)
  float snd_dragk;                 // 1.0 - map_rangeC(soundingForcing, 0.1, 1.0, 0.0, 0.001)   (advectionShader.frag:171)
  float snd_velk;                  // map_rangeC(soundingForcing, 0.9, 1.0, 0.0, 0.001)         (advectionShader.frag:174)
  float a_texX, a_texY;            // vec2(1.) / resolution                 (advectionShader.frag:69)
  float a_invTexY;                 // 1.0 / texelSize.y                     (advectionShader.frag:162)
  float a_aspect;                  // texelSize.y / texelSize.x             (advectionShader.frag:247)
  float chc;                       // cellHeightCompensation = 300. / resolution.y (lightingShader.frag:44)
  // the sun-ray tap of the lighting pass (lightingShader.frag:48-49) when fragCoord is exactly x + 0.5 (quad_scale == 0): offset,
  // filter weights and their four products are then the same for every cell
  int sun_uniform, sun_dx0, sun_fv;
  float sun_w00, sun_w10, sun_w01, sun_w11;
  // display-only second output of the lighting pass (emittedLight, lightingShader.frag:58-60, 98-101): the sunlight colour for
  // this sun angle, sunColor(scattering) (common.glsl:367-378), and "abs(sunAngle) > 85 degrees" (urban areas glow at night)
  float sun_col[3];
  int night_glow;
  // per iteration
  float iterNum;
  int iterI; // int(iterNum)
};

typedef const __attribute__((address_space(4))) Uni CUni;
__device__ __forceinline__ CUni &as_constant(const Uni &u) { return *(CUni *)(&u); }
// the per-row profile arrays (initial_T, sounding) through the same address space: FP is `const float *` or CFloatP
typedef const __attribute__((address_space(4))) float *CFloatP;
__device__ __forceinline__ CFloatP as_constant(const float *p) { return (CFloatP)p; }

struct CellCoord {
  float fx, fy;   // fragCoord
  float tcx, tcy; // texCoord
  int gx;         // global column
};

__device__ __forceinline__ int wrapi(int i, int n)
{
  // i is within (-n, 2n) everywhere this is used with small offsets
  return i < 0 ? i + n : (i >= n ? i - n : i);
}
__device__ __forceinline__ int wrapmod(int i, int n)
{
  int r = i % n;
  return r < 0 ? r + n : r;
}

// simShader.vert:23-24 with the quad of app.js:4770-4788
__device__ __forceinline__ CellCoord cellcoord(const Geo &g, int x, int y)
{
  CellCoord c;
  int gx = g.xoff + x;
  if (gx >= g.Xg) gx -= g.Xg;
  c.gx = gx;
  c.fx = ((float)gx + 0.5f) * g.sx;
  c.fy = ((float)y + 0.5f) * g.sy;
  c.tcx = c.fx * g.texX;
  c.tcy = c.fy * g.texY;
  return c;
}

__device__ __forceinline__ int sat8(int v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }
__device__ __forceinline__ char4 pack_wall(const int w[4])
{
  return make_char4((signed char)sat8(w[0]), (signed char)sat8(w[1]), (signed char)sat8(w[2]), (signed char)sat8(w[3]));
}
__host__ __device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// mix(): lowered as a + t*(b-a), the form pinned by the golden vectors (see oracle/wx_oracle.c)
__device__ __forceinline__ float mixf(float a, float b, float t) { return a + t * (b - a); }
// common.glsl:99-101
__host__ __device__ __forceinline__ float map_range(float v, float min1, float max1, float min2, float max2)
{
  return min2 + (v - min1) * (max2 - min2) / (max1 - min1);
}
__host__ __device__ __forceinline__ float map_rangeC(float v, float min1, float max1, float min2, float max2)
{
  return clampf(map_range(v, min1, max1, min2, max2), fminf(min2, max2), fmaxf(min2, max2));
}
__device__ __forceinline__ float CtoK(float c) { return c + 273.15f; }
__device__ __forceinline__ float KtoC(float k) { return k - 273.15f; }
__device__ __forceinline__ float pow17(float x)
{
  const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
  return x16 * x;
}
__device__ __forceinline__ float pow4(float x)
{
  const float x2 = x * x;
  return x2 * x2;
}
// common.glsl:177-180
__device__ __forceinline__ float maxWater(float T) { return pow17(T / 250.0f); }
// common.glsl:258-261
__device__ __forceinline__ float IR_emitted(float T) { return pow4(T * 0.01f) * 5.670374419f; }

// ------------------------------------------------------------------------------------------------
// velocityShader.frag:32-61.  b: own texel, Pr/Pu: PRESSURE of right / up neighbour
// ------------------------------------------------------------------------------------------------
template <class UT> __device__ __forceinline__ float4 velocity_cell(const UT &u, float4 b, float Pr, float Pu, int wall_dist)
{
  if (wall_dist == 0) {
    b.x = 0.0f;
    b.y = 0.0f;
  } else {
    b.x += b.z - Pr;
    b.y += b.z - Pu;
    b.x *= u.vel_keep;
    b.y *= u.vel_keep;
    b.x += u.wind_add;
  }
  return b;
}

// curlShader.frag:12-19.  c: own (vx,vy); vy_r: vy of right neighbour; vx_u: vx of up neighbour
__device__ __forceinline__ float curl_cell(float vx, float vy, float vy_r, float vx_u) { return vx_u - vx - vy_r + vy; }

// vorticityShader.frag:19-38
__device__ __forceinline__ float2 vorticity_cell(float curl, float cl, float cr, float cd, float cu)
{
  float fx = fabsf(cd) - fabsf(cu);
  float fy = fabsf(cr) - fabsf(cl);
  const float magnitude = sqrtf(fx * fx + fy * fy) + 0.0001f;
  fx /= magnitude;
  fy /= magnitude;
  fx *= curl;
  fy *= curl;
  return make_float2(fx, fy);
}

// pressureShader.frag:16-43.  b: own texel; vx_l: vx of left; bd: texel below; wd: wall below
__device__ __forceinline__ float4 pressure_cell(float4 b, float vx_l, float vy_d, float T_d, int wd_type, int wd_dist)
{
  if (wd_dist == 0 && wd_type == 1) b.w -= T_d - 1000.0f;
  b.z += (vx_l - b.x + vy_d - b.y) * 0.45f;
  return b;
}

// ------------------------------------------------------------------------------------------------
// boundaryShader.frag:72-531
// Accessor A: base(dx,dy) water(dx,dy) -> float4 ; wall(dx,dy) -> char4 ; vort(dx,dy) -> float2 ;
//             light_y0() -> NET_HEATING of lightTexture_0 at the own cell ; light_x0() -> its SUNLIGHT (asked for only next
//             to walls) ; light_xy_up() -> (SUNLIGHT, NET_HEATING) at (x, clamp(y+1)) (surface wall cells only) ;
//             fb() -> float4 ; dep() -> float2 ; has_fb() (wave-uniform)
// iterNum / iterI: the per-iteration uniform (float as the reference passes it, and int(iterNum)); separate from Uni so that
// Uni can live in read-only device memory for a whole wx_step call
// ------------------------------------------------------------------------------------------------
template <class UT> __device__ __forceinline__ float calcEvaporation(const UT &u, float T, float W, float V, float M)
{
  return fmaxf((maxWater(T) - W) * u.landEvaporation * (V / 127.0f + 0.1f) * fminf(M + 1.0f, 50.0f) * 0.05f, 0.0f);
}
__device__ __forceinline__ float calcFireIntensity(int veg, float moist, float precip)
{
  return fmaxf((float)veg * 0.00025f - moist * 0.00020f - precip * 0.02f, 0.0f);
}

// AIR (a wave-uniform fact established by the caller, see air_cell()): the cell and its four neighbours are fluid and the wall
// below is at least 8 rows away -- every surface branch below is then statically dead; the arithmetic that remains is
// the same instruction sequence, so the results are bit-identical to the general instantiation.
__device__ __forceinline__ bool air_cell(char4 w0, char4 wL, char4 wD, char4 wR, char4 wU)
{
  return w0.y != 0 && wL.y != 0 && wD.y != 0 && wR.y != 0 && wU.y != 0 && wD.z >= 8;
}
template <bool AIR = false, class UT, class FP, class A>
__device__ __forceinline__ void boundary_cell(const UT &u, const float iterNum, const int iterI, const Geo &g, const FP initial_T,
                                              int x, int y, const A &a, float4 &base_out, float4 &water_out, char4 &wall_out)
{
  const float gravMult = 0.0001f;
  const float exchangeRate = 0.015f;
  const CellCoord cc = cellcoord(g, x, y);
  const float tcy = cc.tcy;
  const float tcy_up = tcy + g.texY; // texCoordX0Yp.y (simShader.vert:30)

  float4 b = a.base(0, 0);
  float4 w = a.water(0, 0);
  const float realTemp = b.w - tcy * u.dryLapse;
  const char4 w0 = a.wall(0, 0), wL = a.wall(-1, 0), wD = a.wall(0, -1), wR = a.wall(1, 0), wU = a.wall(0, 1);
  int wl[4] = {w0.x, w0.y, w0.z, w0.w};
  bool nextToWall = false;

  wl[VERT_DISTANCE] = wD.z + 1;

  if (AIR || wl[DISTANCE] != 0) { // fluid
    const float light_y = a.light_y0();
    const bool has_fb = a.has_fb(); // wave-uniform: false while no particle has ever written feedback
    const float4 fb = a.fb();
    wl[TYPE] = wD.x;
    if (wl[TYPE] != WALLTYPE_WATER) b.w += light_y; // NET_HEATING
    if (has_fb) {
      b.w += fb.y; // HEAT

      const float precipCoalescence = fmaxf(-fb.z, 0.0f);
      w.y -= precipCoalescence;
      w.x -= precipCoalescence;
      const float precipEvaporation = fmaxf(fb.z, 0.0f);
      w.x += precipEvaporation;

      w.z = fmaxf(w.z * 0.997f - 0.00001f + fb.x * 0.005f, 0.0f);

      w.w /= 1.0f + fmaxf(-fb.z * 0.1f, 0.0f) + fb.x * 0.000f;
      w.w -= fb.x * 0.0001f;
    } else {
      // feedback == 0: every term above is x +- 0 or x / 1 -- the same values without the arithmetic
      w.z = fmaxf(w.z * 0.997f - 0.00001f, 0.0f);
    }
    w.w -= fmaxf((w.w - 4.0f) * 0.01f, 0.0f);
    w.w = fmaxf(w.w, 0.0f);
    if (w.w > 4.0f) w.w -= w.z * 0.02f;

    // GRAVITY :132-148
    const float4 bU = a.base(0, 1);
    const int iy = (int)cc.fy;
    float gravityForce = ((b.w + bU.w) * 0.5f - (initial_T[iy] + initial_T[iy + 1]) * 0.5f) * gravMult;
    gravityForce -= w.y * gravMult * u.waterWeight;
    if (has_fb) gravityForce -= fb.x * gravMult * u.waterWeight;
    b.y += gravityForce;

    float snowCover = 0.0f, soilMoisture = 0.0f;

    if (!AIR && wD.y == 0) { // below is wall
      nextToWall = true;
      wl[DISTANCE] = 1;
      const float4 wtD = a.water(0, -1);
      snowCover = wtD.w;
      soilMoisture = wtD.z;
      wl[VERT_DISTANCE] = 1;
    }
    if (!AIR && wL.y == 0) { // left is wall
      nextToWall = true;
      wl[DISTANCE] = 1;
      if (wL.x == WALLTYPE_WATER) {
        wl[TYPE] = WALLTYPE_LAND;
        wl[DISTANCE] = 0;
      }
      if (wR.y == 0) wl[DISTANCE] = 0;
    } else if (!AIR && wR.y == 0) { // right is wall
      nextToWall = true;
      wl[DISTANCE] = 1;
      if (wR.x == WALLTYPE_WATER) {
        wl[TYPE] = WALLTYPE_LAND;
        wl[DISTANCE] = 0;
      }
    }
    if (!AIR && wU.y == 0) { // above is wall
      nextToWall = true;
      wl[DISTANCE] = 1;
      if (tcy < 0.99f) wl[DISTANCE] = 0;
    }

    // vorticity force :199-208
    const float2 vf00 = a.vort(0, 0), vfL = a.vort(-1, 0), vfD = a.vort(0, -1);
    const float velocityFactor = sqrtf(b.x * b.x + b.y * b.y) * 0.1f;
    b.x += (vf00.x + vfD.x) * (u.vorticity + velocityFactor);
    b.y += (vf00.y + vfL.y) * (u.vorticity + velocityFactor);

    if (!AIR && nextToWall) {
      if (wl[TYPE] != WALLTYPE_WATER) {
        float lightPower = 0.0f;
        const float light_x = a.light_x0();
        if (wD.y == 0) lightPower += fmaxf(light_x * u.cos_a, 0.0f);
        if (wL.y == 0) lightPower += fmaxf(light_x * u.sin_a, 0.0f);
        if (wR.y == 0) lightPower += fmaxf(light_x * u.sin_ma, 0.0f);
        float albedoTotal = 1.0f;
        if (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_FIRE) {
          float albedoSoil = map_rangeC(soilMoisture, 0.0f, 20.0f, ALBEDO_DRYSOIL, ALBEDO_WETSOIL);
          albedoSoil = map_rangeC(snowCover, 0.0f, fullWhiteSnowHeight, albedoSoil, ALBEDO_SNOW);
          const float fullVegetationAlbedo = map_range(snowCover, 0.0f, fullWhiteSnowHeight, ALBEDO_FOREST, ALBEDO_SNOW_FOREST);
          albedoTotal = map_range((float)wD.w, 0.0f, 127.0f, albedoSoil, fullVegetationAlbedo);
        } else if (wl[TYPE] == WALLTYPE_URBAN) {
          albedoTotal = ALBEDO_URBAN;
        } else if (wl[TYPE] == WALLTYPE_INDUSTRIAL) {
          albedoTotal = ALBEDO_INDUSTRIAL;
        } else if (wl[TYPE] == WALLTYPE_RUNWAY) {
          albedoTotal = ALBEDO_RUNWAY;
        }
        lightPower *= (1.0f - albedoTotal);
        lightPower *= lightHeatingConst;
        b.w += lightPower;
      }
    }

    if (AIR || !nextToWall) {
      int nearest = 255;
      if (wD.y < nearest) nearest = wD.y;
      if (wU.y < nearest) nearest = wU.y;
      if (wL.y < nearest) nearest = wL.y;
      if (wR.y < nearest) nearest = wR.y;
      wl[DISTANCE] = nearest + 1;
    }

    if (!AIR && wl[VERT_DISTANCE] <= 5) { // surfaceWindSmootingDist :271-303
      if (wl[VERT_DISTANCE] == 1) {
        float surfaceDrag = 0.0015f;
        if (wl[TYPE] == WALLTYPE_URBAN)
          surfaceDrag = 0.040f;
        else if (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_FIRE)
          surfaceDrag = map_rangeC((float)wl[VEGETATION], 50.0f, 127.0f, 0.0015f, 0.020f);
        b.x -= fabsf(b.x) * b.x * surfaceDrag * 50.0f;
      }
      if (wU.z <= 5) b.x -= (b.x - bU.x) * exchangeRate;
      if (wD.z > 0) b.x -= (b.x - a.base(0, -1).x) * exchangeRate;
    }

    if (!AIR && wl[VERT_DISTANCE] <= 8) { // :305-372
      wl[VEGETATION] = wD.w;
      const float4 waterInSurface = a.water(0, -1);
      const int t = wl[TYPE];
      if (t == WALLTYPE_FIRE) {
        if (wl[VERT_DISTANCE] == 1) {
          float fireIntensity = calcFireIntensity(wl[VEGETATION], waterInSurface.z, w.z);
          fireIntensity = fmaxf(fireIntensity, 0.0f);
          b.w += fireIntensity;
          w.w += fireIntensity * 2.0f;
          w.x += fireIntensity * 0.50f;
        }
      }
      if (t == WALLTYPE_INDUSTRIAL) { // FIRE falls through here too but is excluded (:330)
        const int texFragX = (int)cc.fx % 80;
        if (wl[VERT_DISTANCE] == 5 && (texFragX == 18 || texFragX == 22)) {
          w.x += 0.25f;
          b.x *= 0.5f;
          b.y *= 0.5f;
          b.y += 0.05f;
        } else if (wl[VERT_DISTANCE] == 6 && texFragX == 29) {
          w.w += 0.01f;
          b.w += 0.02f;
          b.x *= 0.5f;
          b.y *= 0.5f;
        }
      }
      if (t == WALLTYPE_FIRE || t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN) w.w += 0.000002f;
      if (t == WALLTYPE_FIRE || t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN || t == WALLTYPE_LAND) {
        if (wl[VERT_DISTANCE] <= 1) {
          const float evaporation = calcEvaporation(u, realTemp, w.x, (float)wl[VEGETATION], waterInSurface.z) / 1.0f;
          w.x += evaporation;
          b.w -= evaporation * u.evapHeat * 0.5f;
          if (wl[VEGETATION] < 10 && w.z < 5.0f) w.w = fminf(w.w + (fmaxf(fabsf(b.x) - 0.12f, 0.0f) * 0.15f), 2.4f);
        }
      } else if (t == WALLTYPE_WATER) {
        if (wl[VERT_DISTANCE] <= 1) {
          const float LocalWaterTemperature = a.base(0, -1).w;
          b.w += (LocalWaterTemperature - realTemp - 1.0f) / 1.0f * waterHeatExchangeRate;
          w.x += fmaxf((maxWater(LocalWaterTemperature) - w.x) * u.waterEvaporation / 1.0f, 0.0f);
        }
      }
    }
  } else { // this is wall :373-530
    wl[VERT_DISTANCE] = wU.z - 1;

    if (wl[VERT_DISTANCE] < 0) {
      const float4 wtU = a.water(0, 1);
      w.z = wtU.z;
      w.w = wtU.w;
      wl[VEGETATION] = wU.w;
      if (wU.y == 0) {
        if (wU.x != WALLTYPE_WATER) {
          wl[TYPE] = wU.x;
        } else if (wl[TYPE] == WALLTYPE_WATER) {
          b.w = a.base(0, 1).w;
        }
      }
    } else if (wl[VERT_DISTANCE] == 0) {
      const float4 waterX0Yp = a.water(0, 1);
      const float2 precipDeposition = a.dep();
      const float2 lightAboveSurface = a.light_xy_up();
      const int t = wl[TYPE];
      if (t == WALLTYPE_INDUSTRIAL) wl[VEGETATION] = min(wl[VEGETATION], 15);
      if (t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN) wl[VEGETATION] = min(wl[VEGETATION], 75);
      if (t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN || t == WALLTYPE_FIRE) {
        if (wl[TYPE] == WALLTYPE_FIRE) {
          const float fireIntensity = calcFireIntensity(wl[VEGETATION], w.z, waterX0Yp.z);
          if (fireIntensity < 0.002f) {
            wl[TYPE] = WALLTYPE_LAND;
          } else if (iterI % ((int)(10.0f / fireIntensity) + 1) == 0) {
            wl[VEGETATION] -= 1;
            if (wl[VEGETATION] < 10) wl[TYPE] = WALLTYPE_LAND;
          }
        }
      }
      if (t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN || t == WALLTYPE_FIRE || t == WALLTYPE_LAND) {
        w.z = clampf(w.z + precipDeposition.x * 0.1f, 0.0f, 1000.0f);
        w.w = clampf(w.w + precipDeposition.y * snowMassToHeight, 0.0f, 4000.0f);

        const float4 baseAboveSurface = a.base(0, 1);
        const float realTempAboveSurface = baseAboveSurface.w - tcy_up * u.dryLapse;
        const float evaporation = calcEvaporation(u, realTempAboveSurface, waterX0Yp.x, (float)wl[VEGETATION], w.z) * 0.10f;
        w.z -= evaporation;

        if (iterI % 100 == 0) {
          const float snowSmoothingRate = 0.02f, moistureSmoothingRate = 0.02f;
          float numNeighbors = 0.0f, totalNeighborSnow = 0.0f, totalNeighborSoilMoisture = 0.0f;
          if (wL.z == 0 && (wL.x == WALLTYPE_LAND || wL.x == WALLTYPE_URBAN)) {
            const float4 wn = a.water(-1, 0);
            totalNeighborSnow += wn.w;
            totalNeighborSoilMoisture += wn.z;
            numNeighbors += 1.0f;
          }
          if (wR.z == 0 && (wR.x == WALLTYPE_LAND || wR.x == WALLTYPE_URBAN)) {
            const float4 wn = a.water(1, 0);
            totalNeighborSnow += wn.w;
            totalNeighborSoilMoisture += wn.z;
            numNeighbors += 1.0f;
          }
          if (numNeighbors > 0.0f) {
            const float avgNeighborSnow = totalNeighborSnow / numNeighbors;
            w.w += (avgNeighborSnow - w.w) * snowSmoothingRate;
            const float avgNeighborSoilMoisture = totalNeighborSoilMoisture / numNeighbors;
            w.z += (avgNeighborSoilMoisture - w.z) * moistureSmoothingRate;
          }
          const int vegetationGrowthRate = (int)(w.z * sqrtf(lightAboveSurface.x) * 0.01f);
          if (vegetationGrowthRate > 0) {
            const int interval = (100 / vegetationGrowthRate) * 100;
            if (interval != 0 && iterI % interval == 0) { // x % 0 is undefined in GLSL -> false
              if ((int)map_rangeC(realTempAboveSurface, CtoK(0.0f), CtoK(25.0f), 0.0f, 127.0f) > wl[VEGETATION]) wl[VEGETATION] += 1;
            }
          }
          const int subInterval = iterI / 100;
          if (subInterval % ((int)(w.z * 0.1f + w.w * 0.5f) + 10) == 0 && wl[VEGETATION] >= 20 &&
              (wL.x == WALLTYPE_FIRE || wR.x == WALLTYPE_FIRE || waterX0Yp.w > 4.5f)) {
            wl[TYPE] = WALLTYPE_FIRE;
          }
        }
      } else if (t == WALLTYPE_WATER) {
        const float waterTempUpdateInterval = 20.0f;
        if (u.dynamicWaterTemperature >= 1.0f &&
            (iterNum - waterTempUpdateInterval * floorf(iterNum / waterTempUpdateInterval)) < 0.5f) {
          float numNeighbors = 0.0f, totalNeighborTemp = 0.0f;
          if (wL.x == WALLTYPE_WATER) {
            totalNeighborTemp += a.base(-1, 0).w;
            numNeighbors += 1.0f;
          }
          if (wR.x == WALLTYPE_WATER) {
            totalNeighborTemp += a.base(1, 0).w;
            numNeighbors += 1.0f;
          }
          if (numNeighbors > 0.0f) {
            const float avgNeighborTemp = totalNeighborTemp / numNeighbors;
            b.w += (avgNeighborTemp - b.w) * 0.10f;
          }
          if (b.w > 500.0f) b.w = CtoK(25.0f);
          const float airTemperature = a.base(0, 1).w - tcy_up * u.dryLapse;
          float netWaterHeating = 0.0f;
          netWaterHeating += (airTemperature - b.w) * waterHeatExchangeRate;
          netWaterHeating -= fmaxf((maxWater(b.w) - waterX0Yp.x) * u.waterEvaporation, 0.0f) * u.evapHeat * 0.5f;
          float lightPower = fmaxf(lightAboveSurface.x * u.cos_a, 0.0f);
          lightPower *= (1.0f - ALBEDO_WATER);
          lightPower *= lightHeatingConst;
          netWaterHeating += lightPower;
          netWaterHeating += lightAboveSurface.y;
          b.w += netWaterHeating / waterHeatCapacity * waterTempUpdateInterval;
        }
        b.w = clampf(b.w, CtoK(0.0f), CtoK(maxWaterTemp));
        wl[VEGETATION] = 20;
        w.z = 100.0f;
        w.w = 0.0f;
      }
    }
  }
  base_out = b;
  water_out = w;
  wall_out = pack_wall(wl);
}

// ------------------------------------------------------------------------------------------------
// advectionShader.frag:65-457
// Accessor A: base(dx,dy) -> float4 (|dx|,|dy| <= 1) ; wall(dx,dy) -> char4 ;
//             base_off(dx,dy) / water_off(dx,dy) / wall_off(dx,dy): texel at an ARBITRARY integer
//             offset from the own cell (data-dependent back-trace footprint)
// ------------------------------------------------------------------------------------------------
struct Taps {
  int dx0, dy0; // offset of tap (i, j) from the own cell; the other taps are +1
  float fx, fy;
};
// common.glsl:194-203 / :216-222. pos is in global fragCoord units.
__device__ __forceinline__ Taps mktaps(const CellCoord &cc, int y, float posx, float posy)
{
  Taps t;
  const float stx = posx - 0.5f, sty = posy - 0.5f;
  const float flx = floorf(stx), fly = floorf(sty);
  t.fx = stx - flx;
  t.fy = sty - fly;
  t.dx0 = (int)flx - cc.gx;
  t.dy0 = (int)fly - y;
  return t;
}
// The four texels (dx0 + i, dy0 + j), i, j in {0, 1}, of a bilinear footprint. Generic form: through the accessor's
// arbitrary-offset reads; an accessor whose storage makes the four taps constant offsets from one address provides
// its own make_fp overload (wx_march.h).
template <class A> struct FpGeneric {
  const A &a;
  int dx0, dy0;
  __device__ __forceinline__ float4 base(int i, int j) const { return a.base_off(dx0 + i, dy0 + j); }
  __device__ __forceinline__ float4 water(int i, int j) const { return a.water_off(dx0 + i, dy0 + j); }
  __device__ __forceinline__ char4 wall(int i, int j) const { return a.wall_off(dx0 + i, dy0 + j); }
};
template <class A> __device__ __forceinline__ FpGeneric<A> make_fp(const A &a, int dx0, int dy0) { return FpGeneric<A>{a, dx0, dy0}; }

__device__ __forceinline__ float bilerp4(float a, float b, float c, float d, float mAB, float mCD, float mY)
{
  return mixf(mixf(a, b, mAB), mixf(c, d, mCD), mY);
}
__device__ __forceinline__ float absHorizontalDist(float a, float b) { return fminf(fminf(fabsf(a - b), fabsf(1.0f + a - b)), 1.0f - a + b); }
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x)
{
  const float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return t * t * (3.0f - 2.0f * t);
}

// NO_WATER: the water texture is identically zero in air cells (BASELINE config 1, checked by the host) and no
// sounding forcing is active: the water interpolation and the phase-change block then provably leave base unchanged
// (condensation = max(negative * 0.2, -0) = -0, dT = -0) and water zero, so they are not evaluated.
// NO_WALL (wave-uniform, established by the caller): neither the cell nor any texel of its back-trace footprints is a wall
// cell, so the wall-aware interpolation reduces to the plain one (same weights, same operations) and the wall branch is dead.
// NO_ZW (wave-uniform, established by the caller; only with NO_WALL): the precipitation-visual channel (z) and the smoke channel (w)
// of the post-boundary water are zero in every texel the footprints can reach -- free air without rain, snow or smoke, i.e. most of
// the sky. Their interpolations are then 0 + t * (0 - 0) = 0 for the finite weights of this path, so they are not evaluated (one
// footprint, eight taps and six lerps less).
// QUIET (launch-uniform, established by the host): no brush input and no airplane event in this iteration (userInputType < 1,
// airplaneValues[3] in [0, 0.9]) -- the two sections are not compiled in: they are uniform branches that are never taken in a running
// simulation, but every one of them is a merge point of the cell's whole state (base, water, wall), i.e. copies and scalar tests on
// the common path.
template <bool NO_WATER = false, bool NO_WALL = false, bool NO_ZW = false, bool QUIET = false, class UT, class FP, class A>
__device__ __forceinline__ void advection_cell(const UT &u, const Geo &g, const FP initial_T, const FP snd_T, const FP snd_W, const FP snd_Vel, int x, int y,
                                               const A &a, float4 &base_out,
                                               float4 &water_out, char4 &wall_out)
{
  const CellCoord cc = cellcoord(g, x, y);
  const float fx = cc.fx, fy = cc.fy, tcx = cc.tcx, tcy = cc.tcy;
  const float a_texY = u.a_texY; // advectionShader.frag:69: texelSize = vec2(1.) / resolution
  const char4 w0 = a.wall(0, 0);
  int wl[4] = {w0.x, w0.y, w0.z, w0.w};
  float4 b, w;

  if (NO_WALL || wl[DISTANCE] != 0) { // not wall
    const float4 c00 = a.base(0, 0), cL = a.base(-1, 0), cD = a.base(0, -1), cR = a.base(1, 0), cU = a.base(0, 1);
    const float4 cLU = a.base(-1, 1), cRD = a.base(1, -1);

    const float velAtP_x = (cL.x + c00.x) / 2.0f, velAtP_y = (cD.y + c00.y) / 2.0f;
    const float velAtVx_x = c00.x, velAtVx_y = (cD.y + cR.y + c00.y + cRD.y) / 4.0f;
    const float velAtVy_x = (cL.x + cU.x + cLU.x + c00.x) / 4.0f, velAtVy_y = c00.y;

    Taps tP; // footprint of the P / T / water sample
    {
      const Taps t = mktaps(cc, y, fx - velAtVx_x, fy - velAtVx_y);
      const auto f = make_fp(a, t.dx0, t.dy0);
      b.x = bilerp4(f.base(0, 0).x, f.base(1, 0).x, f.base(0, 1).x, f.base(1, 1).x, t.fx, t.fx, t.fy);
    }
    {
      const Taps t = mktaps(cc, y, fx - velAtVy_x, fy - velAtVy_y);
      const auto f = make_fp(a, t.dx0, t.dy0);
      b.y = bilerp4(f.base(0, 0).y, f.base(1, 0).y, f.base(0, 1).y, f.base(1, 1).y, t.fx, t.fx, t.fy);
    }
    {
      // bilerpWall at velAtP (common.glsl:216-254): P, T and water.xyw share one footprint
      const Taps t = mktaps(cc, y, fx - velAtP_x, fy - velAtP_y);
      tP = t;
      const auto f = make_fp(a, t.dx0, t.dy0);
      float mAB = t.fx, mCD = t.fx, mY = t.fy;
      if (!NO_WALL) {
        const int wa = f.wall(0, 0).y, wb = f.wall(1, 0).y;
        const int wc = f.wall(0, 1).y, wd = f.wall(1, 1).y;
        if (wa == 0)
          mAB = 1.0f;
        else if (wb == 0)
          mAB = 0.0f;
        if (wc == 0)
          mCD = 1.0f;
        else if (wd == 0)
          mCD = 0.0f;
        if (wa == 0 && wb == 0)
          mY = 1.0f;
        else if (wc == 0 && wd == 0)
          mY = 0.0f;
      }
      const float4 ba = f.base(0, 0), bb = f.base(1, 0);
      const float4 bc = f.base(0, 1), bd = f.base(1, 1);
      b.z = bilerp4(ba.z, bb.z, bc.z, bd.z, mAB, mCD, mY);
      b.w = bilerp4(ba.w, bb.w, bc.w, bd.w, mAB, mCD, mY);
      if (!NO_WATER) {
        const float4 qa = f.water(0, 0), qb = f.water(1, 0);
        const float4 qc = f.water(0, 1), qd = f.water(1, 1);
        w.x = bilerp4(qa.x, qb.x, qc.x, qd.x, mAB, mCD, mY);
        w.y = bilerp4(qa.y, qb.y, qc.y, qd.y, mAB, mCD, mY);
        w.w = (NO_WALL && NO_ZW) ? 0.0f : bilerp4(qa.w, qb.w, qc.w, qd.w, mAB, mCD, mY);
      }
    }
    if (NO_WATER) {
      w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (tcy > u.globalEffectsStartAlt && tcy < u.globalEffectsEndAlt) b.w += u.globalHeating;
    } else {
    if (NO_WALL && NO_ZW) {
      w.z = 0.0f;
    } else {
      // precipitation visualisation channel, +0.05 in y (:103). Its x position is (fx - velAtP_x) + 0.0: the same number as above
      // except -0 -> +0, and both give the same stx = -0.5 -- so the x half of the footprint is the one already computed
      Taps t = mktaps(cc, y, fx - velAtP_x, fy - velAtP_y + 0.05f);
      t.dx0 = tP.dx0;
      t.fx = tP.fx;
      const auto f = make_fp(a, t.dx0, t.dy0);
      float mAB = t.fx, mCD = t.fx, mY = t.fy;
      if (!NO_WALL) {
        const int wa = f.wall(0, 0).y, wb = f.wall(1, 0).y;
        const int wc = f.wall(0, 1).y, wd = f.wall(1, 1).y;
        if (wa == 0)
          mAB = 1.0f;
        else if (wb == 0)
          mAB = 0.0f;
        if (wc == 0)
          mCD = 1.0f;
        else if (wd == 0)
          mCD = 0.0f;
        if (wa == 0 && wb == 0)
          mY = 1.0f;
        else if (wc == 0 && wd == 0)
          mY = 0.0f;
      }
      w.z = bilerp4(f.water(0, 0).z, f.water(1, 0).z, f.water(0, 1).z, f.water(1, 1).z, mAB, mCD, mY);
    }

    float realTemp = b.w - tcy * u.dryLapse;

    const float excessWater = w.x - maxWater(realTemp);
    const float overSaturation = excessWater - w.y;
    float condensation;
    if (overSaturation < 0.0f)
      condensation = overSaturation * 0.20f;
    else
      condensation = overSaturation * u.condensationRate;
    condensation = fmaxf(condensation, -w.y);
    const float dT = condensation * u.evapHeat * 1.0f;
    b.w += dT;
    realTemp += dT;
    w.y += condensation;

    if (tcy > u.globalEffectsStartAlt && tcy < u.globalEffectsEndAlt) { // :154-181
      // clamp(0, 0, hi >= 0) == 0: nothing to evaluate while the slider is at its default 0 (uniform branch)
      if (u.globalDrying != 0.0f) w.x -= clampf(u.globalDrying, 0.0f, fmaxf(w.x - maxWater(fmaxf(realTemp - 20.0f, CtoK(-80.0f))), 0.0f));
      b.w += u.globalHeating;

      // real-sounding forcing: with soundingForcing == 0 every update is x -= d * 0 and x *= 1 (uniform branch;
      // identical values for finite fields)
      if (u.soundingForcing != 0.0f) {
        const int si = (int)(tcy * u.a_invTexY);
        const int si1 = si - 1 < 0 ? 0 : si - 1; // index -1 is undefined in the reference -> clamp
        const float sT = (snd_T[si] + snd_T[si1]) / 2.0f;
        const float sW = (snd_W[si] + snd_W[si1]) / 2.0f;
        const float sV = (snd_Vel[si] + snd_Vel[si1]) / 2.0f;

        const float Tdiff = b.w - sT;
        b.w -= Tdiff * 0.001f * u.soundingForcing;
        const float Wdiff = w.x - sW;
        w.x -= Wdiff * 0.001f * u.soundingForcing;
        b.x *= u.snd_dragk;
        b.y *= u.snd_dragk;
        const float velDiff = b.x - sV;
        b.x -= velDiff * u.snd_velk;
      }
    }
    w.x = fmaxf(w.x, 0.0f);
    } // !NO_WATER
  } else { // wall :189-227
    b = a.base(0, 0);
    w = a.water_off(0, 0);
    if (wl[TYPE] == WALLTYPE_LAND) b.w = 1000.0f;
    const char4 wU = a.wall(0, 1);
    wl[VEGETATION] = max(wl[VEGETATION], 0);
    w.z = fmaxf(w.z, 0.0f);
    if (wU.y != 0) { // surface layer
      const float4 baseX0Yp = a.base(0, 1);
      const float tempC = KtoC(baseX0Yp.w - tcy * u.dryLapse);
      if (w.w > 0.0f && tempC > 0.0f) {
        const float melting = fminf(tempC * snowMeltRate, w.w);
        w.w -= melting;
        b.w += melting / snowMassToHeight * u.meltingHeat;
        w.z += melting;
      }
      if (w.z > 0.0f && tempC > 0.0f) {
        const float evaporation = fmaxf((maxWater(CtoK(tempC)) - w.x) * 0.00001f, 0.0f);
        w.z -= evaporation;
      }
    }
  }

  // USER INPUT :229-401
  if (!QUIET && u.userInputType >= 1) {
    bool inBrush = false;
    float weight = 1.0f;
    const float brushR = u.userInputValues[3] * a_texY;
    if (u.userInputValues[0] < -0.5f) {
      if (fabsf(u.userInputValues[1] - tcy) < brushR) inBrush = true;
    } else {
      float vmx;
      const float vmy = u.userInputValues[1] - tcy;
      if (u.wrapHorizontally)
        vmx = absHorizontalDist(u.userInputValues[0], tcx);
      else
        vmx = fabsf(u.userInputValues[0] - tcx);
      vmx *= u.a_aspect;
      const float distFromMouse = sqrtf(vmx * vmx + vmy * vmy);
      weight = smoothstepf(brushR, 0.0f, distFromMouse);
      if (distFromMouse < brushR) inBrush = true;
    }
    if (inBrush) {
      const int ut = u.userInputType;
      const float inten = u.userInputValues[2];
      const bool aboveIsAir = a.wall(0, 1).y != 0;
      if (ut == 1) {
        b.w += inten;
        if (wl[TYPE] == 2 && wl[DISTANCE] == 0) b.w = clampf(b.w, CtoK(0.0f), CtoK(maxWaterTemp));
      } else if (ut == 2) {
        if (w.y > 0.0f) {
          w.y += inten;
          w.y = fmaxf(w.y, 0.0f);
        }
        w.x += inten;
        w.x = fmaxf(w.x, 0.0f);
      } else if (ut == 3 && wl[DISTANCE] != 0) {
        w.w += inten;
        w.w = fminf(fmaxf(w.w, 0.0f), 2.0f);
      } else if (ut == 4) {
        b.x += u.userInputMove[0] * 5.0f * weight * inten;
        if (!(u.userInputValues[0] < -0.5f)) b.y += u.userInputMove[1] * 5.0f * weight * inten;
      } else if (ut >= 10) {
        if (inten > 0.0f) {
          bool setWall = false;
          const bool isWall = wl[DISTANCE] == 0;
          const int ty = wl[TYPE];
          switch (ut) {
          case 10: wl[TYPE] = WALLTYPE_INERT; setWall = true; break;
          case 11: wl[TYPE] = WALLTYPE_LAND; setWall = true; break;
          case 12: wl[TYPE] = WALLTYPE_WATER; setWall = true; break;
          case 13:
            if (isWall && ty == WALLTYPE_LAND && aboveIsAir) {
              wl[TYPE] = WALLTYPE_FIRE;
              setWall = true;
            }
            break;
          case 14:
            if (isWall && (ty == WALLTYPE_LAND || ty == WALLTYPE_RUNWAY || ty == WALLTYPE_INDUSTRIAL) && aboveIsAir) wl[TYPE] = WALLTYPE_URBAN;
            break;
          case 15:
            if (isWall && (ty == WALLTYPE_LAND || ty == WALLTYPE_URBAN || ty == WALLTYPE_INDUSTRIAL) && aboveIsAir) wl[TYPE] = WALLTYPE_RUNWAY;
            break;
          case 16:
            if (isWall && (ty == WALLTYPE_LAND || ty == WALLTYPE_URBAN || ty == WALLTYPE_RUNWAY) && aboveIsAir) wl[TYPE] = WALLTYPE_INDUSTRIAL;
            break;
          case 20:
            if (isWall && ty != WALLTYPE_WATER && aboveIsAir) w.z += inten * 10.0f;
            break;
          case 21:
            if (isWall && (ty == WALLTYPE_LAND || ty == WALLTYPE_URBAN || ty == WALLTYPE_INDUSTRIAL) && aboveIsAir) w.w += inten * 0.5f;
            break;
          case 22:
            if (isWall && (ty == WALLTYPE_LAND || ty == WALLTYPE_FIRE || ty == WALLTYPE_URBAN || ty == WALLTYPE_INDUSTRIAL) && aboveIsAir) wl[VEGETATION] += 1;
            break;
          default: break;
          }
          if (setWall) {
            wl[DISTANCE] = 0;
            b.w = 1000.0f;
            if (wl[TYPE] == WALLTYPE_LAND)
              w.z = 25.0f;
            else if (wl[TYPE] == WALLTYPE_WATER)
              b.w = u.waterTemperature;
          }
        } else if (wl[DISTANCE] == 0) {
          if (ut == 13) {
            if (wl[TYPE] == WALLTYPE_FIRE) wl[TYPE] = WALLTYPE_LAND;
          } else if (ut == 14) {
            if (wl[TYPE] == WALLTYPE_URBAN) wl[TYPE] = WALLTYPE_LAND;
          } else if (ut == 15) {
            if (wl[TYPE] == WALLTYPE_RUNWAY) wl[TYPE] = WALLTYPE_LAND;
          } else if (ut == 16) {
            if (wl[TYPE] == WALLTYPE_INDUSTRIAL) wl[TYPE] = WALLTYPE_LAND;
          } else if (ut == 20) {
            w.z += inten * 10.0f;
          } else if (ut == 21) {
            w.w += inten * 0.5f;
          } else if (ut == 22) {
            wl[VEGETATION] = max(wl[VEGETATION] - 1, 0);
          } else if (tcy > a_texY) {
            wl[DISTANCE] = 255;
            b.x = 0.0f;
            b.y = 0.0f;
            b.z = 0.0f;
            b.w = initial_T[(int)(tcy * u.a_invTexY)];
            w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          }
        }
      }
    }
  }

  if (wl[DISTANCE] == 0) w.x = (wl[TYPE] == WALLTYPE_WATER) ? 1002.0f : 1001.0f; // :403-409

  // airplane :415-457 (only when the host set a dump (<0) or crash (>0.9) flag)
  if (!QUIET && (u.airplaneValues[3] < 0.0f || u.airplaneValues[3] > 0.9f)) {
    float vpx, vpy = u.airplaneValues[1] - tcy;
    if (u.wrapHorizontally)
      vpx = absHorizontalDist(u.airplaneValues[0], tcx);
    else
      vpx = fabsf(u.airplaneValues[0] - tcx);
    vpx *= u.a_aspect;
    vpx *= (float)g.Y;
    vpy *= (float)g.Y;
    if (u.airplaneValues[3] < 0.0f) vpy += -1.0f;
    const float distFromPlane = sqrtf(vpx * vpx + vpy * vpy);
    const float planeInfluence = fmaxf(1.0f - distFromPlane, 0.0f) * 0.03f;
    if (u.airplaneValues[3] < 0.0f) w.z += planeInfluence * 100.0f;
    if (u.airplaneValues[3] > 0.9f && distFromPlane < 1.5f) {
      if (wl[DISTANCE] == 0) {
        if (wl[TYPE] == WALLTYPE_LAND && wl[VERT_DISTANCE] == 0) wl[TYPE] = WALLTYPE_FIRE;
      } else {
        b.z += 0.05f;
        b.w = CtoK(50.0f);
        w.x += 1.0f;
        w.w += 10.0f;
      }
    }
  }

  base_out = b;
  water_out = w;
  wall_out = pack_wall(wl);
}

// common.glsl:367-372; mix() lowered like mixf()
__host__ __device__ __forceinline__ void hsv2rgb(float h, float sv, float v, float rgb[3])
{
  const float K[3] = {1.0f, 2.0f / 3.0f, 1.0f / 3.0f};
  for (int i = 0; i < 3; i++) {
    const float t = h + K[i];
    const float pch = fabsf((t - floorf(t)) * 6.0f - 3.0f);
    const float c = fminf(fmaxf(pch - 1.0f, 0.0f), 1.0f);
    rgb[i] = v * (1.0f + sv * (c - 1.0f));
  }
}

// ------------------------------------------------------------------------------------------------
// lightingShader.frag:38-170
// Accessor A: T(dy) -> base TEMPERATURE at (x, y+dy) with y REPEAT ; water() -> float4 ; wall() -> char4 ;
//             sun_at(dx, j) / ir_down_at(j) / ir_up_at(j) -> channel x / z / w of the source light texture at column x+dx
//             (wrapped; 0 for the IR channels), ROW j (absolute,
//             caller clamps) ; used for the bilinear sun tap and the IR taps
// ------------------------------------------------------------------------------------------------
// AIR (wave-uniform, established by the caller): the cell is fluid and not directly above a wall (wall.y != 0, wall.z != 1).
// EMIT: also produce the pass's second output (`reflectedLight`, bound to the RGBA16F emittedLight texture that only the display
// shaders read): rgb in *emit, alpha is never written by the shader (stays 0). The accumulating `+=` on the never-initialised
// output variable starts from 0, as SwiftShader (and every WebGL implementation, which must zero-initialise) does.
template <bool AIR = false, bool EMIT = false, class UT, class A>
__device__ __forceinline__ float4 lighting_cell(const UT &u, const Geo &g, int x, int y, const A &a, float4 *emit = nullptr)
{
  const CellCoord cc = cellcoord(g, x, y);
  const float fy = cc.fy, tcy = cc.tcy;
  const int Y = g.Y;
  const float resY = (float)Y;
  if (EMIT) *emit = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (fy >= resY - 1.0f) return make_float4(u.sunIntensity, 0.0f, 0.0f, 0.0f); // :40-41 (the top row leaves emittedLight untouched: 0)
  const float cellHeightCompensation = u.chc;

  // :48-49 LINEAR tap at texCoord + (sin a, cos a) texels, S=REPEAT T=CLAMP_TO_EDGE. The filter weights are
  // the exact fp32 fractions of the offset (weight precision is implementation defined in GL).
  float sunlight;
  if (u.sun_uniform) { // (uniform branch) the same taps and the same products as below, computed once on the host
    const int dx0 = u.sun_dx0;
    int j0 = y + u.sun_fv, j1 = y + u.sun_fv + 1;
    j0 = j0 < 0 ? 0 : (j0 > Y - 1 ? Y - 1 : j0);
    j1 = j1 < 0 ? 0 : (j1 > Y - 1 ? Y - 1 : j1);
    const float t00 = a.sun_at(dx0, j0), t10 = a.sun_at(dx0 + 1, j0);
    const float t01 = a.sun_at(dx0, j1), t11 = a.sun_at(dx0 + 1, j1);
    sunlight = u.sun_w00 * t00 + u.sun_w10 * t10 + u.sun_w01 * t01 + u.sun_w11 * t11;
  } else {
    const float ox = (cc.fx - ((float)cc.gx + 0.5f)) + u.sin_a;
    const float oy = (fy - ((float)y + 0.5f)) + u.cos_a;
    const float fu = floorf(ox), fv = floorf(oy);
    const float al = ox - fu, be = oy - fv;
    const int dx0 = (int)fu;
    int j0 = y + (int)fv, j1 = y + (int)fv + 1;
    j0 = j0 < 0 ? 0 : (j0 > Y - 1 ? Y - 1 : j0);
    j1 = j1 < 0 ? 0 : (j1 > Y - 1 ? Y - 1 : j1);
    const float t00 = a.sun_at(dx0, j0), t10 = a.sun_at(dx0 + 1, j0);
    const float t01 = a.sun_at(dx0, j1), t11 = a.sun_at(dx0 + 1, j1);
    sunlight = (1.0f - al) * (1.0f - be) * t00 + al * (1.0f - be) * t10 + (1.0f - al) * be * t01 + al * be * t11;
  }

  const float realTemp = a.T(0) - tcy * u.dryLapse;
  const float4 water = a.water();
  const char4 wall = a.wall();

  const float standardSunBrightness = 1250.0f; // common.glsl:11
  if (AIR || wall.y != 0) {
    float net_heating = 0.0f;
    float er = 0.0f, eg = 0.0f, eb = 0.0f;
    if (EMIT) { // scattering in air (:60)
      er = u.sun_col[0] * sunlight * (1.0f - tcy) * 2.0f / standardSunBrightness;
      eg = u.sun_col[1] * sunlight * (1.0f - tcy) * 2.0f / standardSunBrightness;
      eb = u.sun_col[2] * sunlight * (1.0f - tcy) * 2.0f / standardSunBrightness;
    }
    if (fy < resY - 2.0f) {
      float reflection = fminf(sqrtf(water.y * 0.0010f + water.z * 0.00020f) * cellHeightCompensation, 1.0f); // pow(x, 0.5)
      reflection += 0.0002f;
      const float absorbtion = fminf(water.w * 0.020f * cellHeightCompensation, 1.0f);
      const float lightReflected = sunlight * reflection;
      const float lightAbsorbed = sunlight * absorbtion;
      sunlight = fmaxf(0.0f, sunlight - lightReflected - lightAbsorbed);
      net_heating += lightAbsorbed * lightHeatingConst;
      if (EMIT) { // sunlight reflected by clouds and precipitation REPLACES the scattering term (:78)
        er = u.sun_col[0] * lightReflected / standardSunBrightness;
        eg = u.sun_col[1] * lightReflected / standardSunBrightness;
        eb = u.sun_col[2] * lightReflected / standardSunBrightness;
      }
    }
    const int yu = (y + 1 >= Y) ? Y - 1 : y + 1; // light texture: CLAMP_TO_EDGE in T
    const int yd = (y == 0) ? 0 : y - 1;
    float IR_down = a.ir_down_at(yu);
    float IR_up = 0.0f; // unassigned for air above an INERT wall (:90) -> 0
    if (!AIR && wall.z == 1) {
      if (EMIT && u.night_glow && (wall.x == WALLTYPE_RUNWAY || wall.x == WALLTYPE_URBAN || wall.x == WALLTYPE_INDUSTRIAL)) { // :98-101
        er += 1.00f * 0.03f;
        eg += 0.97f * 0.03f;
        eb += 0.57f * 0.03f;
      }
      switch (wall.x) {
      case WALLTYPE_RUNWAY:
      case WALLTYPE_URBAN:
      case WALLTYPE_INDUSTRIAL:
      case WALLTYPE_LAND:
        IR_up = IR_emitted(realTemp);
        net_heating += (IR_down - IR_up) * lightHeatingConst;
        break;
      case WALLTYPE_WATER: {
        const float waterTemperature = a.T(-1);
        IR_up = IR_emitted(waterTemperature);
        net_heating += (IR_down - IR_up) * lightHeatingConst;
        break;
      }
      case WALLTYPE_FIRE:
        IR_up = IR_emitted(realTemp + 100.0f);
        net_heating = 0.0f;
        break;
      default: break;
      }
    } else {
      IR_up = a.ir_up_at(yd);
      float emissivity = u.greenhouseGases;
      emissivity += water.x * u.waterGreenHouseEffect;
      emissivity += water.y * 5.0f;
      emissivity *= cellHeightCompensation;
      emissivity = fminf(emissivity, 1.0f);
      const float absorbedDown = IR_down * emissivity;
      const float absorbedUp = IR_up * emissivity;
      const float emitted = IR_emitted(realTemp) * emissivity;
      net_heating += (absorbedDown + absorbedUp - emitted * 2.0f) * lightHeatingConst;
      IR_down -= absorbedDown;
      IR_down += emitted;
      IR_up -= absorbedUp;
      IR_up += emitted;
    }
    if (EMIT) { // glow of thick smoke = fire (:143-148)
      const float smokeOpacity = clampf(1.0f - (1.0f / (water.w + 1.0f)), 0.0f, 1.0f);
      const float fireIntensity = clampf((smokeOpacity - 0.8f) * 25.0f, 0.0f, 1.0f);
      float fc[3];
      hsv2rgb(fireIntensity * 0.008f, 0.98f, 5.0f, fc);
      er += (fireIntensity * fc[0]) * 0.1f; // mix(vec3(0), fireCol, fireIntensity)
      eg += (fireIntensity * fc[1]) * 0.1f;
      eb += (fireIntensity * fc[2]) * 0.1f;
      *emit = make_float4(er, eg, eb, 0.0f);
    }
    net_heating *= u.IR_rate;
    return make_float4(sunlight, net_heating, IR_down, IR_up);
  }
  if (EMIT && wall.x != WALLTYPE_WATER) // land: part of the light is reflected by the ground (:158-166)
    *emit = make_float4(sunlight * 0.60f / standardSunBrightness, sunlight * 0.5f / standardSunBrightness, sunlight * 0.4f / standardSunBrightness, 0.0f);
  return make_float4((wall.x == WALLTYPE_WATER) ? sunlight * 0.90f : 0.0f, 0.0f, 0.0f, 0.0f);
}

} // namespace wx
