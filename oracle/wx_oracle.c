/*
 * wx_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see wx_oracle.h header).
 *
 * Plain-C restatement of the reference simulation iteration. One function per shader, written
 * from the GLSL under the semantics of SURVEY.md Appendix A:
 *   - NEAREST sampling, REPEAT wrap on both axes for base/water/wall/curl/vort/feedback
 *     (app.js:5191-5305 never sets TEXTURE_WRAP_*); light textures LINEAR, S=REPEAT,
 *     T=CLAMP_TO_EDGE (app.js:5276-5290)
 *   - fp32 arithmetic evaluated left to right with NO fused multiply-add
 *     (build with -ffp-contract=off)
 *   - RGBA8I stores saturate to [-128,127]
 *   - mix(a,b,t) = a + t*(b-a) (pinned by the goldens) ; clamp = min(max()) ; mod(x,y) = x - y*floor(x/y)
 * Deliberate, documented deviations (all inside GLSL's "undefined"/implementation-defined room):
 *   - integer modulo by zero (boundaryShader.frag:462 when vegetationGrowthRate > 100) -> condition false
 *   - sounding index y-1 < 0 (advectionShader.frag:59-61, row 0) -> clamped to 0
 *   - sin/cos of the uniform sunAngle are evaluated once on the host (sinf/cosf) per pass
 *   - pow(x, c) with constant c in {17, 4, 2, 0.5, 1/3}: fixed multiply chains / sqrt / Newton cbrt
 *     (see pow17, pow4, det_cbrt)
 *   - unassigned IR_up (lightingShader.frag:90, air above an INERT wall) = 0
 *   - additive particle splats are summed in droplet-index order
 */
#include "wx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* channel enums: common.glsl:42-95 */
enum { VX = 0, VY = 1, PRESSURE = 2, TEMPERATURE = 3 };
enum { TOTAL = 0, CLOUD = 1, PRECIPITATION = 2, SOIL_MOISTURE = 2, SMOKE = 3, SNOW = 3 };
enum { TYPE = 0, DISTANCE = 1, VERT_DISTANCE = 2, VEGETATION = 3 };
enum { SUNLIGHT = 0, NET_HEATING = 1, IR_DOWN = 2, IR_UP = 3 };
enum { MASS = 0, HEAT = 1, VAPOR = 2 };
enum {
  WALLTYPE_INERT = 0,
  WALLTYPE_LAND = 1,
  WALLTYPE_WATER = 2,
  WALLTYPE_FIRE = 3,
  WALLTYPE_URBAN = 4,
  WALLTYPE_RUNWAY = 5,
  WALLTYPE_INDUSTRIAL = 6
};

/* common.glsl:9-35 */
#define lightHeatingConst 0.000002f
#define waterHeatExchangeRate 0.0002f
#define waterHeatCapacity 50.0f
#define fullWhiteSnowHeight 10.0f
#define snowMassToHeight 0.05f
#define snowMeltRate 0.000015f
#define maxWaterTemp 40.0f
#define ALBEDO_SNOW 0.85f
#define ALBEDO_SNOW_FOREST 0.30f
#define ALBEDO_FOREST 0.10f
#define ALBEDO_DRYSOIL 0.30f
#define ALBEDO_WETSOIL 0.15f
#define ALBEDO_URBAN 0.08f
#define ALBEDO_INDUSTRIAL 0.08f
#define ALBEDO_RUNWAY 0.04f
#define ALBEDO_WATER 0.05f

typedef struct {
  int X, Y, Xg, xoff;
  float sx, sy;     /* quad UV scale per axis */
  float texX, texY; /* texelSize uniform = f32(1/res) */
  const float *vary; /* optional measured varyings (fragCoord.xy, texCoord.xy) per cell, or NULL */
} geo_t;

static geo_t mkgeo(const wxo_params *p)
{
  geo_t g;
  g.X = p->X;
  g.Y = p->Y;
  g.Xg = p->X_global > 0 ? p->X_global : p->X;
  g.xoff = p->x_off;
  if (p->quad_scale) {
    /* app.js:4770-4788: U runs 0 .. f32(res*1.0000001); fragCoord = U*(i+0.5)/res */
    g.sx = (float)((double)g.Xg * 1.0000001) / (float)g.Xg;
    g.sy = (float)((double)g.Y * 1.0000001) / (float)g.Y;
  } else {
    g.sx = 1.0f;
    g.sy = 1.0f;
  }
  g.texX = (float)(1.0 / (double)g.Xg); /* app.js:5436-5437 */
  g.texY = (float)(1.0 / (double)g.Y);
  g.vary = p->varyings;
  return g;
}

static inline int wrapmod(int i, int n)
{
  int r = i % n;
  return r < 0 ? r + n : r;
}
static inline int gx_of(const geo_t *g, int x) { return wrapmod(g->xoff + x, g->Xg); }
/* simShader.vert:23 fragCoord, :24 texCoord */
typedef struct {
  float fx, fy, tcx, tcy;
} cc_t;
static inline cc_t cellcoord(const geo_t *g, int x, int y)
{
  cc_t c;
  if (g->vary) { /* what the reference's rasteriser actually interpolated (golden runs only) */
    const float *v = g->vary + 4 * ((size_t)y * g->X + x);
    c.fx = v[0];
    c.fy = v[1];
    c.tcx = v[2];
    c.tcy = v[3];
  } else {
    c.fx = ((float)gx_of(g, x) + 0.5f) * g->sx;
    c.fy = ((float)y + 0.5f) * g->sy;
    c.tcx = c.fx * g->texX;
    c.tcy = c.fy * g->texY;
  }
  return c;
}

static inline int8_t sat8(int v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
/* mix(): GLSL defines x*(1-a)+y*a; the SwiftShader goldens are reproduced (T bit-exact in 96 % of cells
 * instead of 68 %) by the equally common lowering a + t*(b-a), which is what is pinned here. */
static inline float mixf(float a, float b, float t) { return a + t * (b - a); }
/* common.glsl:99-101 */
static inline float map_range(float v, float min1, float max1, float min2, float max2)
{
  return min2 + (v - min1) * (max2 - min2) / (max1 - min1);
}
static inline float map_rangeC(float v, float min1, float max1, float min2, float max2)
{
  return clampf(map_range(v, min1, max1, min2, max2), fminf(min2, max2), fmaxf(min2, max2));
}
static inline float CtoK(float c) { return c + 273.15f; }
static inline float KtoC(float k) { return k - 273.15f; }
/* pow() with the constant exponents the grid passes use. GLSL pow is not correctly rounded on any GPU
 * (SURVEY Appendix A), so its value is only defined to a few ulp; the restatement fixes ONE evaluation
 * order built from exactly-rounded fp32 multiplies / sqrt, which CPU and GPU reproduce bit for bit. */
static inline float pow17(float x)
{
  const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
  return x16 * x;
}
static inline float pow4(float x)
{
  const float x2 = x * x;
  return x2 * x2;
}
/* common.glsl:177-180: pow(T / 250.0, 17.0) */
static inline float maxWater(float T) { return pow17(T / 250.0f); }
/* common.glsl:258-261: pow(T * 0.01, 4.) * IR_constant */
static inline float IR_emitted(float T) { return pow4(T * 0.01f) * 5.670374419f; }

static inline uint32_t f2u(float f);
static inline float u2f(uint32_t u);
/* common.glsl:103-111 */
uint32_t wxo_hash(uint32_t x)
{
  x += (x << 10u);
  x ^= (x >> 6u);
  x += (x << 3u);
  x ^= (x >> 11u);
  x += (x << 15u);
  return x;
}
static inline uint32_t f2u(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float u2f(uint32_t u)
{
  float f;
  memcpy(&f, &u, 4);
  return f;
}
/* pow(x, 1./3.) for x > 0 (precipitationShader.vert:195) as a fixed sequence of exactly-rounded
 * operations: bit-level seed + 4 Newton steps (bit-reproducible on CPU and GPU, ~1 ulp) */
static inline float det_cbrt(float x)
{
  float y = u2f(f2u(x) / 3u + 709921077u);
  for (int i = 0; i < 4; i++) y = (2.0f * y + x / (y * y)) / 3.0f;
  return y;
}
/* common.glsl:126-137 */
float wxo_random2d(float sx, float sy)
{
  uint32_t h = wxo_hash(f2u(sx) + wxo_hash(f2u(sy)));
  h &= 0x007FFFFFu;
  h |= 0x3F800000u;
  float r2 = u2f(h);
  return r2 - 1.0f * floorf(r2 / 1.0f); /* mod(r2, 1.0) */
}

#define C4(arr, x, y) ((arr) + 4 * ((size_t)(y) * X + (x)))

/* ------------------------------------------------------------------------------------------
 * velocityShader.frag:32-61
 * ---------------------------------------------------------------------------------------- */
void wxo_velocity(const wxo_params *p, const float *base_in, const int8_t *wall_in, float *base_out,
                  int8_t *wall_out)
{
  const int X = p->X, Y = p->Y;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < Y; y++) {
    const int yu = (y + 1 == Y) ? 0 : y + 1;
    for (int x = 0; x < X; x++) {
      const int xr = (x + 1 == X) ? 0 : x + 1;
      const float *b = C4(base_in, x, y);
      const int8_t *w = C4(wall_in, x, y);
      float vx = b[VX], vy = b[VY];
      const float P = b[PRESSURE];
      if (w[DISTANCE] == 0) {
        vx = 0.0f;
        vy = 0.0f;
      } else {
        vx += P - C4(base_in, xr, y)[PRESSURE];
        vy += P - C4(base_in, x, yu)[PRESSURE];
        vx *= 1.0f - p->dragMultiplier * 0.0002f;
        vy *= 1.0f - p->dragMultiplier * 0.0002f;
        vx += p->wind * 0.000001f;
      }
      float *o = C4(base_out, x, y);
      o[VX] = vx;
      o[VY] = vy;
      o[PRESSURE] = P;
      o[TEMPERATURE] = b[TEMPERATURE];
      memcpy(C4(wall_out, x, y), w, 4);
    }
  }
}

/* curlShader.frag:12-19 */
void wxo_curl(const wxo_params *p, const float *base_in, float *curl_out)
{
  const int X = p->X, Y = p->Y;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < Y; y++) {
    const int yu = (y + 1 == Y) ? 0 : y + 1;
    for (int x = 0; x < X; x++) {
      const int xr = (x + 1 == X) ? 0 : x + 1;
      const float *c = C4(base_in, x, y);
      curl_out[(size_t)y * X + x] = C4(base_in, x, yu)[0] - c[0] - C4(base_in, xr, y)[1] + c[1];
    }
  }
}

/* vorticityShader.frag:19-38 */
void wxo_vorticity(const wxo_params *p, const float *curl_in, float *vort_out)
{
  const int X = p->X, Y = p->Y;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < Y; y++) {
    const int yu = (y + 1 == Y) ? 0 : y + 1, yd = (y == 0) ? Y - 1 : y - 1;
    for (int x = 0; x < X; x++) {
      const int xr = (x + 1 == X) ? 0 : x + 1, xl = (x == 0) ? X - 1 : x - 1;
      const float curl = curl_in[(size_t)y * X + x];
      const float cl = curl_in[(size_t)y * X + xl], cr = curl_in[(size_t)y * X + xr];
      const float cd = curl_in[(size_t)yd * X + x], cu = curl_in[(size_t)yu * X + x];
      float fx = fabsf(cd) - fabsf(cu);
      float fy = fabsf(cr) - fabsf(cl);
      const float magnitude = sqrtf(fx * fx + fy * fy) + 0.0001f;
      fx /= magnitude;
      fy /= magnitude;
      fx *= curl;
      fy *= curl;
      vort_out[2 * ((size_t)y * X + x) + 0] = fx;
      vort_out[2 * ((size_t)y * X + x) + 1] = fy;
    }
  }
}

/* pressureShader.frag:16-43 */
void wxo_pressure(const wxo_params *p, const float *base_in, const int8_t *wall_in, float *base_out,
                  int8_t *wall_out)
{
  const int X = p->X, Y = p->Y;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < Y; y++) {
    const int yd = (y == 0) ? Y - 1 : y - 1;
    for (int x = 0; x < X; x++) {
      const int xl = (x == 0) ? X - 1 : x - 1;
      const float *b = C4(base_in, x, y);
      const float *bl = C4(base_in, xl, y);
      const float *bd = C4(base_in, x, yd);
      const int8_t *wd = C4(wall_in, x, yd);
      float T = b[3];
      if (wd[1] == 0 && wd[0] == 1) T -= bd[3] - 1000.0f;
      float *o = C4(base_out, x, y);
      o[0] = b[0];
      o[1] = b[1];
      o[2] = b[2] + (bl[0] - b[0] + bd[1] - b[1]) * 0.45f;
      o[3] = T;
      memcpy(C4(wall_out, x, y), C4(wall_in, x, y), 4);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * boundaryShader.frag:72-531
 * ---------------------------------------------------------------------------------------- */
/* boundaryShader.frag:65-68 */
static inline float calcEvaporation(const wxo_params *p, float T, float W, float V, float M)
{
  return fmaxf((maxWater(T) - W) * p->landEvaporation * (V / 127.0f + 0.1f) * fminf(M + 1.0f, 50.0f) * 0.05f, 0.0f);
}
/* boundaryShader.frag:70 */
static inline float calcFireIntensity(int veg, float moist, float precip)
{
  return fmaxf((float)veg * 0.00025f - moist * 0.00020f - precip * 0.02f, 0.0f);
}

void wxo_boundary(const wxo_params *p, const float *initial_T, float iterNum, const float *base_in,
                  const float *water_in, const float *vort_in, const int8_t *wall_in,
                  const float *light_in, const float *fb_in, const float *dep_in, float *base_out,
                  float *water_out, int8_t *wall_out)
{
  const geo_t g = mkgeo(p);
  const int X = p->X, Y = p->Y;
  const float gravMult = 0.0001f;
  const float exchangeRate = 0.015f;
  const float cos_a = cosf(p->sunAngle), sin_a = sinf(p->sunAngle), sin_ma = sinf(-p->sunAngle);
  const int iterI = (int)iterNum;

#pragma omp parallel for schedule(static)
  for (int y = 0; y < Y; y++) {
    const int yu = (y + 1 == Y) ? 0 : y + 1, yd = (y == 0) ? Y - 1 : y - 1;
    const int yu_clamp = (y + 1 >= Y) ? Y - 1 : y + 1; /* light texture: CLAMP_TO_EDGE in T */
    for (int x = 0; x < X; x++) {
      const int xr = (x + 1 == X) ? 0 : x + 1, xl = (x == 0) ? X - 1 : x - 1;
      const cc_t cc = cellcoord(&g, x, y);
      const float tcy = cc.tcy;
      const float tcy_up = tcy + g.texY; /* texCoordX0Yp.y, simShader.vert:30 */
      float b[4], w[4];
      int wl[4];
      memcpy(b, C4(base_in, x, y), 16);
      memcpy(w, C4(water_in, x, y), 16);
      const float *fb = C4(fb_in, x, y);
      const float realTemp = b[TEMPERATURE] - tcy * p->dryLapse;
      const int8_t *w0 = C4(wall_in, x, y), *wL = C4(wall_in, xl, y), *wD = C4(wall_in, x, yd),
                   *wR = C4(wall_in, xr, y), *wU = C4(wall_in, x, yu);
      for (int c = 0; c < 4; c++) wl[c] = w0[c];
      const float *light = C4(light_in, x, y);
      int nextToWall = 0;

      wl[VERT_DISTANCE] = wD[VERT_DISTANCE] + 1;

      if (wl[DISTANCE] != 0) { /* fluid */
        wl[TYPE] = wD[TYPE];
        if (wl[TYPE] != WALLTYPE_WATER) b[TEMPERATURE] += light[NET_HEATING];
        b[TEMPERATURE] += fb[HEAT];

        const float precipCoalescence = fmaxf(-fb[VAPOR], 0.0f);
        w[CLOUD] -= precipCoalescence;
        w[TOTAL] -= precipCoalescence;
        const float precipEvaporation = fmaxf(fb[VAPOR], 0.0f);
        w[TOTAL] += precipEvaporation;

        w[PRECIPITATION] = fmaxf(w[PRECIPITATION] * 0.997f - 0.00001f + fb[MASS] * 0.005f, 0.0f);

        w[SMOKE] /= 1.0f + fmaxf(-fb[VAPOR] * 0.1f, 0.0f) + fb[MASS] * 0.000f;
        w[SMOKE] -= fb[MASS] * 0.0001f;
        w[SMOKE] -= fmaxf((w[SMOKE] - 4.0f) * 0.01f, 0.0f);
        w[SMOKE] = fmaxf(w[SMOKE], 0.0f);
        if (w[SMOKE] > 4.0f) w[SMOKE] -= w[PRECIPITATION] * 0.02f;

        /* GRAVITY :132-148 */
        const float *bU = C4(base_in, x, yu);
        float gravityForce =
          ((b[TEMPERATURE] + bU[TEMPERATURE]) * 0.5f - (initial_T[(int)cc.fy] + initial_T[(int)cc.fy + 1]) * 0.5f) * gravMult;
        gravityForce -= w[CLOUD] * gravMult * p->waterWeight;
        gravityForce -= fb[MASS] * gravMult * p->waterWeight;
        b[VY] += gravityForce;

        float snowCover = 0.0f, soilMoisture = 0.0f;

        if (wD[DISTANCE] == 0) { /* below is wall */
          nextToWall = 1;
          wl[DISTANCE] = 1;
          const float *wtD = C4(water_in, x, yd);
          snowCover = wtD[SNOW];
          soilMoisture = wtD[SOIL_MOISTURE];
          wl[VERT_DISTANCE] = 1;
        }
        if (wL[DISTANCE] == 0) { /* left is wall */
          nextToWall = 1;
          wl[DISTANCE] = 1;
          if (wL[TYPE] == WALLTYPE_WATER) {
            wl[TYPE] = WALLTYPE_LAND;
            wl[DISTANCE] = 0;
          }
          if (wR[DISTANCE] == 0) wl[DISTANCE] = 0;
        } else if (wR[DISTANCE] == 0) { /* right is wall */
          nextToWall = 1;
          wl[DISTANCE] = 1;
          if (wR[TYPE] == WALLTYPE_WATER) {
            wl[TYPE] = WALLTYPE_LAND;
            wl[DISTANCE] = 0;
          }
        }
        if (wU[DISTANCE] == 0) { /* above is wall */
          nextToWall = 1;
          wl[DISTANCE] = 1;
          if (tcy < 0.99f) wl[DISTANCE] = 0;
        }

        /* vorticity force :199-208 */
        const float *vf00 = vort_in + 2 * ((size_t)y * X + x);
        const float *vfL = vort_in + 2 * ((size_t)y * X + xl);
        const float *vfD = vort_in + 2 * ((size_t)yd * X + x);
        const float velocityFactor = sqrtf(b[VX] * b[VX] + b[VY] * b[VY]) * 0.1f;
        b[VX] += (vf00[0] + vfD[0]) * (p->vorticity + velocityFactor);
        b[VY] += (vf00[1] + vfL[1]) * (p->vorticity + velocityFactor);

        if (nextToWall) {
          if (wl[TYPE] != WALLTYPE_WATER) {
            float lightPower = 0.0f;
            if (wD[DISTANCE] == 0) lightPower += fmaxf(light[SUNLIGHT] * cos_a, 0.0f);
            if (wL[DISTANCE] == 0) lightPower += fmaxf(light[SUNLIGHT] * sin_a, 0.0f);
            if (wR[DISTANCE] == 0) lightPower += fmaxf(light[SUNLIGHT] * sin_ma, 0.0f);
            float albedoTotal = 1.0f;
            if (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_FIRE) {
              float albedoSoil = map_rangeC(soilMoisture, 0.0f, 20.0f, ALBEDO_DRYSOIL, ALBEDO_WETSOIL);
              albedoSoil = map_rangeC(snowCover, 0.0f, fullWhiteSnowHeight, albedoSoil, ALBEDO_SNOW);
              const float fullVegetationAlbedo =
                map_range(snowCover, 0.0f, fullWhiteSnowHeight, ALBEDO_FOREST, ALBEDO_SNOW_FOREST);
              albedoTotal = map_range((float)wD[VEGETATION], 0.0f, 127.0f, albedoSoil, fullVegetationAlbedo);
            } else if (wl[TYPE] == WALLTYPE_URBAN) {
              albedoTotal = ALBEDO_URBAN;
            } else if (wl[TYPE] == WALLTYPE_INDUSTRIAL) {
              albedoTotal = ALBEDO_INDUSTRIAL;
            } else if (wl[TYPE] == WALLTYPE_RUNWAY) {
              albedoTotal = ALBEDO_RUNWAY;
            }
            lightPower *= (1.0f - albedoTotal);
            lightPower *= lightHeatingConst;
            b[TEMPERATURE] += lightPower;
          }
        }

        if (!nextToWall) {
          int nearest = 255;
          if (wD[DISTANCE] < nearest) nearest = wD[DISTANCE];
          if (wU[DISTANCE] < nearest) nearest = wU[DISTANCE];
          if (wL[DISTANCE] < nearest) nearest = wL[DISTANCE];
          if (wR[DISTANCE] < nearest) nearest = wR[DISTANCE];
          wl[DISTANCE] = nearest + 1;
        }

        if (wl[VERT_DISTANCE] <= 5) { /* surfaceWindSmootingDist :271-303 */
          if (wl[VERT_DISTANCE] == 1) {
            float surfaceDrag = 0.0015f;
            if (wl[TYPE] == WALLTYPE_URBAN)
              surfaceDrag = 0.040f;
            else if (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_FIRE)
              surfaceDrag = map_rangeC((float)wl[VEGETATION], 50.0f, 127.0f, 0.0015f, 0.020f);
            b[VX] -= fabsf(b[VX]) * b[VX] * surfaceDrag * 50.0f;
          }
          if (wU[VERT_DISTANCE] <= 5) b[VX] -= (b[VX] - C4(base_in, x, yu)[VX]) * exchangeRate;
          if (wD[VERT_DISTANCE] > 0) b[VX] -= (b[VX] - C4(base_in, x, yd)[VX]) * exchangeRate;
        }

        if (wl[VERT_DISTANCE] <= 8) { /* :305-372 */
          wl[VEGETATION] = wD[VEGETATION];
          const float *waterInSurface = C4(water_in, x, yd);
          const int t = wl[TYPE];
          if (t == WALLTYPE_FIRE) {
            if (wl[VERT_DISTANCE] == 1) {
              float fireIntensity =
                calcFireIntensity(wl[VEGETATION], waterInSurface[SOIL_MOISTURE], w[PRECIPITATION]);
              fireIntensity = fmaxf(fireIntensity, 0.0f);
              b[TEMPERATURE] += fireIntensity;
              w[SMOKE] += fireIntensity * 2.0f;
              w[TOTAL] += fireIntensity * 0.50f;
            }
          }
          if (t == WALLTYPE_INDUSTRIAL) { /* (FIRE falls through but is excluded :330) */
            const int texFragX = (int)cc.fx % 80;
            if (wl[VERT_DISTANCE] == 5 && (texFragX == 18 || texFragX == 22)) {
              w[TOTAL] += 0.25f;
              b[VX] *= 0.5f;
              b[VY] *= 0.5f;
              b[VY] += 0.05f;
            } else if (wl[VERT_DISTANCE] == 6 && texFragX == 29) {
              w[SMOKE] += 0.01f;
              b[TEMPERATURE] += 0.02f;
              b[VX] *= 0.5f;
              b[VY] *= 0.5f;
            }
          }
          if (t == WALLTYPE_FIRE || t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN) w[SMOKE] += 0.000002f;
          if (t == WALLTYPE_FIRE || t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN || t == WALLTYPE_LAND) {
            if (wl[VERT_DISTANCE] <= 1) {
              const float evaporation =
                calcEvaporation(p, realTemp, w[TOTAL], (float)wl[VEGETATION], waterInSurface[SOIL_MOISTURE]) / 1.0f;
              w[TOTAL] += evaporation;
              b[TEMPERATURE] -= evaporation * p->evapHeat * 0.5f;
              if (wl[VEGETATION] < 10 && w[SOIL_MOISTURE] < 5.0f)
                w[SMOKE] = fminf(w[SMOKE] + (fmaxf(fabsf(b[VX]) - 0.12f, 0.0f) * 0.15f), 2.4f);
            }
          } else if (t == WALLTYPE_WATER) {
            if (wl[VERT_DISTANCE] <= 1) {
              const float LocalWaterTemperature = C4(base_in, x, yd)[TEMPERATURE];
              b[TEMPERATURE] += (LocalWaterTemperature - realTemp - 1.0f) / 1.0f * waterHeatExchangeRate;
              w[TOTAL] += fmaxf((maxWater(LocalWaterTemperature) - w[TOTAL]) * p->waterEvaporation / 1.0f, 0.0f);
            }
          }
        }
      } else { /* this is wall :373-530 */
        wl[VERT_DISTANCE] = wU[VERT_DISTANCE] - 1;

        if (wl[VERT_DISTANCE] < 0) {
          const float *wtU = C4(water_in, x, yu);
          w[2] = wtU[2];
          w[3] = wtU[3];
          wl[VEGETATION] = wU[VEGETATION];
          if (wU[DISTANCE] == 0) {
            if (wU[TYPE] != WALLTYPE_WATER) {
              wl[TYPE] = wU[TYPE];
            } else if (wl[TYPE] == WALLTYPE_WATER) {
              b[TEMPERATURE] = C4(base_in, x, yu)[TEMPERATURE];
            }
          }
        } else if (wl[VERT_DISTANCE] == 0) {
          const float *waterX0Yp = C4(water_in, x, yu);
          const float *precipDeposition = dep_in + 2 * ((size_t)y * X + x);
          const float *lightAboveSurface = C4(light_in, x, yu_clamp);
          const int t = wl[TYPE];
          if (t == WALLTYPE_INDUSTRIAL) wl[VEGETATION] = wl[VEGETATION] < 15 ? wl[VEGETATION] : 15;
          if (t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN) wl[VEGETATION] = wl[VEGETATION] < 75 ? wl[VEGETATION] : 75;
          if (t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN || t == WALLTYPE_FIRE) {
            if (wl[TYPE] == WALLTYPE_FIRE) {
              const float fireIntensity =
                calcFireIntensity(wl[VEGETATION], w[SOIL_MOISTURE], waterX0Yp[PRECIPITATION]);
              if (fireIntensity < 0.002f) {
                wl[TYPE] = WALLTYPE_LAND;
              } else if (iterI % ((int)(10.0f / fireIntensity) + 1) == 0) {
                wl[VEGETATION] -= 1;
                if (wl[VEGETATION] < 10) wl[TYPE] = WALLTYPE_LAND;
              }
            }
          }
          if (t == WALLTYPE_INDUSTRIAL || t == WALLTYPE_URBAN || t == WALLTYPE_FIRE || t == WALLTYPE_LAND) {
            w[SOIL_MOISTURE] = clampf(w[SOIL_MOISTURE] + precipDeposition[0] * 0.1f, 0.0f, 1000.0f);
            w[SNOW] = clampf(w[SNOW] + precipDeposition[1] * snowMassToHeight, 0.0f, 4000.0f);

            const float *baseAboveSurface = C4(base_in, x, yu);
            const float *waterAboveSurface = waterX0Yp;
            const float realTempAboveSurface = baseAboveSurface[TEMPERATURE] - tcy_up * p->dryLapse;
            const float evaporation = calcEvaporation(p, realTempAboveSurface, waterAboveSurface[TOTAL],
                                                      (float)wl[VEGETATION], w[SOIL_MOISTURE]) *
                                      0.10f;
            w[SOIL_MOISTURE] -= evaporation;

            if (iterI % 100 == 0) {
              const float snowSmoothingRate = 0.02f, moistureSmoothingRate = 0.02f;
              float numNeighbors = 0.0f, totalNeighborSnow = 0.0f, totalNeighborSoilMoisture = 0.0f;
              if (wL[VERT_DISTANCE] == 0 && (wL[TYPE] == WALLTYPE_LAND || wL[TYPE] == WALLTYPE_URBAN)) {
                totalNeighborSnow += C4(water_in, xl, y)[SNOW];
                totalNeighborSoilMoisture += C4(water_in, xl, y)[SOIL_MOISTURE];
                numNeighbors += 1.0f;
              }
              if (wR[VERT_DISTANCE] == 0 && (wR[TYPE] == WALLTYPE_LAND || wR[TYPE] == WALLTYPE_URBAN)) {
                totalNeighborSnow += C4(water_in, xr, y)[SNOW];
                totalNeighborSoilMoisture += C4(water_in, xr, y)[SOIL_MOISTURE];
                numNeighbors += 1.0f;
              }
              if (numNeighbors > 0.0f) {
                const float avgNeighborSnow = totalNeighborSnow / numNeighbors;
                w[SNOW] += (avgNeighborSnow - w[SNOW]) * snowSmoothingRate;
                const float avgNeighborSoilMoisture = totalNeighborSoilMoisture / numNeighbors;
                w[SOIL_MOISTURE] += (avgNeighborSoilMoisture - w[SOIL_MOISTURE]) * moistureSmoothingRate;
              }
              const int vegetationGrowthRate =
                (int)(w[SOIL_MOISTURE] * sqrtf(lightAboveSurface[SUNLIGHT]) * 0.01f);
              if (vegetationGrowthRate > 0) {
                const int interval = (100 / vegetationGrowthRate) * 100;
                if (interval != 0 && iterI % interval == 0) { /* %0 undefined in GLSL -> false */
                  if ((int)map_rangeC(realTempAboveSurface, CtoK(0.0f), CtoK(25.0f), 0.0f, 127.0f) > wl[VEGETATION])
                    wl[VEGETATION] += 1;
                }
              }
              const int subInterval = iterI / 100;
              if (subInterval % ((int)(w[SOIL_MOISTURE] * 0.1f + w[SNOW] * 0.5f) + 10) == 0 &&
                  wl[VEGETATION] >= 20 &&
                  (wL[TYPE] == WALLTYPE_FIRE || wR[TYPE] == WALLTYPE_FIRE || waterX0Yp[SMOKE] > 4.5f)) {
                wl[TYPE] = WALLTYPE_FIRE;
              }
            }
          } else if (t == WALLTYPE_WATER) {
            const float waterTempUpdateInterval = 20.0f;
            if (p->dynamicWaterTemperature >= 1.0f &&
                (iterNum - waterTempUpdateInterval * floorf(iterNum / waterTempUpdateInterval)) < 0.5f) {
              float numNeighbors = 0.0f, totalNeighborTemp = 0.0f;
              if (wL[TYPE] == WALLTYPE_WATER) {
                totalNeighborTemp += C4(base_in, xl, y)[TEMPERATURE];
                numNeighbors += 1.0f;
              }
              if (wR[TYPE] == WALLTYPE_WATER) {
                totalNeighborTemp += C4(base_in, xr, y)[TEMPERATURE];
                numNeighbors += 1.0f;
              }
              if (numNeighbors > 0.0f) {
                const float avgNeighborTemp = totalNeighborTemp / numNeighbors;
                b[TEMPERATURE] += (avgNeighborTemp - b[TEMPERATURE]) * 0.10f;
              }
              if (b[TEMPERATURE] > 500.0f) b[TEMPERATURE] = CtoK(25.0f);
              const float airTemperature = C4(base_in, x, yu)[TEMPERATURE] - tcy_up * p->dryLapse;
              float netWaterHeating = 0.0f;
              netWaterHeating += (airTemperature - b[TEMPERATURE]) * waterHeatExchangeRate;
              netWaterHeating -=
                fmaxf((maxWater(b[TEMPERATURE]) - waterX0Yp[TOTAL]) * p->waterEvaporation, 0.0f) * p->evapHeat * 0.5f;
              float lightPower = fmaxf(lightAboveSurface[SUNLIGHT] * cos_a, 0.0f);
              lightPower *= (1.0f - ALBEDO_WATER);
              lightPower *= lightHeatingConst;
              netWaterHeating += lightPower;
              netWaterHeating += lightAboveSurface[NET_HEATING];
              b[TEMPERATURE] += netWaterHeating / waterHeatCapacity * waterTempUpdateInterval;
            }
            b[TEMPERATURE] = clampf(b[TEMPERATURE], CtoK(0.0f), CtoK(maxWaterTemp));
            wl[VEGETATION] = 20;
            w[SOIL_MOISTURE] = 100.0f;
            w[SNOW] = 0.0f;
          }
        }
      }
      memcpy(C4(base_out, x, y), b, 16);
      memcpy(C4(water_out, x, y), w, 16);
      int8_t *wo = C4(wall_out, x, y);
      for (int c = 0; c < 4; c++) wo[c] = sat8(wl[c]);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * advectionShader.frag:65-227 (+ :403-409 wall markers). Brush (:229-401) and airplane
 * (:415-457) inputs: see wxo_advection's tail.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int ix0, ix1, iy0, iy1; /* local wrapped tap indices */
  float fx, fy;
} taps_t;

/* common.glsl:194-203 / :216-222: st = pos - 0.5; ipos = floor(st); fpos = fract(st).
 * pos is in GLOBAL fragCoord units; taps are returned as local indices (offset from own cell). */
static inline taps_t mktaps(const geo_t *g, int x, int y, float posx, float posy)
{
  taps_t t;
  const float stx = posx - 0.5f, sty = posy - 0.5f;
  const float flx = floorf(stx), fly = floorf(sty);
  t.fx = stx - flx;
  t.fy = sty - fly;
  const int dx = (int)flx - gx_of(g, x); /* tap offset relative to own (global) column */
  t.ix0 = wrapmod(x + dx, g->X);
  t.ix1 = wrapmod(x + dx + 1, g->X);
  t.iy0 = wrapmod((int)fly, g->Y);
  t.iy1 = wrapmod((int)fly + 1, g->Y);
  return t;
}

static inline float bilerp_c(const float *tex, int X, const taps_t *t, int c)
{
  const float a = C4(tex, t->ix0, t->iy0)[c], b = C4(tex, t->ix1, t->iy0)[c];
  const float cc = C4(tex, t->ix0, t->iy1)[c], d = C4(tex, t->ix1, t->iy1)[c];
  return mixf(mixf(a, b, t->fx), mixf(cc, d, t->fx), t->fy);
}

typedef struct {
  float mAB, mCD, mY;
} wallmix_t;

/* common.glsl:229-251 */
static inline wallmix_t mkwallmix(const int8_t *wall, int X, const taps_t *t)
{
  wallmix_t m;
  const int wa = C4(wall, t->ix0, t->iy0)[1], wb = C4(wall, t->ix1, t->iy0)[1];
  const int wc = C4(wall, t->ix0, t->iy1)[1], wd = C4(wall, t->ix1, t->iy1)[1];
  m.mAB = t->fx;
  m.mCD = t->fx;
  m.mY = t->fy;
  if (wa == 0)
    m.mAB = 1.0f;
  else if (wb == 0)
    m.mAB = 0.0f;
  if (wc == 0)
    m.mCD = 1.0f;
  else if (wd == 0)
    m.mCD = 0.0f;
  if (wa == 0 && wb == 0)
    m.mY = 1.0f;
  else if (wc == 0 && wd == 0)
    m.mY = 0.0f;
  return m;
}

static inline float bilerpWall_c(const float *tex, int X, const taps_t *t, const wallmix_t *m, int c)
{
  const float a = C4(tex, t->ix0, t->iy0)[c], b = C4(tex, t->ix1, t->iy0)[c];
  const float cc = C4(tex, t->ix0, t->iy1)[c], d = C4(tex, t->ix1, t->iy1)[c];
  return mixf(mixf(a, b, m->mAB), mixf(cc, d, m->mCD), m->mY);
}

/* common.glsl:268-271 */
static inline float absHorizontalDist(float a, float b)
{
  return fminf(fminf(fabsf(a - b), fabsf(1.0f + a - b)), 1.0f - a + b);
}
static inline float smoothstepf(float e0, float e1, float x)
{
  const float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return t * t * (3.0f - 2.0f * t);
}

void wxo_advection(const wxo_params *p, const float *initial_T, const float *snd_T,
                   const float *snd_W, const float *snd_Vel, const float *base_in,
                   const float *water_in, const int8_t *wall_in, float *base_out, float *water_out,
                   int8_t *wall_out)
{
  const geo_t g = mkgeo(p);
  const int X = p->X, Y = p->Y;
  /* advectionShader.frag:69: texelSize = vec2(1.) / resolution (in-shader fp32 division) */
  const float a_texX = 1.0f / (float)g.Xg, a_texY = 1.0f / (float)Y;

#pragma omp parallel for schedule(static)
  for (int y = 0; y < Y; y++) {
    const int yu = (y + 1 == Y) ? 0 : y + 1, yd = (y == 0) ? Y - 1 : y - 1;
    for (int x = 0; x < X; x++) {
      const int xr = (x + 1 == X) ? 0 : x + 1, xl = (x == 0) ? X - 1 : x - 1;
      const cc_t cc = cellcoord(&g, x, y);
      const float fx = cc.fx, fy = cc.fy, tcx = cc.tcx, tcy = cc.tcy;
      int wl[4];
      const int8_t *w0 = C4(wall_in, x, y);
      for (int c = 0; c < 4; c++) wl[c] = w0[c];
      float b[4], w[4];
      float realTemp = 0.0f;

      if (wl[DISTANCE] != 0) { /* not wall */
        const float *c00 = C4(base_in, x, y), *cL = C4(base_in, xl, y), *cD = C4(base_in, x, yd);
        const float *cR = C4(base_in, xr, y), *cU = C4(base_in, x, yu);
        const float *cLU = C4(base_in, xl, yu), *cRD = C4(base_in, xr, yd);

        const float velAtP_x = (cL[0] + c00[0]) / 2.0f, velAtP_y = (cD[1] + c00[1]) / 2.0f;
        const float velAtVx_x = c00[0], velAtVx_y = (cD[1] + cR[1] + c00[1] + cRD[1]) / 4.0f;
        const float velAtVy_x = (cL[0] + cU[0] + cLU[0] + c00[0]) / 4.0f, velAtVy_y = c00[1];

        taps_t t;
        t = mktaps(&g, x, y, fx - velAtVx_x, fy - velAtVx_y);
        b[VX] = bilerp_c(base_in, X, &t, VX);
        t = mktaps(&g, x, y, fx - velAtVy_x, fy - velAtVy_y);
        b[VY] = bilerp_c(base_in, X, &t, VY);

        t = mktaps(&g, x, y, fx - velAtP_x, fy - velAtP_y);
        wallmix_t m = mkwallmix(wall_in, X, &t);
        b[PRESSURE] = bilerpWall_c(base_in, X, &t, &m, PRESSURE);
        b[TEMPERATURE] = bilerpWall_c(base_in, X, &t, &m, TEMPERATURE);
        w[0] = bilerpWall_c(water_in, X, &t, &m, 0);
        w[1] = bilerpWall_c(water_in, X, &t, &m, 1);
        w[3] = bilerpWall_c(water_in, X, &t, &m, 3);

        t = mktaps(&g, x, y, fx - velAtP_x + 0.0f, fy - velAtP_y + 0.05f);
        m = mkwallmix(wall_in, X, &t);
        w[PRECIPITATION] = bilerpWall_c(water_in, X, &t, &m, PRECIPITATION);

        realTemp = b[TEMPERATURE] - tcy * p->dryLapse;

        const float excessWater = w[TOTAL] - maxWater(realTemp);
        const float overSaturation = excessWater - w[CLOUD];
        float condensation;
        if (overSaturation < 0.0f)
          condensation = overSaturation * 0.20f;
        else
          condensation = overSaturation * p->condensationRate;
        condensation = fmaxf(condensation, -w[CLOUD]);
        const float dT = condensation * p->evapHeat * 1.0f;
        b[TEMPERATURE] += dT;
        realTemp += dT;
        w[CLOUD] += condensation;

        if (tcy > p->globalEffectsStartAlt && tcy < p->globalEffectsEndAlt) { /* :154-181 */
          w[TOTAL] -= clampf(p->globalDrying, 0.0f,
                             fmaxf(w[TOTAL] - maxWater(fmaxf(realTemp - 20.0f, CtoK(-80.0f))), 0.0f));
          b[TEMPERATURE] += p->globalHeating;

          const int si = (int)(tcy * (1.0f / a_texY));
          const int si1 = si - 1 < 0 ? 0 : si - 1;
          const float sT = snd_T ? (snd_T[si] + snd_T[si1]) / 2.0f : 0.0f;
          const float sW = snd_W ? (snd_W[si] + snd_W[si1]) / 2.0f : 0.0f;
          const float sV = snd_Vel ? (snd_Vel[si] + snd_Vel[si1]) / 2.0f : 0.0f;

          const float Tdiff = b[TEMPERATURE] - sT;
          b[TEMPERATURE] -= Tdiff * 0.001f * p->soundingForcing;
          const float Wdiff = w[TOTAL] - sW;
          w[TOTAL] -= Wdiff * 0.001f * p->soundingForcing;
          const float dragk = 1.0f - map_rangeC(p->soundingForcing, 0.1f, 1.0f, 0.0f, 0.001f);
          b[VX] *= dragk;
          b[VY] *= dragk;
          const float velDiff = b[VX] - sV;
          b[VX] -= velDiff * map_rangeC(p->soundingForcing, 0.9f, 1.0f, 0.0f, 0.001f);
        }
        w[TOTAL] = fmaxf(w[TOTAL], 0.0f);
      } else { /* wall :189-227 */
        memcpy(b, C4(base_in, x, y), 16);
        memcpy(w, C4(water_in, x, y), 16);
        if (wl[TYPE] == WALLTYPE_LAND) b[TEMPERATURE] = 1000.0f;
        const int8_t *wU = C4(wall_in, x, yu);
        wl[VEGETATION] = wl[VEGETATION] > 0 ? wl[VEGETATION] : 0;
        w[SOIL_MOISTURE] = fmaxf(w[SOIL_MOISTURE], 0.0f);
        if (wU[DISTANCE] != 0) { /* surface layer */
          const float *baseX0Yp = C4(base_in, x, yu);
          const float tempC = KtoC(baseX0Yp[TEMPERATURE] - tcy * p->dryLapse);
          if (w[SNOW] > 0.0f && tempC > 0.0f) {
            const float melting = fminf(tempC * snowMeltRate, w[SNOW]);
            w[SNOW] -= melting;
            b[TEMPERATURE] += melting / snowMassToHeight * p->meltingHeat;
            w[SOIL_MOISTURE] += melting;
          }
          if (w[SOIL_MOISTURE] > 0.0f && tempC > 0.0f) {
            const float evaporation = fmaxf((maxWater(CtoK(tempC)) - w[TOTAL]) * 0.00001f, 0.0f);
            w[SOIL_MOISTURE] -= evaporation;
          }
        }
      }

      /* USER INPUT :229-401 */
      int inBrush = 0;
      float weight = 1.0f;
      const float brushR = p->userInputValues[3] * a_texY;
      if (p->userInputType >= 0) { /* nothing below has an effect for type < 1 */
        if (p->userInputValues[0] < -0.5f) {
          if (fabsf(p->userInputValues[1] - tcy) < brushR) inBrush = 1;
        } else {
          float vmx, vmy = p->userInputValues[1] - tcy;
          if (p->wrapHorizontally)
            vmx = absHorizontalDist(p->userInputValues[0], tcx);
          else
            vmx = fabsf(p->userInputValues[0] - tcx);
          vmx *= a_texY / a_texX;
          const float distFromMouse = sqrtf(vmx * vmx + vmy * vmy);
          weight = smoothstepf(brushR, 0.0f, distFromMouse);
          if (distFromMouse < brushR) inBrush = 1;
        }
      }
      if (inBrush) {
        const int ut = p->userInputType;
        const float inten = p->userInputValues[2];
        const int aboveIsAir = C4(wall_in, x, yu)[DISTANCE] != 0;
        if (ut == 1) {
          b[3] += inten;
          if (wl[TYPE] == 2 && wl[DISTANCE] == 0) b[3] = clampf(b[3], CtoK(0.0f), CtoK(maxWaterTemp));
        } else if (ut == 2) {
          if (w[CLOUD] > 0.0f) {
            w[CLOUD] += inten;
            w[CLOUD] = fmaxf(w[CLOUD], 0.0f);
          }
          w[TOTAL] += inten;
          w[TOTAL] = fmaxf(w[TOTAL], 0.0f);
        } else if (ut == 3 && wl[DISTANCE] != 0) {
          w[SMOKE] += inten;
          w[SMOKE] = fminf(fmaxf(w[SMOKE], 0.0f), 2.0f);
        } else if (ut == 4) {
          if (p->userInputValues[0] < -0.5f) {
            b[VX] += p->userInputMove[0] * 5.0f * weight * inten;
          } else {
            b[VX] += p->userInputMove[0] * 5.0f * weight * inten;
            b[VY] += p->userInputMove[1] * 5.0f * weight * inten;
          }
        } else if (ut >= 10) {
          if (inten > 0.0f) {
            int setWall = 0;
            switch (ut) {
            case 10: wl[TYPE] = WALLTYPE_INERT; setWall = 1; break;
            case 11: wl[TYPE] = WALLTYPE_LAND; setWall = 1; break;
            case 12: wl[TYPE] = WALLTYPE_WATER; setWall = 1; break;
            case 13:
              if (wl[DISTANCE] == 0 && wl[TYPE] == WALLTYPE_LAND && aboveIsAir) {
                wl[TYPE] = WALLTYPE_FIRE;
                setWall = 1;
              }
              break;
            case 14:
              if (wl[DISTANCE] == 0 &&
                  (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_RUNWAY || wl[TYPE] == WALLTYPE_INDUSTRIAL) && aboveIsAir)
                wl[TYPE] = WALLTYPE_URBAN;
              break;
            case 15:
              if (wl[DISTANCE] == 0 &&
                  (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_URBAN || wl[TYPE] == WALLTYPE_INDUSTRIAL) && aboveIsAir)
                wl[TYPE] = WALLTYPE_RUNWAY;
              break;
            case 16:
              if (wl[DISTANCE] == 0 &&
                  (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_URBAN || wl[TYPE] == WALLTYPE_RUNWAY) && aboveIsAir)
                wl[TYPE] = WALLTYPE_INDUSTRIAL;
              break;
            case 20:
              if (wl[DISTANCE] == 0 && wl[TYPE] != WALLTYPE_WATER && aboveIsAir) w[SOIL_MOISTURE] += inten * 10.0f;
              break;
            case 21:
              if (wl[DISTANCE] == 0 &&
                  (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_URBAN || wl[TYPE] == WALLTYPE_INDUSTRIAL) && aboveIsAir)
                w[SNOW] += inten * 0.5f;
              break;
            case 22:
              if (wl[DISTANCE] == 0 &&
                  (wl[TYPE] == WALLTYPE_LAND || wl[TYPE] == WALLTYPE_FIRE || wl[TYPE] == WALLTYPE_URBAN ||
                   wl[TYPE] == WALLTYPE_INDUSTRIAL) && aboveIsAir)
                wl[VEGETATION] += 1;
              break;
            default: break;
            }
            if (setWall) {
              wl[DISTANCE] = 0;
              b[TEMPERATURE] = 1000.0f;
              if (wl[TYPE] == WALLTYPE_LAND)
                w[SOIL_MOISTURE] = 25.0f;
              else if (wl[TYPE] == WALLTYPE_WATER)
                b[TEMPERATURE] = p->waterTemperature;
            }
          } else {
            if (wl[DISTANCE] == 0) {
              if (ut == 13) {
                if (wl[TYPE] == WALLTYPE_FIRE) wl[TYPE] = WALLTYPE_LAND;
              } else if (ut == 14) {
                if (wl[TYPE] == WALLTYPE_URBAN) wl[TYPE] = WALLTYPE_LAND;
              } else if (ut == 15) {
                if (wl[TYPE] == WALLTYPE_RUNWAY) wl[TYPE] = WALLTYPE_LAND;
              } else if (ut == 16) {
                if (wl[TYPE] == WALLTYPE_INDUSTRIAL) wl[TYPE] = WALLTYPE_LAND;
              } else if (ut == 20) {
                w[SOIL_MOISTURE] += inten * 10.0f;
              } else if (ut == 21) {
                w[SNOW] += inten * 0.5f;
              } else if (ut == 22) {
                wl[VEGETATION] = wl[VEGETATION] - 1 > 0 ? wl[VEGETATION] - 1 : 0;
              } else if (tcy > a_texY) {
                wl[DISTANCE] = 255;
                b[VX] = 0.0f;
                b[VY] = 0.0f;
                b[PRESSURE] = 0.0f;
                b[TEMPERATURE] = initial_T[(int)(tcy * (1.0f / a_texY))];
                w[TOTAL] = 0.0f;
                w[CLOUD] = 0.0f;
                w[PRECIPITATION] = 0.0f;
                w[SMOKE] = 0.0f;
              }
            }
          }
        }
      }

      if (wl[DISTANCE] == 0) { /* :403-409 */
        w[TOTAL] = (wl[TYPE] == WALLTYPE_WATER) ? 1002.0f : 1001.0f;
      }

      /* airplane :415-457 */
      {
        float vpx, vpy = p->airplaneValues[1] - tcy;
        if (p->wrapHorizontally)
          vpx = absHorizontalDist(p->airplaneValues[0], tcx);
        else
          vpx = fabsf(p->airplaneValues[0] - tcx);
        vpx *= a_texY / a_texX;
        vpx *= (float)Y;
        vpy *= (float)Y;
        if (p->airplaneValues[3] < 0.0f) vpy += -1.0f;
        const float distFromPlane = sqrtf(vpx * vpx + vpy * vpy);
        const float planeInfluence = fmaxf(1.0f - distFromPlane, 0.0f) * 0.03f;
        if (p->airplaneValues[3] < 0.0f) w[PRECIPITATION] += planeInfluence * 100.0f;
        if (p->airplaneValues[3] > 0.9f) {
          if (distFromPlane < 1.5f) {
            if (wl[DISTANCE] == 0) {
              if (wl[TYPE] == WALLTYPE_LAND && wl[VERT_DISTANCE] == 0) wl[TYPE] = WALLTYPE_FIRE;
            } else {
              b[PRESSURE] += 0.05f;
              b[TEMPERATURE] = CtoK(50.0f);
              w[TOTAL] += 1.0f;
              w[SMOKE] += 10.0f;
            }
          }
        }
      }

      memcpy(C4(base_out, x, y), b, 16);
      memcpy(C4(water_out, x, y), w, 16);
      int8_t *wo = C4(wall_out, x, y);
      for (int c = 0; c < 4; c++) wo[c] = sat8(wl[c]);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * lightingShader.frag:38-170. emit_out (optional, X*Y*4 floats): the second render target
 * `reflectedLight` (:15), bound to the RGBA16F emittedLight texture (app.js:5283, 5294), as the
 * unrounded fp32 values the shader computes; the attachment stores them as binary16. Its alpha is
 * never written and the accumulating `+=` starts from the zero-initialised output variable.
 * ---------------------------------------------------------------------------------------- */
/* common.glsl:367-372 */
static inline void hsv2rgb(float h, float sv, float v, float rgb[3])
{
  const float K[3] = {1.0f, 2.0f / 3.0f, 1.0f / 3.0f};
  for (int i = 0; i < 3; i++) {
    const float t = h + K[i];
    const float pch = fabsf((t - floorf(t)) * 6.0f - 3.0f);
    rgb[i] = v * mixf(1.0f, clampf(pch - 1.0f, 0.0f, 1.0f), sv);
  }
}

/* GL LINEAR filtering (GLES 3.0 spec 3.8.10), S=REPEAT, T=CLAMP_TO_EDGE, channel SUNLIGHT, of the
 * sample at (texel centre of (x,y)) + (ox, oy) texels. The filter-weight precision is implementation
 * defined in GL; here the weights are the exact fp32 fractions of the offsets, which also makes the
 * result independent of how the domain is cut into slabs. */
static inline float light_linear_sun(const float *light, int X, int Y, int x, int y, float ox, float oy)
{
  const float fu = floorf(ox), fv = floorf(oy);
  const float a = ox - fu, bb = oy - fv;
  const int i0 = wrapmod(x + (int)fu, X), i1 = wrapmod(x + (int)fu + 1, X);
  int j0 = y + (int)fv, j1 = y + (int)fv + 1;
  j0 = j0 < 0 ? 0 : (j0 > Y - 1 ? Y - 1 : j0);
  j1 = j1 < 0 ? 0 : (j1 > Y - 1 ? Y - 1 : j1);
  const float t00 = C4(light, i0, j0)[0], t10 = C4(light, i1, j0)[0];
  const float t01 = C4(light, i0, j1)[0], t11 = C4(light, i1, j1)[0];
  return (1.0f - a) * (1.0f - bb) * t00 + a * (1.0f - bb) * t10 + (1.0f - a) * bb * t01 + a * bb * t11;
}

void wxo_lighting_mrt(const wxo_params *p, const float *base_in, const float *water_in,
                      const int8_t *wall_in, const float *light_in, float *light_out, float *emit_out)
{
  const geo_t g = mkgeo(p);
  const int X = p->X, Y = p->Y;
  const float resY = (float)Y;
  const float cellHeightCompensation = 300.0f / resY;
  const float sin_a = sinf(p->sunAngle), cos_a = cosf(p->sunAngle);
  /* :58-59 how red the sunlight is; sunColor() common.glsl:374-378 */
  const float deg2rad = 0.0174533f, standardSunBrightness = 1250.0f; /* common.glsl:6, 11 */
  const float scattering = clampf(map_range(fabsf(p->sunAngle), 75.0f * deg2rad, 90.0f * deg2rad, 0.0f, 1.0f), 0.0f, 1.0f);
  float sunCol[3];
  {
    const float val = 1.0f - scattering;
    hsv2rgb(0.015f + val * 0.15f, fminf(2.0f - val * 2.0f, 1.0f), 1.0f, sunCol);
  }
  const int night_glow = fabsf(p->sunAngle) > 85.0f * deg2rad;

#pragma omp parallel for schedule(static)
  for (int y = 0; y < Y; y++) {
    const int yu = (y + 1 >= Y) ? Y - 1 : y + 1; /* light: clamp */
    const int yd = (y == 0) ? 0 : y - 1;         /* light: clamp */
    const int yd_wrap = (y == 0) ? Y - 1 : y - 1; /* base: repeat */
    for (int x = 0; x < X; x++) {
      float *lo = C4(light_out, x, y);
      const cc_t cc = cellcoord(&g, x, y);
      const float fy = cc.fy, tcy = cc.tcy;
      float em[3] = {0.0f, 0.0f, 0.0f};
      float *eo = emit_out ? C4(emit_out, x, y) : NULL;
      if (eo) eo[0] = eo[1] = eo[2] = eo[3] = 0.0f;
      if (fy >= resY - 1.0f) {
        lo[0] = p->sunIntensity;
        lo[1] = 0.0f;
        lo[2] = 0.0f;
        lo[3] = 0.0f;
        continue;
      }
      /* :48-49 sample at texCoord + (sin a, cos a) texels; the quad-scale offset of fragCoord from
       * the texel centre is carried along */
      const float ox = (cc.fx - ((float)gx_of(&g, x) + 0.5f)) + sin_a;
      const float oy = (fy - ((float)y + 0.5f)) + cos_a;
      float sunlight = light_linear_sun(light_in, X, Y, x, y, ox, oy);

      const float realTemp = C4(base_in, x, y)[TEMPERATURE] - tcy * p->dryLapse;
      const float *water = C4(water_in, x, y);
      const int8_t *wall = C4(wall_in, x, y);

      if (wall[DISTANCE] != 0) {
        for (int c = 0; c < 3; c++) em[c] = sunCol[c] * sunlight * (1.0f - tcy) * 2.0f / standardSunBrightness; /* :60 scattering in air */
        float net_heating = 0.0f;
        if (fy < resY - 2.0f) {
          float reflection =
            fminf(sqrtf(water[CLOUD] * 0.0010f + water[PRECIPITATION] * 0.00020f) * cellHeightCompensation, 1.0f); /* pow(x, 0.5) */
          reflection += 0.0002f;
          const float absorbtion = fminf(water[SMOKE] * 0.020f * cellHeightCompensation, 1.0f);
          const float lightReflected = sunlight * reflection;
          const float lightAbsorbed = sunlight * absorbtion;
          sunlight = fmaxf(0.0f, sunlight - lightReflected - lightAbsorbed);
          net_heating += lightAbsorbed * lightHeatingConst;
          for (int c = 0; c < 3; c++) em[c] = sunCol[c] * lightReflected / standardSunBrightness; /* :78 `=`: replaces the air term */
        }
        float IR_down = C4(light_in, x, yu)[IR_DOWN];
        float IR_up = 0.0f;
        if (wall[VERT_DISTANCE] == 1) {
          if (night_glow && (wall[TYPE] == WALLTYPE_RUNWAY || wall[TYPE] == WALLTYPE_URBAN || wall[TYPE] == WALLTYPE_INDUSTRIAL)) { /* :98-101 */
            em[0] += 1.00f * 0.03f;
            em[1] += 0.97f * 0.03f;
            em[2] += 0.57f * 0.03f;
          }
          switch (wall[TYPE]) {
          case WALLTYPE_RUNWAY:
          case WALLTYPE_URBAN:
          case WALLTYPE_INDUSTRIAL:
          case WALLTYPE_LAND:
            IR_up = IR_emitted(realTemp);
            net_heating += (IR_down - IR_up) * lightHeatingConst;
            break;
          case WALLTYPE_WATER: {
            const float waterTemperature = C4(base_in, x, yd_wrap)[TEMPERATURE];
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
          IR_up = C4(light_in, x, yd)[IR_UP];
          float emissivity = p->greenhouseGases;
          emissivity += water[TOTAL] * p->waterGreenHouseEffect;
          emissivity += water[CLOUD] * 5.0f;
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
        if (eo) { /* :143-148 thick smoke glows (fire) */
          const float smokeOpacity = clampf(1.0f - (1.0f / (water[SMOKE] + 1.0f)), 0.0f, 1.0f);
          const float fireIntensity = clampf((smokeOpacity - 0.8f) * 25.0f, 0.0f, 1.0f);
          float fireCol[3];
          hsv2rgb(fireIntensity * 0.008f, 0.98f, 5.0f, fireCol);
          for (int c = 0; c < 3; c++) eo[c] = em[c] + mixf(0.0f, fireCol[c], fireIntensity) * 0.1f;
        }
        net_heating *= p->IR_rate;
        lo[0] = sunlight;
        lo[1] = net_heating;
        lo[2] = IR_down;
        lo[3] = IR_up;
      } else {
        if (eo && wall[TYPE] != WALLTYPE_WATER) { /* :158-166 land reflects part of the light */
          eo[0] = sunlight * 0.60f / standardSunBrightness;
          eo[1] = sunlight * 0.5f / standardSunBrightness;
          eo[2] = sunlight * 0.4f / standardSunBrightness;
        }
        lo[0] = (wall[TYPE] == WALLTYPE_WATER) ? sunlight * 0.90f : 0.0f;
        lo[1] = 0.0f;
        lo[2] = 0.0f;
        lo[3] = 0.0f;
      }
    }
  }
}

void wxo_lighting(const wxo_params *p, const float *base_in, const float *water_in,
                  const int8_t *wall_in, const float *light_in, float *light_out)
{
  wxo_lighting_mrt(p, base_in, water_in, wall_in, light_in, light_out, NULL);
}

/* ------------------------------------------------------------------------------------------
 * precipitationShader.vert:66-293 + point rasterisation with ONE,ONE blending (app.js:5940-5953)
 * Whole-domain only (X == X_global, x_off == 0).
 * ---------------------------------------------------------------------------------------- */
static inline const float *texfetch4(const float *tex, int X, int Y, float u, float v)
{
  const int ix = wrapmod((int)floorf(u * (float)X), X);
  const int iy = wrapmod((int)floorf(v * (float)Y), Y);
  return C4(tex, ix, iy);
}

/* accumulators of splat_order 1: one deposit per sprite at its anchor (i0 + 6, j0 + 6) of an (X+1) x (Y+1) grid, the 1-px
 * sprites (inactive count into texel (0,0), lightning request into texel (1,0)) in their own slots */
typedef struct {
  float *acc; /* 5 floats per anchor: mass, heat, vapor | rain, snow */
  float count, light[4];
} anchors_t;

static void splat(const wxo_params *p, float gposx, float gposy, float size, const float *feedback,
                  const float *deposition, float *fb, float *dep, anchors_t *an)
{
  const int X = p->X, Y = p->Y;
  if (!(gposx >= -1.0f && gposx <= 1.0f && gposy >= -1.0f && gposy <= 1.0f)) return; /* clipped */
  float xw = (gposx + 1.0f) * 0.5f * (float)X, yw = (gposy + 1.0f) * 0.5f * (float)Y;
  if (p->subpixel_bits > 0) { /* rasteriser snaps window coordinates to a 1/2^bits sub-pixel grid */
    const float q = (float)(1 << p->subpixel_bits);
    xw = floorf(xw * q + 0.5f) / q;
    yw = floorf(yw * q + 0.5f) / q;
  }
  int i0, i1, j0, j1;
  if (size <= 1.0f) {
    i0 = i1 = (int)floorf(xw);
    j0 = j1 = (int)floorf(yw);
  } else {
    /* pixel (i,j) is covered when its centre lies in [w - size/2, w + size/2) */
    const float h = size * 0.5f;
    i0 = (int)ceilf(xw - h - 0.5f);
    i1 = i0 + (int)size - 1;
    j0 = (int)ceilf(yw - h - 0.5f);
    j1 = j0 + (int)size - 1;
  }
  if (an) {
    if (size <= 1.0f) { /* the only 1-px sprites: texel (0,0) counts inactive droplets, texel (1,0) takes lightning requests */
      if (j0 != 0 || i0 < 0 || i0 > 1) return;
      if (i0 == 0)
        an->count += feedback[0];
      else
        for (int c = 0; c < 4; c++) an->light[c] += feedback[c];
      return;
    }
    const int q = i0 + 6, r = j0 + 6;
    if (q < 0 || q > X || r < 0 || r > Y) return;
    float *a = an->acc + 5 * ((size_t)r * (X + 1) + q);
    a[0] += feedback[0];
    a[1] += feedback[1];
    a[2] += feedback[2];
    a[3] += deposition[0];
    a[4] += deposition[1];
    return;
  }
  for (int j = j0; j <= j1; j++) {
    if (j < 0 || j >= Y) continue;
    for (int i = i0; i <= i1; i++) {
      if (i < 0 || i >= X) continue;
      float *f = C4(fb, i, j);
      f[0] += feedback[0];
      f[1] += feedback[1];
      f[2] += feedback[2];
      f[3] += feedback[3];
      float *d = dep + 2 * ((size_t)j * X + i);
      d[0] += deposition[0];
      d[1] += deposition[1];
    }
  }
}

/* 12 consecutive values with the index-anchored tree of the HIP engine (csrc/wx_kernels.h splat_box_tile): pairs
 * p[i] = a[i] + a[i+1], quads q[i] = p[i] + p[i+2], s = (q[0] + q[4]) + q[8]; values outside the grid are 0 */
static inline float tree12(const float *a) /* a[0..11] */
{
  const float q0 = (a[0] + a[1]) + (a[2] + a[3]);
  const float q4 = (a[4] + a[5]) + (a[6] + a[7]);
  const float q8 = (a[8] + a[9]) + (a[10] + a[11]);
  return (q0 + q4) + q8;
}

/* splat_order 1: out(i, j) = sum of the anchors q in [i-5, i+6], r in [j-5, j+6]; columns (vertical sums) first, then rows.
 * The textures were cleared before (app.js:5933-5934), so they are written, then the 1-px sprites are added. */
static void box_sum_anchors(const wxo_params *p, const anchors_t *an, float *fb, float *dep)
{
  const int X = p->X, Y = p->Y, AP = X + 1;
  float *V = (float *)calloc((size_t)AP * Y * 5, 4); /* vertical 12-sums: V[j][q][c] */
#pragma omp parallel for schedule(static)
  for (int j = 0; j < Y; j++)
    for (int q = 0; q <= X; q++)
      for (int c = 0; c < 5; c++) {
        float a[12];
        for (int k = 0; k < 12; k++) {
          const int r = j - 5 + k;
          a[k] = (r >= 0 && r <= Y) ? an->acc[5 * ((size_t)r * AP + q) + c] : 0.0f;
        }
        V[5 * ((size_t)j * AP + q) + c] = tree12(a);
      }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < Y; j++)
    for (int i = 0; i < X; i++) {
      float o[5];
      for (int c = 0; c < 5; c++) {
        float a[12];
        for (int k = 0; k < 12; k++) {
          const int q = i - 5 + k;
          a[k] = (q >= 0 && q <= X) ? V[5 * ((size_t)j * AP + q) + c] : 0.0f;
        }
        o[c] = tree12(a);
      }
      float *f = C4(fb, i, j);
      f[0] = o[0];
      f[1] = o[1];
      f[2] = o[2];
      f[3] = 0.0f;
      dep[2 * ((size_t)j * X + i)] = o[3];
      dep[2 * ((size_t)j * X + i) + 1] = o[4];
    }
  free(V);
  C4(fb, 0, 0)[0] += an->count;
  for (int c = 0; c < 4; c++) C4(fb, 1, 0)[c] += an->light[c];
}

void wxo_precipitation(const wxo_params *p, float iterNum, int n_drops, const float *drops_in,
                       const float *base_in, const float *water_in, const float *lightning_in,
                       float *drops_out, float *fb, float *dep)
{
  const geo_t g = mkgeo(p);
  const int X = p->X, Y = p->Y;
  const float resX = (float)X, resY = (float)Y;
  const float initalMass = 0.15f;
  anchors_t an_store = {NULL, 0.0f, {0.0f, 0.0f, 0.0f, 0.0f}}, *an = NULL;
  if (p->splat_order == 1) {
    an_store.acc = (float *)calloc((size_t)(X + 1) * (Y + 1) * 5, 4);
    an = &an_store;
  }

  for (int i = 0; i < n_drops; i++) {
    const float dropPosition[2] = {drops_in[5 * i], drops_in[5 * i + 1]};
    const float mass[2] = {drops_in[5 * i + 2], drops_in[5 * i + 3]};
    const float density = drops_in[5 * i + 4];
    float newPos[2] = {dropPosition[0], dropPosition[1]};
    float newMass[2] = {mass[0], mass[1]};
    float newDensity = density;
    float feedback[4] = {0, 0, 0, 0}, deposition[2] = {0, 0};
    int isActive = 1, spawned = 0, lightningSpawned = 0;
    float size = 1.0f, gpos[2] = {0, 0};
    float tc[2] = {0, 0};
    const float *base = NULL, *water = NULL;
    float realTemp = 0.0f;

    if (mass[0] < 0.0f) { /* inactive :72-162 */
      tc[0] = wxo_random2d(mass[0], dropPosition[0] + iterNum * 0.3754f);
      tc[1] = wxo_random2d(mass[1], dropPosition[0] + iterNum * 0.073162f);
      base = texfetch4(base_in, X, Y, tc[0], tc[1]);
      water = texfetch4(water_in, X, Y, tc[0], tc[1]);
      realTemp = base[TEMPERATURE] - tc[1] * p->dryLapse;
      float threshold;
      if (realTemp > CtoK(0.0f))
        threshold = p->aboveZeroThreshold;
      else
        threshold = p->subZeroThreshold;

      if (water[CLOUD] > threshold && base[TEMPERATURE] < 500.0f) {
        const float spawnChance =
          ((water[CLOUD] - threshold) / (p->inactiveDroplets + 10.0f)) * resX * resY * p->spawnChanceMult;
        const float c10 = water[CLOUD] * 10.0f;
        const float pw = c10 * c10; /* pow(x, 2.0) */
        const float nrmRand = pw - floorf(pw);
        if (spawnChance > nrmRand) {
          spawned = 1;
          newPos[0] = (tc[0] - 0.5f) * 2.0f;
          newPos[1] = (tc[1] - 0.5f) * 2.0f;
          if (realTemp < CtoK(0.0f)) {
            newMass[0] = 0.0f;
            newMass[1] = initalMass;
            feedback[HEAT] += newMass[1] * p->meltingHeat;
            newDensity = p->snowDensity;
            const float cloudPlusPrecipDensity = water[CLOUD] + water[PRECIPITATION];
            const float lightningSpawnChance = fmaxf((cloudPlusPrecipDensity - 2.5f) * 0.0033f, 0.0f);
            if (lightning_in[2] < iterNum - 30.0f &&
                wxo_random2d(base[TEMPERATURE] * 0.2324f, water[TOTAL] * 7.7f) < lightningSpawnChance) {
              lightningSpawned = 1;
              isActive = 0;
              size = 1.0f;
              feedback[0] = tc[0];
              feedback[1] = tc[1];
              feedback[2] = iterNum;
              feedback[3] = clampf(cloudPlusPrecipDensity / 10.0f + (wxo_random2d(tc[0], tc[1]) - 0.5f), 0.01f, 4.0f);
              gpos[0] = -1.0f + g.texX * 3.0f;
              gpos[1] = -1.0f + g.texY;
            }
          } else {
            newMass[0] = initalMass;
            newMass[1] = 0.0f;
            newDensity = 1.0f;
          }
          feedback[VAPOR] -= initalMass;
        }
      }
      if (spawned) {
        if (!lightningSpawned) {
          size = 1.0f;
          gpos[0] = newPos[0];
          gpos[1] = newPos[1];
        }
      } else {
        isActive = 0;
        size = 1.0f;
        feedback[MASS] = 1.0f;
        gpos[0] = -1.0f + g.texX;
        gpos[1] = -1.0f + g.texY;
      }
    }

    if (isActive) { /* :164-288 */
      if (!spawned) {
        tc[0] = dropPosition[0] / 2.0f + 0.5f;
        tc[1] = dropPosition[1] / 2.0f + 0.5f;
        water = texfetch4(water_in, X, Y, tc[0], tc[1]);
        base = texfetch4(base_in, X, Y, tc[0], tc[1]);
        realTemp = base[TEMPERATURE] - tc[1] * p->dryLapse;
      }
      const float totalMass = newMass[0] + newMass[1];
      if (totalMass < 0.04f) {
        feedback[HEAT] = -(totalMass * p->evapHeat);
        feedback[VAPOR] = totalMass;
        newMass[0] = -2.0f - dropPosition[0];
        newMass[1] = dropPosition[1];
      } else if (newPos[1] < -1.0f || water[TOTAL] > 1000.0f) {
        if (texfetch4(base_in, X, Y, tc[0], tc[1] + g.texY)[TEMPERATURE] > 500.0f) newPos[1] += g.texY * 1.0f;
        deposition[0] = newMass[0];
        deposition[1] = newMass[1];
        newMass[0] = -2.0f - dropPosition[0];
        newMass[1] = dropPosition[1];
      } else {
        const float surfaceArea = det_cbrt(totalMass);
        const float growthRate =
          fmaxf(map_range(realTemp, CtoK(0.0f), CtoK(-30.0f), p->growthRate0C, p->growthRate_30C), p->growthRate0C);
        float growth = water[CLOUD] * growthRate * surfaceArea;
        if (realTemp < CtoK(0.0f) && water[CLOUD] > 0.0f && density == 1.0f)
          growth += surfaceArea * water[PRECIPITATION] * 0.0030f;
        feedback[VAPOR] -= growth * 1.0f;
        if (realTemp < CtoK(0.0f)) {
          newMass[1] += growth;
          feedback[HEAT] += growth * p->meltingHeat;
          const float freezing = fminf((CtoK(0.0f) - realTemp) * p->freezingRate * surfaceArea, newMass[0]);
          newMass[0] -= freezing;
          newMass[1] += freezing;
          feedback[HEAT] += freezing * p->meltingHeat;
        } else {
          newMass[0] += growth;
          const float melting = fminf((realTemp - CtoK(0.0f)) * p->meltingRate * surfaceArea, newMass[1]);
          newMass[1] -= melting;
          newMass[0] += melting;
          feedback[HEAT] -= melting * p->meltingHeat;
          newDensity = fminf(newDensity + (melting / totalMass) * 1.00f, 1.0f);
        }
        float dropletTemp = base[TEMPERATURE] - tc[1] * p->dryLapse;
        if (newMass[1] > 0.0f) dropletTemp = fminf(dropletTemp, CtoK(0.0f));
        const float evapAndSubli = fmaxf((maxWater(dropletTemp) - water[TOTAL]) * surfaceArea * p->evapRate, 0.0f);
        const float evap = fminf(newMass[0], evapAndSubli);
        const float subli = fminf(newMass[1], evapAndSubli - evap);
        newMass[0] -= evap;
        newMass[1] -= subli;
        feedback[VAPOR] += evap;
        feedback[VAPOR] += subli;
        feedback[HEAT] -= evap * p->evapHeat;
        feedback[HEAT] -= subli * p->evapHeat;
        feedback[HEAT] -= subli * p->meltingHeat;

        newPos[0] += base[VX] / resX * 2.0f;
        newPos[1] += base[VY] / resY * 2.0f;
        newPos[1] -= p->fallSpeed * newDensity * sqrtf(totalMass / surfaceArea);
        {
          const float t = newPos[0] + 1.0f;
          newPos[0] = (t - 2.0f * floorf(t / 2.0f)) - 1.0f;
        }
        feedback[MASS] = totalMass;
      }
      const float pntSize = 12.0f, pntSurface = 12.0f * 12.0f;
      feedback[MASS] /= pntSurface;
      feedback[HEAT] /= pntSurface;
      feedback[VAPOR] /= pntSurface;
      deposition[0] /= pntSize;
      deposition[1] /= pntSize;
      size = pntSize;
      gpos[0] = newPos[0];
      gpos[1] = newPos[1];
    }

    drops_out[5 * i + 0] = newPos[0];
    drops_out[5 * i + 1] = newPos[1];
    drops_out[5 * i + 2] = newMass[0];
    drops_out[5 * i + 3] = newMass[1];
    drops_out[5 * i + 4] = fmaxf(newDensity, 0.0f);
    splat(p, gpos[0], gpos[1], size, feedback, deposition, fb, dep, an);
  }
  if (an) {
    box_sum_anchors(p, an, fb, dep);
    free(an->acc);
  }
}

/* lightningLocationShader.frag:24-38 */
void wxo_lightning_location(const wxo_params *p, float iterNum, const float *fb, float *lightning)
{
  const float *n = fb + 4 * 1; /* texel (1,0) */
  (void)p;
  if (n[2] < fmaxf(iterNum - 1.0f, 1.0f) || n[2] > iterNum) return; /* discard */
  memcpy(lightning, n, 16);
}

/* ------------------------------------------------------------------------------------------
 * Whole-simulation object: texture set + ping-pong of app.js:5830-6005
 * ---------------------------------------------------------------------------------------- */
struct wxo_sim {
  int X, Y, n_drops;
  wxo_params p;
  float *initial_T, *snd_T, *snd_W, *snd_Vel;
  float *base[2], *water[2];
  int8_t *wall[2];
  float *light[2], *curl, *vort, *fb, *dep;
  float *emitted; /* emittedLight, unrounded */
  float lightning[4];
  float *drops[2];
  int even;     /* app.js: `even` */
  int drop_cur; /* buffer holding the most recent particle state */
  int64_t iter;
};

wxo_sim *wxo_create(int X, int Y, int n_drops)
{
  wxo_sim *s = (wxo_sim *)calloc(1, sizeof(*s));
  const size_t n = (size_t)X * Y;
  s->X = X;
  s->Y = Y;
  s->n_drops = n_drops;
  for (int i = 0; i < 2; i++) {
    s->base[i] = (float *)calloc(n * 4, 4);
    s->water[i] = (float *)calloc(n * 4, 4);
    s->wall[i] = (int8_t *)calloc(n * 4, 1);
    s->light[i] = (float *)calloc(n * 4, 4);
    s->drops[i] = (float *)calloc((size_t)(n_drops > 0 ? n_drops : 1) * 5, 4);
  }
  s->curl = (float *)calloc(n, 4);
  s->emitted = (float *)calloc(n, 16);
  s->vort = (float *)calloc(n * 2, 4);
  s->fb = (float *)calloc(n * 4, 4);
  s->dep = (float *)calloc(n * 2, 4);
  s->initial_T = (float *)calloc((size_t)Y + 8, 4);
  s->snd_T = (float *)calloc((size_t)Y + 8, 4);
  s->snd_W = (float *)calloc((size_t)Y + 8, 4);
  s->snd_Vel = (float *)calloc((size_t)Y + 8, 4);
  s->even = 1;
  s->p.X = X;
  s->p.Y = Y;
  s->p.X_global = X;
  return s;
}

void wxo_destroy(wxo_sim *s)
{
  if (!s) return;
  for (int i = 0; i < 2; i++) {
    free(s->base[i]);
    free(s->water[i]);
    free(s->wall[i]);
    free(s->light[i]);
    free(s->drops[i]);
  }
  free(s->curl);
  free(s->emitted);
  free(s->vort);
  free(s->fb);
  free(s->dep);
  free(s->initial_T);
  free(s->snd_T);
  free(s->snd_W);
  free(s->snd_Vel);
  free(s);
}

/* setupTextures() app.js:5189-5234 (same data into _0 and _1) + setupPrecipitationBuffers() */
void wxo_upload(wxo_sim *s, const float *base, const float *water, const int8_t *wall, const float *drops)
{
  const size_t n = (size_t)s->X * s->Y;
  for (int i = 0; i < 2; i++) {
    memcpy(s->base[i], base, n * 16);
    memcpy(s->water[i], water, n * 16);
    memcpy(s->wall[i], wall, n * 4);
    memset(s->light[i], 0, n * 16);
    if (drops && s->n_drops > 0) memcpy(s->drops[i], drops, (size_t)s->n_drops * 20);
  }
  memset(s->curl, 0, n * 4);
  memset(s->emitted, 0, n * 16);
  memset(s->vort, 0, n * 8);
  memset(s->fb, 0, n * 16);
  memset(s->dep, 0, n * 8);
  memset(s->lightning, 0, 16);
  s->even = 1;
  s->drop_cur = 0;
}

void wxo_set_params(wxo_sim *s, const wxo_params *p, const float *initial_T, const float *snd_T,
                    const float *snd_W, const float *snd_Vel)
{
  const float keep_inactive = s->p.inactiveDroplets;
  s->p = *p;
  s->p.X = s->X;
  s->p.Y = s->Y;
  if (s->p.X_global <= 0) s->p.X_global = s->X;
  if (p->inactiveDroplets < 0.0f) s->p.inactiveDroplets = keep_inactive;
  if (initial_T) memcpy(s->initial_T, initial_T, ((size_t)s->Y + 1) * 4);
  if (snd_T) memcpy(s->snd_T, snd_T, ((size_t)s->Y + 1) * 4);
  if (snd_W) memcpy(s->snd_W, snd_W, ((size_t)s->Y + 1) * 4);
  if (snd_Vel) memcpy(s->snd_Vel, snd_Vel, ((size_t)s->Y + 1) * 4);
}

void wxo_step_ex(wxo_sim *s, int n_iter, unsigned mask)
{
  const size_t n = (size_t)s->X * s->Y;
  for (int it = 0; it < n_iter; it++) {
    const float iterNum = (float)s->iter;
    /* 1 velocity: base_0, wall_0 -> base_1, wall_1 (app.js:5832-5839) */
    if (mask & 1u)
      wxo_velocity(&s->p, s->base[0], s->wall[0], s->base[1], s->wall[1]);
    else {
      memcpy(s->base[1], s->base[0], n * 16);
      memcpy(s->wall[1], s->wall[0], n * 4);
    }
    /* 2,3 curl, vorticity (5842-5855) */
    if (mask & 2u) {
      wxo_curl(&s->p, s->base[1], s->curl);
      wxo_vorticity(&s->p, s->curl, s->vort);
    }
    /* 4 boundary: base_1, water_1, vort, wall_1, light_0, fb, dep -> base_0, water_0, wall_0 */
    if (mask & 4u)
      wxo_boundary(&s->p, s->initial_T, iterNum, s->base[1], s->water[1], s->vort, s->wall[1], s->light[0],
                   s->fb, s->dep, s->base[0], s->water[0], s->wall[0]);
    else {
      memcpy(s->base[0], s->base[1], n * 16);
      memcpy(s->water[0], s->water[1], n * 16);
      memcpy(s->wall[0], s->wall[1], n * 4);
    }
    /* 5 advection: _0 -> _1 (5881-5890) */
    if (mask & 8u)
      wxo_advection(&s->p, s->initial_T, s->snd_T, s->snd_W, s->snd_Vel, s->base[0], s->water[0], s->wall[0],
                    s->base[1], s->water[1], s->wall[1]);
    else {
      memcpy(s->base[1], s->base[0], n * 16);
      memcpy(s->water[1], s->water[0], n * 16);
      memcpy(s->wall[1], s->wall[0], n * 4);
    }
    /* 6 pressure: base_1, wall_1 -> base_0, wall_0 (5893-5900) */
    if (mask & 16u)
      wxo_pressure(&s->p, s->base[1], s->wall[1], s->base[0], s->wall[0]);
    else {
      memcpy(s->base[0], s->base[1], n * 16);
      memcpy(s->wall[0], s->wall[1], n * 4);
    }
    /* 7 lighting (5903-5930) */
    const int src = s->even ? 0 : 1, dst = s->even ? 1 : 0;
    if (mask & 32u) wxo_lighting_mrt(&s->p, s->base[1], s->water[1], s->wall[1], s->light[src], s->light[dst], s->emitted);
    s->even = !s->even;
    /* 8 clear (5933-5934) */
    memset(s->fb, 0, n * 16);
    memset(s->dep, 0, n * 8);
    /* 9,10 precipitation + lightning location (5936-5983) */
    if ((mask & 64u) && s->p.enablePrecipitation && s->n_drops > 0) {
      wxo_precipitation(&s->p, iterNum, s->n_drops, s->drops[src], s->base[1], s->water[1], s->lightning,
                        s->drops[dst], s->fb, s->dep);
      s->drop_cur = dst;
      if (s->iter % 600 == 0) s->p.inactiveDroplets = s->fb[0];
      wxo_lightning_location(&s->p, iterNum, s->fb, s->lightning);
    }
    s->iter++;
  }
}

void wxo_step(wxo_sim *s, int n_iter) { wxo_step_ex(s, n_iter, 0x7Fu); }

int64_t wxo_get_iter(const wxo_sim *s) { return s->iter; }
void wxo_set_iter(wxo_sim *s, int64_t it) { s->iter = it; }

const void *wxo_field(const wxo_sim *s, int field)
{
  switch (field) {
  case 0: return s->base[0];
  case 1: return s->base[1];
  case 2: return s->water[0];
  case 3: return s->water[1];
  case 4: return s->wall[0];
  case 5: return s->wall[1];
  case 6: return s->light[0];
  case 7: return s->light[1];
  case 8: return s->curl;
  case 9: return s->vort;
  case 10: return s->fb;
  case 11: return s->dep;
  case 12: return s->lightning;
  case 13: return s->drops[s->drop_cur];
  case 14: return s->emitted;
  default: return NULL;
  }
}
