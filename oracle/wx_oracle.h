/*
 * wx_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the weather-sandbox hot path.
 *
 * This is a plain-C restatement of the reference's simulation-iteration shaders
 * (reference = niels747/2D-Weather-Sandbox, the sim fragment shaders, the
 * precipitationShader.vert, shaders/common.glsl and the draw() loop app.js:5830-6005).
 * It is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * NOTHING in the product path (2d-weather-sandbox_amd/, include/, host/) may include,
 * link, import or execute it.
 *
 * Parity pin: validated against per-pass and multi-iteration golden vectors produced by the
 * reference shaders themselves under SwiftShader (oracle/golden/, tests/golden/).
 */
#ifndef WX_ORACLE_H
#define WX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Uniform values of all simulation programs. All fp32, as after gl.uniform1f. */
typedef struct wxo_params {
  /* geometry */
  int32_t X;          /* local width in cells (slab width incl. halo columns when decomposed) */
  int32_t Y;          /* height in cells */
  int32_t X_global;   /* width of the whole periodic domain (== X when not decomposed) */
  int32_t x_off;      /* global x of local column 0 (may be negative: wrapped mod X_global) */
  int32_t quad_scale; /* 1: model the 1.0000001 quad-UV scale (app.js:4770-4788); 0: exact x+0.5 */
  /* velocityShader.frag:14-16 */
  float dragMultiplier, wind;
  /* boundaryShader.frag:22-38 */
  float vorticity, landEvaporation, waterEvaporation, dynamicWaterTemperature;
  float evapHeat, waterWeight, sunAngle, dryLapse;
  /* advectionShader.frag:33-43 */
  float meltingHeat, condensationRate, globalDrying, globalHeating, soundingForcing;
  float globalEffectsStartAlt, globalEffectsEndAlt, waterTemperature;
  /* lightingShader.frag:22-29 */
  float sunIntensity, greenhouseGases, waterGreenHouseEffect, IR_rate;
  /* precipitationShader.vert:31-48 */
  float aboveZeroThreshold, subZeroThreshold, spawnChanceMult, snowDensity, fallSpeed;
  float growthRate0C, growthRate_30C, freezingRate, meltingRate, evapRate, inactiveDroplets;
  /* advectionShader.frag:21-31 (brush / airplane) */
  float userInputValues[4];
  float userInputMove[2];
  int32_t userInputType;
  int32_t wrapHorizontally;
  float airplaneValues[4];
  /* host switches (app.js:5936) */
  int32_t enablePrecipitation;
  /* optional: measured simShader.vert varyings (fragCoord.xy, texCoord.xy) per cell, Y*X*4 floats, for
   * comparing against goldens from a rasteriser whose interpolation is not (i+0.5)*scale; NULL = analytic */
  const float *varyings;
  /* particle splats: 0 = exact window coordinates; n>0 = snap to 1/2^n pixel first (SwiftShader: 4) */
  int32_t subpixel_bits;
  /* particle splats: 0 = every covered texel receives the droplets' values in droplet-index order (what an in-order blend
   * unit does; the pinned mode); 1 = the summation tree of the HIP engine's deterministic mode (WX_OPT_SPLAT_ORDER 1): per
   * sprite ANCHOR texel the deposits are added in droplet-index order, then a 12x12 box sum with index-anchored trees
   * (pairs, quads, (q0 + q4) + q8; columns first) -- the same real sum, a different fp32 association; validated against
   * mode 0 to rounding in tests/test_oracle_golden.py */
  int32_t splat_order;
} wxo_params;

typedef struct wxo_sim wxo_sim;

/* ---- single passes (each restates one shader; arrays are row-major, y=0 bottom, 4 ch/texel) ---- */
void wxo_velocity(const wxo_params *p, const float *base_in, const int8_t *wall_in, float *base_out,
                  int8_t *wall_out);
void wxo_curl(const wxo_params *p, const float *base_in, float *curl_out);
void wxo_vorticity(const wxo_params *p, const float *curl_in, float *vort_out);
void wxo_boundary(const wxo_params *p, const float *initial_T, float iterNum, const float *base_in,
                  const float *water_in, const float *vort_in, const int8_t *wall_in,
                  const float *light_in, const float *fb_in, const float *dep_in, float *base_out,
                  float *water_out, int8_t *wall_out);
void wxo_advection(const wxo_params *p, const float *initial_T, const float *snd_T,
                   const float *snd_W, const float *snd_Vel, const float *base_in,
                   const float *water_in, const int8_t *wall_in, float *base_out, float *water_out,
                   int8_t *wall_out);
void wxo_pressure(const wxo_params *p, const float *base_in, const int8_t *wall_in, float *base_out,
                  int8_t *wall_out);
void wxo_lighting(const wxo_params *p, const float *base_in, const float *water_in,
                  const int8_t *wall_in, const float *light_in, float *light_out);
/* the same pass with its second render target (emittedLight; emit_out may be NULL) */
void wxo_lighting_mrt(const wxo_params *p, const float *base_in, const float *water_in,
                      const int8_t *wall_in, const float *light_in, float *light_out, float *emit_out);
/* particles: drops are 5 floats each (pos.xy, mass.xy, density); fb RGBA32F, dep RG32F are += */
void wxo_precipitation(const wxo_params *p, float iterNum, int n_drops, const float *drops_in,
                       const float *base_in, const float *water_in, const float *lightning_in,
                       float *drops_out, float *fb, float *dep);
void wxo_lightning_location(const wxo_params *p, float iterNum, const float *fb, float *lightning);

/* hash known-answer entry points (common.glsl:103-137) */
uint32_t wxo_hash(uint32_t x);
float wxo_random2d(float sx, float sy);

/* ---- whole simulation object: exact ping-pong of app.js:5830-6005 ---- */
wxo_sim *wxo_create(int X, int Y, int n_drops);
void wxo_destroy(wxo_sim *s);
void wxo_upload(wxo_sim *s, const float *base, const float *water, const int8_t *wall,
                const float *drops);
void wxo_set_params(wxo_sim *s, const wxo_params *p, const float *initial_T /*Y+1*/,
                    const float *snd_T, const float *snd_W, const float *snd_Vel /*Y+1 or NULL*/);
void wxo_step(wxo_sim *s, int n_iter);
/* pass_mask for wxo_step_ex: bit0 velocity, 1 curl+vorticity, 2 boundary, 3 advection, 4 pressure,
 * 5 lighting, 6 precipitation(+clear+lightning). Masked-off passes copy through (dry config C2). */
void wxo_step_ex(wxo_sim *s, int n_iter, unsigned pass_mask);
int64_t wxo_get_iter(const wxo_sim *s);
void wxo_set_iter(wxo_sim *s, int64_t it);
/* field ids follow include/wxsim.h WX_FIELD_* */
const void *wxo_field(const wxo_sim *s, int field);

#ifdef __cplusplus
}
#endif
#endif
