/*
 * wxsim_napi.c -- thin Node.js N-API (raw C, N-API v8 / Node 12+) shim over the C ABI of include/wxsim.h.
 *
 * This is the binding a maintainer of the reference adds so that app.js' per-frame simulation block
 * (app.js:5830-6005) and its gl.readPixels consumers call the MI355X engine instead of WebGL:
 *   JS (host/sim_host.js)  ->  this N-API addon  ->  libwxsim.so (C ABI)  ->  HIP kernels.
 * libwxsim.so is dlopen()ed at require() time from ../2d-weather-sandbox_amd/csrc (or $WXSIM_LIB).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/wxsim.h"

static struct {
  void *dl;
  int (*create)(int, int, int, wx_sim **);
  void (*destroy)(wx_sim *);
  const char *(*last_error)(const wx_sim *);
  int (*abi_version)(void);
  int (*upload)(wx_sim *, const float *, const float *, const int8_t *, const float *);
  int (*set_params)(wx_sim *, const wx_params *, const float *, const float *, const float *, const float *);
  int (*step)(wx_sim *, int);
  int (*sync)(wx_sim *);
  int (*set_option)(wx_sim *, int, int);
  int (*tune_placement)(wx_sim *, int, int, float *, float *);
  int (*placement_info)(const wx_sim *, float *, float *);
  int (*pair_stats)(wx_sim *, int64_t *, int64_t *);
  int64_t (*get_iter)(const wx_sim *);
  int (*set_iter)(wx_sim *, int64_t);
  int (*read_rect)(wx_sim *, int, int, int, int, int, void *, int);
  int (*read_particles)(wx_sim *, int, int, float *);
  int (*setup_columns)(wx_sim *, const int32_t *, const uint8_t *, const double *, const float *, const float *, const float *, const float *,
                       const float *);
  int (*setup_terrain)(wx_sim *, double, double, int, double, const float *, const float *, const float *, const float *);
  int (*init_droplets)(wx_sim *, uint32_t);
  size_t (*stream_bytes)(int, int);
  void *(*host_alloc)(size_t);
  void (*host_free)(void *);
  int (*stream_frame)(wx_sim *, int, int, int, int, void *);
  int (*stream_wait)(wx_sim *);
  /* N slabs in one process (wx_group_*: the decomposed domain from a JavaScript host) */
  int (*group_create)(int, const int *, int, int, int, int, int, wx_group **);
  void (*group_destroy)(wx_group *);
  const char *(*group_last_error)(const wx_group *);
  int (*group_count)(const wx_group *);
  int (*group_transport)(const wx_group *);
  wx_sim *(*group_slab)(wx_group *, int);
  int (*group_agree)(wx_group *);
  int (*group_step)(wx_group *, int);
  int (*group_sync)(wx_group *);
  int (*group_set_option)(wx_group *, int, int);
  int (*group_exchange)(wx_group *);
  int (*pool_flags)(wx_sim *, uint8_t *);
  int (*local_width)(const wx_sim *);
} L;

#define NAPI_CALL(env, call)                                                        \
  do {                                                                              \
    if ((call) != napi_ok) {                                                        \
      napi_throw_error((env), NULL, "N-API call failed: " #call);                   \
      return NULL;                                                                  \
    }                                                                               \
  } while (0)

static napi_value throw_wx(napi_env env, wx_sim *s, int rc, const char *what)
{
  char buf[640];
  snprintf(buf, sizeof(buf), "%s failed (%d): %s", what, rc, L.last_error ? L.last_error(s) : "");
  napi_throw_error(env, NULL, buf);
  return NULL;
}

/* what the external wraps: the C handle plus the dimensions the array arguments are checked against. wx_sim * stays the
 * first member, so a slot pointer also reads as wx_sim ** */
struct wx_gslot;
typedef struct wx_slot {
  wx_sim *s;
  int32_t X, Y, N;
  int32_t borrowed; /* a slab of a group: the group destroys it */
  /* borrowed slots: the group they belong to (NULL once it is gone), their link in its list, and a strong reference to the group's
   * external -- a slab handle JS still holds keeps its group alive, and destroying the group turns every slab handle it lent out
   * into a dead one ("expected a simulation handle") instead of a dangling pointer */
  struct wx_gslot *group;
  struct wx_slot *next_in_group;
  napi_ref group_ref;
} wx_slot;
typedef struct wx_gslot {
  wx_group *g;
  int32_t N; /* droplets of the pool every slab holds */
  int32_t Y; /* grid height, as given to groupCreate: what the slabs' array arguments are checked against */
  wx_slot *slabs; /* the slab handles lent out (groupSlab) */
} wx_gslot;

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "wrong number of arguments");
    return -1;
  }
  return 0;
}

static wx_sim *get_handle(napi_env env, napi_value v)
{
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_type_error(env, NULL, "expected a simulation handle");
    return NULL;
  }
  if (!*(wx_sim **)p) napi_throw_type_error(env, NULL, "the simulation handle was destroyed (or the slab group it belonged to)");
  return *(wx_sim **)p;
}
static wx_slot *get_slot(napi_env env, napi_value v)
{
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((wx_slot *)p)->s) {
    napi_throw_type_error(env, NULL, "expected a simulation handle");
    return NULL;
  }
  return (wx_slot *)p;
}
/* a typed array of the wanted type holding at least `need` elements, else a thrown RangeError / TypeError and NULL */
static void *typed_data(napi_env env, napi_value v, napi_typedarray_type want, size_t *len);
static void *typed_at_least(napi_env env, napi_value v, napi_typedarray_type want, size_t need, const char *what)
{
  size_t len = 0;
  void *d = typed_data(env, v, want, &len);
  char buf[160];
  if (!d) {
    snprintf(buf, sizeof(buf), "%s: typed array of the wrong type (or not a typed array)", what);
    napi_throw_type_error(env, NULL, buf);
    return NULL;
  }
  if (len < need) {
    snprintf(buf, sizeof(buf), "%s: %zu elements, %zu needed", what, len, need);
    napi_throw_range_error(env, NULL, buf);
    return NULL;
  }
  return d;
}
static int is_nullish(napi_env env, napi_value v)
{
  napi_valuetype t;
  return napi_typeof(env, v, &t) != napi_ok || t == napi_null || t == napi_undefined;
}

static void finalize_handle(napi_env env, void *data, void *hint)
{
  wx_slot *slot = (wx_slot *)data;
  if (slot->s && !slot->borrowed) L.destroy(slot->s);
  if (slot->borrowed) {
    if (slot->group) { /* leave the group's list */
      wx_slot **pp = &slot->group->slabs;
      while (*pp && *pp != slot) pp = &(*pp)->next_in_group;
      if (*pp) *pp = slot->next_in_group;
    }
    if (slot->group_ref) napi_delete_reference(env, slot->group_ref);
  }
  free(slot);
}

static void *typed_data(napi_env env, napi_value v, napi_typedarray_type want, size_t *len)
{
  bool is = false;
  napi_typedarray_type t;
  void *data = NULL;
  if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return NULL;
  if (napi_get_typedarray_info(env, v, &t, len, &data, NULL, NULL) != napi_ok || t != want) return NULL;
  return data;
}

/* create(X, Y, nDroplets) -> handle   [replaces texture/FBO/particle-buffer creation, app.js:5149-5317, 4891-5002] */
static napi_value Create(napi_env env, napi_callback_info info)
{
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  int32_t X, Y, N;
  NAPI_CALL(env, napi_get_value_int32(env, a[0], &X));
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &Y));
  NAPI_CALL(env, napi_get_value_int32(env, a[2], &N));
  wx_sim *s = NULL;
  int rc = L.create(X, Y, N, &s);
  if (rc) return throw_wx(env, NULL, rc, "wx_create");
  wx_slot *slot = (wx_slot *)malloc(sizeof(*slot));
  slot->s = s;
  slot->X = X;
  slot->Y = Y;
  slot->N = N;
  slot->borrowed = 0;
  slot->group = NULL;
  slot->next_in_group = NULL;
  slot->group_ref = NULL;
  napi_value ext;
  NAPI_CALL(env, napi_create_external(env, slot, finalize_handle, NULL, &ext));
  return ext;
}

static napi_value Destroy(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  void *p = NULL;
  if (napi_get_value_external(env, a[0], &p) == napi_ok && p && *(wx_sim **)p) {
    if (!((wx_slot *)p)->borrowed) L.destroy(*(wx_sim **)p);
    *(wx_sim **)p = NULL;
  }
  return NULL;
}

/* upload(h, Float32Array base, Float32Array water, Int8Array wall, Float32Array drops|null)  [setupTextures()] */
static napi_value Upload(napi_env env, napi_callback_info info)
{
  napi_value a[5];
  if (get_args(env, info, 5, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  const wx_slot *sl = get_slot(env, a[0]);
  if (!sl) return NULL;
  const size_t n4 = (size_t)sl->X * sl->Y * 4; /* wx_upload reads X*Y texels of 4 channels from each grid array */
  const float *base = (const float *)typed_at_least(env, a[1], napi_float32_array, n4, "upload: base (Float32Array, X*Y*4)");
  if (!base) return NULL;
  const float *water = (const float *)typed_at_least(env, a[2], napi_float32_array, n4, "upload: water (Float32Array, X*Y*4)");
  if (!water) return NULL;
  const int8_t *wall = (const int8_t *)typed_at_least(env, a[3], napi_int8_array, n4, "upload: wall (Int8Array, X*Y*4)");
  if (!wall) return NULL;
  const float *drops = NULL;
  if (!is_nullish(env, a[4]) && sl->N > 0) {
    drops = (const float *)typed_at_least(env, a[4], napi_float32_array, (size_t)sl->N * 5, "upload: drops (Float32Array, nDroplets*5)");
    if (!drops) return NULL;
  }
  int rc = L.upload(s, base, water, wall, drops);
  if (rc) return throw_wx(env, s, rc, "wx_upload");
  return NULL;
}

static int get_f(napi_env env, napi_value obj, const char *k, float *out)
{
  napi_value v;
  double d;
  bool has = false;
  if (napi_has_named_property(env, obj, k, &has) != napi_ok || !has) return 0;
  if (napi_get_named_property(env, obj, k, &v) != napi_ok || napi_get_value_double(env, v, &d) != napi_ok) return 0;
  *out = (float)d;
  return 1;
}
static void get_fv(napi_env env, napi_value obj, const char *k, float *out, uint32_t n)
{
  napi_value arr, e;
  bool has = false;
  if (napi_has_named_property(env, obj, k, &has) != napi_ok || !has) return;
  if (napi_get_named_property(env, obj, k, &arr) != napi_ok) return;
  for (uint32_t i = 0; i < n; i++) {
    double d;
    if (napi_get_element(env, arr, i, &e) == napi_ok && napi_get_value_double(env, e, &d) == napi_ok) out[i] = (float)d;
  }
}

/* setParams(h, uniformsObject, Float32Array initial_T [, snd_T, snd_W, snd_Vel])   [gl.uniform* pushes] */
static napi_value SetParams(napi_env env, napi_callback_info info)
{
  napi_value a[6];
  size_t argc = 6;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, NULL, NULL));
  if (argc < 3) {
    napi_throw_type_error(env, NULL, "setParams(handle, uniforms, initial_T[, sndT, sndW, sndVel])");
    return NULL;
  }
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  wx_params p;
  memset(&p, 0, sizeof(p));
  p.userInputType = -1;
  p.pass_mask = WX_PASS_ALL;
  napi_value o = a[1];
#define F(name) get_f(env, o, #name, &p.name)
  F(dragMultiplier); F(wind); F(vorticity); F(landEvaporation); F(waterEvaporation); F(dynamicWaterTemperature);
  F(evapHeat); F(waterWeight); F(sunAngle); F(dryLapse); F(meltingHeat); F(condensationRate); F(globalDrying);
  F(globalHeating); F(soundingForcing); F(globalEffectsStartAlt); F(globalEffectsEndAlt); F(waterTemperature);
  F(sunIntensity); F(greenhouseGases); F(waterGreenHouseEffect); F(IR_rate); F(aboveZeroThreshold); F(subZeroThreshold);
  F(spawnChanceMult); F(snowDensity); F(fallSpeed); F(growthRate0C); F(growthRate_30C); F(freezingRate); F(meltingRate);
  F(evapRate); F(inactiveDroplets);
#undef F
  get_fv(env, o, "userInputValues", p.userInputValues, 4);
  get_fv(env, o, "userInputMove", p.userInputMove, 2);
  get_fv(env, o, "airplaneValues", p.airplaneValues, 4);
  float t;
  if (get_f(env, o, "userInputType", &t)) p.userInputType = (int32_t)t;
  if (get_f(env, o, "wrapHorizontally", &t)) p.wrapHorizontally = (int32_t)t;
  if (get_f(env, o, "enablePrecipitation", &t)) p.enablePrecipitation = (int32_t)t;
  if (get_f(env, o, "quad_scale", &t)) p.quad_scale = (int32_t)t;
  if (get_f(env, o, "pass_mask", &t)) p.pass_mask = (uint32_t)t;
  const wx_slot *sl = get_slot(env, a[0]);
  if (!sl) return NULL;
  const size_t ny = (size_t)sl->Y + 1; /* wx_set_params copies Y+1 entries of every profile array it is given */
  const float *T0 = NULL;
  if (!is_nullish(env, a[2])) {
    T0 = (const float *)typed_at_least(env, a[2], napi_float32_array, ny, "setParams: initial_T (Float32Array, Y+1)");
    if (!T0) return NULL;
  }
  const float *snd[3] = {NULL, NULL, NULL};
  for (size_t i = 3; i < argc && i < 6; i++) {
    if (is_nullish(env, a[i])) continue;
    snd[i - 3] = (const float *)typed_at_least(env, a[i], napi_float32_array, ny, "setParams: sounding array (Float32Array, Y+1)");
    if (!snd[i - 3]) return NULL;
  }
  int rc = L.set_params(s, &p, T0, snd[0], snd[1], snd[2]);
  if (rc) return throw_wx(env, s, rc, "wx_set_params");
  return NULL;
}

/* step(h, nIter)   [the loop body app.js:5830-6005, asynchronous] */
static napi_value Step(napi_env env, napi_callback_info info)
{
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int32_t n;
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &n));
  int rc = L.step(s, n);
  if (rc) return throw_wx(env, s, rc, "wx_step");
  return NULL;
}

/* setOption(h, option, value): wx_set_option (1 = WX_OPT_SPLAT_ORDER: deterministic particle splats, 2 = WX_OPT_CHECK_LAUNCHES) */
static napi_value SetOption(napi_env env, napi_callback_info info)
{
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int32_t opt, val;
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &opt));
  NAPI_CALL(env, napi_get_value_int32(env, a[2], &val));
  int rc = L.set_option(s, opt, val);
  if (rc) return throw_wx(env, s, rc, "wx_set_option");
  return NULL;
}

/* tunePlacement(h, tries, itersPerTry) -> [msBefore, msAfter]: wx_tune_placement (the state is unchanged) */
static napi_value TunePlacement(napi_env env, napi_callback_info info)
{
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int32_t tries, iters;
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &tries));
  NAPI_CALL(env, napi_get_value_int32(env, a[2], &iters));
  float before = 0.f, after = 0.f;
  int rc = L.tune_placement(s, tries, iters, &before, &after);
  if (rc) return throw_wx(env, s, rc, "wx_tune_placement");
  napi_value arr, v0, v1;
  NAPI_CALL(env, napi_create_array_with_length(env, 2, &arr));
  NAPI_CALL(env, napi_create_double(env, before, &v0));
  NAPI_CALL(env, napi_create_double(env, after, &v1));
  NAPI_CALL(env, napi_set_element(env, arr, 0, v0));
  NAPI_CALL(env, napi_set_element(env, arr, 1, v1));
  return arr;
}

/* placementInfo(h) -> [msFirst, msKept] of the handle's placement search (tunePlacement, or the implicit one inside the first step of a
 * big whole-domain handle: WX_OPT_PLACEMENT_SEARCH), or null if none has run */
static napi_value PlacementInfo(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  float first = 0.f, kept = 0.f;
  napi_value arr, v0, v1;
  if (L.placement_info(s, &first, &kept) != 1) {
    NAPI_CALL(env, napi_get_null(env, &arr));
    return arr;
  }
  NAPI_CALL(env, napi_create_array_with_length(env, 2, &arr));
  NAPI_CALL(env, napi_create_double(env, first, &v0));
  NAPI_CALL(env, napi_create_double(env, kept, &v1));
  NAPI_CALL(env, napi_set_element(env, arr, 0, v0));
  NAPI_CALL(env, napi_set_element(env, arr, 1, v1));
  return arr;
}

/* pairStats(h) -> [cellsRecomputed, pairsRepeated]: wx_pair_stats (the dry pair kernel's exact path since the last call) */
static napi_value PairStats(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int64_t fixed = 0, repeated = 0;
  int rc = L.pair_stats(s, &fixed, &repeated);
  if (rc) return throw_wx(env, s, rc, "wx_pair_stats");
  napi_value arr, v0, v1;
  NAPI_CALL(env, napi_create_array_with_length(env, 2, &arr));
  NAPI_CALL(env, napi_create_double(env, (double)fixed, &v0));
  NAPI_CALL(env, napi_create_double(env, (double)repeated, &v1));
  NAPI_CALL(env, napi_set_element(env, arr, 0, v0));
  NAPI_CALL(env, napi_set_element(env, arr, 1, v1));
  return arr;
}

static napi_value Sync(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int rc = L.sync(s);
  if (rc) return throw_wx(env, s, rc, "wx_sync");
  return NULL;
}

static napi_value GetIter(napi_env env, napi_callback_info info)
{
  napi_value a[1], r;
  if (get_args(env, info, 1, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  NAPI_CALL(env, napi_create_double(env, (double)L.get_iter(s), &r));
  return r;
}

static napi_value SetIter(napi_env env, napi_callback_info info)
{
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  double d;
  NAPI_CALL(env, napi_get_value_double(env, a[1], &d));
  int rc = L.set_iter(s, (int64_t)d);
  if (rc) return throw_wx(env, s, rc, "wx_set_iter");
  return NULL;
}

/* readRect(h, field, x, y, w, h, dstTypedArray)  [every gl.readPixels of SURVEY 3.5]; dst type selects dtype */
static napi_value ReadRect(napi_env env, napi_callback_info info)
{
  napi_value a[7];
  if (get_args(env, info, 7, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int32_t v[5];
  for (int i = 0; i < 5; i++) NAPI_CALL(env, napi_get_value_int32(env, a[1 + i], &v[i]));
  bool is = false;
  napi_typedarray_type t;
  size_t len;
  void *data;
  if (napi_is_typedarray(env, a[6], &is) != napi_ok || !is ||
      napi_get_typedarray_info(env, a[6], &t, &len, &data, NULL, NULL) != napi_ok) {
    napi_throw_type_error(env, NULL, "readRect: dst must be a typed array");
    return NULL;
  }
  int dtype = t == napi_float32_array ? WX_DTYPE_F32 : t == napi_int8_array ? WX_DTYPE_I8 : t == napi_int32_array ? WX_DTYPE_I32 : t == napi_uint16_array ? WX_DTYPE_F16 /* raw binary16 */ : -1;
  int ch = (v[0] == WX_FIELD_CURL) ? 1 : (v[0] == WX_FIELD_VORT || v[0] == WX_FIELD_PRECIP_DEP) ? 2 : 4;
  if (dtype < 0 || len < (size_t)v[3] * v[4] * ch) {
    napi_throw_range_error(env, NULL, "readRect: destination too small or of unsupported type");
    return NULL;
  }
  int rc = L.read_rect(s, v[0], v[1], v[2], v[3], v[4], data, dtype);
  if (rc) return throw_wx(env, s, rc, "wx_read_rect");
  return a[6];
}

/* readParticles(h, first, count, Float32Array dst)  [gl.getBufferSubData, app.js:5019, 5086, 6597] */
static napi_value ReadParticles(napi_env env, napi_callback_info info)
{
  napi_value a[4];
  if (get_args(env, info, 4, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int32_t first, count;
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &first));
  NAPI_CALL(env, napi_get_value_int32(env, a[2], &count));
  size_t len;
  float *dst = (float *)typed_data(env, a[3], napi_float32_array, &len);
  if (!dst || len < (size_t)count * 5) {
    napi_throw_range_error(env, NULL, "readParticles: dst must be a Float32Array of count*5");
    return NULL;
  }
  int rc = L.read_particles(s, first, count, dst);
  if (rc) return throw_wx(env, s, rc, "wx_read_particles");
  return a[3];
}

/* setupColumns(h, Int32Array wallRows, Uint8Array sea, Float64Array vegNoise, Float32Array snow, Float32Array T_air,
 *              Float32Array totalWater, Float32Array cloudWater, Float32Array drops|null)
 * [the setup draw of a new simulation, setupShader.frag:36-92, filled on the device from the 1-D descriptors] */
static napi_value SetupColumns(napi_env env, napi_callback_info info)
{
  napi_value a[9];
  if (get_args(env, info, 9, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  const wx_slot *sl = get_slot(env, a[0]);
  if (!sl) return NULL;
  const size_t nx = (size_t)sl->X, ny = (size_t)sl->Y; /* per-column descriptors: X entries; per-row sounding: Y entries */
  const int32_t *rows = (const int32_t *)typed_at_least(env, a[1], napi_int32_array, nx, "setupColumns: wallRows (Int32Array, X)");
  if (!rows) return NULL;
  const uint8_t *sea = (const uint8_t *)typed_at_least(env, a[2], napi_uint8_array, nx, "setupColumns: sea (Uint8Array, X)");
  if (!sea) return NULL;
  const double *veg = (const double *)typed_at_least(env, a[3], napi_float64_array, nx, "setupColumns: vegNoise (Float64Array, X)");
  if (!veg) return NULL;
  const float *snow = (const float *)typed_at_least(env, a[4], napi_float32_array, nx, "setupColumns: snow (Float32Array, X)");
  if (!snow) return NULL;
  const float *T = (const float *)typed_at_least(env, a[5], napi_float32_array, ny, "setupColumns: T_air (Float32Array, Y)");
  if (!T) return NULL;
  const float *tot = (const float *)typed_at_least(env, a[6], napi_float32_array, ny, "setupColumns: totalWater (Float32Array, Y)");
  if (!tot) return NULL;
  const float *cloud = (const float *)typed_at_least(env, a[7], napi_float32_array, ny, "setupColumns: cloudWater (Float32Array, Y)");
  if (!cloud) return NULL;
  const float *drops = NULL;
  if (!is_nullish(env, a[8]) && sl->N > 0) {
    drops = (const float *)typed_at_least(env, a[8], napi_float32_array, (size_t)sl->N * 5, "setupColumns: drops (Float32Array, nDroplets*5)");
    if (!drops) return NULL;
  }
  int rc = L.setup_columns(s, rows, sea, veg, snow, T, tot, cloud, drops);
  if (rc) return throw_wx(env, s, rc, "wx_setup_columns");
  return NULL;
}

/* setupTerrain(h, seed, heightMult, snap, simHeight, Float32Array T_air, totalWater, cloudWater, Float32Array drops | null): a new
 * simulation generated entirely on the device (wx_setup_terrain: setupShader.frag:26-92) */
static napi_value SetupTerrain(napi_env env, napi_callback_info info)
{
  napi_value a[9];
  if (get_args(env, info, 9, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  const wx_slot *sl = get_slot(env, a[0]);
  if (!sl) return NULL;
  double seed, mult, snap, simh;
  NAPI_CALL(env, napi_get_value_double(env, a[1], &seed));
  NAPI_CALL(env, napi_get_value_double(env, a[2], &mult));
  NAPI_CALL(env, napi_get_value_double(env, a[3], &snap));
  NAPI_CALL(env, napi_get_value_double(env, a[4], &simh));
  const size_t ny = (size_t)sl->Y;
  const float *T = (const float *)typed_at_least(env, a[5], napi_float32_array, ny, "setupTerrain: T_air (Float32Array, Y)");
  if (!T) return NULL;
  const float *tot = (const float *)typed_at_least(env, a[6], napi_float32_array, ny, "setupTerrain: totalWater (Float32Array, Y)");
  if (!tot) return NULL;
  const float *cloud = (const float *)typed_at_least(env, a[7], napi_float32_array, ny, "setupTerrain: cloudWater (Float32Array, Y)");
  if (!cloud) return NULL;
  const float *drops = NULL;
  if (!is_nullish(env, a[8]) && sl->N > 0) {
    drops = (const float *)typed_at_least(env, a[8], napi_float32_array, (size_t)sl->N * 5, "setupTerrain: drops (Float32Array, nDroplets*5)");
    if (!drops) return NULL;
  }
  int rc = L.setup_terrain(s, seed, mult, (int)snap, simh, T, tot, cloud, drops);
  if (rc) return throw_wx(env, s, rc, "wx_setup_terrain");
  return NULL;
}

/* initDroplets(h, seed): initRainDrops() on the device (wx_init_droplets) */
static napi_value InitDroplets(napi_env env, napi_callback_info info)
{
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  uint32_t seed;
  NAPI_CALL(env, napi_get_value_uint32(env, a[1], &seed));
  int rc = L.init_droplets(s, seed);
  if (rc) return throw_wx(env, s, rc, "wx_init_droplets");
  return NULL;
}

static void finalize_pinned(napi_env env, void *data, void *hint) { L.host_free(data); }

/* streamFrame(h, x, y, w, h) -> ArrayBuffer over pinned host memory that the copies fill asynchronously; streamWait(h)
 * blocks until it is complete. Layout: BASE_DISP f32x4, WATER_CUR f32x4, WALL_DISP i8x4, LIGHT_0 f32x4, CURL f32, PRECIP_FB f32x4
 * [what the renderer binds per frame, app.js:6081-6219] */
static napi_value StreamFrame(napi_env env, napi_callback_info info)
{
  napi_value a[5];
  if (get_args(env, info, 5, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int32_t r[4];
  for (int i = 0; i < 4; i++) NAPI_CALL(env, napi_get_value_int32(env, a[1 + i], &r[i]));
  const size_t bytes = L.stream_bytes(r[2], r[3]);
  void *buf = bytes ? L.host_alloc(bytes) : NULL;
  if (!buf) {
    napi_throw_error(env, NULL, "streamFrame: empty rectangle or wx_host_alloc failed");
    return NULL;
  }
  int rc = L.stream_frame(s, r[0], r[1], r[2], r[3], buf);
  if (rc) {
    L.host_free(buf);
    return throw_wx(env, s, rc, "wx_stream_frame");
  }
  napi_value ab;
  NAPI_CALL(env, napi_create_external_arraybuffer(env, buf, bytes, finalize_pinned, NULL, &ab));
  return ab;
}

static napi_value StreamWait(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  if (!s) return NULL;
  int rc = L.stream_wait(s);
  if (rc) return throw_wx(env, s, rc, "wx_stream_wait");
  return NULL;
}

/* ---- the decomposed domain: N column slabs in this process (wx_group_*, include/wxsim.h) ---- */
/* the group is going away: every slab handle it lent out becomes a dead handle */
static void orphan_slabs(wx_gslot *gs)
{
  for (wx_slot *sl = gs->slabs; sl;) {
    wx_slot *next = sl->next_in_group;
    sl->s = NULL;
    sl->group = NULL;
    sl->next_in_group = NULL;
    sl = next;
  }
  gs->slabs = NULL;
}
static void finalize_group(napi_env env, void *data, void *hint)
{
  wx_gslot *gs = (wx_gslot *)data;
  orphan_slabs(gs);
  if (gs->g) L.group_destroy(gs->g);
  free(gs);
}
static wx_gslot *get_group(napi_env env, napi_value v)
{
  void *p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((wx_gslot *)p)->g) {
    napi_throw_type_error(env, NULL, "expected a slab group");
    return NULL;
  }
  return (wx_gslot *)p;
}
static napi_value throw_group(napi_env env, wx_group *g, int rc, const char *what)
{
  char buf[640];
  snprintf(buf, sizeof(buf), "%s failed (%d): %s", what, rc, L.group_last_error(g));
  napi_throw_error(env, NULL, buf);
  return NULL;
}
/* groupCreate(nSlabs, Xglobal, Y, halo, transport[, nDroplets]) -> group   (transport: 0 auto, 1 RCCL, 2 local copies) */
static napi_value GroupCreate(napi_env env, napi_callback_info info)
{
  napi_value a[6];
  size_t argc = 6;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, NULL, NULL));
  if (argc < 5) {
    napi_throw_type_error(env, NULL, "groupCreate(nSlabs, Xglobal, Y, halo, transport[, nDroplets])");
    return NULL;
  }
  int32_t v[6] = {0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < argc && i < 6; i++) NAPI_CALL(env, napi_get_value_int32(env, a[i], &v[i]));
  wx_group *g = NULL;
  int rc = L.group_create(v[0], NULL, v[1], v[2], v[3], v[5], v[4], &g);
  if (rc) return throw_group(env, NULL, rc, "wx_group_create");
  wx_gslot *gs = (wx_gslot *)malloc(sizeof(*gs));
  gs->g = g;
  gs->N = v[5];
  gs->Y = v[2];
  gs->slabs = NULL;
  napi_value ext;
  NAPI_CALL(env, napi_create_external(env, gs, finalize_group, NULL, &ext));
  return ext;
}
/* groupSlab(group, i) -> handle of slab i, usable with upload / setParams / readRect ... (owned by the group; dead once the group is
 * destroyed). The grid height its array arguments are checked against is the one groupCreate was given -- a third argument, which
 * earlier versions took as Y, is ignored. */
static napi_value GroupSlab(napi_env env, napi_callback_info info)
{
  napi_value a[3];
  if (get_args(env, info, 2, a)) return NULL;
  wx_gslot *gs = get_group(env, a[0]);
  if (!gs) return NULL;
  int32_t i;
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &i));
  wx_sim *s = L.group_slab(gs->g, i);
  if (!s) {
    napi_throw_range_error(env, NULL, "groupSlab: no such slab");
    return NULL;
  }
  wx_slot *slot = (wx_slot *)malloc(sizeof(*slot));
  slot->s = s;
  slot->X = L.local_width(s);
  slot->Y = gs->Y;
  slot->N = gs->N;
  slot->borrowed = 1;
  slot->group = NULL;
  slot->next_in_group = NULL;
  slot->group_ref = NULL;
  napi_value ext;
  if (napi_create_external(env, slot, finalize_handle, NULL, &ext) != napi_ok) {
    free(slot);
    napi_throw_error(env, NULL, "groupSlab: napi_create_external failed");
    return NULL;
  }
  /* from here on the finalizer owns the slot */
  if (napi_create_reference(env, a[0], 1, &slot->group_ref) != napi_ok) slot->group_ref = NULL;
  slot->group = gs;
  slot->next_in_group = gs->slabs;
  gs->slabs = slot;
  return ext;
}
static napi_value GroupInfo(napi_env env, napi_callback_info info)
{
  napi_value a[1], o, v;
  if (get_args(env, info, 1, a)) return NULL;
  wx_gslot *gs = get_group(env, a[0]);
  if (!gs) return NULL;
  NAPI_CALL(env, napi_create_object(env, &o));
  NAPI_CALL(env, napi_create_int32(env, L.group_count(gs->g), &v));
  NAPI_CALL(env, napi_set_named_property(env, o, "slabs", v));
  NAPI_CALL(env, napi_create_int32(env, L.group_transport(gs->g), &v));
  NAPI_CALL(env, napi_set_named_property(env, o, "transport", v));
  return o;
}
static napi_value GroupAgree(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  wx_gslot *gs = get_group(env, a[0]);
  if (!gs) return NULL;
  int rc = L.group_agree(gs->g);
  if (rc) return throw_group(env, gs->g, rc, "wx_group_agree");
  return NULL;
}
/* groupStep(group, n): n iterations of every slab with the ring halo exchange every halo / 6 iterations  [loop body app.js:5830-6005] */
static napi_value GroupStep(napi_env env, napi_callback_info info)
{
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  wx_gslot *gs = get_group(env, a[0]);
  if (!gs) return NULL;
  int32_t n;
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &n));
  int rc = L.group_step(gs->g, n);
  if (rc) return throw_group(env, gs->g, rc, "wx_group_step");
  return NULL;
}
/* groupSetOption(group, option, value): wx_set_option on every slab */
static napi_value GroupSetOption(napi_env env, napi_callback_info info)
{
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  wx_gslot *gs = get_group(env, a[0]);
  if (!gs) return NULL;
  int32_t o, v;
  NAPI_CALL(env, napi_get_value_int32(env, a[1], &o));
  NAPI_CALL(env, napi_get_value_int32(env, a[2], &v));
  int rc = L.group_set_option(gs->g, o, v);
  if (rc) return throw_group(env, gs->g, rc, "wx_group_set_option");
  return NULL;
}
/* groupExchange(group): an exchange now -- afterwards every active droplet is owned by exactly one slab */
static napi_value GroupExchange(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  wx_gslot *gs = get_group(env, a[0]);
  if (!gs) return NULL;
  int rc = L.group_exchange(gs->g);
  if (rc) return throw_group(env, gs->g, rc, "wx_group_exchange");
  return NULL;
}
/* poolFlags(h, Uint8Array nDroplets) -> the array: 0 tracked by another slab, 1 inactive, 2 active and owned here, 3 ghost copy */
static napi_value PoolFlags(napi_env env, napi_callback_info info)
{
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  wx_sim *s = get_handle(env, a[0]);
  const wx_slot *sl = s ? get_slot(env, a[0]) : NULL;
  if (!sl) return NULL;
  uint8_t *dst = (uint8_t *)typed_at_least(env, a[1], napi_uint8_array, (size_t)sl->N, "poolFlags: Uint8Array, nDroplets");
  if (!dst) return NULL;
  int rc = L.pool_flags(s, dst);
  if (rc) return throw_wx(env, s, rc, "wx_pool_flags");
  return a[1];
}
static napi_value GroupSync(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  wx_gslot *gs = get_group(env, a[0]);
  if (!gs) return NULL;
  int rc = L.group_sync(gs->g);
  if (rc) return throw_group(env, gs->g, rc, "wx_group_sync");
  return NULL;
}
static napi_value GroupDestroy(napi_env env, napi_callback_info info)
{
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  void *p = NULL;
  if (napi_get_value_external(env, a[0], &p) == napi_ok && p && ((wx_gslot *)p)->g) {
    orphan_slabs((wx_gslot *)p);
    L.group_destroy(((wx_gslot *)p)->g);
    ((wx_gslot *)p)->g = NULL;
  }
  return NULL;
}

static napi_value AbiVersion(napi_env env, napi_callback_info info)
{
  napi_value r;
  NAPI_CALL(env, napi_create_int32(env, L.abi_version(), &r));
  return r;
}

static int load_lib(napi_env env)
{
  const char *path = getenv("WXSIM_LIB");
  char buf[4096];
  if (!path) {
    Dl_info di;
    if (dladdr((void *)&load_lib, &di) && di.dli_fname) {
      snprintf(buf, sizeof(buf), "%s", di.dli_fname);
      char *slash = strrchr(buf, '/');
      if (slash) *slash = 0;
      strncat(buf, "/../2d-weather-sandbox_amd/csrc/libwxsim.so", sizeof(buf) - strlen(buf) - 1);
      path = buf;
    }
  }
  L.dl = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!L.dl) {
    char msg[4600];
    snprintf(msg, sizeof(msg), "cannot load libwxsim.so (%s): %s -- there is no CPU fallback", path ? path : "?", dlerror());
    napi_throw_error(env, NULL, msg);
    return -1;
  }
#define SYM(field, name)                                          \
  *(void **)&L.field = dlsym(L.dl, name);                         \
  if (!L.field) {                                                 \
    napi_throw_error(env, NULL, "libwxsim.so lacks symbol " name); \
    return -1;                                                    \
  }
  SYM(create, "wx_create") SYM(destroy, "wx_destroy") SYM(last_error, "wx_last_error") SYM(abi_version, "wx_abi_version")
  SYM(upload, "wx_upload") SYM(set_params, "wx_set_params") SYM(step, "wx_step") SYM(sync, "wx_sync") SYM(get_iter, "wx_get_iter")
  SYM(set_iter, "wx_set_iter") SYM(read_rect, "wx_read_rect") SYM(read_particles, "wx_read_particles")
  SYM(setup_columns, "wx_setup_columns") SYM(setup_terrain, "wx_setup_terrain") SYM(init_droplets, "wx_init_droplets") SYM(stream_bytes, "wx_stream_bytes") SYM(host_alloc, "wx_host_alloc") SYM(host_free, "wx_host_free")
  SYM(stream_frame, "wx_stream_frame") SYM(stream_wait, "wx_stream_wait") SYM(set_option, "wx_set_option") SYM(tune_placement, "wx_tune_placement")
  SYM(group_create, "wx_group_create") SYM(group_destroy, "wx_group_destroy") SYM(group_last_error, "wx_group_last_error") SYM(group_count, "wx_group_count")
  SYM(group_transport, "wx_group_transport") SYM(group_slab, "wx_group_slab") SYM(group_agree, "wx_group_agree") SYM(group_step, "wx_group_step")
  SYM(group_sync, "wx_group_sync") SYM(local_width, "wx_local_width") SYM(group_set_option, "wx_group_set_option") SYM(group_exchange, "wx_group_exchange")
  SYM(pool_flags, "wx_pool_flags") SYM(placement_info, "wx_placement_info") SYM(pair_stats, "wx_pair_stats")
#undef SYM
  return 0;
}

static napi_value Init(napi_env env, napi_value exports)
{
  if (load_lib(env)) return NULL;
  napi_property_descriptor d[] = {
    {"create", 0, Create, 0, 0, 0, napi_default, 0},       {"destroy", 0, Destroy, 0, 0, 0, napi_default, 0},
    {"upload", 0, Upload, 0, 0, 0, napi_default, 0},       {"setParams", 0, SetParams, 0, 0, 0, napi_default, 0},
    {"step", 0, Step, 0, 0, 0, napi_default, 0},           {"sync", 0, Sync, 0, 0, 0, napi_default, 0},
    {"getIter", 0, GetIter, 0, 0, 0, napi_default, 0},     {"setIter", 0, SetIter, 0, 0, 0, napi_default, 0},
    {"readRect", 0, ReadRect, 0, 0, 0, napi_default, 0},   {"readParticles", 0, ReadParticles, 0, 0, 0, napi_default, 0},
    {"abiVersion", 0, AbiVersion, 0, 0, 0, napi_default, 0}, {"setupColumns", 0, SetupColumns, 0, 0, 0, napi_default, 0}, {"setupTerrain", 0, SetupTerrain, 0, 0, 0, napi_default, 0}, {"initDroplets", 0, InitDroplets, 0, 0, 0, napi_default, 0},
    {"streamFrame", 0, StreamFrame, 0, 0, 0, napi_default, 0}, {"streamWait", 0, StreamWait, 0, 0, 0, napi_default, 0},
    {"setOption", 0, SetOption, 0, 0, 0, napi_default, 0},   {"tunePlacement", 0, TunePlacement, 0, 0, 0, napi_default, 0},
    {"placementInfo", 0, PlacementInfo, 0, 0, 0, napi_default, 0}, {"pairStats", 0, PairStats, 0, 0, 0, napi_default, 0},
    {"groupCreate", 0, GroupCreate, 0, 0, 0, napi_default, 0}, {"groupSlab", 0, GroupSlab, 0, 0, 0, napi_default, 0},
    {"groupInfo", 0, GroupInfo, 0, 0, 0, napi_default, 0},     {"groupAgree", 0, GroupAgree, 0, 0, 0, napi_default, 0},
    {"groupStep", 0, GroupStep, 0, 0, 0, napi_default, 0},     {"groupSync", 0, GroupSync, 0, 0, 0, napi_default, 0},
    {"groupDestroy", 0, GroupDestroy, 0, 0, 0, napi_default, 0}, {"groupSetOption", 0, GroupSetOption, 0, 0, 0, napi_default, 0},
    {"groupExchange", 0, GroupExchange, 0, 0, 0, napi_default, 0}, {"poolFlags", 0, PoolFlags, 0, 0, 0, napi_default, 0},
  };
  NAPI_CALL(env, napi_define_properties(env, exports, sizeof(d) / sizeof(d[0]), d));
  return exports;
}

NAPI_MODULE(wxsim_napi, Init)
