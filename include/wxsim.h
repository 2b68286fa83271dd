/*
 * wxsim.h -- C ABI of the MI355X-native simulation engine (libwxsim.so).
 *
 * Drop-in boundary for the reference's "simulation step + field readback" seam
 * (niels747/2D-Weather-Sandbox has no FFI/plugin API: the seam is draw()'s simulation block and the
 * gl.readPixels / gl.getBufferSubData call sites, all talking to module-scope WebGL objects).
 * Every entry point cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * WX_E_* code, the message is available from wx_last_error(); no exceptions cross the ABI.
 * One caller thread per handle (the reference is single-threaded JS issuing an in-order GL command
 * stream). All grids are row-major with y = 0 at the BOTTOM, x fastest, 4 interleaved channels per
 * cell -- byte-identical to the reference's textures, readPixels results and save files
 * (app.js:1297-1312, 6586-6593).
 */
#ifndef WXSIM_H
#define WXSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WX_ABI_VERSION 11

/* error codes */
#define WX_OK 0
#define WX_E_INVALID (-1)   /* bad argument */
#define WX_E_DEVICE (-2)    /* HIP runtime error (message in wx_last_error) */
#define WX_E_NOMEM (-3)
#define WX_E_RANGE (-4)     /* rectangle / particle range outside the grid (no wrap, as readPixels) */
#define WX_E_STATE (-5)     /* call sequence error (e.g. step before upload) */

/* Uniform values of the simulation programs (what gl.uniform* pushes):
 *   velocityShader.frag:14-16, boundaryShader.frag:22-38, advectionShader.frag:21-43,
 *   lightingShader.frag:22-29, precipitationShader.vert:31-48; host side app.js:3401-3443
 *   (setGuiUniforms), app.js:5479-5635 (constants), app.js:6557-6561 (sun), app.js:5804-5808 (brush),
 *   app.js:3335 (airplane). All values are fp32 exactly as the GL uniform would hold them. */
typedef struct wx_params {
  float dragMultiplier, wind;
  float vorticity, landEvaporation, waterEvaporation, dynamicWaterTemperature;
  float evapHeat, waterWeight;
  float sunAngle;  /* solar zenith angle, rad (app.js:6538-6539) */
  float dryLapse;  /* simHeight * dryLapseRate / 1000 (app.js:5439) */
  float meltingHeat, condensationRate, globalDrying, globalHeating, soundingForcing;
  float globalEffectsStartAlt, globalEffectsEndAlt; /* normalised by simHeight (app.js:3425-3426) */
  float waterTemperature;                           /* K (app.js:3427) */
  float sunIntensity;                               /* W/m2 (app.js:6550) */
  float greenhouseGases, waterGreenHouseEffect, IR_rate;
  float aboveZeroThreshold, subZeroThreshold, spawnChanceMult, snowDensity, fallSpeed;
  float growthRate0C, growthRate_30C, freezingRate, meltingRate, evapRate;
  float inactiveDroplets; /* < 0: keep the engine's own measured value (app.js:5957-5966) */
  float userInputValues[4]; /* xpos ypos intensity brushSize (advectionShader.frag:21) */
  float userInputMove[2];
  int32_t userInputType;    /* -1 none (app.js:5749) */
  int32_t wrapHorizontally;
  float airplaneValues[4];
  int32_t enablePrecipitation; /* guiControls.enablePrecipitation (app.js:5936) */
  int32_t quad_scale;          /* 0: fragCoord = (x+.5, y+.5) exactly (the reference author's intent,
                                  app.js:4766-4769); 1: model the *1.0000001 quad UV scale */
  uint32_t pass_mask;          /* WX_PASS_* bits; WX_PASS_ALL = the reference loop */
} wx_params;

#define WX_PASS_VELOCITY 1u
#define WX_PASS_VORTICITY 2u   /* curl + vorticity */
#define WX_PASS_BOUNDARY 4u
#define WX_PASS_ADVECTION 8u
#define WX_PASS_PRESSURE 16u
#define WX_PASS_LIGHTING 32u
#define WX_PASS_PRECIPITATION 64u
#define WX_PASS_ALL 0x7Fu
#define WX_PASS_DRY (WX_PASS_VELOCITY | WX_PASS_ADVECTION | WX_PASS_PRESSURE)

/* Fields for wx_read_rect / wx_device_ptr. "FB0"/"FB1" are the reference's frameBuff_0 / frameBuff_1
 * (app.js:5243-5252); which one a consumer reads is listed in SURVEY.md section 3.5. */
enum {
  WX_FIELD_BASE_CUR = 0,   /* baseTexture_0: post-pressure state (FB0; weather stations, save) */
  WX_FIELD_BASE_DISP = 1,  /* baseTexture_1: post-advection, pre-pressure (FB1; display, sounding) */
  WX_FIELD_WATER_0 = 2,    /* waterTexture_0: post-boundary (what a save stores, app.js:6587-6589) */
  WX_FIELD_WATER_CUR = 3,  /* waterTexture_1: post-advection (current) */
  WX_FIELD_WALL_CUR = 4,   /* wallTexture_0 */
  WX_FIELD_WALL_DISP = 5,  /* wallTexture_1 */
  WX_FIELD_LIGHT_0 = 6,    /* lightTexture_0 (the one boundaryShader samples, app.js:5868-5869) */
  WX_FIELD_LIGHT_1 = 7,
  WX_FIELD_CURL = 8,       /* R32F */
  WX_FIELD_VORT = 9,       /* RG32F vortForce (an intermediate: only the per-pass kernel set, WX_FUSED=0, stores it) */
  WX_FIELD_PRECIP_FB = 10, /* RGBA32F precipitationFeedbackTexture (stored with its three written channels; alpha -- 0, and the lightning
                            * request's fourth component at texel (1,0), precipitationShader.vert:135-139 -- is added when the field is read) */
  WX_FIELD_PRECIP_DEP = 11,/* RG32F precipitationDepositionTexture */
  WX_FIELD_LIGHTNING = 12, /* 1x1 RGBA32F lightningDataTexture */
  /* RGBA16F emittedLight: the lighting pass's second render target (app.js:838, 5283, 5294; lightingShader.frag:15, 60-78,
   * 98-101, 143-166), read only by the renderer's ambient-light pyramid (app.js:6096). Not stored per iteration: computed for
   * the requested rectangle when read, from what the most recent lighting pass sampled (zero before the first one). */
  WX_FIELD_EMITTED = 13,
  WX_FIELD_COUNT = 14
};

/* destination element types of wx_read_rect (WX_DTYPE_F16: IEEE binary16, WX_FIELD_EMITTED only) */
enum { WX_DTYPE_F32 = 0, WX_DTYPE_I8 = 1, WX_DTYPE_I32 = 2, WX_DTYPE_F16 = 3 };

typedef struct wx_sim wx_sim;

/* Replaces texture/FBO creation app.js:5149-5317 and particle buffers app.js:4891-5002.
 * Allocates both ping-pong copies of every field on the current HIP device. */
int wx_create(int X, int Y, int n_droplets, wx_sim **out);

/* Column-slab variant (no reference counterpart: the reference is single-GPU). The handle owns columns
 * [x0, x0 + X_owned) of a periodic domain X_global wide and stores `halo` ghost columns on each side
 * (local width X_owned + 2*halo; local column i is global column (x0 - halo + i) mod X_global). */
int wx_create_slab(int X_global, int Y, int x0, int X_owned, int halo, int n_droplets, wx_sim **out);
/* Dependency cone of one iteration, columns per side: `halo` ghost columns stay exact for halo / WX_SLAB_CONE
 * iterations, then the halo must be exchanged (wx_halo_pack / wx_halo_unpack).
 * With particles the cone is wider from the second iteration of a period on: a droplet deposits within a sprite radius (6 columns) of
 * where it is, and the next boundary pass feeds that deposit through advection and pressure (3 more) -- 9 columns per iteration, after 6
 * for the first (whose feedback texture came with the exchange). Iteration j (0-based) of a period is exact on the owned columns +
 * halo - 6 - 9j ghost columns, droplets are processed there, and the owned columns need a sprite radius of that left in the last
 * iteration: WX_SLAB_PERIOD_PARTICLES(halo) iterations per exchange (at most 15: the flip history is a 16-bit mask). (For flows of one
 * cell per iteration and more wx_slab_period() is the authority: the cone is 6 + floor|vx|, and the last iteration keeps cone + 1 columns
 * instead of 6 -- a droplet is tested where it starts from and deposits where it arrives.)
 * (Rounds 2-3 assumed 6 per iteration throughout, which the measured spread -- 6 to 7 columns -- satisfied in every test but the
 * worst case does not.) */
#define WX_SLAB_CONE 6
#define WX_SLAB_CONE_PARTICLES 9
#define WX_SLAB_PERIOD_PARTICLES(halo) ((halo) < 12 ? 0 : ((1 + ((halo) - 12) / WX_SLAB_CONE_PARTICLES) < 15 ? (1 + ((halo) - 12) / WX_SLAB_CONE_PARTICLES) : 15))
/* Slabs exact at ANY speed (round 5). The figures above hold for |vx| < 1 cell / iteration -- the shaders' documented range
 * (common.glsl:40-41), which the reference never enforces: advectionShader.frag:85-99 back-traces `fragCoord - vel` for any vel, so one
 * iteration's cone really is 6 + floor|vx| columns. The marching kernels therefore keep the largest |vx| they produce, and the HOSTS OF ALL
 * SLABS agree on a bound per exchange period:
 *   wx_slab_vx_take(s, &v)       the largest |vx| this slab has seen since the last take (0 below 0.5; synchronises; after an upload /
 *                                wx_device_ptr(BASE_CUR) the state itself is scanned first)
 *   wx_slab_set_vx_bound(s, v)   v = the maximum over ALL slabs: the coming period assumes |vx| < 1.25 v + 0.25 (a flow measured at v may
 *                                have accelerated by the time the period is over), i.e. cone = 6 + floor(1.25 v + 0.25) once that
 *                                reaches 1 -- WX_E_STATE if the halo is too thin for it
 *   wx_slab_cone(s), wx_slab_period(s)   ghost columns per iteration / iterations per exchange under the current bound (with particles:
 *                                cone for the first iteration, cone + 3 for every further one, a sprite radius left in the last)
 * A |vx| that reaches the bound inside a period is REPORTED by the next blocking call (WX_E_STATE), never silent; so is a velocity of
 * 2 * halo - 8 cells / iteration and more anywhere in the slab (beyond that even the strips three halo widths from the edges read ghost columns). wx_slab_step /
 * wx_group_step do all of this themselves without a host round trip inside a period: the maxima travel with the exchange (one word per
 * slab, all-gathered) and size the period after the next; slab.py's host-driven exchange all-reduces them. Hosts that drive
 * wx_step_overlap / wx_halo_* themselves call the three functions once per period. */
int wx_slab_vx_take(wx_sim *s, float *vmax);
int wx_slab_set_vx_bound(wx_sim *s, float v_measured);
int wx_slab_cone(const wx_sim *s);
int wx_slab_period(const wx_sim *s);

void wx_destroy(wx_sim *s);
const char *wx_last_error(const wx_sim *s); /* also valid with s == NULL for create failures */
int wx_abi_version(void);
/* (ABI 11) Which arithmetic this library was built with. WX_ARITH_EXACT (libwxsim.so, the default and the one every parity claim is about):
 * no FMA contraction, correctly rounded division and sqrt -- bit-identical to the CPU oracle. WX_ARITH_FAST (libwxsim_fast.so, `make -C
 * csrc fast`; SURVEY.md Appendix A's opt-in tolerance build): contraction allowed, 1-ulp hardware reciprocal / sqrt -- the freedom a GL
 * driver takes with the reference's shaders (common.glsl:177-180, advectionShader.frag:111-131); masks stay bit-exact, float fields stay
 * inside the calibrated rounding envelope of the exact build (tests/test_fast_arith.py). A host picks one by the file it loads. */
#define WX_ARITH_EXACT 0
#define WX_ARITH_FAST 1
int wx_arith(void);

/* Replaces setupTextures() app.js:5189-5234 (the same data goes into BOTH _0 and _1) and
 * setupPrecipitationBuffers() app.js:4915-5002. Host arrays are copied; the caller keeps ownership.
 * Resets even=true, light/curl/vort/feedback/lightning = 0; iterNum is NOT reset (the reference keeps
 * it across KeyL reloads, app.js:4628-4640). Arrays cover the handle's LOCAL width. drops may be NULL. */
int wx_upload(wx_sim *s, const float *base, const float *water, const int8_t *wall, const float *drops);

/* Device-side initialiser of a new simulation: replaces the setup draw (shaders/fragment/setupShader.frag:36-92,
 * app.js `setupTextures` path for a new simulation) WITHOUT three X*Y host arrays. The caller passes the 1-D part of the
 * generator -- per local column: number of wall rows (0..Y), sea (1) / land (0), the vegetation noise term
 * `noise(x*0.01 + rand(seed)*10) * 150` (double) and the snow height; per row: air temperature, total and cloud water of
 * the initial sounding (setupShader.frag:78-89) -- and the textures are filled on the device. Equivalent to wx_upload of
 * the arrays that weather_sandbox_amd.synth.terrain_grid builds from the same descriptors (tested bit for bit); resets
 * the same state as wx_upload. */
int wx_setup_columns(wx_sim *s, const int32_t *wall_rows, const uint8_t *sea_column, const double *veg_noise, const float *snow,
                     const float *T_air, const float *total_water, const float *cloud_water, const float *drops);

/* The same with the 1-D part generated on the device too (k_terrain_columns: rand / noise of setupShader.frag:26-33, the octave sum of
 * :52-59, vegetation noise :72, snow :74 -- in double, operation for operation what the host generators evaluate, so the result equals
 * wx_setup_columns of weather_sandbox_amd.synth.terrain_columns(X, Y, seed=, height_mult=, snap=) unless the device's sin() differs in
 * the last bit at a column whose height sits on a row boundary). seed / height_mult: the shader's uniforms (app.js: Math.random() and the
 * new-simulation dialog); snap: terrain constant over `snap` columns and an even number of rows thick (1 = the shader's raw terrain);
 * sim_height: simHeight [m]; the three per-row arrays (Y entries) are the initial sounding as for wx_setup_columns. A slab handle
 * generates its own local columns (global column (x0 - halo + i) mod X_global): no host array scales with the grid. */
int wx_setup_terrain(wx_sim *s, double seed, double height_mult, int snap, double sim_height, const float *T_air, const float *total_water,
                     const float *cloud_water, const float *drops);

/* initRainDrops() (app.js:4901-4913) on the device: a fresh pool of inactive droplets whose five fields are random seeds, (r, r, -10 + r,
 * r, r) with r in [0, 1). Field c of droplet i = 24 bits of hash(seed + hash(5 i + c)) with the shaders' integer hash (common.glsl:103-111):
 * a pure function of the seed, so every slab of a decomposed domain generates the same pool (the partitioned pool wants the WHOLE pool
 * on every slab) and no host array scales with the droplet count. The grid is untouched; the pool bookkeeping of slab handles is reset
 * as after wx_upload. */
int wx_init_droplets(wx_sim *s, uint32_t seed);

/* Replaces the uniform pushes (see wx_params) plus the `initial_Tv` / realWorldSounding_* arrays
 * (app.js:5444-5474, 5485-5537). initial_T has Y+1 entries; sounding arrays Y+1 entries or NULL (= 0).
 * Takes effect at the next wx_step. */
int wx_set_params(wx_sim *s, const wx_params *p, const float *initial_T, const float *sounding_T,
                  const float *sounding_W, const float *sounding_Vel);

/* Replaces the loop body app.js:5830-6005: n_iter iterations, enqueued asynchronously on the handle's
 * stream; iterNum++ per iteration; the 600-iteration inactive-droplet count (app.js:5957-5966) is
 * refreshed on the device without a host round trip. */
int wx_step(wx_sim *s, int n_iter);
int wx_sync(wx_sim *s);

/* Engine options (no reference counterpart).
 * WX_OPT_SPLAT_ORDER: how the additive particle splats (gl.blendFunc(ONE, ONE) point sprites, app.js:5940-5953) are summed.
 *   0 (default): fp32 atomics in arrival order -- like the reference's blend unit the result depends on the order;
 *   1: deterministic -- every droplet records its deposit, the records are sorted by anchor texel (stable radix sort) and each
 *      texel's deposits are added in droplet-index order, so the feedback / deposition textures (and everything downstream)
 *      are a pure function of the state. Slower; meant for tests that compare coupled particle runs bit for bit.
 * WX_OPT_CHECK_LAUNCHES: 1 = synchronise after every kernel launch of wx_step and report a fault with the kernel's name
 *   (debugging; the launch status itself is always checked).
 * WX_OPT_KERNEL_SET: 1 (default) = the row-marching single-kernel iteration; 0 = one kernel per reference pass -- the independent
 *   cross-check every parity test also runs. WX_OPT_DRY_KERNEL: the water-free dry iteration as the row-marching kernel (1, default)
 *   or the LDS-tiled one (0). WX_OPT_ROW_BANDS: launch shape of the marching wet kernel -- 0 column blocks per XCD, 1 (default) row
 *   bands per XCD on grids at least 512 rows high, 2 row bands on any grid (tests). WX_OPT_FIX_CAP: entries of the exact-path cell
 *   list (default: a quarter of the grid's cells; tests provoke the overflow report), before the first step.
 * With s == NULL an option becomes the process-wide default of handles created afterwards (the library itself reads NO environment
 * variable; tuning switches exist only in -DWX_DEBUG builds). */
#define WX_OPT_SPLAT_ORDER 1
#define WX_OPT_CHECK_LAUNCHES 2
#define WX_OPT_KERNEL_SET 3
#define WX_OPT_DRY_KERNEL 4
#define WX_OPT_ROW_BANDS 5
#define WX_OPT_FIX_CAP 6
/* WX_OPT_POOL_EXACT (slab handles with particles): 1 = the partitioned droplet pool reproduces the undecomposed run EXACTLY. The
 *   default protocol exchanges nothing inside an exchange period, so a droplet that retires is not probed for re-spawning by the other
 *   ranks until the next exchange, a droplet can spawn a second time from a stale record, and the lightning state / inactive count are
 *   a period late. In exact mode the hosts run ONE iteration per wx_step and follow it with wx_pool_events_pack -> all-gather ->
 *   wx_pool_events_apply: the buffers then also carry every rank's lightning request (summed over the ranks before the accept test of
 *   lightningLocationShader.frag:24-38, so two requests in one iteration cancel as in the reference) and, every 600 iterations, what the
 *   `inactiveDroplets` refresh needs (app.js:5957-5966). Grid halos and edge droplets still travel once per period. With
 *   WX_OPT_SPLAT_ORDER 1 the slabs are then bit-identical to one handle (SURVEY 8e's determinism check); the price is a latency-bound
 *   all-gather of a few KB per iteration. */
#define WX_OPT_POOL_EXACT 7
/* WX_OPT_EXCHANGE_OVERLAP (slab handles driven by the library's transport: wx_slab_step / wx_group_step): 1 (default) = pack, transfers
 *   and unpack run on a side stream of the handle and hide behind the interior strips of the neighbouring iterations (grid-only: the
 *   iteration before AND after; with particles the iteration after, because precipitation needs its whole iteration and the exchange
 *   needs precipitation's feedback texture); 0 = everything in order on the compute stream -- same results bit for bit (tests), and
 *   what the exact particle mode always does. */
#define WX_OPT_EXCHANGE_OVERLAP 8
/* WX_OPT_SPLIT_LAUNCH (slab handles with a comm stream): how wx_step_overlap runs a split iteration. 0 (DEFAULT) = the edge strips and the
 *   interior as two launch groups on two streams joined by events (rounds 2-4: measured faster, profiles/r05_slab_protocol_cost.txt).
 *   1 (experimental, round 5) = ONE launch over all strips whose dispatch order puts the edge strips first / last, with device-side
 *   hand-offs: the edge strips report on a device word that a one-wave gate kernel on the comm stream polls before the halo is packed, and
 *   poll an epoch word the comm stream bumps behind the unpack -- no second stream, no join events on the compute stream. Same results. */
#define WX_OPT_SPLIT_LAUNCH 9
/* WX_OPT_DRY_PAIRS (round 5; default 1): the water-free dry stencil (pass_mask WX_PASS_DRY, no water anywhere, no brush, wall texture
 *   constant) runs TWO iterations per launch wherever two are left in a wx_step call and neither is a split iteration -- the second
 *   iteration's input never leaves the wavefront (csrc/wx_march2.h): 18 instead of 36 bytes per cell-step, 0.71 instead of 0.86 ms per
 *   iteration at 32768 x 4096. Same results bit for bit: cells whose back-trace is 0.9 cells or more (no exact path inside the march) are
 *   handled by tiles -- first-iteration ones taint what depends on them (NaN), second-iteration cells that are fast or tainted put their
 *   8 x 8 tile on a list, and a small fix kernel behind the pair recomputes both iterations for those tiles from the pair's untouched
 *   inputs (round 6; wx_pair_stats); only a back-trace of three cells or more -- or an overflowing list -- makes one further launch repeat
 *   the whole pair with the one-iteration stencil. 0 = one iteration per launch. */
#define WX_OPT_DRY_PAIRS 10
/* (ABI 10) 1 (default): waterTexture_0 -- the post-boundary water of a step's last iteration, which only saves read (app.js:6587-6589) --
 * is made when somebody asks for WX_FIELD_WATER_0 (per-pass kernels on the retained inputs of that iteration, with its parameters)
 * instead of being stored by every frame's last iteration: 16 of the 36 display-side bytes per cell. Applies to the marching wet kernel
 * without particles; 0 = always stored by the iteration itself. Bit-identical either way (tests/test_gpu_parity.py). */
#define WX_OPT_WATER0_ON_DEMAND 11
/* (ABI 11) WX_OPT_PLACEMENT_SEARCH: the number of further allocation sets the handle's ONE placement search tries (default 6; 0 = never).
 * A whole-domain handle of WX_PLACEMENT_AUTO_CELLS cells or more runs wx_tune_placement(s, tries, 20, ..) by itself inside its first
 * wx_step (a quarter of a second at 16384 x 2048; state, counters and results untouched; skipped silently if the device has no room for
 * two more copies of the state) unless the host has called wx_tune_placement before -- see "Placement tuning" below. Device pointers
 * taken before that first step are invalid after it. Slab handles never search by themselves (their hosts call wx_tune_placement). With
 * s == NULL: the default of handles created afterwards (tests switch the search off). */
#define WX_OPT_PLACEMENT_SEARCH 12
#define WX_PLACEMENT_AUTO_CELLS (8u << 20)
int wx_set_option(wx_sim *s, int option, int value);

/* iterNum global (app.js:440) */
int64_t wx_get_iter(const wx_sim *s);
int wx_set_iter(wx_sim *s, int64_t iter);

/* Replaces every gl.readPixels of SURVEY.md section 3.5 (app.js:1084-1092, 3931-3943, 3020-3065,
 * 1840-1906, 4215-4237, 4347-4357, 5958-5961, 5986-5988, 6584-6593). Rows bottom-up, no wrap
 * (WX_E_RANGE outside the local grid); float fields accept WX_DTYPE_F32, wall fields WX_DTYPE_I8 or
 * WX_DTYPE_I32 (both are used by the reference: app.js:6593 vs 3943), WX_FIELD_EMITTED WX_DTYPE_F16 (the texture's own
 * format) or WX_DTYPE_F32 (what readPixels(..., gl.FLOAT) of a half-float attachment returns). Synchronises the stream. */
int wx_read_rect(wx_sim *s, int field, int x, int y, int w, int h, void *dst, int dtype);

/* Replaces gl.getBufferSubData on the transform-feedback buffers (app.js:5019, 5086, 6597):
 * 5 floats per droplet (pos.xy, mass.xy, density) from the destination buffer of the last step. */
int wx_read_particles(wx_sim *s, int first, int count, float *dst);

/* Field streaming for a display consumer: what the reference's renderer binds every frame (app.js:6081-6219)
 * -- BASE_DISP, WATER_CUR, WALL_DISP (int8), LIGHT_0, CURL, PRECIP_FB, EMITTED (binary16) of the viewport rect, in this order,
 * each field contiguous (w*h texels, rows bottom-up, no wrap) -- copied asynchronously into ONE host buffer of wx_stream_bytes(w, h)
 * bytes, ideally pinned (wx_host_alloc). wx_stream_frame returns at once: the copies run on the handle's own copy
 * stream after everything enqueued so far; later wx_step calls are ordered after them on the device, so the host never
 * blocks; wx_stream_wait blocks until the frame is complete in host memory. One frame in flight per handle. */
size_t wx_stream_bytes(int w, int h);
void *wx_host_alloc(size_t bytes); /* pinned host memory; NULL on failure */
void wx_host_free(void *p);
int wx_stream_frame(wx_sim *s, int x, int y, int w, int h, void *host_dst);
int wx_stream_wait(wx_sim *s);

/* Placement tuning (no reference counterpart). Where a handle's planes lie in physical device memory decides how the ~13 streams of
 * the marching kernels spread over the HBM channels: the same binary runs the same iteration in 0.72 .. 0.87 ms depending on the
 * allocations (profiles/r03_alloc_probe.txt). wx_tune_placement times the handle's own iteration (current parameters and
 * pass mask; 4 untimed + iters_per_try timed iterations) on the allocations it has and on up to `tries` further sets that receive a
 * copy of the state, keeps the fastest as the handle's storage and restores the state from a backup taken at the start: state,
 * iteration counter and all fields are unchanged;
 * device pointers obtained from wx_device_ptr before the call are invalid afterwards. ms_before / ms_after (may be NULL): the
 * iteration time on the allocation the handle had and on the winner. Needs three times the handle's memory while it runs (more is used
 * if free: rejected candidates are kept until the end so that the allocator does not hand the same memory out again). */
int wx_tune_placement(wx_sim *s, int tries, int iters_per_try, float *ms_before, float *ms_after);
/* (ABI 11) Returns 1 and the two iteration times [ms] of the handle's placement search -- the host's call above or the implicit one of
 * the first wx_step (WX_OPT_PLACEMENT_SEARCH) -- if one has run, else 0. Either pointer may be NULL. */
int wx_placement_info(const wx_sim *s, float *ms_first, float *ms_kept);

/* ---- plumbing for hosts that own device memory / streams (PyTorch, multi-GPU halo exchange) ---- */
int wx_set_stream(wx_sim *s, void *hip_stream);     /* NULL = legacy default stream */
void *wx_device_ptr(wx_sim *s, int field);          /* device address of a field's current storage; WX_FIELD_LIGHT_0/1, WX_FIELD_EMITTED,
                                                     * WX_FIELD_PRECIP_FB and (after the marching wet kernel, round 6) WX_FIELD_BASE_DISP are
                                                     * stored in another form (planes / on demand / three channels / post-advection P only): the
                                                     * pointer is to the RGBA texture made at the time of the call, valid until the next wx_step.
                                                     * Slabs: velocities written through WX_FIELD_BASE_CUR are looked at by the next exchange
                                                     * (a |vx| beyond the current period's bound is REPORTED, WX_E_STATE); only wx_upload /
                                                     * wx_setup_* re-size the first period -- and on the slabs of an initialised ring those are
                                                     * COLLECTIVE: every rank calls them alike before the next wx_slab_step / wx_exchange (the
                                                     * ranks all-gather their |vx| once after an upload: a rank that skipped it would pair the
                                                     * ring's later collectives off by one) */
int wx_local_width(const wx_sim *s);                /* X_owned + 2*halo */
/* Halo exchange of the state carried across iterations (base_0, wall_0, water_1, both light textures; with particles also
 * the feedback and deposition textures):
 * pack the `halo` outermost OWNED columns of one side into a contiguous device buffer / unpack a
 * neighbour's buffer into this handle's ghost columns. side: 0 = left (low x), 1 = right. */
size_t wx_halo_bytes(const wx_sim *s);
/* (ABI 10) How much of such a buffer a message of the CURRENT period occupies, from its first byte: wx_halo_bytes normally; between slabs
 * that agreed on the water-free dry stencil (wx_slab_assert_water_free(s, 1) below) the base texture alone -- 16 of the 68 bytes per cell:
 * that iteration writes nothing else, so the ghost columns of water, wall and light stay what the upload made them. A host transport
 * sends / receives this many bytes of the buffers it packs / unpacks (buffers are still sized by wx_halo_bytes). The value changes only
 * through calls every rank makes alike -- wx_slab_assert_water_free, or a wx_step whose parameters leave the water-free dry stencil --, so
 * neighbours never disagree about a message's size; a slab that was given new contents in between is refused by wx_halo_pack
 * (WX_E_STATE) until the slabs have agreed again. Such slabs also run their periods in order (iterations in pairs, none split:
 * wx_step_overlap's flags are ignored), which is faster there than the overlap (profiles/r05_dry_slab_inorder.txt). */
size_t wx_halo_message_bytes(const wx_sim *s);
int wx_halo_pack(wx_sim *s, int side, void *dev_buf);
int wx_halo_unpack(wx_sim *s, int side, const void *dev_buf);
/* both sides in ONE launch each (ABI 8): next to a marching kernel that holds every wave slot of the chip a second small launch queues
 * behind thousands of workgroups (profiles/r04_slab_protocol_cost.txt); dev_left / dev_right as side 0 / 1 above */
int wx_halo_pack_both(wx_sim *s, void *dev_left, void *dev_right);
int wx_halo_unpack_both(wx_sim *s, const void *dev_left, const void *dev_right);
/* Overlap of the halo exchange with compute (no reference counterpart; BASELINE north_star: "halo exchange ... overlapped on a
 * side HIP stream"). After wx_set_comm_stream(s, stream) the pack / unpack kernels run on `stream` -- the stream the host also
 * issues its send / recv on -- fenced against the handle's compute stream by events inside the library:
 *   wx_step_overlap(s, n, WX_OVERLAP_EDGES_FIRST): in the LAST iteration of the call the edge strips (every column
 *     wx_halo_pack reads) are launched first and an event is recorded behind them; wx_halo_pack waits for that event only,
 *     so packing and sending run while the interior strips of that iteration still compute;
 *   wx_step_overlap(s, n, WX_OVERLAP_EDGES_LAST): in the FIRST iteration the interior strips are launched first; the compute
 *     stream then waits for the event wx_halo_unpack recorded on the comm stream and launches the edge strips, the only ones
 *     that read ghost columns.
 * Both flags may be combined. The split needs a row-marching kernel (default kernel set; all grid passes on, or the water-free
 * dry iteration; particles off) and a slab wide enough to have interior strips; otherwise the call degrades to the in-order exchange: wx_halo_pack waits
 * for everything enqueued on the compute stream, the next wx_step waits for the unpack. wx_step(s, n) == wx_step_overlap(s, n, 0). */
/* The water-free dry iteration on slabs. pass_mask == WX_PASS_DRY runs a kernel that does not touch the water texture at all
 * (36 B/cell) while the handle KNOWS it is trivial (0 in air, only the wall marker in walls: established by wx_upload, dropped by
 * anything that can create water). A slab also receives its neighbours' ghost columns, so it relies on that only after the host
 * has established it for every slab: wx_water_free(s) reports what the last wx_upload found for THIS handle, the host combines
 * the answers of all ranks (slab.py: all-reduce MIN) and passes the result to wx_slab_assert_water_free on every handle. A
 * wrong assertion is never silent: the slab whose own contents contradict it refuses its next wx_halo_pack (WX_E_STATE; ABI 10 --
 * agreed slabs exchange the base texture alone, wx_halo_message_bytes), and handles that carry particles, which keep the full
 * message, validate the arriving water on the device at every wx_halo_unpack (reported by the next blocking call). Without the
 * assertion a slab handle runs the water-carrying dry kernel. */
int wx_water_free(const wx_sim *s);
int wx_slab_assert_water_free(wx_sim *s, int agreed);

#define WX_OVERLAP_EDGES_FIRST 1u
#define WX_OVERLAP_EDGES_LAST 2u
/* (ABI 10) This call is one piece of a longer step -- another wx_step / wx_step_overlap follows before anything reads a display-side
 * field (WX_FIELD_BASE_DISP, _WATER_0, _CURL, the emitted-light image): its last iteration does not store them. A slab host cuts
 * a frame of 10 iterations into exchange periods of 6 or 7; without the flag every piece ends with an iteration that stores 36 B per
 * cell nobody looks at -- 4 of 30 iterations, +4 % on the metric's slab (profiles/r05_slab_protocol_cost.txt). wx_slab_step /
 * wx_group_step set it themselves for every piece but the last. */
#define WX_OVERLAP_MORE_TO_COME 4u
int wx_set_comm_stream(wx_sim *s, void *hip_stream); /* NULL: pack / unpack on the compute stream again */
int wx_step_overlap(wx_sim *s, int n_iter, unsigned flags);

/* Particles on slabs (n_droplets > 0 with halo > 0; halo and X_owned multiples of 64): the PARTITIONED droplet pool.
 * wx_upload hands every rank the whole pool once; from then on an active droplet is tracked by the rank whose owned columns contain
 * it (and, as a ghost copy, by the neighbour while it is within `halo` columns of the common edge); the other ranks only know
 * "active elsewhere" and skip it. Inactive droplets are static records that every rank holds; every rank tests every inactive
 * record's hashed spawn probe (precipitationShader.vert:82-84: anywhere in the domain) against its own columns and acts on the ones
 * that land where its grid is valid. Nothing is communicated inside an exchange period (WX_SLAB_PERIOD_PARTICLES(halo) iterations, at
 * most 15). At the halo exchange (all buffers are DEVICE pointers; calls are enqueued on the handle's stream):
 *   wx_pool_events_pack  -> all-gather of the buffers -> wx_pool_events_apply(gathered, n_ranks, stride_bytes):
 *       a buffer = 16-byte header (first int32: number of events) + 32-byte events, wx_pool_event_bytes() in all (room for every
 *       droplet: the start-up burst of an all-inactive pool flips most of them in one period); the hosts normally gather only the
 *       filled part -- stride_bytes = bytes per rank in `gathered` (0 = whole buffers), the same on every rank, >= the largest count;
 *       the droplets whose active / inactive status flipped (spawned, evaporated, deposited) with their final records; per droplet
 *       the report with the earliest first flip wins, then the longest flip history, then the rank that processed it last -- a rank
 *       that spawned a droplet from a stale inactive record after another rank had (it cannot know inside a period) drops its phantom;
 *   wx_pool_edges_pack(left, right, refresh_inactive) -> send / recv with the ring neighbours (same batch as the grid halos)
 *       -> wx_pool_edges_apply(buffer received from either neighbour): active droplets that left the owned columns are handed over,
 *       the ones within `halo` columns of an edge become the neighbour's ghost copies. refresh_inactive != 0 also refreshes the
 *       `inactiveDroplets` uniform (app.js:5957-5966) from the inactive records -- every rank holds all of them: no collective;
 *   lightning state: wx_lightning_get -> pick the latest strike -> wx_lightning_set;  then wx_slab_period_begin.
 * Buffer overflows (more status flips / edge droplets than the fixed capacities) are reported by the next blocking call (WX_E_STATE).
 * wx_read_particles returns the LOCAL view of the pool; wx_pool_flags says per droplet what it is worth: 0 = tracked by another rank
 * (stale here), 1 = inactive (the same record on every rank), 2 = active inside this rank's owned columns (THE record), 3 = ghost copy. */
int wx_slab_set_rank(wx_sim *s, int rank);
int wx_slab_period_begin(wx_sim *s);
size_t wx_pool_event_bytes(const wx_sim *s);
size_t wx_pool_edge_bytes(const wx_sim *s);
int wx_pool_events_pack(wx_sim *s, void *dev_buf);
int wx_pool_events_apply(wx_sim *s, const void *dev_bufs, int n_ranks, size_t stride_bytes);
int wx_pool_edges_pack(wx_sim *s, void *dev_left, void *dev_right, int refresh_inactive);
int wx_pool_edges_apply(wx_sim *s, const void *dev_buf);
int wx_pool_flags(wx_sim *s, uint8_t *host_dst);
int wx_lightning_get(wx_sim *s, float out[4]);
int wx_lightning_set(wx_sim *s, const float in[4]);

/* ---- the halo exchange inside the library (no reference counterpart: the reference is single-GPU; BASELINE north_star: "halo
 * exchange on RCCL send/recv over xGMI overlapped on a side HIP stream; host code stays in JavaScript"). RCCL is bound at run time
 * (dlopen), so nothing here is needed to run on one GPU.
 *
 * One rank per process (what `bench.py --gpus N` runs under torchrun, which then only launches the ranks and carries the 128-byte
 * id from rank 0 to the others):
 *   wx_comm_unique_id(id)                  ncclGetUniqueId
 *   wx_comm_init(s, id, rank, world)       ncclCommInitRank on the handle's device; rank r owns columns [r, r + 1) * X_global / world
 *   wx_exchange(s)                         pack both edges -> ncclGroupStart; ncclSend x 2; ncclRecv x 2; ncclGroupEnd -> unpack both
 *                                          ghost strips, all enqueued on the handle's comm stream (one of the library's own unless
 *                                          wx_set_comm_stream named one); never blocks the host
 *   wx_slab_step(s, n)                     n iterations with one wx_exchange per wx_slab_period(s) iterations; the iteration before
 *                                          an exchange launches its edge strips first, the one after it its interior strips first
 *                                          (wx_step_overlap), so the transfer runs behind the interior of both. Replaces the loop
 *                                          body app.js:5830-6005 for one slab of a decomposed domain.
 * One process, N slabs (the Node host: `node host/sim_host.js --gpus N`): */
#define WX_UNIQUE_ID_BYTES 128
int wx_comm_unique_id(void *id128);
int wx_comm_init(wx_sim *s, const void *id128, int rank, int world);
int wx_exchange(wx_sim *s);
int wx_slab_step(wx_sim *s, int n_iter);

typedef struct wx_group wx_group;
#define WX_TRANSPORT_AUTO 0  /* RCCL if every slab has a device of its own, else local */
#define WX_TRANSPORT_RCCL 1  /* ncclCommInitAll over the slabs' devices (RCCL refuses two ranks on one device) */
#define WX_TRANSPORT_LOCAL 2 /* device-to-device copies between the slabs' halo buffers, event-fenced: any number of slabs per device */
/* n_slabs handles of X_global / n_slabs columns each (+ `halo` ghost columns per side), slab i on devices[i] (NULL: i modulo the
 * device count), every slab with compute and comm streams of its own. Upload / set parameters / read through the per-slab handles
 * (wx_group_slab; local arrays incl. ghost columns, as for wx_create_slab), then wx_group_agree once, then step the group. */
int wx_group_create(int n_slabs, const int *devices, int X_global, int Y, int halo, int n_droplets, int transport, wx_group **out);
void wx_group_destroy(wx_group *g); /* destroys the slab handles too */
const char *wx_group_last_error(const wx_group *g);
int wx_group_count(const wx_group *g);
int wx_group_transport(const wx_group *g);
wx_sim *wx_group_slab(wx_group *g, int i);
int wx_group_agree(wx_group *g);            /* after the uploads: wx_water_free of every slab -> wx_slab_assert_water_free on all */
int wx_group_set_option(wx_group *g, int option, int value); /* wx_set_option on every slab (WX_OPT_SPLAT_ORDER, WX_OPT_POOL_EXACT ...) */
int wx_group_step(wx_group *g, int n_iter); /* like wx_slab_step, for all slabs; asynchronous */
int wx_group_exchange(wx_group *g);         /* an exchange now: afterwards every active droplet is owned by exactly one slab (read the pool then) */
int wx_group_sync(wx_group *g);
/* Slabs with particles (n_droplets > 0; halo and X_global / n_slabs multiples of 64) on the library's transport: wx_upload hands every
 * slab the WHOLE pool; wx_exchange / wx_slab_step / wx_group_step then also run the droplet-pool protocol above -- status flips (+ every
 * rank's lightning state) all-gathered with a stride every rank derives from the headers of the PREVIOUS period's rounds (the whole
 * event buffer at first, then 4 x the largest count seen, at least 65536 events: no host round trip inside a period; a burst beyond
 * that is reported as WX_E_STATE by the next blocking call), edge droplets in the same batch of transfers as the grid halos -- every
 * WX_SLAB_PERIOD_PARTICLES(halo) iterations, on the handle's side stream behind the interior strips of the next iteration
 * (WX_OPT_EXCHANGE_OVERLAP); with WX_OPT_POOL_EXACT in order, one iteration at a time, each followed by the all-gather of its flips and
 * lightning requests. */

/* The largest |velocity component| [cells / iteration] among the cells the marching wet kernel handed to its exact path (back-traces
 * of 0.9 cells and more) since the last call; 0 if there was none; NaN if a velocity was NaN. Resets the value; synchronises the
 * handle's stream. Whole-domain handles and slabs are exact at any speed (slabs size their exchange period by the |vx| they measure:
 * wx_slab_set_vx_bound above); the value is a diagnostic of the flow, nothing more. */
int wx_fastest_velocity(wx_sim *s, float *cells_per_iteration);

/* (ABI 11) The pair kernel's exact path (WX_OPT_DRY_PAIRS) since the last call: how many output cells k_dry2_fix recomputed (81 per
 * recorded 8 x 8 tile of cells with back-traces of 0.9 .. 3 cells in either iteration: one wavefront per tile) and how many pairs were repeated WHOLE with the
 * one-iteration kernel (a second-iteration back-trace of three cells or more, or more recorded cells than the list holds: 1/64 of the
 * grid's cells, 64 Ki .. 1 Mi tiles, WX_OPT_FIX_CAP). Either pointer may be NULL. Resets both; synchronises the handle's stream. The
 * reference has no velocity clamp (advectionShader.frag:85-99); results are bit-identical to one iteration per launch whatever these say. */
int wx_pair_stats(wx_sim *s, int64_t *cells_recomputed, int64_t *pairs_repeated);

/* Per-kernel device time from HIP events recorded on the handle's stream around every launch.
 * wx_profile(s, 1) starts collecting, wx_profile_read returns accumulated milliseconds and launch counts
 * for up to `cap` kernels (names via wx_kernel_name) and resets the accumulators. */
int wx_profile(wx_sim *s, int enable);
int wx_profile_read(wx_sim *s, int cap, float *ms, int *launches);
int wx_kernel_count(void);
const char *wx_kernel_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* WXSIM_H */
