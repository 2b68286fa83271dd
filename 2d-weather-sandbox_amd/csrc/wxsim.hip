// wxsim.hip -- libwxsim.so: the C ABI of include/wxsim.h on top of the HIP kernels (gfx950 only).
//
// Host side of the hot path: texture set, ping-pong bookkeeping and per-iteration launch sequence that
// restate the reference's draw() simulation block (app.js:5830-6005), plus the readback entry points.
// There is NO CPU fallback: every entry point that computes needs a HIP device and fails with
// WX_E_DEVICE otherwise.
#include "../../include/wxsim.h"
#include "wx_tile.h"
#include "wx_wet.h"
#include "wx_dry.h"
#include "wx_march.h"
#include "wx_march2.h"
#include "wx_kernels.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstring>
#include <utility>
#include <string>
#include <vector>

using namespace wx;

#ifndef WX_SPLIT_PRIO
#define WX_SPLIT_PRIO 0 // s_setprio level of the edge waves of a split iteration (StripOrder::prio)
#endif

namespace {

thread_local std::string g_create_error;

enum KernelId {
  K_VELOCITY = 0,
  K_CURL,
  K_VORTICITY,
  K_BOUNDARY,
  K_ADVECTION,
  K_PRESSURE,
  K_LIGHTING,
  K_PRECIP,
  K_LIGHTNING,
  K_SPLAT, // box filter of the particle deposits + accumulation clear
  K_COPY,
  K_HALO,
  K_FUSED_DRY,  // velocity + advection + pressure (pass_mask WX_PASS_DRY)
  K_MARCH_DRY,  // the same as a row-marching wavefront kernel (wx_march.h)
  K_MARCH_WET,  // the whole iteration as one row-marching wavefront kernel (wx_wet.h)
  K_MARCH_DRY2, // TWO dry iterations per launch (wx_march2.h)
  K_COUNT
};
const char *const kKernelNames[K_COUNT] = {"velocity", "curl", "vorticity", "boundary", "advection", "pressure", "lighting",
                                           "precipitation", "lightning", "splat_box", "copy", "halo", "fused_dry_vel_advect_pressure", "march_dry_vel_advect_pressure",
                                           "march_wet_full_iteration", "march_dry2_two_iterations_per_launch"};

struct ProfRec {
  hipEvent_t a, b;
  int kid;
  bool shared_a; // `a` is the previous scope's `b` (one event per scope boundary: an event record costs the stream ~5 us)
};

} // namespace

struct Block {
  char *p;
  size_t bytes;
};

struct wx_sim {
  int X = 0, Y = 0;      // local width (owned + 2*halo), height
  int Xg = 0, x0 = 0;    // global width, global x of the first OWNED column
  int halo = 0, n_drops = 0;
  bool uploaded = false, have_params = false;
  int even = 1;          // app.js `even`
  // emittedLight on demand (k_emitted): did the most recent iteration run the lighting pass, and with which uniforms
  bool emit_lit = false;
  Uni emit_uni;
  // waterTexture_0 (post-boundary water: what a save stores, app.js:6587-6589 -- no display pass samples it) is made ON DEMAND after a
  // step of the marching wet kernel without particles: the display iteration of a frame then stores 20 instead of 36 extra bytes per
  // cell (958 -> ~830 us at 16384 x 2048). The inputs of the last iteration stay where the ping-pong left them until the next step
  // (base[1], wall[1], water[2], light_0's planes); materialize_water0 runs velocity -> curl -> vorticity -> boundary of the per-pass
  // kernel set on them, with the parameters of THAT iteration (w0_uni; the initial_T row is copied aside if wx_set_params replaces it).
  bool water0_pending = false, w0_even = false, w0_initT_saved = false, lazy_water0 = true;
  Uni w0_uni;
  float4 *w0_b1 = nullptr, *w0_b2 = nullptr, *w0_light = nullptr;
  char4 *w0_w1 = nullptr, *w0_w2 = nullptr;
  float *w0_initT = nullptr, *w0_curl = nullptr;
  half4 *emitted = nullptr; // RGBA16F, allocated by the first read
  int drop_cur = 0;      // particle buffer holding the latest state
  // slab handles with particles (see SlabP in wx_kernels.h)
  // the partitioned droplet pool (SlabP): per droplet "tracked by another rank", status flips / ownership of the current period,
  // scratch for the event resolution; capacities of the exchange buffers
  unsigned char *pool_remote = nullptr, *pool_owned = nullptr;
  unsigned short *pool_flips = nullptr;
  int *pool_best = nullptr;
  int pool_event_cap = 0, pool_edge_cap = 0;
  bool pool_check = false; // an exchange buffer may have overflowed since the flags were last looked at
  int period_j = 0;      // iterations since wx_slab_period_begin: halo - 6*j ghost columns are still valid
  int pool_exact = 0;    // WX_OPT_POOL_EXACT
  int exact_pending = 0; // iterations since the last wx_pool_events_apply (exact mode allows one)
  int rank = 0;          // tie-break of the claim keys (wx_slab_set_rank)
  int seam = 0;          // local column of global column 0 if strictly inside the local array
  int64_t iter = 0;
  int dry_march = 1;     // water-free dry iteration: 1 = row-marching wavefront kernel (wx_march.h, default: 0.32 vs 0.38 ms at
                         // 16384x2048), 0 = LDS-tiled kernel (wx_dry.h); env WX_DRY_MARCH
  int dry_pairs = 1;     // WX_OPT_DRY_PAIRS: the water-free dry stencil two iterations per launch where it can (wx_march2.h; round 5)
  // placement (wx_tune_placement): the search runs ONCE per handle -- by the host's call, or inside the first wx_step of a whole-domain
  // handle of WX_PLACEMENT_AUTO_CELLS cells and more (WX_OPT_PLACEMENT_SEARCH tries; 0: never)
  int place_tries = 6;
  bool place_done = false, place_busy = false;
  float place_first_ms = 0.f, place_kept_ms = 0.f;
  // the pair kernel's exact path (wx_march2.h, Dry2Fix): control words, the list of recorded cells, the host-visible length of the last list
  int *pair_ctl = nullptr;
  int2 *pair_cells = nullptr;
  int pair_cap = 0, pair_epoch = 0;
  int *pair_hint_host = nullptr, *pair_hint_dev = nullptr;
  int bands_mode = 1;    // WX_OPT_ROW_BANDS
  int fix_cap_request = 0; // WX_OPT_FIX_CAP (0: a quarter of the grid)
  int fused = 2;         // non-zero (default): the whole iteration as one row-marching kernel (wx_wet.h); 0: one kernel per reference
                         // pass (env WX_FUSED; the independent cross-check of the parity tests)
  wx_params p{};
  Geo geo{};
  Uni uni{};
  hipStream_t stream = nullptr;
  // display streaming (wx_stream_frame): copies run on their own stream, fenced against the compute stream by events
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_fields_ready = nullptr, ev_copy_done = nullptr;
  bool copy_in_flight = false;
  // device storage
  float4 *base[3] = {nullptr, nullptr, nullptr}; // [2]: post-advection base of the single-kernel paths (= baseTexture_1)
  float4 *water[3] = {nullptr, nullptr, nullptr}, *light[2] = {nullptr, nullptr}; // water[2]: spare of the single-kernel ping-pong
  bool ran_fused = false; // the last iteration used the single-kernel ping-pong (field mapping differs)
  char4 *wall[2] = {nullptr, nullptr};
  float *curl = nullptr;
  float2 *vort = nullptr, *dep = nullptr;
  float3 *fb = nullptr;      // precipitationFeedbackTexture stored with its three used channels (12-byte texels); RGBA on demand: fb_rgba
  float4 *fb_rgba = nullptr; // (not in the placement blocks: allocated by the first reader of WX_FIELD_PRECIP_FB)
  float *drops[2] = {nullptr, nullptr};
  float *initial_T = nullptr, *snd_T = nullptr, *snd_W = nullptr, *snd_Vel = nullptr;
  DevState *state = nullptr;
  // light_0 / light_1 as planes (wx_tile.h, LightPlanes): the representation of the marching wet kernel. `light_planar`
  // says which copy is current; the other kernels, readback and streaming use the interleaved light[] buffers.
  LightPlanes lp[3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}; // [2]: spare of the marching kernel
  bool light_planar = false;
  float *tdisp = nullptr;   // post-advection temperature where the pressure pass changes it (marching wet kernel): for the droplets, and for baseTexture_1 on demand
  float *pdisp = nullptr;   // post-advection pressure of the display iteration (marching wet kernel): baseTexture_1 is assembled from base_0, this and tdisp when asked for
  bool disp_lazy = false;   // BASE_DISP has to be assembled (field_info): the last display iteration was the marching wet kernel's
  bool wall_veg_ok = false; // no negative vegetation byte anywhere (advection would clamp it: the one wall change it can make unasked)
  FullCtx *full_ctx = nullptr;
  std::vector<void *> alloc_pads;          // (WX_ALLOC_PADS experiment: pads between the planes' allocations)
  float4 *zero_row = nullptr;              // X texels of zeros (marching wet kernel: rows of feedback tiles that are known to be zero)
  // marching wet kernel: output cells fed by a back-trace longer than 0.9 cells, recomputed exactly by k_wet_fix (wx_wet.h)
  int *fix_count = nullptr;
  int2 *fix_cells = nullptr;
  int *fix_hint_host = nullptr, *fix_hint_dev = nullptr; // pinned + mapped word: length of the last exact-path list (WetFixList::hint)
  int fix_cap = 0;
  bool fix_check = false; // a marching iteration ran since the overflow flag was last looked at
  // overlap of the halo exchange with compute (wx_set_comm_stream / wx_step_overlap)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_edges = nullptr, ev_unpacked = nullptr; // edge strips of the last iteration done (compute stream) / ghosts written (comm stream)
  bool edges_recorded = false;   // ev_edges was recorded behind the edge strips of the latest iteration
  // Split iterations (edge strips / interior, wx_step_overlap): the two launch groups read the same inputs and write disjoint columns,
  // so the edge group -- a handful of strips that cannot fill the chip -- runs on a stream of its own NEXT TO the interior group
  // instead of in front of it (in order it cost a 2132-column slab +0.045 ms per iteration, profiles/r04_slab_protocol_cost.txt)
  hipStream_t edge_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // Round 5: a split iteration as ONE ordered launch with device-side hand-offs (StripOrder, wx_tile.h) instead of two launch groups on two
  // streams: sync_words[0] = arrivals of edge items (cumulative; a gate kernel on the comm stream waits for arrive_target),
  // sync_words[1] = the ghost epoch (bumped on the comm stream behind every unpack; edge strips dispatched last poll for epoch_host)
  int split_launch = 0;          // WX_OPT_SPLIT_LAUNCH: 0 (default) = the two launch groups of rounds 2-4; 1 = one ordered launch (round 5: measured, not faster)
  unsigned *sync_words = nullptr;
  unsigned arrive_target = 0, epoch_host = 0;
  bool gate_pending = false;     // the latest iteration's edge strips report through sync_words[0]; their exact-path list (fix_count2) is still
                                 // to be consumed -- by the halo pack on the comm stream, else by settle_edges on the compute stream
  bool gate_wet = false;         // ... and it was the wet kernel (there is a list to consume)
  bool split_check = false;      // a split iteration ran since the give-up flag (sync_words[2]) was last looked at
  bool gate_passed = false;      // the comm stream already holds the gate of the latest iteration: a further pack of the same exchange is ordered behind it
  WetIn gate_in{};
  WetOut gate_out{};
  float gate_iter = 0.f;
  bool gate_opt_out = false;
  int *fix_count2 = nullptr;     // the edge group's own list of exact-path cells (the groups run concurrently)
  int2 *fix_cells2 = nullptr;
  int fix_cap2 = 0;
  bool unpack_pending = false;   // ghost columns were unpacked on the comm stream since the compute stream last waited
  // transport inside the library (wx_comm_init / wx_exchange / wx_slab_step; in-process groups: wx_group_*)
  int device = 0;                 // HIP device the handle lives on
  void *comm = nullptr;           // ncclComm_t of a one-rank-per-process job (wx_comm_init)
  int comm_rank = 0, comm_world = 1;
  char *xsend[2] = {nullptr, nullptr}, *xrecv[2] = {nullptr, nullptr}; // halo buffers [left, right] of wx_exchange / wx_group_step
  size_t xbytes = 0;
  bool exchange_in_order = false; // WX_OPT_EXCHANGE_OVERLAP 0: the library's transport keeps the exchange on the compute stream
  hipStream_t own_stream = nullptr, own_comm_stream = nullptr; // streams the library created itself (groups, wx_comm_init)
  hipEvent_t ev_packed = nullptr, ev_copied = nullptr;         // in-process transport: my send buffers are full / my recv copies are done
  char *ev_mine = nullptr, *ev_all = nullptr;                           // slabs with particles: my status-flip events / everybody's (all-gathered)
  int ev_world = 0;
  char *psend[2] = {nullptr, nullptr}, *precv[2] = {nullptr, nullptr}; // ... and the edge droplets to / from the ring neighbours
  hipEvent_t ev_evpacked = nullptr, ev_evcopied = nullptr;
  // events per rank the all-gathers of the library's transport carry: everything at first, then 4 x the largest count the PREVIOUS
  // period's rounds carried, at least 65536 (pool_stride_update: no host round trip inside a period)
  int pool_stride_events = 1 << 30;
  int *ev_seen_host = nullptr; // pinned: pool_seen_max as of the last round
  hipEvent_t ev_counted = nullptr;
  bool count_pending = false;
  // Slabs exact at any speed (VxTrack, wx_tile.h): one iteration invalidates 6 + floor|vx| ghost columns per side. The hosts of all slabs
  // agree on a bound for |vx| per exchange period (measured by the kernels a period or two earlier, with a margin: wx_slab_set_vx_bound);
  // the period is sized by it and the kernels report a |vx| that reaches it (cone_violation: never a silent divergence).
  int cone = WX_SLAB_CONE;
  float vx_bound = 0.f;
  bool vx_stale = true;           // the state was replaced from outside since |vx| was last looked at: scan it before the next period is sized
  bool vx_untracked = false;      // an iteration since the last roll ran a kernel that does not track |vx| (tiled dry kernel, per-pass kernels)
  bool vx_check = false;          // a limit was in force since the violation word was last looked at
  float vx_known = 0.f;           // the latest measurement the bound was derived from
  int *vx_dev = nullptr;          // device: [0] my rolled maximum, [1 .. world] everybody's (all-gathered)
  int *vx_host = nullptr;         // pinned: [2][world] the maxima of the two latest rolls
  hipEvent_t ev_vx[2] = {nullptr, nullptr};
  bool vx_have[2] = {false, false};
  int vx_slot = 0, vx_world = 0;
  int since_exchange = 0;         // iterations since the ghost columns were last fresh (upload or exchange)
  bool exchanged = false;         // ... and they came from an exchange (the next step may run its interior strips first)
  int air_from_row = -1;       // lowest row above which every cell is free air (cost model of the row segmentation); -1: to be measured
  WetLaunch wet_shape{};       // cached launch shape for that value
  bool wet_shape_valid = false;
  SplatGrid sg{};              // particle splat accumulation (allocated when the handle has droplets)
  int splat_par = 0;           // which set of work-list counters the next iteration fills (see SplatGrid::work)
  bool fb_dirty = false; // feedback/deposition hold non-zero data (particles ran last iteration)
  // wx_set_option
  int splat_order = 0;        // WX_OPT_SPLAT_ORDER: 1 = deterministic (records sorted by anchor, summed in droplet-index order)
  int check_launches = 0;     // WX_OPT_CHECK_LAUNCHES: synchronise and check after every kernel launch of wx_step (debugging)
  int *det_key[2] = {nullptr, nullptr}, *det_idx[2] = {nullptr, nullptr}; // deposit records: keys / droplet indices, unsorted and sorted
  float *det_val = nullptr;
  void *det_tmp = nullptr;
  size_t det_tmp_bytes = 0;
  bool water_trivial = false; // the water texture is known to be 0 in air cells and only the wall marker in wall cells: set by wx_upload,
                              // cleared by every step that can put water there (anything but the water-free dry iteration)
  bool ghost_check = false;   // ghost columns were unpacked while water_trivial: DevState::ghost_nontrivial is validated by the next blocking call
  bool local_water_free = false; // what the last upload established for THIS handle's cells (wx_water_free)
  bool slab_dry_agreed = false;  // the host asserted that every slab of the domain was uploaded water-free (wx_slab_assert_water_free)
  // ... and from then on this slab's halo messages carry the base texture alone (wx_halo_message_bytes): the water-free dry iteration
  // writes nothing else, so the ghost columns of water, wall and light stay what the upload made them. Latched by the (collective)
  // assertion and dropped by it or by a step that runs anything else -- the same call with the same parameters on every rank --, never
  // by a rank-local event: two neighbours must not disagree about the size of a message
  bool halo_base_only = false;
  // (set per exchange by the library's transport, wx_comm.h: a period that runs in order anyway keeps its exchange on the compute stream --
  // two cross-stream event hops per period cost the north star's slab 0.005 ms per iteration, profiles/r05_dry_slab_inorder.txt)
  bool xchg_inline = false;
  // device storage of the handle (see storage_begin): blocks[0] = the arena of the small objects (or of everything: WX_ARENA=1), then one
  // allocation per large plane; every pointer member that lives in a block is registered, so that wx_tune_placement can move the
  // whole state to another set of allocations
  std::vector<Block> blocks;
  size_t small_used = 0;
  bool one_arena = false;
  size_t arena_skew = 0;
  int arena_count = 0;
  std::vector<void **> slots;
  // profiling
  bool profiling = false;
  hipEvent_t prof_tail = nullptr; // the event that closed the latest scope of the current wx_step call: the next scope starts from it
  bool prof_chain = false;        // inside wx_step: scopes follow each other on the stream (other entry points time their scope alone)
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  double prof_ms[K_COUNT] = {0};
  int prof_n[K_COUNT] = {0};
  std::string err;
};

// makes the handle's device current for the scope (the slabs of a wx_group live on different devices)
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(const wx_sim *s)
  {
    int cur = -1;
    if (s && hipGetDevice(&cur) == hipSuccess && cur != s->device) {
      prev = cur;
      (void)hipSetDevice(s->device);
    }
  }
  ~DeviceScope()
  {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

extern "C" void transport_release(wx_sim *s); // wx_comm.h

namespace {

int fail(wx_sim *s, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (s)
    s->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIPCHK(s, expr)                                                                                  \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail((s), WX_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));           \
  } while (0)

size_t ncell(const wx_sim *s) { return (size_t)s->X * s->Y; }

// after every kernel launch of the iteration loop: the launch itself is always checked (hipGetLastError is a thread-local read);
// with WX_OPT_CHECK_LAUNCHES the stream is also synchronised, so that an asynchronous fault is reported with the kernel's name
#define LAUNCH_CHECK(s, what)                                                                                         \
  do {                                                                                                                \
    hipError_t e_ = hipGetLastError();                                                                                \
    if (e_ == hipSuccess && (s)->check_launches) e_ = hipStreamSynchronize((s)->stream);                              \
    if (e_ != hipSuccess) return fail((s), WX_E_DEVICE, "wx_step: %s (iteration %lld): %s", what, (long long)(s)->iter, hipGetErrorString(e_)); \
  } while (0)

hipEvent_t get_event(wx_sim *s)
{
  if (!s->ev_pool.empty()) {
    hipEvent_t e = s->ev_pool.back();
    s->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

struct ProfScope {
  wx_sim *s;
  ProfRec r;
  bool on;
  ProfScope(wx_sim *s_, int kid) : s(s_), on(s_->profiling)
  {
    if (on) {
      // Scopes of one wx_step call follow each other on the stream with nothing but the odd memset or event wait in between: the event that
      // closed one opens the next. Two records per boundary were a 10 us bubble between the kernels (rocprofv3 timeline of the bench:
      // 1.5 % of a grid-only iteration, 4 % of one with particles); one is half of that.
      r.kid = kid;
      r.shared_a = s->prof_tail != nullptr;
      r.a = r.shared_a ? s->prof_tail : get_event(s);
      r.b = get_event(s);
      if (!r.shared_a) hipEventRecord(r.a, s->stream);
    }
  }
  ~ProfScope()
  {
    if (on) {
      hipEventRecord(r.b, s->stream);
      s->prof_tail = s->prof_chain ? r.b : nullptr;
      s->prof.push_back(r);
    }
  }
};

void collect_profile(wx_sim *s)
{
  if (s->prof.empty()) return;
  hipStreamSynchronize(s->stream);
  for (auto &r : s->prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      s->prof_ms[r.kid] += ms;
      s->prof_n[r.kid] += 1;
    }
    if (!r.shared_a) s->ev_pool.push_back(r.a); // (a shared one goes back with the record it closed)
    s->ev_pool.push_back(r.b);
  }
  s->prof.clear();
  s->prof_tail = nullptr;
}

dim3 grid2d(const wx_sim *s) { return dim3((s->X + BX - 1) / BX, (s->Y + BY - 1) / BY); }

void build_geo(wx_sim *s)
{
  Geo &g = s->geo;
  g.X = s->X;
  g.Y = s->Y;
  g.Xg = s->Xg;
  int xoff = (s->x0 - s->halo) % s->Xg;
  if (xoff < 0) xoff += s->Xg;
  g.xoff = xoff;
  if (s->p.quad_scale) {
    // app.js:4770-4788: the quad's U runs 0 .. f32(res * 1.0000001)
    g.sx = (float)((double)s->Xg * 1.0000001) / (float)s->Xg;
    g.sy = (float)((double)s->Y * 1.0000001) / (float)s->Y;
  } else {
    g.sx = 1.0f;
    g.sy = 1.0f;
  }
  g.texX = (float)(1.0 / (double)s->Xg); // app.js:5436-5437
  g.texY = (float)(1.0 / (double)s->Y);
}

void build_uni(wx_sim *s)
{
  const wx_params &p = s->p;
  Uni &u = s->uni;
  u.dragMultiplier = p.dragMultiplier;
  u.wind = p.wind;
  u.vorticity = p.vorticity;
  u.landEvaporation = p.landEvaporation;
  u.waterEvaporation = p.waterEvaporation;
  u.dynamicWaterTemperature = p.dynamicWaterTemperature;
  u.evapHeat = p.evapHeat;
  u.waterWeight = p.waterWeight;
  u.dryLapse = p.dryLapse;
  u.meltingHeat = p.meltingHeat;
  u.condensationRate = p.condensationRate;
  u.globalDrying = p.globalDrying;
  u.globalHeating = p.globalHeating;
  u.soundingForcing = p.soundingForcing;
  u.globalEffectsStartAlt = p.globalEffectsStartAlt;
  u.globalEffectsEndAlt = p.globalEffectsEndAlt;
  u.waterTemperature = p.waterTemperature;
  u.sunIntensity = p.sunIntensity;
  u.greenhouseGases = p.greenhouseGases;
  u.waterGreenHouseEffect = p.waterGreenHouseEffect;
  u.IR_rate = p.IR_rate;
  u.aboveZeroThreshold = p.aboveZeroThreshold;
  u.subZeroThreshold = p.subZeroThreshold;
  u.spawnChanceMult = p.spawnChanceMult;
  u.snowDensity = p.snowDensity;
  u.fallSpeed = p.fallSpeed;
  u.growthRate0C = p.growthRate0C;
  u.growthRate_30C = p.growthRate_30C;
  u.freezingRate = p.freezingRate;
  u.meltingRate = p.meltingRate;
  u.evapRate = p.evapRate;
  for (int i = 0; i < 4; i++) u.userInputValues[i] = p.userInputValues[i];
  for (int i = 0; i < 2; i++) u.userInputMove[i] = p.userInputMove[i];
  u.userInputType = p.userInputType;
  u.wrapHorizontally = p.wrapHorizontally;
  for (int i = 0; i < 4; i++) u.airplaneValues[i] = p.airplaneValues[i];
  // sin/cos of the uniform sunAngle, evaluated once here instead of per fragment
  u.cos_a = cosf(p.sunAngle);
  u.sin_a = sinf(p.sunAngle);
  u.sin_ma = sinf(-p.sunAngle);
  // uniform-only sub-expressions of the shaders, evaluated once with the same fp32 operations
  u.vel_keep = 1.0f - p.dragMultiplier * 0.0002f;
  u.wind_add = p.wind * 0.000001f;
  u.snd_dragk = 1.0f - map_rangeC(p.soundingForcing, 0.1f, 1.0f, 0.0f, 0.001f);
  u.snd_velk = map_rangeC(p.soundingForcing, 0.9f, 1.0f, 0.0f, 0.001f);
  u.a_texX = 1.0f / (float)s->Xg;
  u.a_texY = 1.0f / (float)s->Y;
  u.a_invTexY = 1.0f / u.a_texY;
  u.a_aspect = u.a_texY / u.a_texX;
  u.chc = 300.0f / (float)s->Y;
  u.sun_uniform = s->p.quad_scale == 0;
  if (u.sun_uniform) { // lighting_cell's per-cell expressions with fragCoord - (x + 0.5) == 0, in the same fp32 operations
    const float ox = 0.0f + u.sin_a, oy = 0.0f + u.cos_a;
    const float fu = floorf(ox), fv = floorf(oy);
    const float al = ox - fu, be = oy - fv;
    u.sun_dx0 = (int)fu;
    u.sun_fv = (int)fv;
    u.sun_w00 = (1.0f - al) * (1.0f - be);
    u.sun_w10 = al * (1.0f - be);
    u.sun_w01 = (1.0f - al) * be;
    u.sun_w11 = al * be;
  } else {
    u.sun_dx0 = u.sun_fv = 0;
    u.sun_w00 = u.sun_w10 = u.sun_w01 = u.sun_w11 = 0.0f;
  }
  { // lightingShader.frag:58-59: how red the sunlight is; common.glsl:374-378
    const float deg2rad = 0.0174533f;
    const float scattering = clampf(map_range(fabsf(p.sunAngle), 75.0f * deg2rad, 90.0f * deg2rad, 0.0f, 1.0f), 0.0f, 1.0f);
    const float val = 1.0f - scattering;
    hsv2rgb(0.015f + val * 0.15f, fminf(2.0f - val * 2.0f, 1.0f), 1.0f, u.sun_col);
    u.night_glow = fabsf(p.sunAngle) > 85.0f * deg2rad;
  }
}

// Device storage. The marching kernels stream ~13 planes at once; where those planes lie relative to each other in PHYSICAL memory
// decides how the streams spread over the HBM channels: the SAME binary ran the 16384 x 2048 iteration in 0.72 .. 0.87 ms depending on
// the addresses the allocator happened to hand out (several handles alive in one process, each time reproducible to 0.1 %;
// profiles/r03_alloc_probe.txt) -- the "box to box" spread of rounds 1 and 2. Planes at power-of-two distances in one physically
// contiguous range alias in the channel hash (hipDeviceMallocContiguous: always the slowest level); ONE allocation carved into planes
// tends to be like that (0.76 - 0.84), one allocation PER PLANE scatters them (0.72 - 0.75 most of the time, 0.80 - 0.84 sometimes).
// Hence: one allocation per large plane (default; WX_ARENA=1: everything in one arena, WX_ARENA_SKEW / WX_ARENA_CONTIG: experiments),
// the small objects share blocks[0], and every pointer is registered so that wx_tune_placement can try other sets of allocations.
static int block_of(const std::vector<Block> &B, const void *p)
{
  for (size_t i = 0; i < B.size(); i++)
    if ((const char *)p >= B[i].p && (const char *)p < B[i].p + B[i].bytes) return (int)i;
  return -1;
}
static hipError_t block_alloc(size_t bytes, Block *out)
{
  out->p = nullptr;
  out->bytes = bytes;
  hipError_t e = hipErrorUnknown;
  if (const char *c = wx_tune_env("WX_ARENA_CONTIG")) { // experiment: physically contiguous VRAM (the largest page-table fragments)
    if (atoi(c) != 0) e = hipExtMallocWithFlags((void **)&out->p, bytes, hipDeviceMallocContiguous);
    if (e != hipSuccess) (void)hipGetLastError();
  }
  if (e != hipSuccess) e = hipMalloc((void **)&out->p, bytes);
  return e;
}
static int storage_begin(wx_sim *s, size_t total_bytes, size_t small_bytes)
{
  s->arena_skew = 0;
  if (const char *e = wx_tune_env("WX_ARENA_SKEW")) s->arena_skew = (size_t)atoll(e);
  if (const char *e = wx_tune_env("WX_ARENA")) s->one_arena = atoi(e) != 0;
  Block b;
  const size_t bytes = s->one_arena ? total_bytes + 128 * (4096 + s->arena_skew) : small_bytes;
  if (block_alloc(bytes, &b) != hipSuccess) return fail(s, WX_E_NOMEM, "wx_create: %zu bytes of device memory", bytes);
  HIPCHK(s, hipMemset(b.p, 0, bytes));
  s->blocks.push_back(b);
  return WX_OK;
}
static void dfree(wx_sim *s, void *p)
{
  if (p && block_of(s->blocks, p) < 0) hipFree(p);
}
template <class T> int dalloc(wx_sim *s, T **p, size_t n)
{
  const size_t bytes = n * sizeof(T);
  const bool large = bytes >= (1u << 20);
  if (!s->blocks.empty() && (s->one_arena || !large)) { // carve it out of blocks[0]
    size_t off = (s->small_used + 255) & ~(size_t)255;
    if (large) off += s->arena_skew * (size_t)(++s->arena_count);
    off = (off + 255) & ~(size_t)255;
    if (off + bytes <= s->blocks[0].bytes) {
      *p = reinterpret_cast<T *>(s->blocks[0].p + off);
      s->small_used = off + bytes;
      s->slots.push_back(reinterpret_cast<void **>(p));
      return WX_OK; // (zeroed with the block)
    }
  }
  Block b;
  if (const char *e = wx_tune_env("WX_ALLOC_PADS")) { // experiment: "a,b,m" = a pad of (a + b * (k mod m)) MB in front of the k-th plane
    int a = 0, bb = 0, m = 1; // (no pad scheme makes a placement fast: profiles/r04_placement_walk.txt)
    if (sscanf(e, "%d,%d,%d", &a, &bb, &m) >= 1 && m > 0) {
      void *pad = nullptr;
      const size_t mb = (size_t)(a + bb * ((int)s->alloc_pads.size() % m));
      if (mb > 0 && hipMalloc(&pad, mb << 20) == hipSuccess) s->alloc_pads.push_back(pad);
      else (void)hipGetLastError();
    }
  }
  if (block_alloc(bytes ? bytes : 256, &b) != hipSuccess) return fail(s, WX_E_NOMEM, "%zu bytes of device memory", bytes);
  HIPCHK(s, hipMemset(b.p, 0, b.bytes));
  s->blocks.push_back(b);
  *p = reinterpret_cast<T *>(b.p);
  s->slots.push_back(reinterpret_cast<void **>(p));
  return WX_OK;
}

int copy_field(wx_sim *s, const float4 *src, float4 *dst)
{
  ProfScope ps(s, K_COPY);
  hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, s->stream, src, dst, ncell(s));
  LAUNCH_CHECK(s, "copy");
  return WX_OK;
}
int copy_wall(wx_sim *s, const char4 *src, char4 *dst)
{
  ProfScope ps(s, K_COPY);
  hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, s->stream, src, dst, ncell(s));
  LAUNCH_CHECK(s, "copy");
  return WX_OK;
}

// One iteration with the reference's pass structure (one kernel per draw call).
int iterate_per_pass(wx_sim *s, unsigned mask)
{
  s->vx_untracked = true; // (these kernels do not report their |vx|: the exchange scans the state instead)
  const dim3 grid = grid2d(s), block(BX, BY);
  const Geo g = s->geo;
  Uni u = s->uni;
  u.iterNum = (float)s->iter; // iterNum is passed as float (app.js:5859) and cast int(iterNum) in the shaders
  u.iterI = (int)u.iterNum;

  // 1 velocity: base_0, wall_0 -> base_1, wall_1
  if (mask & WX_PASS_VELOCITY) {
    ProfScope ps(s, K_VELOCITY);
    hipLaunchKernelGGL(k_velocity, grid, block, 0, s->stream, g, u, s->base[0], s->wall[0], s->base[1], s->wall[1]);
    LAUNCH_CHECK(s, "velocity");
  } else {
    if (int rc_ = copy_field(s, s->base[0], s->base[1])) return rc_;
    if (int rc_ = copy_wall(s, s->wall[0], s->wall[1])) return rc_;
  }
  // 2, 3 curl + vorticity
  if (mask & WX_PASS_VORTICITY) {
    {
      ProfScope ps(s, K_CURL);
      hipLaunchKernelGGL(k_curl, grid, block, 0, s->stream, g, s->base[1], s->curl);
      LAUNCH_CHECK(s, "curl");
    }
    {
      ProfScope ps(s, K_VORTICITY);
      hipLaunchKernelGGL(k_vorticity, grid, block, 0, s->stream, g, s->curl, s->vort);
      LAUNCH_CHECK(s, "vorticity");
    }
  }
  // 4 boundary: base_1, water_1, vort, wall_1, light_0 (always _0), feedback, deposition -> base_0, water_0, wall_0
  if (mask & WX_PASS_BOUNDARY) {
    GridPtrs in{s->base[1], s->water[1], s->wall[1], s->vort, s->light[0], s->fb_dirty ? s->fb : nullptr,
                s->fb_dirty ? s->dep : nullptr};
    ProfScope ps(s, K_BOUNDARY);
    hipLaunchKernelGGL(k_boundary, grid, block, 0, s->stream, g, u, s->initial_T, in, s->base[0], s->water[0], s->wall[0]);
    LAUNCH_CHECK(s, "boundary");
  } else {
    if (int rc_ = copy_field(s, s->base[1], s->base[0])) return rc_;
    if (int rc_ = copy_field(s, s->water[1], s->water[0])) return rc_;
    if (int rc_ = copy_wall(s, s->wall[1], s->wall[0])) return rc_;
  }
  // 5 advection: _0 -> _1
  if (mask & WX_PASS_ADVECTION) {
    GridPtrs in{s->base[0], s->water[0], s->wall[0], nullptr, nullptr, nullptr, nullptr};
    ProfScope ps(s, K_ADVECTION);
    hipLaunchKernelGGL(k_advection, grid, block, 0, s->stream, g, u, s->initial_T, s->snd_T, s->snd_W, s->snd_Vel, in, s->base[1],
                       s->water[1], s->wall[1]);
    LAUNCH_CHECK(s, "advection");
  } else {
    if (int rc_ = copy_field(s, s->base[0], s->base[1])) return rc_;
    if (int rc_ = copy_field(s, s->water[0], s->water[1])) return rc_;
    if (int rc_ = copy_wall(s, s->wall[0], s->wall[1])) return rc_;
  }
  // 6 pressure: base_1, wall_1 -> base_0, wall_0
  if (mask & WX_PASS_PRESSURE) {
    ProfScope ps(s, K_PRESSURE);
    hipLaunchKernelGGL(k_pressure, grid, block, 0, s->stream, g, s->base[1], s->wall[1], s->base[0], s->wall[0]);
    LAUNCH_CHECK(s, "pressure");
  } else {
    if (int rc_ = copy_field(s, s->base[1], s->base[0])) return rc_;
    if (int rc_ = copy_wall(s, s->wall[1], s->wall[0])) return rc_;
  }
  // 7 lighting: base_1 (pre-pressure!), water_1, wall_1, light_src -> light_dst ; even = !even
  const int src = s->even ? 0 : 1, dst = s->even ? 1 : 0;
  if (mask & WX_PASS_LIGHTING) {
    ProfScope ps(s, K_LIGHTING);
    hipLaunchKernelGGL(k_lighting, grid, block, 0, s->stream, g, u, s->base[1], s->water[1], s->wall[1], s->light[src], s->light[dst]);
    LAUNCH_CHECK(s, "lighting");
  }
  return WX_OK;
}

// light_0 / light_1: switch between the interleaved textures and the planes of the marching wet kernel
static void light_to_planes(wx_sim *s)
{
  if (s->light_planar) return;
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_light_to_planes, dim3(2048), dim3(256), 0, s->stream, ncell(s), s->light[i], s->lp[i]);
  s->light_planar = true;
}
static void light_to_rgba(wx_sim *s)
{
  if (!s->light_planar) return;
  for (int i = 0; i < 2; i++) {
    LightPlanesC src{s->lp[i].x, s->lp[i].y, s->lp[i].zw};
    hipLaunchKernelGGL(k_light_from_planes, dim3(2048), dim3(256), 0, s->stream, ncell(s), src, s->light[i]);
  }
  s->light_planar = false;
}

// The stream EVERYTHING of a halo / droplet-pool exchange runs on (pack, transfers, unpack, the pool kernels): the handle's comm stream --
// whoever created it, the library or the host (wx_set_comm_stream) -- unless the exchange has to stay in order with the iterations:
// the exact particle mode (its per-iteration rounds need the finished iteration and are needed by the next one) and
// WX_OPT_EXCHANGE_OVERLAP 0. One function, so that no two pieces of an exchange can ever disagree about their stream (with a
// host-supplied comm stream and WX_OPT_POOL_EXACT the pool kernels ran on the compute stream and the transfers on the comm stream, unfenced).
static hipStream_t exchange_stream(const wx_sim *s)
{
  const bool in_order = (s->pool_remote && s->pool_exact) || s->exchange_in_order || s->xchg_inline;
  return (s->comm_stream && !in_order) ? s->comm_stream : s->stream;
}

// the compute stream may not touch ghost columns before the comm stream has written them
static void wait_unpacked(wx_sim *s)
{
  if (!s->unpack_pending) return;
  hipStreamWaitEvent(s->stream, s->ev_unpacked, 0);
  s->unpack_pending = false;
}

// device words of the ordered split launch + the edge strips' own exact-path list (created by the first split iteration)
static int split_launch_ready(wx_sim *s, bool need_fix_list)
{
  if (!s->sync_words) {
    if (hipMalloc((void **)&s->sync_words, 4 * sizeof(unsigned)) != hipSuccess || hipMemsetAsync(s->sync_words, 0, 4 * sizeof(unsigned), s->stream) != hipSuccess)
      return fail(s, WX_E_NOMEM, "wx_step: the split launch's device words");
    s->arrive_target = s->epoch_host = 0;
  }
  if (need_fix_list && !s->fix_count2) {
    const size_t cap = std::min<size_t>(std::max<size_t>((size_t)s->Y * 8 * 64, 1u << 14), 1u << 20); // (a few strips' worth of cells)
    if (hipMalloc((void **)&s->fix_count2, 16) != hipSuccess || hipMalloc((void **)&s->fix_cells2, cap * sizeof(int2)) != hipSuccess ||
        hipMemsetAsync(s->fix_count2, 0, 16, s->stream) != hipSuccess)
      return fail(s, WX_E_NOMEM, "wx_step: the edge strips' exact-path cell list");
    s->fix_cap2 = (int)cap;
    if (wx_tune_env("WX_SPLIT_PREWARM") && s->comm_stream && s->full_ctx) {
      // (experiment) the fix kernel needs 652 bytes of scratch per lane: let the comm stream's hardware queue get its scratch ring now,
      // while nothing on the chip is polling for the exchange this kernel is part of
      hipStreamSynchronize(s->stream);
      const WetFixList fix2{s->fix_count2, s->fix_cells2, s->fix_cap2, nullptr, &s->state->fastest_bits, nullptr};
      WetIn in{};
      WetOut out{};
      launch_wet_fix(0.f, s->full_ctx, in, out, fix2, &s->state->fix_overflow, false, s->comm_stream, 64);
      hipStreamSynchronize(s->comm_stream);
    }
  }
  return WX_OK;
}
// The edge strips of the latest split iteration left exact-path cells on their own list and nobody has packed the halo since (a host
// that asked for WX_OVERLAP_EDGES_FIRST and then did something else): consume the list on the compute stream -- it is ordered behind
// the whole launch, so no gate is needed -- before anything looks at those cells.
static void settle_edges(wx_sim *s)
{
  if (!s->gate_pending) return;
  if (s->gate_wet) {
    const WetFixList fix2{s->fix_count2, s->fix_cells2, s->fix_cap2, nullptr, &s->state->fastest_bits, nullptr};
    launch_wet_fix(s->gate_iter, s->full_ctx, s->gate_in, s->gate_out, fix2, &s->state->fix_overflow, s->gate_opt_out, s->stream, 64);
  }
  s->gate_pending = false;
}
// comm stream: the ghost columns (and whatever else an exchange writes) are in place -- the event for the compute stream's stream-ordered
// consumers, the epoch word for edge strips that are already resident and polling
static int mark_unpacked(wx_sim *s, hipStream_t st)
{
  HIPCHK(s, hipEventRecord(s->ev_unpacked, st));
  s->unpack_pending = true;
  if (s->sync_words) {
    s->epoch_host += 1;
    hipLaunchKernelGGL(k_strip_epoch, dim3(1), dim3(1), 0, st, s->sync_words + 1, s->epoch_host);
  }
  return WX_OK;
}

// what the marching kernels report their |vx| to: the accumulator always, the limit of the current exchange period on slabs
static VxTrack vx_track(wx_sim *s)
{
  if (s->halo > 0) s->vx_check = true;
  // slabs watch three halo widths from either edge of the local array (in columns here; the launch functions turn them into strips)
  const int zl = s->halo > 0 ? 3 * s->halo : 0, zr = s->halo > 0 ? s->X - 3 * s->halo : 0;
  return VxTrack{&s->state->vx_max_bits, &s->state->cone_violation, s->halo > 0 ? (float)(s->cone - 5) : 0.0f, zl, zr, s->halo > 0 ? (float)std::max(1, 2 * s->halo - 8) : 0.0f};
}

// stream / events / second exact-path list of the concurrent edge group (created by the first split iteration)
static int edge_stream_ready(wx_sim *s, bool need_fix_list)
{
  if (!s->edge_stream) {
    // (highest priority: the handful of edge waves should get the first free slots next to the interior group's thousands -- the
    // neighbours wait for what they produce)
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    HIPCHK(s, hipStreamCreateWithPriority(&s->edge_stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(s, hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    HIPCHK(s, hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  }
  if (need_fix_list && !s->fix_count2) {
    const size_t cap = std::min<size_t>(std::max<size_t>((size_t)s->Y * 8 * 64, 1u << 14), 1u << 20); // (a few strips' worth of cells)
    if (hipMalloc((void **)&s->fix_count2, 16) != hipSuccess || hipMalloc((void **)&s->fix_cells2, cap * sizeof(int2)) != hipSuccess ||
        hipMemsetAsync(s->fix_count2, 0, 16, s->stream) != hipSuccess)
      return fail(s, WX_E_NOMEM, "wx_step: the edge group's exact-path cell list");
    s->fix_cap2 = (int)cap;
  }
  return WX_OK;
}

// The whole iteration as ONE row-marching kernel (wx_wet.h). Reads base[0], wall[0], water[1], the light_0 planes and the source
// light planes; writes the other buffer of each pair and swaps the pointers, so that afterwards the usual roles hold again
// (base[0] / wall[0] = post-pressure state, water[1] = post-advection water, lp[0] / lp[1] = light_0 / light_1 as in the reference).
// edge_mode: 0 = one launch over all strips; 1 = edge strips first, event, interior (last iteration before an exchange);
// 2 = interior first, wait for the unpack event, edge strips (first iteration after an exchange); 3 = a one-iteration period: both
int iterate_march_wet(wx_sim *s, bool opt_out, bool precip, int edge_mode = 0)
{
  const int src = s->even ? 0 : 1;
  light_to_planes(s);
  // lighting writes light_1 (even) or light_0 (odd); waves that are still at their boundary stage read light_0, so the odd
  // iterations write the spare plane set, which then becomes light_0
  const LightPlanes light_dst = s->even ? s->lp[1] : s->lp[2];
  const bool fb = s->fb_dirty;
  WetIn in{s->base[0], s->wall[0], s->water[1], LightPlanesC{s->lp[0].x, s->lp[0].y, s->lp[0].zw},
           LightPlanesC{s->lp[src].x, s->lp[src].y, s->lp[src].zw}, fb ? s->fb : nullptr, fb ? s->dep : nullptr, fb ? s->sg.fb_zero : nullptr,
           s->zero_row, s->sg.TXn};
  // (waterTexture_0 on demand: see water0_pending; with particles the feedback textures the boundary stage read are gone after the iteration)
  const bool lazy_w0 = opt_out && !precip && !fb && s->lazy_water0;
  // (round 6) the display iteration stores 4 instead of 16 bytes per cell of baseTexture_1: the post-advection PRESSURE (+ the temperature in rows
  // directly above land, as for the droplets) -- vx, vy and T everywhere else are the post-pressure texture's (pressure_cell, wx_cells.h); the
  // RGBA texels are assembled when a reader asks for WX_FIELD_BASE_DISP (k_base_disp_assemble)
  WetOut out{s->base[1], s->wall[1], s->water[2], light_dst, s->pdisp, lazy_w0 ? nullptr : s->water[0], s->curl, (precip || opt_out) ? s->tdisp : nullptr};
  if (opt_out) s->disp_lazy = true;
  if (opt_out) {
    s->water0_pending = lazy_w0;
    s->w0_even = s->even;
    s->w0_initT_saved = false;
    s->w0_uni = s->uni;
    s->w0_uni.iterNum = (float)s->iter;
    s->w0_uni.iterI = (int)s->w0_uni.iterNum;
  }
#ifdef WX_WET_TIMING
  static unsigned long long *dbg_cycles = nullptr;
  if (!dbg_cycles && hipMalloc((void **)&dbg_cycles, 16 * 8192 * WMAXSEG) != hipSuccess) return fail(s, WX_E_NOMEM, "wx_step: timing buffer");
  out.cycles = dbg_cycles;
#endif
  if (s->air_from_row < 0 && wet_alpha() == 1.0) { // rows cost the same with or without terrain (the default): nothing to measure
    s->air_from_row = 0;
    s->wet_shape_valid = false;
  }
  if (s->air_from_row < 0) { // after an upload or a wall edit: where does the terrain end? (one small kernel + a 4-byte readback)
    int *d = &s->state->scratch_int, v = 0;
    hipMemsetAsync(d, 0, 4, s->stream);
    hipLaunchKernelGGL(k_air_from_row, dim3(1024), dim3(256), 0, s->stream, s->X, s->Y, s->wall[0], d);
    hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, s->stream);
    if (hipStreamSynchronize(s->stream) != hipSuccess) return fail(s, WX_E_DEVICE, "wx_step: terrain scan failed");
    s->air_from_row = v < s->Y ? v : s->Y;
    s->wet_shape_valid = false;
  }
  if (!s->wet_shape_valid) {
    s->wet_shape = wet_launch_shape(s->geo, s->air_from_row, s->bands_mode);
    s->wet_shape_valid = true;
    if (wx_tune_env("WX_MARCH_DEBUG")) fprintf(stderr, "[wx_wet] air_from_row=%d of %d\n", s->air_from_row, s->Y);
  }
  const WetLaunch &shape = s->wet_shape;
  if (!s->fix_cells) { // (once per handle) room for a quarter of the grid's cells, at most 8 M entries
    size_t cap = std::min<size_t>(std::max<size_t>(ncell(s) / 4, 1u << 16), 1u << 23);
    if (s->fix_cap_request > 0) cap = (size_t)s->fix_cap_request; // (WX_OPT_FIX_CAP; tests: provoke the overflow report)
    if (hipMalloc((void **)&s->fix_count, 16) != hipSuccess || hipMalloc((void **)&s->fix_cells, cap * sizeof(int2)) != hipSuccess ||
        hipMemsetAsync(s->fix_count, 0, 16, s->stream) != hipSuccess) // {entries, arrival ticket of the fix pass, what the hint word was last told, -}: the fix pass leaves the first two at 0
      return fail(s, WX_E_NOMEM, "wx_step: %zu bytes for the exact-path cell list", cap * sizeof(int2));
    s->fix_cap = (int)cap;
    // (optional: without the mapped word every fix launch covers the whole chip)
    if (!(wx_tune_env("WX_FIX_HINT") && atoi(wx_tune_env("WX_FIX_HINT")) == 0) && // (WX_FIX_HINT=0: tuning)
        hipHostMalloc((void **)&s->fix_hint_host, sizeof(int), hipHostMallocMapped) == hipSuccess && s->fix_hint_host) {
      *s->fix_hint_host = 1;
      if (hipHostGetDevicePointer((void **)&s->fix_hint_dev, s->fix_hint_host, 0) != hipSuccess) s->fix_hint_dev = nullptr;
      static const int one = 1; // (count[2]: what the hint word was last told -- the first empty launch sets it back to 0)
      if (s->fix_hint_dev) hipMemcpyAsync(s->fix_count + 2, &one, 4, hipMemcpyHostToDevice, s->stream);
    } else {
      s->fix_hint_host = nullptr;
      (void)hipGetLastError();
    }
  }
  {
    ProfScope ps(s, K_MARCH_WET);
    const WetFixList fix{s->fix_count, s->fix_cells, s->fix_cap, s->fix_hint_host, &s->state->fastest_bits, s->fix_hint_dev};
    // one launch group: marching kernel over a strip range (or two) -> the fix pass over what it recorded (leaves the list empty)
    // no brush input, no airplane event (the common case: a running simulation): the instantiation without those sections
    const bool quiet = !(s->uni.userInputType >= 1) && !(s->uni.airplaneValues[3] < 0.0f || s->uni.airplaneValues[3] > 0.9f);
    const VxTrack vt = vx_track(s);
    auto group_on = [&](hipStream_t st, const WetFixList &fl, int lo0, int cnt0, int lo1 = 0, int cnt1 = 0, bool halved = false) {
      StripOrder eo{}; // the edge group of the two-launch protocol: its waves may get a higher issue priority (StripOrder::prio; mode 4)
      if (halved) {
        eo.mode = 4;
        eo.prio = WX_SPLIT_PRIO;
        if (const char *e = wx_tune_env("WX_SPLIT_PRIO")) eo.prio = atoi(e);
        if (const char *e = wx_tune_env("WX_SPLIT_HALVE")) halved = atoi(e) != 0;
        if (eo.prio == 0) eo.mode = 0;
      }
      launch_march_wet(halved ? wet_shape_halved(shape) : shape, (float)s->iter, s->full_ctx, in, out, fl, opt_out, quiet, st, lo0, cnt0, lo1, cnt1, eo.mode ? &eo : nullptr, nullptr, &vt); // (both ranges in ONE launch)
      launch_wet_fix((float)s->iter, s->full_ctx, in, out, fl, &s->state->fix_overflow, opt_out, st, halved ? 64 : 0);
    };
    auto group = [&](int lo0, int cnt0, int lo1 = 0, int cnt1 = 0) { group_on(s->stream, fix, lo0, cnt0, lo1, cnt1); };
    // edge strips: every output column wx_halo_pack reads ([halo, 2*halo) and its mirror) and every strip that reads ghost columns
    // (a wave records -- and the fix pass rewrites -- only output cells of its own strip: the groups never write each other's columns)
    const int nl = s->halo > 0 ? (2 * s->halo - 1) / WOUT + 1 : 0, nr0 = s->halo > 0 ? (s->X - 2 * s->halo) / WOUT : shape.n_strips;
    if (edge_mode == 0 || s->halo == 0 || nl >= nr0) {
      if (edge_mode & 2) wait_unpacked(s);
      group(0, -1);
    } else if (s->split_launch) {
      // ONE launch over all strips (round 5): the edge strips first in dispatch order when an exchange follows (they report on a device
      // word the comm stream's gate kernel polls), last when one precedes (they poll the epoch word the comm stream bumps behind the
      // unpack). No second stream, no join events, one fix pass on this stream; the edge strips' own exact-path list is consumed on the
      // comm stream in front of the pack (halo_pack_impl) -- B's strips next to them are ordered behind that through the epoch too.
      if (int rc = split_launch_ready(s, true)) return rc;
      StripOrder ord{};
      ord.mode = (edge_mode & 1) ? 1 : 2;
      ord.nl = nl;
      ord.nr0 = nr0;
      if (edge_mode == 3) wait_unpacked(s); // (a one-iteration period: the whole launch behind the unpack, the edge strips first all the same)
      if ((edge_mode & 2) && s->unpack_pending) {
        ord.epoch = s->sync_words + 1;
        ord.epoch_want = s->epoch_host;
        // (the interior strips bordering the edge strips read four of their columns -- cells the comm stream's fix pass may still be
        // rewriting: they wait with the edges)
        if (ord.nl + 1 < ord.nr0 - 1) {
          ord.nl += 1;
          ord.nr0 -= 1;
        }
        s->unpack_pending = false; // (this launch waits on the device; everything behind it on the stream is ordered behind it)
      }
      if (edge_mode & 1) {
        ord.arrive = s->sync_words;
        ord.edge_list = 1;
      }
      ord.prio = WX_SPLIT_PRIO;
      if (const char *e = wx_tune_env("WX_SPLIT_PRIO")) ord.prio = atoi(e);
      if (const char *e = wx_tune_env("WX_SPLIT_NOFENCE")) ord.nofence = atoi(e);
      const WetFixList fix2{s->fix_count2, s->fix_cells2, s->fix_cap2, nullptr, &s->state->fastest_bits, nullptr};
      s->split_check = true;
      const int items = launch_march_wet(shape, (float)s->iter, s->full_ctx, in, out, fix, opt_out, quiet, s->stream, 0, -1, 0, 0, &ord, &fix2, &vt);
      launch_wet_fix((float)s->iter, s->full_ctx, in, out, fix, &s->state->fix_overflow, opt_out, s->stream, 0);
      if (edge_mode & 1) {
        s->arrive_target += (unsigned)items;
        s->gate_pending = true;
        s->gate_wet = true;
        s->gate_in = in;
        s->gate_out = out;
        s->gate_iter = (float)s->iter;
        s->gate_opt_out = opt_out;
      }
    } else {
      // (rounds 2-4, WX_OPT_SPLIT_LAUNCH 0) The edge group on its own stream, the interior group on the compute stream, side by side: the edge stream starts behind
      // everything the compute stream holds so far (and behind the unpack where the ghosts are still in flight), the compute stream
      // goes on behind both groups. What the comm stream waits for before it packs (ev_edges) is the edge group alone.
      if (int rc = edge_stream_ready(s, true)) return rc;
      const WetFixList fix2{s->fix_count2, s->fix_cells2, s->fix_cap2, nullptr, &s->state->fastest_bits, nullptr};
      hipEventRecord(s->ev_fork, s->stream);
      hipStreamWaitEvent(s->edge_stream, s->ev_fork, 0);
      if ((edge_mode & 2) && s->unpack_pending) {
        hipStreamWaitEvent(s->edge_stream, s->ev_unpacked, 0);
        s->unpack_pending = false; // (the compute stream joins the edge stream below)
      }
      group_on(s->edge_stream, fix2, 0, nl, nr0, shape.n_strips - nr0, true); // (half-height segments: done before the interior)
      if (edge_mode & 1) {
        hipEventRecord(s->ev_edges, s->edge_stream);
        s->edges_recorded = true;
      }
      hipEventRecord(s->ev_join, s->edge_stream);
      group(nl, nr0 - nl);
      hipStreamWaitEvent(s->stream, s->ev_join, 0);
    }
    s->fix_check = true;
    LAUNCH_CHECK(s, "march_wet");
  }
#ifdef WX_WET_TIMING
  if (s->iter == 40) { // per segment: start offset and duration of its waves (s_memtime ticks, 100 MHz)
    hipStreamSynchronize(s->stream);
    const int ns = shape.n_strips, nseg = shape.segs.n_seg * (shape.segs.bands ? 8 : 1), per_band = shape.segs.n_seg;
    std::vector<unsigned long long> c(2 * (size_t)ns * nseg);
    hipMemcpy(c.data(), dbg_cycles, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t i = 0; i < c.size(); i += 2) { t0 = c[i] < t0 ? c[i] : t0; t1 = c[i + 1] > t1 ? c[i + 1] : t1; }
    fprintf(stderr, "[wx_wet timing] kernel span %llu ticks, %d strips x %d segments\n", t1 - t0, ns, nseg);
    for (int sg = 0; sg < nseg; sg++) {
      double st = 0, du = 0, dmin = 1e30, dmax = 0, en = 0;
      for (int k = 0; k < ns; k++) {
        const size_t i = 2 * ((size_t)sg * ns + k);
        const double d = (double)(c[i + 1] - c[i]);
        st += (double)(c[i] - t0); du += d; en += (double)(c[i + 1] - t0);
        dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax;
      }
      const int sl = sg % per_band; // (row bands: the table is relative to the band of XCD sg / per_band)
      fprintf(stderr, "  seg %3d rows %4d..%4d (%3d): start %7.0f end %7.0f  dur avg %6.0f min %6.0f max %6.0f  per row %5.1f\n", sg, shape.segs.start[sl], shape.segs.start[sl + 1],
              shape.segs.start[sl + 1] - shape.segs.start[sl], st / ns, en / ns, du / ns, dmin, dmax, du / ns / (shape.segs.start[sl + 1] - shape.segs.start[sl] + 8));
    }
  }
#endif
  std::swap(s->base[0], s->base[1]);
  std::swap(s->wall[0], s->wall[1]);
  std::swap(s->water[1], s->water[2]);
  if (!s->even) std::swap(s->lp[0], s->lp[2]);
  return WX_OK;
}

// BASELINE configs[1] (pass_mask == WX_PASS_DRY): velocity + advection + pressure in one kernel.
// The masked-off boundary pass makes water_0 a copy of water_1, so advection's water input is water[1].
// Does the dry iteration carry water (brush / airplane / sounding forcing can create it, an upload can hold it)?
static bool dry_has_water(const wx_sim *s)
{
  return !(s->water_trivial && s->p.userInputType < 0 && s->p.airplaneValues[3] == 0.0f && s->p.soundingForcing == 0.0f);
}
// ... and does it run the row-marching kernel (wx_march.h: the water-free state, strips of 60 columns), which can be launched
// per strip range, or the tiled one (wx_dry.h)?
static bool dry_marches(const wx_sim *s) { return !dry_has_water(s) && s->dry_march && s->X >= 64; }

// Does a slab run its exchange periods IN ORDER (no split iterations, wx_step_overlap's flags ignored)? The agreed water-free dry stencil
// with base-only halo messages does: its iterations then go in pairs (two per launch), which split iterations cannot.
static bool dry_runs_in_order(const wx_sim *s)
{
  const bool dry = s->fused && (s->p.pass_mask & 0x3Fu) == WX_PASS_DRY;
  const bool precip = (s->p.pass_mask & WX_PASS_PRECIPITATION) && s->p.enablePrecipitation && s->n_drops > 0;
  return dry && s->halo_base_only && s->dry_pairs && !precip && dry_marches(s) && s->wall_veg_ok && s->p.userInputType < 0 && !(s->p.airplaneValues[3] > 0.9f) &&
         s->Y >= 16;
}

// edge_mode: as iterate_march_wet (marching kernel only)
int iterate_dry(wx_sim *s, bool write_disp, int edge_mode = 0)
{
  if (write_disp) s->disp_lazy = false; // (the dry kernels store baseTexture_1 whole, into base[2])
  const bool water = dry_has_water(s);
  DryIn in{s->base[0], s->wall[0], s->water[1]};
  DryOut out{s->base[1], s->wall[1], s->water[2], s->base[2]};
  // advection changes the wall texture only through the brush, an airplane crash or by clamping a negative vegetation byte
  const bool wall_const = s->wall_veg_ok && s->p.userInputType < 0 && !(s->p.airplaneValues[3] > 0.9f);
  bool wall_written = true;
  if (dry_marches(s)) {
    ProfScope ps(s, K_MARCH_DRY);
    const int n_strips = march_dry_strips(s->geo);
    // edge strips: every output column wx_halo_pack reads ([halo, 2*halo) and its mirror) and every strip that reads ghost columns
    const int nl = s->halo > 0 ? (2 * s->halo - 1) / MOUT + 1 : 0, nr0 = s->halo > 0 ? (s->X - 2 * s->halo) / MOUT : n_strips;
    const VxTrack vt = vx_track(s);
    auto launch_on = [&](hipStream_t st, int lo, int cnt, int lo2 = 0, int cnt2 = 0) {
      launch_march_dry(s->geo, s->uni, s->full_ctx, in, out, write_disp, !wall_const, st, lo, cnt, lo2, cnt2, nullptr, &vt);
    };
    auto launch = [&](int lo, int cnt) { launch_on(s->stream, lo, cnt); };
    if (edge_mode == 0 || s->halo == 0 || nl >= nr0) {
      if (edge_mode & 2) wait_unpacked(s);
      launch(0, -1);
    } else if (s->split_launch) { // one ordered launch with device-side hand-offs (see iterate_march_wet)
      if (int rc = split_launch_ready(s, false)) return rc;
      StripOrder ord{};
      ord.mode = (edge_mode & 1) ? 1 : 2;
      ord.nl = nl;
      ord.nr0 = nr0;
      if (edge_mode == 3) wait_unpacked(s);
      if ((edge_mode & 2) && s->unpack_pending) {
        ord.epoch = s->sync_words + 1;
        ord.epoch_want = s->epoch_host;
        s->unpack_pending = false;
      }
      if (edge_mode & 1) ord.arrive = s->sync_words;
      ord.prio = WX_SPLIT_PRIO;
      if (const char *e = wx_tune_env("WX_SPLIT_PRIO")) ord.prio = atoi(e);
      if (const char *e = wx_tune_env("WX_SPLIT_NOFENCE")) ord.nofence = atoi(e);
      s->split_check = true;
      const int items = launch_march_dry(s->geo, s->uni, s->full_ctx, in, out, write_disp, !wall_const, s->stream, 0, -1, 0, 0, &ord, &vt);
      if (edge_mode & 1) {
        s->arrive_target += (unsigned)items;
        s->gate_pending = true;
        s->gate_wet = false;
      }
    } else { // (rounds 2-4) edge strips and interior side by side (see iterate_march_wet)
      if (int rc = edge_stream_ready(s, false)) return rc;
      hipEventRecord(s->ev_fork, s->stream);
      hipStreamWaitEvent(s->edge_stream, s->ev_fork, 0);
      if ((edge_mode & 2) && s->unpack_pending) {
        hipStreamWaitEvent(s->edge_stream, s->ev_unpacked, 0);
        s->unpack_pending = false;
      }
      launch_on(s->edge_stream, 0, nl, nr0, n_strips - nr0); // (both edges in one launch)
      if (edge_mode & 1) {
        hipEventRecord(s->ev_edges, s->edge_stream);
        s->edges_recorded = true;
      }
      hipEventRecord(s->ev_join, s->edge_stream);
      launch(nl, nr0 - nl);
      hipStreamWaitEvent(s->stream, s->ev_join, 0);
    }
    LAUNCH_CHECK(s, "march_dry");
    wall_written = !wall_const;
  } else {
    ProfScope ps(s, K_FUSED_DRY);
    s->vx_untracked = true; // (the tiled kernel does not report its |vx|: the exchange scans the state instead)
    launch_fused_dry(s->geo, s->uni, s->full_ctx, in, out, water, write_disp, s->stream);
    LAUNCH_CHECK(s, "fused_dry");
  }
  std::swap(s->base[0], s->base[1]);
  if (wall_written) std::swap(s->wall[0], s->wall[1]);
  if (water) {
    // reference ping-pong with the boundary pass masked off: water_0 = previous water_1, water_1 = advected water
    std::swap(s->water[0], s->water[1]);
    std::swap(s->water[1], s->water[2]);
  }
  return WX_OK;
}

// The reference clears the feedback / deposition textures every iteration (app.js:5933-5937); here they are rewritten by the box sum while
// particles run, and cleared ONCE by the first iteration after they were switched off (whichever kernel runs it).
static void clear_particle_textures(wx_sim *s)
{
  if (!s->fb_dirty) return;
  const size_t n = ncell(s);
  hipMemsetAsync(s->fb, 0, n * 12, s->stream);
  hipMemsetAsync(s->dep, 0, n * 8, s->stream);
  if (s->sg.fb_zero) hipMemsetAsync(s->sg.fb_zero, 1, 2 * (size_t)s->sg.TXn * s->sg.TYn, s->stream);
  s->fb_dirty = false;
}

// Two iterations of the water-free dry stencil in one launch (wx_march2.h): base[0] -> base[1], one swap; the wall texture is constant
// (the caller checked). write_disp: the post-advection base of the SECOND iteration goes to base[2].
int iterate_dry_pair(wx_sim *s, bool write_disp)
{
  if (!s->pair_ctl) { // (once per handle) the second iteration's exact-path list: room for 1/64 of the grid's cells, 64 Ki .. 1 Mi entries
    size_t cap = std::min<size_t>(std::max<size_t>(ncell(s) / 64, 1u << 16), 1u << 20);
    if (s->fix_cap_request > 0) cap = (size_t)s->fix_cap_request; // (WX_OPT_FIX_CAP; tests: provoke the overflow -> the whole pair is repeated)
    if (const char *e = wx_tune_env("WX_MARCH2_FIX_CAP")) cap = (size_t)atoi(e); // (0: every recorded cell repeats the whole pair -- round 5's behaviour, for A/B timing)
    if (hipMalloc((void **)&s->pair_ctl, D2_WORDS * sizeof(int)) != hipSuccess || hipMalloc((void **)&s->pair_cells, std::max<size_t>(cap, 1) * sizeof(int2)) != hipSuccess ||
        hipMemsetAsync(s->pair_ctl, 0, D2_WORDS * sizeof(int), s->stream) != hipSuccess)
      return fail(s, WX_E_NOMEM, "wx_step: %zu bytes for the pair kernel's exact-path list", cap * sizeof(int2));
    s->pair_cap = (int)cap;
    s->pair_epoch = 0;
    if (hipHostMalloc((void **)&s->pair_hint_host, sizeof(int), hipHostMallocMapped) == hipSuccess && s->pair_hint_host) {
      *s->pair_hint_host = 1;
      if (hipHostGetDevicePointer((void **)&s->pair_hint_dev, s->pair_hint_host, 0) != hipSuccess) s->pair_hint_dev = nullptr;
      static const int one = 1; // (ctl[D2_TOLD]: what the hint word was last told -- the first empty fix launch sets it back to 0)
      if (s->pair_hint_dev) hipMemcpyAsync(s->pair_ctl + D2_TOLD, &one, 4, hipMemcpyHostToDevice, s->stream);
    } else {
      s->pair_hint_host = nullptr;
      (void)hipGetLastError();
    }
  }
  if (write_disp) s->disp_lazy = false;
  DryIn in{s->base[0], s->wall[0], s->water[1]};
  DryOut out{s->base[1], s->wall[1], s->water[2], s->base[2]};
  {
    ProfScope ps(s, K_MARCH_DRY2);
    const VxTrack vt = vx_track(s);
    if (++s->pair_epoch == 0x7fffffff) { // (the device keeps the LARGEST epoch that asked for a repeat: start over, a few times per century)
      hipMemsetAsync(s->pair_ctl + D2_REDO_EPOCH, 0, 4, s->stream);
      s->pair_epoch = 1;
    }
    const Dry2Fix fix{s->pair_ctl, s->pair_cells, s->pair_cap, s->pair_epoch, s->pair_hint_dev, s->pair_hint_dev ? s->pair_hint_host : nullptr};
    // (water[2] -- the spare of the wet ping-pong, dead in the water-free dry state -- holds the intermediate state of a pair repeated whole)
    launch_march_dry2(s->geo, s->uni, s->full_ctx, in, out, write_disp, s->stream, &vt, fix, s->water[2]);
    LAUNCH_CHECK(s, "march_dry2");
  }
  std::swap(s->base[0], s->base[1]);
  return WX_OK;
}

} // namespace

extern "C" {

int wx_abi_version(void) { return WX_ABI_VERSION; }

#ifndef WX_FAST_ARITH
#define WX_FAST_ARITH 0
#endif
int wx_arith(void) { return WX_FAST_ARITH ? WX_ARITH_FAST : WX_ARITH_EXACT; }

const char *wx_last_error(const wx_sim *s) { return s ? s->err.c_str() : g_create_error.c_str(); }

int wx_kernel_count(void) { return K_COUNT; }
const char *wx_kernel_name(int k) { return (k >= 0 && k < K_COUNT) ? kKernelNames[k] : ""; }

// process-wide defaults of the options a handle takes at creation (wx_set_option(NULL, ...))
static int g_opt_kernel_set = 1, g_opt_dry_kernel = 1, g_opt_row_bands = 1, g_opt_fix_cap = 0, g_opt_place_tries = 6;

int wx_create_slab(int X_global, int Y, int x0, int X_owned, int halo, int n_droplets, wx_sim **out)
{
  if (!out) return fail(nullptr, WX_E_INVALID, "wx_create: out is NULL");
  *out = nullptr;
  if (X_global < 2 || Y < 4 || X_owned < 1 || X_owned > X_global || halo < 0 || n_droplets < 0 || X_global > 65535 * 16 || Y > 65535)
    return fail(nullptr, WX_E_INVALID, "wx_create: bad geometry X_global=%d Y=%d X_owned=%d halo=%d n_droplets=%d", X_global, Y, X_owned, halo,
                n_droplets);
  if (halo > 0 && X_owned < halo) return fail(nullptr, WX_E_INVALID, "wx_create_slab: X_owned (%d) < halo (%d)", X_owned, halo);
  if (halo == 0 && X_owned != X_global)
    return fail(nullptr, WX_E_INVALID, "wx_create_slab: a slab narrower than the domain needs halo > 0");
  if (n_droplets > 0 && halo > 0 && (halo % 64 || X_owned % 64 || X_owned + 2 * halo > X_global))
    return fail(nullptr, WX_E_INVALID,
                "wx_create_slab: particles on a slab need halo and X_owned to be multiples of 64 (the sprite clip at the domain edge must fall "
                "on a splat tile boundary) and X_owned + 2*halo <= X_global; got halo=%d X_owned=%d X_global=%d", halo, X_owned, X_global);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, WX_E_DEVICE, "no HIP device available (%s): libwxsim has no CPU fallback", hipGetErrorString(e));
  wx_sim *s = new wx_sim();
  (void)hipGetDevice(&s->device);
  s->X = X_owned + 2 * halo;
  s->Y = Y;
  s->Xg = X_global;
  s->x0 = x0;
  s->halo = halo;
  s->n_drops = n_droplets;
  s->fused = g_opt_kernel_set ? 2 : 0;
  s->dry_march = g_opt_dry_kernel;
  s->bands_mode = g_opt_row_bands;
  s->fix_cap_request = g_opt_fix_cap;
  s->place_tries = g_opt_place_tries;
  const size_t n = ncell(s);
  int rc = WX_OK;
  { // everything allocated below, to the byte (plus the splat grids of handles with droplets)
    const size_t nd = (size_t)(n_droplets > 0 ? n_droplets : 1);
    size_t total = n * (3 * 16 + 3 * 16 + 2 * 16 + 2 * 4 + 4 + 8 + 16 + 8 + 3 * 16 + (n_droplets > 0 ? 4 : 0)) + 2 * nd * 20 + 4 * ((size_t)Y + 8) * 4 + (size_t)s->X * 16 + 65536;
    if (n_droplets > 0) total += ((size_t)s->X + 8) * ((size_t)Y + 8) * 24 + n / 64 + (size_t)n_droplets * 8 + (1u << 20);
    rc = storage_begin(s, total, (size_t)16 << 20);
  }
  for (int i = 0; i < 2 && rc == WX_OK; i++) {
    if ((rc = dalloc(s, &s->base[i], n))) break;
    if ((rc = dalloc(s, &s->water[i], n))) break;
    if ((rc = dalloc(s, &s->light[i], n))) break;
    if ((rc = dalloc(s, &s->wall[i], n))) break;
    if ((rc = dalloc(s, &s->drops[i], (size_t)(n_droplets > 0 ? n_droplets : 1) * 5))) break;
  }
  if (rc == WX_OK) rc = dalloc(s, &s->base[2], n);
  if (rc == WX_OK) rc = dalloc(s, &s->curl, n);
  if (rc == WX_OK) rc = dalloc(s, &s->vort, n);
  if (rc == WX_OK) rc = dalloc(s, &s->fb, n);
  if (rc == WX_OK) rc = dalloc(s, &s->dep, n);
  if (rc == WX_OK) rc = dalloc(s, &s->initial_T, (size_t)Y + 8);
  if (rc == WX_OK) rc = dalloc(s, &s->snd_T, (size_t)Y + 8);
  if (rc == WX_OK) rc = dalloc(s, &s->snd_W, (size_t)Y + 8);
  if (rc == WX_OK) rc = dalloc(s, &s->snd_Vel, (size_t)Y + 8);
  if (rc == WX_OK) rc = dalloc(s, &s->state, 1);
  if (rc == WX_OK) rc = dalloc(s, &s->tdisp, n);
  if (rc == WX_OK) rc = dalloc(s, &s->pdisp, n);
  for (int i = 0; i < 3 && rc == WX_OK; i++) {
    if ((rc = dalloc(s, &s->lp[i].x, n))) break;
    if ((rc = dalloc(s, &s->lp[i].y, n))) break;
    rc = dalloc(s, &s->lp[i].zw, n);
  }
  if (rc == WX_OK) rc = dalloc(s, &s->full_ctx, 1);
  if (rc == WX_OK) rc = dalloc(s, &s->zero_row, (size_t)s->X);
  if (rc == WX_OK) rc = dalloc(s, &s->water[2], n);
  if (rc == WX_OK && n_droplets > 0) {
    SplatGrid &sg = s->sg;
    sg.AP = s->X + 8;
    sg.AH = Y + 8;
    sg.TXn = (s->X + 2 + STX - 1) / STX; // anchors 0 .. X (+1 right of a seam)
    sg.TYn = (Y + 1 + STY - 1) / STY;
    rc = dalloc(s, &sg.acc3, (size_t)sg.AP * sg.AH);
    if (rc == WX_OK) rc = dalloc(s, &sg.acc2, (size_t)sg.AP * sg.AH);
    if (rc == WX_OK) rc = dalloc(s, &sg.dirty, 2 * (size_t)sg.TXn * sg.TYn);   // [t]: deposits, [T + t]: rain / snow deposits
    if (rc == WX_OK) rc = dalloc(s, &sg.fb_zero, 2 * (size_t)sg.TXn * sg.TYn); // [2t]: feedback tile zero, [2t + 1]: deposition tile zero
    if (rc == WX_OK) rc = dalloc(s, &sg.work, 16 + 5 * (size_t)sg.TXn * sg.TYn);
    if (rc == WX_OK && halo > 0) rc = dalloc(s, &s->pool_remote, (size_t)n_droplets);
    if (rc == WX_OK && halo > 0) rc = dalloc(s, &s->pool_owned, (size_t)n_droplets);
    if (rc == WX_OK && halo > 0) rc = dalloc(s, &s->pool_flips, (size_t)n_droplets);
    if (rc == WX_OK && halo > 0) rc = dalloc(s, &s->pool_best, (size_t)n_droplets);
    if (rc == WX_OK && halo > 0) {
      if (hipMemset(s->pool_best, 0x7f, (size_t)n_droplets * 4) != hipSuccess) rc = WX_E_DEVICE; // (0x7f7f7f7f: larger than any event key)
      // capacities of the exchange buffers: status flips of one period (a few hundred in a storm, at most every droplet) and the
      // droplets within `halo` columns of an edge (generous, fixed, checked: an overflow is reported by the next blocking call)
      s->pool_event_cap = n_droplets; // (worst case: every droplet flips in one period -- the start-up burst of an all-inactive pool; the
                                      // hosts only transfer the filled part, see wx_pool_events_apply)
      s->pool_edge_cap = std::max(4096, n_droplets / 8);
      int xoff = (x0 - halo) % X_global;
      if (xoff < 0) xoff += X_global;
      s->seam = (xoff > 0 && xoff + s->X > X_global) ? X_global - xoff : 0;
    }
  }
  if (rc != WX_OK) {
    g_create_error = s->err;
    wx_destroy(s);
    return rc == WX_E_DEVICE ? WX_E_NOMEM : rc;
  }
  s->p.pass_mask = WX_PASS_ALL;
  build_geo(s);
  // The allocations above were zeroed with hipMemset on the default stream, which may return before the device is done. A host that
  // then works on a NON-BLOCKING stream of its own (wx_set_stream; the slabs of a wx_group) is not ordered behind the default stream:
  // its first upload could be overwritten by a zeroing that is still in flight (seen: a slab of a freshly created group reading back
  // as zeros). The handle is complete when this returns.
  if (hipStreamSynchronize(nullptr) != hipSuccess) {
    g_create_error = "wx_create: device error while zeroing the handle's storage";
    wx_destroy(s);
    return WX_E_DEVICE;
  }
  *out = s;
  return WX_OK;
}

int wx_create(int X, int Y, int n_droplets, wx_sim **out) { return wx_create_slab(X, Y, 0, X, 0, n_droplets, out); }

static void water0_scratch_free(wx_sim *s);
static int materialize_water0(wx_sim *s);

void wx_destroy(wx_sim *s)
{
  if (!s) return;
  DeviceScope ds(s);
  if (s->stream) hipStreamSynchronize(s->stream);
  else hipDeviceSynchronize();
  if (s->comm_stream) hipStreamSynchronize(s->comm_stream);
  transport_release(s);
  for (auto &r : s->prof) {
    if (!r.shared_a) hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  for (auto e : s->ev_pool) hipEventDestroy(e);
  for (int i = 0; i < 2; i++) {
    dfree(s, s->base[i]);
    dfree(s, s->water[i]);
    dfree(s, s->light[i]);
    dfree(s, s->wall[i]);
    dfree(s, s->drops[i]);
  }
  water0_scratch_free(s);
  hipFree(s->w0_initT);
  s->w0_initT = nullptr;
  dfree(s, s->base[2]);
  dfree(s, s->curl);
  hipFree(s->emitted);
  dfree(s, s->vort);
  dfree(s, s->fb);
  if (s->fb_rgba) hipFree(s->fb_rgba);
  s->fb_rgba = nullptr;
  dfree(s, s->dep);
  dfree(s, s->initial_T);
  dfree(s, s->snd_T);
  dfree(s, s->snd_W);
  dfree(s, s->snd_Vel);
  if (s->copy_stream) {
    hipStreamSynchronize(s->copy_stream);
    hipEventDestroy(s->ev_fields_ready);
    hipEventDestroy(s->ev_copy_done);
    hipStreamDestroy(s->copy_stream);
  }
  if (s->ev_edges) {
    hipEventDestroy(s->ev_edges);
    hipEventDestroy(s->ev_unpacked);
  }
  if (s->edge_stream) {
    hipStreamSynchronize(s->edge_stream);
    hipEventDestroy(s->ev_fork);
    hipEventDestroy(s->ev_join);
    hipStreamDestroy(s->edge_stream);
  }
  hipFree(s->fix_count2);
  hipFree(s->fix_cells2);
  hipFree(s->sync_words);
  hipFree(s->pair_ctl);
  hipFree(s->pair_cells);
  if (s->pair_hint_host) hipHostFree(s->pair_hint_host);
  dfree(s, s->state);
  dfree(s, s->pool_remote);
  dfree(s, s->pool_owned);
  dfree(s, s->pool_flips);
  dfree(s, s->pool_best);
  dfree(s, s->tdisp);
  dfree(s, s->pdisp);
  for (int i = 0; i < 3; i++) {
    dfree(s, s->lp[i].x);
    dfree(s, s->lp[i].y);
    dfree(s, s->lp[i].zw);
  }
  dfree(s, s->full_ctx);
  dfree(s, s->zero_row);
  for (void *pad : s->alloc_pads) hipFree(pad);
  s->alloc_pads.clear();
  hipFree(s->fix_count);
  hipFree(s->fix_cells);
  if (s->fix_hint_host) hipHostFree(s->fix_hint_host);
  dfree(s, s->water[2]);
  dfree(s, s->sg.acc3);
  dfree(s, s->sg.acc2);
  dfree(s, s->sg.dirty);
  dfree(s, s->sg.fb_zero);
  dfree(s, s->sg.work);
  for (int i = 0; i < 2; i++) {
    dfree(s, s->det_key[i]);
    dfree(s, s->det_idx[i]);
  }
  dfree(s, s->det_val);
  hipFree(s->det_tmp);
  for (Block &b : s->blocks) hipFree(b.p);
  delete s;
}

static int reset_after_upload(wx_sim *s, const float *drops);
// a halo unpack still running on the comm stream would overwrite freshly uploaded ghost columns: drain it first
static int drain_comm(wx_sim *s)
{
  if (s->comm_stream) {
    HIPCHK(s, hipStreamSynchronize(s->comm_stream));
    s->unpack_pending = s->edges_recorded = false;
  }
  return WX_OK;
}

int wx_upload(wx_sim *s, const float *base, const float *water, const int8_t *wall, const float *drops)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (!base || !water || !wall) return fail(s, WX_E_INVALID, "wx_upload: NULL grid array");
  if (int rc = drain_comm(s)) return rc;
  const size_t n = ncell(s);
  {
    bool trivial = true;
    for (size_t i = 0; i < n && trivial; i++) {
      const float *w = water + 4 * i;
      const int8_t *wl = wall + 4 * i;
      const float x_expected = wl[1] == 0 ? (wl[0] == 2 ? 1002.0f : 1001.0f) : 0.0f;
      trivial = (w[0] == x_expected) && w[1] == 0.0f && w[2] == 0.0f && w[3] == 0.0f && wl[3] >= 0;
    }
    s->local_water_free = trivial;
    // a slab may rely on it only once the host has established it for the neighbours too (their ghost columns flow in)
    s->water_trivial = trivial && (s->halo == 0 || s->slab_dry_agreed);
    bool veg_ok = true;
    for (size_t i = 0; i < n && veg_ok; i++) veg_ok = wall[4 * i + 3] >= 0;
    s->wall_veg_ok = veg_ok;
  }
  for (int i = 0; i < 2; i++) {
    HIPCHK(s, hipMemcpyAsync(s->base[i], base, n * 16, hipMemcpyHostToDevice, s->stream));
    HIPCHK(s, hipMemcpyAsync(s->water[i], water, n * 16, hipMemcpyHostToDevice, s->stream));
    HIPCHK(s, hipMemcpyAsync(s->wall[i], wall, n * 4, hipMemcpyHostToDevice, s->stream));
  }
  return reset_after_upload(s, drops);
}

// partitioned pool of a slab handle: every rank was handed the whole pool; keep what lies in the local array (owned + ghost columns)
static int pool_reset(wx_sim *s)
{
  if (!s->pool_remote) return WX_OK;
  HIPCHK(s, hipMemsetAsync(s->pool_remote, 0, (size_t)s->n_drops, s->stream));
  HIPCHK(s, hipMemsetAsync(s->pool_owned, 0, (size_t)s->n_drops, s->stream));
  HIPCHK(s, hipMemsetAsync(s->pool_flips, 0, (size_t)s->n_drops * 2, s->stream));
  hipLaunchKernelGGL(k_pool_edges_pack, dim3((s->n_drops + 255) / 256), dim3(256), 0, s->stream, s->geo, s->n_drops, 0, s->X, 0, 0, s->drops[0], s->pool_remote,
                     nullptr, nullptr, nullptr, nullptr, s->state, 0);
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}

// initRainDrops() on the device (k_init_droplets): a fresh, all-inactive pool from a seed; the grid is untouched
int wx_init_droplets(wx_sim *s, uint32_t seed)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (s->n_drops <= 0) return fail(s, WX_E_STATE, "wx_init_droplets: the handle has no droplets");
  if (int rc = drain_comm(s)) return rc;
  hipLaunchKernelGGL(k_init_droplets, dim3((s->n_drops + 255) / 256), dim3(256), 0, s->stream, s->n_drops, seed, s->drops[0], s->drops[1]);
  HIPCHK(s, hipGetLastError());
  if (int rc = pool_reset(s)) return rc;
  HIPCHK(s, hipStreamSynchronize(s->stream));
  s->period_j = 0;
  s->exact_pending = 0;
  return WX_OK;
}

// everything wx_upload resets besides the three grids (shared with wx_setup_columns)
static int reset_after_upload(wx_sim *s, const float *drops)
{
  const size_t n = ncell(s);
  s->water0_pending = false; // (waterTexture_0 is what was uploaded)
  for (int i = 0; i < 2; i++) {
    HIPCHK(s, hipMemsetAsync(s->light[i], 0, n * 16, s->stream));
    if (drops && s->n_drops > 0)
      HIPCHK(s, hipMemcpyAsync(s->drops[i], drops, (size_t)s->n_drops * 20, hipMemcpyHostToDevice, s->stream));
  }
  HIPCHK(s, hipMemsetAsync(s->curl, 0, n * 4, s->stream));
  HIPCHK(s, hipMemsetAsync(s->vort, 0, n * 8, s->stream));
  HIPCHK(s, hipMemsetAsync(s->fb, 0, n * 12, s->stream));
  HIPCHK(s, hipMemsetAsync(s->dep, 0, n * 8, s->stream));
  HIPCHK(s, hipMemsetAsync(s->state->lightning, 0, 16, s->stream));
  HIPCHK(s, hipMemsetAsync(&s->state->px_count, 0, 60, s->stream)); // px_count, px_light[4], scratch_int, ghost_nontrivial, fix_overflow, pool_overflow, pool_seen_max, fastest_bits, pool_retired, mailbox_w, vx_max_bits, cone_violation
  if (int rc = pool_reset(s)) return rc;
  if (s->sg.acc3) {
    HIPCHK(s, hipMemsetAsync(s->sg.acc3, 0, (size_t)s->sg.AP * s->sg.AH * 12, s->stream));
    HIPCHK(s, hipMemsetAsync(s->sg.acc2, 0, (size_t)s->sg.AP * s->sg.AH * 8, s->stream));
    HIPCHK(s, hipMemsetAsync(s->sg.dirty, 0, 2 * (size_t)s->sg.TXn * s->sg.TYn, s->stream));
    HIPCHK(s, hipMemsetAsync(s->sg.fb_zero, 1, 2 * (size_t)s->sg.TXn * s->sg.TYn, s->stream));
    HIPCHK(s, hipMemsetAsync(s->sg.work, 0, 64, s->stream));
    s->splat_par = 0;
  }
  HIPCHK(s, hipStreamSynchronize(s->stream)); // the caller keeps ownership of the host arrays
  s->fb_dirty = false;
  s->air_from_row = -1;
  s->ghost_check = false;
  s->light_planar = false; // the interleaved light textures were just zeroed
  s->period_j = 0;
  s->exact_pending = 0;
  s->since_exchange = 0; // (wx_slab_step / wx_group_step: the ghost columns are fresh, and not from an exchange)
  s->exchanged = false;
  s->vx_stale = true;    // (whatever velocities the new state holds: looked at before the next exchange period is sized)
  s->vx_untracked = false;
  s->vx_have[0] = s->vx_have[1] = false;
  s->ran_fused = false;
  s->even = 1;
  s->emit_lit = false; // (the reference re-creates the emittedLight texture with the others)
  s->drop_cur = 0;
  s->uploaded = true;
  return WX_OK;
}

// Device-side initialiser of a new simulation: what the reference's setup pass draws (setupShader.frag:36-92), filled
// from the per-column terrain description and the per-row sounding instead of uploading three X*Y textures.
// device scratch of the setup fill: the column descriptors and the per-row sounding
struct SetupScratch {
  char *mem = nullptr;
  double *veg;
  int32_t *rows;
  float *snow, *T, *tot, *cloud;
  unsigned int *sea;
};
static int setup_scratch(wx_sim *s, SetupScratch &q, const char *who)
{
  const int X = s->X, Y = s->Y;
  const size_t bytes = (size_t)X * (4 + 8 + 4 + 4) + (size_t)Y * 12;
  if (hipMalloc(&q.mem, bytes) != hipSuccess) return fail(s, WX_E_NOMEM, "%s: %zu bytes of device scratch", who, bytes);
  q.veg = (double *)q.mem;
  q.rows = (int32_t *)(q.veg + X);
  q.snow = (float *)(q.rows + X);
  q.T = q.snow + X;
  q.tot = q.T + Y;
  q.cloud = q.tot + Y;
  q.sea = (unsigned int *)(q.cloud + Y);
  return WX_OK;
}
// descriptors on the device -> textures (k_setup_columns), both buffers of each pair, state reset as after wx_upload
static int setup_fill(wx_sim *s, SetupScratch &q, const float *T_air, const float *total_water, const float *cloud_water, const float *drops, const char *who)
{
  const int X = s->X, Y = s->Y;
  hipMemcpyAsync(q.T, T_air, (size_t)Y * 4, hipMemcpyHostToDevice, s->stream);
  hipMemcpyAsync(q.tot, total_water, (size_t)Y * 4, hipMemcpyHostToDevice, s->stream);
  hipMemcpyAsync(q.cloud, cloud_water, (size_t)Y * 4, hipMemcpyHostToDevice, s->stream);
  hipLaunchKernelGGL(k_setup_columns, grid2d(s), dim3(BX, BY), 0, s->stream, X, Y, q.rows, q.sea, q.veg, q.snow, q.T, q.tot, q.cloud, s->base[0], s->water[0],
                     s->wall[0]);
  const size_t n = ncell(s);
  hipMemcpyAsync(s->base[1], s->base[0], n * 16, hipMemcpyDeviceToDevice, s->stream);
  hipMemcpyAsync(s->water[1], s->water[0], n * 16, hipMemcpyDeviceToDevice, s->stream);
  hipMemcpyAsync(s->wall[1], s->wall[0], n * 4, hipMemcpyDeviceToDevice, s->stream);
  hipError_t e = hipStreamSynchronize(s->stream); // the caller's arrays may go away now
  hipFree(q.mem);
  q.mem = nullptr;
  if (e != hipSuccess) return fail(s, WX_E_DEVICE, "%s: %s", who, hipGetErrorString(e));
  s->water_trivial = s->local_water_free = false;
  s->wall_veg_ok = true; // k_setup_columns clamps the vegetation to 0..127
  return reset_after_upload(s, drops);
}

int wx_setup_columns(wx_sim *s, const int32_t *wall_rows, const uint8_t *sea_column, const double *veg_noise, const float *snow,
                     const float *T_air, const float *total_water, const float *cloud_water, const float *drops)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (!wall_rows || !sea_column || !veg_noise || !snow || !T_air || !total_water || !cloud_water)
    return fail(s, WX_E_INVALID, "wx_setup_columns: NULL descriptor array");
  if (int rc = drain_comm(s)) return rc;
  const int X = s->X, Y = s->Y;
  for (int x = 0; x < X; x++)
    if (wall_rows[x] < 0 || wall_rows[x] > Y) return fail(s, WX_E_RANGE, "wx_setup_columns: wall_rows[%d] = %d outside 0..%d", x, wall_rows[x], Y);
  static_assert(sizeof(double) == 8, "");
  SetupScratch q;
  if (int rc = setup_scratch(s, q, "wx_setup_columns")) return rc;
  std::vector<unsigned int> sea32(X);
  for (int x = 0; x < X; x++) sea32[x] = sea_column[x] ? 1u : 0u;
  hipMemcpyAsync(q.veg, veg_noise, (size_t)X * 8, hipMemcpyHostToDevice, s->stream);
  hipMemcpyAsync(q.rows, wall_rows, (size_t)X * 4, hipMemcpyHostToDevice, s->stream);
  hipMemcpyAsync(q.snow, snow, (size_t)X * 4, hipMemcpyHostToDevice, s->stream);
  hipMemcpyAsync(q.sea, sea32.data(), (size_t)X * 4, hipMemcpyHostToDevice, s->stream);
  return setup_fill(s, q, T_air, total_water, cloud_water, drops, "wx_setup_columns"); // (synchronises: sea32 stays alive until then)
}

int wx_setup_terrain(wx_sim *s, double seed, double height_mult, int snap, double sim_height, const float *T_air, const float *total_water,
                     const float *cloud_water, const float *drops)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (!T_air || !total_water || !cloud_water) return fail(s, WX_E_INVALID, "wx_setup_terrain: NULL sounding array");
  if (snap < 1 || !(sim_height > 0.0) || !(height_mult >= 0.0) || !(seed == seed))
    return fail(s, WX_E_INVALID, "wx_setup_terrain: snap >= 1, sim_height > 0, height_mult >= 0 (got snap=%d sim_height=%g height_mult=%g seed=%g)", snap, sim_height, height_mult, seed);
  if (s->Y < 16) return fail(s, WX_E_INVALID, "wx_setup_terrain: at least 16 rows (the terrain leaves 8 rows of air)");
  if (int rc = drain_comm(s)) return rc;
  SetupScratch q;
  if (int rc = setup_scratch(s, q, "wx_setup_terrain")) return rc;
  hipLaunchKernelGGL(k_terrain_columns, dim3((s->X + 255) / 256), dim3(256), 0, s->stream, s->X, s->Xg, s->x0 - s->halo, s->Y, seed, height_mult, snap, sim_height, q.rows,
                     q.sea, q.veg, q.snow);
  return setup_fill(s, q, T_air, total_water, cloud_water, drops, "wx_setup_terrain");
}

int wx_set_params(wx_sim *s, const wx_params *p, const float *initial_T, const float *sounding_T, const float *sounding_W,
                  const float *sounding_Vel)
{
  if (!s || !p) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (!initial_T && !s->have_params) return fail(s, WX_E_INVALID, "wx_set_params: initial_T is required on the first call");
  s->p = *p;
  build_geo(s);
  build_uni(s);
  const size_t nb = ((size_t)s->Y + 1) * 4;
  if (initial_T && s->water0_pending && !s->w0_initT_saved) { // waterTexture_0 of the last iteration is still to be made: with ITS initial_T
    if (!s->w0_initT) HIPCHK(s, hipMalloc((void **)&s->w0_initT, nb));
    HIPCHK(s, hipMemcpyAsync(s->w0_initT, s->initial_T, nb, hipMemcpyDeviceToDevice, s->stream));
    s->w0_initT_saved = true;
  }
  if (initial_T) HIPCHK(s, hipMemcpyAsync(s->initial_T, initial_T, nb, hipMemcpyHostToDevice, s->stream));
  if (sounding_T) HIPCHK(s, hipMemcpyAsync(s->snd_T, sounding_T, nb, hipMemcpyHostToDevice, s->stream));
  if (sounding_W) HIPCHK(s, hipMemcpyAsync(s->snd_W, sounding_W, nb, hipMemcpyHostToDevice, s->stream));
  if (sounding_Vel) HIPCHK(s, hipMemcpyAsync(s->snd_Vel, sounding_Vel, nb, hipMemcpyHostToDevice, s->stream));
  {
    FullCtx fc{s->geo, s->uni, s->initial_T, s->snd_T, s->snd_W, s->snd_Vel};
    HIPCHK(s, hipMemcpyAsync(s->full_ctx, &fc, sizeof(fc), hipMemcpyHostToDevice, s->stream));
  }
  if (p->inactiveDroplets >= 0.0f)
    HIPCHK(s, hipMemcpyAsync(&s->state->inactiveDroplets, &p->inactiveDroplets, 4, hipMemcpyHostToDevice, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  s->have_params = true;
  return WX_OK;
}

int wx_step(wx_sim *s, int n_iter) { return wx_step_overlap(s, n_iter, 0u); }

int wx_step_overlap(wx_sim *s, int n_iter, unsigned flags)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (n_iter < 0) return fail(s, WX_E_INVALID, "wx_step: n_iter < 0");
  if (!s->uploaded || !s->have_params) return fail(s, WX_E_STATE, "wx_step before wx_upload / wx_set_params");
  struct TailReset { // profile scopes share their boundary events only within one call (between calls the stream may idle)
    wx_sim *s;
    ~TailReset()
    {
      s->prof_tail = nullptr;
      s->prof_chain = false;
    }
  } tail_reset{s};
  s->prof_tail = nullptr;
  s->prof_chain = true;
  // Where the planes lie in physical memory is worth +-8 % of the iteration (include/wxsim.h, "Placement tuning"): big whole-domain handles
  // look for a good set of allocations once, before their first iteration (state and counters are untouched; round-5 verdict: a host that
  // follows INTEGRATION.md should not have to know). A failed search (no room for two more copies of the state) is not an error of the step.
  if (!s->place_done && !s->place_busy && n_iter > 0) {
    s->place_done = true;
    if (s->place_tries > 0 && s->halo == 0 && ncell(s) >= WX_PLACEMENT_AUTO_CELLS && !s->blocks.empty()) {
      const std::string err_keep = s->err;
      if (wx_tune_placement(s, s->place_tries, 20, nullptr, nullptr) != WX_OK) {
        s->err = err_keep;
        (void)hipGetLastError();
      }
    }
  }
  settle_edges(s); // (the previous call ended with an edges-first iteration that no halo pack followed)
  const unsigned mask = s->p.pass_mask;
  const bool precip = (mask & WX_PASS_PRECIPITATION) && s->p.enablePrecipitation && s->n_drops > 0;
  // the marching kernel implements the full grid-pass set; any other pass_mask (but the dry one) runs the per-pass kernels
  const bool fused = s->fused && (mask & 0x3Fu) == 0x3Fu;
  const bool dry = s->fused && (mask & 0x3Fu) == WX_PASS_DRY;
  if (s->copy_in_flight) { // a streamed frame still reads the display fields: order this step after it (device-side wait)
    HIPCHK(s, hipStreamWaitEvent(s->stream, s->ev_copy_done, 0));
    s->copy_in_flight = false;
  }
  // the vegetation byte stays known non-negative only while nothing but the brush-free dry iteration touches the wall texture
  if (!dry || s->p.userInputType >= 0 || s->p.airplaneValues[3] > 0.9f) s->wall_veg_ok = false;
  // ... and the water texture stays known trivial only through water-free dry iterations (iterate_dry's `water == false` case)
  // (local_water_free goes with it: it is what wx_slab_assert_water_free re-arms water_trivial from, and after a step that can create
  // water only a new upload may establish it again -- ADVICE round 3)
  if (!dry || s->p.userInputType >= 0 || s->p.airplaneValues[3] != 0.0f || s->p.soundingForcing != 0.0f) s->water_trivial = s->local_water_free = s->halo_base_only = false;
  if (s->p.userInputType >= 10) s->air_from_row = -1; // wall tools: the terrain may grow (re-measured after the call)
  if (precip && s->pool_remote && s->pool_exact && n_iter > 0 && (n_iter > 1 || s->exact_pending > 0))
    return fail(s, WX_E_STATE, "wx_step: WX_OPT_POOL_EXACT takes one iteration per call, each followed by wx_pool_events_pack / all-gather / wx_pool_events_apply");
  if (precip && s->pool_remote && n_iter > 0) { // slab with particles: refuse an over-long call BEFORE any iteration runs
    const int allowed = wx_slab_period(s) - s->period_j; // (include/wxsim.h: `cone` columns for the first iteration, cone + 3 for every further one)
    if (n_iter > allowed)
      return fail(s, WX_E_STATE, "wx_step: %d iterations asked, %d done since the last halo exchange; %d ghost columns allow %d per period with particles",
                  n_iter, s->period_j, s->halo, wx_slab_period(s));
  }
  // overlap needs the kernel that can be launched per strip range; everything else orders the exchange on the compute stream
  // (with particles only the iteration AFTER an exchange splits: precipitation needs the whole grid of its iteration, and the
  // exchange needs the feedback texture precipitation leaves behind -- so the exchange hides behind the next interior strips)
  // The agreed water-free dry stencil runs its periods IN ORDER: iterations in pairs (two per launch), none of them split, and a
  // base-only halo message a quarter the size -- measured on the north star's slab (4096 + 2 x 42 columns x 4096 rows, tools/
  // slab_protocol_cost.py, profiles/r05_dry_slab_inorder.txt): 0.115 ms per iteration + the link time of 2.75 MB per side and period
  // against 0.133 with split edge / interior iterations that must run one iteration per launch
  const bool dry_in_order = dry_runs_in_order(s);
  const bool can_split = (fused || (dry && dry_marches(s))) && s->comm_stream != nullptr && s->halo > 0 && !(precip && s->pool_exact) && !dry_in_order;
  if (!can_split) wait_unpacked(s);
  if (!fused && s->light_planar) { // the per-pass / dry kernels take the light textures interleaved (the conversion reads ghost columns)
    wait_unpacked(s);
    light_to_rgba(s);
  }
  s->edges_recorded = false;
  if (n_iter > 0) s->gate_passed = false;
  if (n_iter > 0) s->water0_pending = false; // (the inputs of the previous step's last iteration are about to be overwritten)
  const int show_at = (flags & WX_OVERLAP_MORE_TO_COME) ? -1 : n_iter - 1; // the iteration that also stores the display-side fields
  for (int it = 0; it < n_iter; it++) {
    int edge_mode = 0;
    if (can_split) edge_mode = ((flags & WX_OVERLAP_EDGES_LAST) && it == 0 ? 2 : 0) | ((flags & WX_OVERLAP_EDGES_FIRST) && it == n_iter - 1 && !precip ? 1 : 0);
    if (can_split && it == 0 && !(edge_mode & 2)) wait_unpacked(s);
    int rc;
    // the water-free dry stencil in PAIRS (WX_OPT_DRY_PAIRS): two iterations per launch while two are left, none of them split
    const bool pair = dry && s->dry_pairs && !precip && edge_mode == 0 && it + 1 < n_iter && dry_marches(s) && s->wall_veg_ok && s->p.userInputType < 0 &&
                      !(s->p.airplaneValues[3] > 0.9f) && s->Y >= 16 &&
                      !(can_split && (flags & WX_OVERLAP_EDGES_FIRST) && it + 1 == n_iter - 1);
    if (pair) {
      rc = iterate_dry_pair(s, it + 1 == show_at);
      if (rc != WX_OK) return rc;
      s->ran_fused = true;
      s->iter += 2; // (`even` toggles twice)
      it += 1;
      clear_particle_textures(s); // (particles were switched off: the reference's clear, app.js:5933-5937, as below)
      continue;
    }
    if (dry)
      rc = iterate_dry(s, precip || it == show_at, edge_mode);
    else if (fused)
      rc = iterate_march_wet(s, it == show_at, precip, edge_mode);
    else
      rc = iterate_per_pass(s, mask);
    if (rc != WX_OK) return rc;
    s->ran_fused = fused || dry;
    const int src = s->even ? 0 : 1, dst = s->even ? 1 : 0;
    s->even = !s->even;
    // 8-10 clear feedback/deposition, precipitation, lightning location (app.js:5933-5983). The clear and the
    // blend-unit splats are replaced by: deposit at the sprite anchors -> 12x12 box sum that (re)writes both textures.
    if (precip) {
      const bool two_kernel = fused && !dry; // marching wet kernel: droplets sample base[0] (velocity) + tdisp (temperature)
      Uni u = s->uni;
      u.iterNum = (float)s->iter;
      u.iterI = (int)u.iterNum;
      {
        ProfScope ps(s, K_PRECIP);
        SlabP sp{0, s->X, 0, s->X, 0, 0, nullptr, nullptr, nullptr, 0};
        const float *d_in = s->drops[src];
        float *d_out = s->drops[dst];
        if (s->pool_remote) { // slab: the grid of this iteration is valid on the owned columns + (halo - 6*(j+1)) ghost columns; the
          // feedback texture is exact where every droplet within a sprite radius (6 px) was processed, so the owned columns
          // need 6 valid ghost columns even in the last iteration of a period
          const int margin = s->halo - s->cone - (s->cone + 3) * s->period_j; // >= cone + 1 (wx_slab_period): checked before the loop
          sp = SlabP{s->halo - margin, s->X - s->halo + margin, s->halo, s->X - s->halo, s->seam, s->period_j + 1, s->pool_remote, s->pool_flips, s->pool_owned, s->pool_exact};
          d_in = d_out = s->drops[0]; // the partitioned pool is updated in place
        }
        static const int precip_wgs = [] { const char *e = wx_tune_env("WX_PRECIP_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();
        const int chunks = (s->n_drops + 255) / 256;
        hipLaunchKernelGGL(k_precipitation, dim3(chunks < precip_wgs ? chunks : precip_wgs), dim3(256), 0, s->stream, s->geo, u, s->n_drops, d_in,
                           two_kernel ? s->base[0] : ((fused || dry) ? s->base[2] : s->base[1]), s->water[1], s->state, d_out, s->sg, sp,
                           two_kernel ? s->tdisp : nullptr, DetSplat{s->splat_order ? s->det_key[0] : nullptr, s->det_val}, s->wall[0]);
        LAUNCH_CHECK(s, "precipitation");
      }
      if (s->splat_order) { // deterministic order: sort the deposit records by anchor (stable), add each run in droplet-index order
        ProfScope ps(s, K_SPLAT);
        HIPCHK(s, hipcub::DeviceRadixSort::SortPairs(s->det_tmp, s->det_tmp_bytes, s->det_key[0], s->det_key[1], s->det_idx[0], s->det_idx[1], s->n_drops, 0, 31,
                                                     s->stream));
        hipLaunchKernelGGL(k_splat_runs, dim3((s->n_drops + 255) / 256), dim3(256), 0, s->stream, s->n_drops, s->det_key[1], s->det_idx[1], s->det_val, s->sg,
                           s->state);
        LAUNCH_CHECK(s, "splat_runs");
      }
      {
        ProfScope ps(s, K_SPLAT);
        const int T = s->sg.TXn * s->sg.TYn, par = s->splat_par;
        hipLaunchKernelGGL(k_splat_classify, dim3((T + 255) / 256), dim3(256), 0, s->stream, s->X, s->Y, s->sg, s->pool_remote ? 0 : 1, par);
        hipLaunchKernelGGL(k_splat_box, dim3(std::min(T, splat_box_grid())), dim3(256), 0, s->stream, s->X, s->Y, s->sg, s->state, s->fb, s->dep, s->seam,
                           s->pool_remote ? 0 : 1, par);
        LAUNCH_CHECK(s, "splat_classify / splat_box");
      }
      {
        // lightningLocation pass (one thread: it reads texel (1,0) the box sum just wrote and the request in the state) + the clear of
        // the accumulation tiles that held deposits, in one launch
        ProfScope ps(s, K_SPLAT);
        const int T = s->sg.TXn * s->sg.TYn;
        const LightningArgs la{u.iterNum, (int)(s->iter % 600 == 0), s->pool_remote ? 0 : 1, s->pool_remote && s->pool_exact ? 1 : 0, s->fb, s->state};
        hipLaunchKernelGGL(k_splat_clear, dim3(T < 2048 ? T : 2048), dim3(256), 0, s->stream, s->X, s->Y, s->sg, s->splat_par, la);
        LAUNCH_CHECK(s, "splat_clear");
        s->splat_par ^= 1;
      }
      s->drop_cur = s->pool_remote ? 0 : dst;
      s->fb_dirty = true;
      if (s->pool_remote) s->period_j++;
      if (s->pool_remote && s->pool_exact) s->exact_pending++;
    } else {
      clear_particle_textures(s); // particles were switched off: the reference's clear leaves both textures zero
    }
    s->iter++;
  }
  if (n_iter > 0) {
    s->emit_lit = !dry && (mask & WX_PASS_LIGHTING) != 0;
    s->emit_uni = s->uni;
  }
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}

int wx_set_option(wx_sim *s, int option, int value)
{
  if (!s) { // defaults of handles created from now on
    switch (option) {
    case WX_OPT_KERNEL_SET: g_opt_kernel_set = value != 0; return WX_OK;
    case WX_OPT_DRY_KERNEL: g_opt_dry_kernel = value != 0; return WX_OK;
    case WX_OPT_ROW_BANDS: if (value < 0 || value > 2) return WX_E_INVALID; g_opt_row_bands = value; return WX_OK;
    case WX_OPT_FIX_CAP: if (value < 0) return WX_E_INVALID; g_opt_fix_cap = value; return WX_OK;
    case WX_OPT_PLACEMENT_SEARCH: if (value < 0 || value > 64) return WX_E_INVALID; g_opt_place_tries = value; return WX_OK;
    default: return WX_E_INVALID;
    }
  }
  DeviceScope dev_scope(s);
  switch (option) {
  case WX_OPT_KERNEL_SET:
    if (int rc = wx_sync(s)) return rc;
    s->fused = value != 0 ? 2 : 0;
    return WX_OK;
  case WX_OPT_DRY_KERNEL:
    s->dry_march = value != 0;
    return WX_OK;
  case WX_OPT_ROW_BANDS:
    if (value < 0 || value > 2) return fail(s, WX_E_INVALID, "wx_set_option: WX_OPT_ROW_BANDS takes 0, 1 or 2");
    s->bands_mode = value;
    s->wet_shape_valid = false;
    return WX_OK;
  case WX_OPT_POOL_EXACT:
    if (!s->pool_remote) return fail(s, WX_E_STATE, "wx_set_option: WX_OPT_POOL_EXACT is for slab handles with particles");
    if (int rc = wx_sync(s)) return rc;
    s->pool_exact = value != 0;
    s->exact_pending = 0;
    return WX_OK;
  case WX_OPT_DRY_PAIRS:
    if (int rc = wx_sync(s)) return rc;
    s->dry_pairs = value != 0;
    return WX_OK;
  case WX_OPT_PLACEMENT_SEARCH:
    if (value < 0 || value > 64) return fail(s, WX_E_INVALID, "wx_set_option: WX_OPT_PLACEMENT_SEARCH takes 0 (never) .. 64 tries");
    s->place_tries = value;
    return WX_OK;
  case WX_OPT_WATER0_ON_DEMAND:
    if (int rc = materialize_water0(s)) return rc;
    s->lazy_water0 = value != 0;
    return WX_OK;
  case WX_OPT_SPLIT_LAUNCH: // split iterations (wx_step_overlap): 1 = one ordered launch + device-side hand-offs, 0 = two launch groups on two streams
    if (int rc = wx_sync(s)) return rc;
    s->split_launch = value != 0;
    return WX_OK;
  case WX_OPT_EXCHANGE_OVERLAP: // (takes effect at the next wx_slab_step / wx_exchange / wx_group_step: transport_prepare)
    s->exchange_in_order = value == 0;
    return WX_OK;
  case WX_OPT_FIX_CAP:
    if (value < 0) return fail(s, WX_E_INVALID, "wx_set_option: WX_OPT_FIX_CAP >= 0");
    if (s->fix_cells) return fail(s, WX_E_STATE, "wx_set_option: WX_OPT_FIX_CAP before the first step");
    s->fix_cap_request = value;
    return WX_OK;
  case WX_OPT_SPLAT_ORDER: {
    if (value != 0 && value != 1) return fail(s, WX_E_INVALID, "wx_set_option: WX_OPT_SPLAT_ORDER takes 0 (atomics) or 1 (deterministic)");
    if (value && s->n_drops > 0 && !s->det_val) { // the deposit records and the sort's scratch space, once
      const size_t n = (size_t)s->n_drops;
      for (int i = 0; i < 2; i++) {
        if (int rc = dalloc(s, &s->det_key[i], n)) return rc;
        if (int rc = dalloc(s, &s->det_idx[i], n)) return rc;
      }
      if (int rc = dalloc(s, &s->det_val, 5 * n)) return rc;
      std::vector<int> iota(n);
      for (size_t i = 0; i < n; i++) iota[i] = (int)i;
      HIPCHK(s, hipMemcpy(s->det_idx[0], iota.data(), n * 4, hipMemcpyHostToDevice));
      HIPCHK(s, hipcub::DeviceRadixSort::SortPairs(nullptr, s->det_tmp_bytes, s->det_key[0], s->det_key[1], s->det_idx[0], s->det_idx[1], s->n_drops, 0, 31,
                                                   s->stream));
      HIPCHK(s, hipMalloc(&s->det_tmp, s->det_tmp_bytes ? s->det_tmp_bytes : 16));
    }
    s->splat_order = value;
    return WX_OK;
  }
  case WX_OPT_CHECK_LAUNCHES:
    s->check_launches = value != 0;
    return WX_OK;
  default:
    return fail(s, WX_E_INVALID, "wx_set_option: unknown option %d", option);
  }
}

// Ghost columns were unpacked while the handle relied on the host's assertion that the whole domain is water-free: k_halo_unpack
// checked every ghost texel on the device; a blocking call is where the verdict is collected (never a silent divergence).
static int validate_ghost_flag(wx_sim *s)
{
  if (s->vx_check) { // did a velocity reach the bound the exchange period was sized for?
    int bits = 0;
    HIPCHK(s, hipMemcpyAsync(&bits, &s->state->cone_violation, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    s->vx_check = false;
    if (bits) {
      float v;
      memcpy(&v, &bits, 4);
      HIPCHK(s, hipMemsetAsync(&s->state->cone_violation, 0, 4, s->stream));
      return fail(s, WX_E_STATE, "|vx| reached %.3f cells / iteration in an exchange period sized for |vx| < %d (%d ghost columns per iteration): the flow accelerated "
                                 "faster than the margin of wx_slab_set_vx_bound allows, ghost columns were consumed faster than assumed and the slab may differ "
                                 "from the undecomposed run since (a velocity of 2 * halo - 8 = %d cells / iteration and more is reported wherever it occurs: from three halo widths "
                                 "inside the slab it reaches the ghost columns)", v, s->cone - 5, s->cone, std::max(1, 2 * s->halo - 8));
    }
  }
  if (s->sync_words && s->split_check) { // did a device-side hand-off of a split iteration give up polling?
    unsigned w[4] = {0, 0, 0, 0};
    HIPCHK(s, hipMemcpyAsync(w, s->sync_words, 16, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    s->split_check = false;
    if (w[2]) {
      HIPCHK(s, hipMemsetAsync(s->sync_words + 2, 0, 8, s->stream));
      return fail(s, WX_E_STATE, "a device-side hand-off of a split iteration (edge strips <-> halo exchange) was not answered within its poll limit (%s; arrivals %u of %u "
                                 "(gate wanted %u), epoch %u of %u): the exchange did not run or a peer is gone; the results since are invalid",
                  w[2] == 1 ? "the gate kernel waiting for the edge strips" : "an edge strip waiting for the ghost columns", w[0], s->arrive_target, w[3], w[1], s->epoch_host);
    }
  }
  if (s->fix_check) { // did a marching iteration find more cells with |v| >= 0.9 than the exact-path list holds?
    int over = 0;
    HIPCHK(s, hipMemcpyAsync(&over, &s->state->fix_overflow, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    s->fix_check = false;
    if (over) {
      HIPCHK(s, hipMemsetAsync(&s->state->fix_overflow, 0, 4, s->stream));
      return fail(s, WX_E_STATE, "%d output cells of one iteration were fed by velocities >= 0.9 cells/iteration; the exact path holds %d: the state has "
                                 "left the simulation's range (velocities are documented as -1 .. 1) and the results since are invalid", over, s->fix_cap);
    }
  }
  if (s->pool_check) {
    int over = 0;
    HIPCHK(s, hipMemcpyAsync(&over, &s->state->pool_overflow, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    s->pool_check = false;
    if (over) {
      HIPCHK(s, hipMemsetAsync(&s->state->pool_overflow, 0, 4, s->stream));
      return fail(s, WX_E_STATE, "a droplet-pool exchange buffer received %d entries (event capacity %d, edge capacity %d): droplets were lost", over,
                  s->pool_event_cap, s->pool_edge_cap);
    }
  }
  if (!s->ghost_check) return WX_OK;
  int flag = 0;
  wait_unpacked(s);
  HIPCHK(s, hipMemcpyAsync(&flag, &s->state->ghost_nontrivial, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  s->ghost_check = false;
  if (flag) {
    s->water_trivial = s->local_water_free = s->wall_veg_ok = false;
    return fail(s, WX_E_STATE, "a neighbour's ghost columns carry water (or a negative vegetation byte) although wx_slab_assert_water_free(1) was "
                               "called: the water-free dry iteration that ran since is invalid");
  }
  return WX_OK;
}

// ---- placement tuning ----
namespace {
struct Ref { // a pointer as (block, offset)
  int block;
  size_t off;
};
struct TuneSnap { // host-side state that iterations change; the rotating plane pointers as block references
  Ref base[3], water[3], light[2], wall[2], drops[2], lpx[3], lpy[3], lpzw[3];
  int even, drop_cur, splat_par, period_j, air_from_row;
  int64_t iter;
  bool ran_fused, light_planar, fb_dirty, water_trivial, local_water_free, wall_veg_ok, ghost_check, emit_lit, fix_check, wet_shape_valid, halo_base_only, water0_pending, disp_lazy;
  Uni emit_uni;
};
bool snap_take(const wx_sim *s, TuneSnap &t)
{
  bool ok = true;
  auto ref = [&](const void *p) {
    const int i = block_of(s->blocks, p);
    ok = ok && i >= 0;
    return Ref{i, i >= 0 ? (size_t)((const char *)p - s->blocks[i].p) : 0};
  };
  for (int i = 0; i < 3; i++) { t.base[i] = ref(s->base[i]); t.water[i] = ref(s->water[i]); t.lpx[i] = ref(s->lp[i].x); t.lpy[i] = ref(s->lp[i].y); t.lpzw[i] = ref(s->lp[i].zw); }
  for (int i = 0; i < 2; i++) { t.light[i] = ref(s->light[i]); t.wall[i] = ref(s->wall[i]); t.drops[i] = ref(s->drops[i]); }
  t.even = s->even; t.drop_cur = s->drop_cur; t.splat_par = s->splat_par; t.period_j = s->period_j; t.air_from_row = s->air_from_row; t.iter = s->iter;
  t.ran_fused = s->ran_fused; t.light_planar = s->light_planar; t.fb_dirty = s->fb_dirty; t.water_trivial = s->water_trivial; t.local_water_free = s->local_water_free; t.wall_veg_ok = s->wall_veg_ok; t.halo_base_only = s->halo_base_only; t.water0_pending = s->water0_pending; t.disp_lazy = s->disp_lazy;
  t.ghost_check = s->ghost_check; t.emit_lit = s->emit_lit; t.fix_check = s->fix_check; t.wet_shape_valid = s->wet_shape_valid; t.emit_uni = s->emit_uni;
  return ok;
}
void snap_put(wx_sim *s, const TuneSnap &t) // (into the block set the handle points at NOW)
{
  auto at = [&](const Ref &r) { return s->blocks[r.block].p + r.off; };
  for (int i = 0; i < 3; i++) {
    s->base[i] = (float4 *)at(t.base[i]); s->water[i] = (float4 *)at(t.water[i]);
    s->lp[i].x = (float *)at(t.lpx[i]); s->lp[i].y = (float *)at(t.lpy[i]); s->lp[i].zw = (float2 *)at(t.lpzw[i]);
  }
  for (int i = 0; i < 2; i++) { s->light[i] = (float4 *)at(t.light[i]); s->wall[i] = (char4 *)at(t.wall[i]); s->drops[i] = (float *)at(t.drops[i]); }
  s->even = t.even; s->drop_cur = t.drop_cur; s->splat_par = t.splat_par; s->period_j = t.period_j; s->air_from_row = t.air_from_row; s->iter = t.iter;
  s->ran_fused = t.ran_fused; s->light_planar = t.light_planar; s->fb_dirty = t.fb_dirty; s->water_trivial = t.water_trivial; s->local_water_free = t.local_water_free; s->wall_veg_ok = t.wall_veg_ok; s->halo_base_only = t.halo_base_only; s->water0_pending = t.water0_pending; s->disp_lazy = t.disp_lazy;
  s->ghost_check = t.ghost_check; s->emit_lit = t.emit_lit; s->fix_check = t.fix_check; s->wet_shape_valid = t.wet_shape_valid; s->emit_uni = t.emit_uni;
}
// every registered pointer of the handle moves from the current block set to `to` (same block, same offset)
int rebase(wx_sim *s, const std::vector<Block> &to)
{
  for (void **slot : s->slots) {
    const int i = block_of(s->blocks, *slot);
    if (i >= 0) *slot = to[i].p + (reinterpret_cast<char *>(*slot) - s->blocks[i].p);
  }
  s->blocks = to;
  FullCtx fc{s->geo, s->uni, s->initial_T, s->snd_T, s->snd_W, s->snd_Vel}; // (holds device pointers into the blocks)
  HIPCHK(s, hipMemcpyAsync(s->full_ctx, &fc, sizeof(fc), hipMemcpyHostToDevice, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  return WX_OK;
}
void free_set(std::vector<Block> &B)
{
  for (Block &b : B) hipFree(b.p);
  B.clear();
}
size_t set_bytes(const std::vector<Block> &B)
{
  size_t n = 0;
  for (const Block &b : B) n += b.bytes;
  return n;
}
// a new set of allocations with the sizes of `like`; allocated in a different order each time (`variant`) with an odd-sized pad in
// between, so that the new set does not simply land at a fixed stride behind the previous one (pads are appended to `pads`)
bool alloc_set(const std::vector<Block> &like, std::vector<Block> &out, int variant, std::vector<char *> &pads)
{
  const size_t n = like.size();
  out.assign(n, Block{nullptr, 0});
  for (size_t k = 0; k < n; k++) {
    const size_t i = (variant & 1) ? n - 1 - (k + (size_t)variant * 5) % n : (k + (size_t)variant * 5) % n;
    if (k % 3 == 0) {
      char *pad = nullptr;
      if (hipMalloc((void **)&pad, (size_t)(3 + 14 * ((variant + (int)k) % 7)) << 20) == hipSuccess) pads.push_back(pad);
      else (void)hipGetLastError();
    }
    if (block_alloc(like[i].bytes, &out[i]) != hipSuccess) {
      (void)hipGetLastError();
      free_set(out);
      return false;
    }
  }
  return true;
}
int copy_set(wx_sim *s, const std::vector<Block> &dst, const std::vector<Block> &src)
{
  for (size_t i = 0; i < src.size(); i++) HIPCHK(s, hipMemcpyAsync(dst[i].p, src[i].p, src[i].bytes, hipMemcpyDeviceToDevice, s->stream));
  return WX_OK;
}
} // namespace

// Where the planes lie in physical memory decides how the kernel's ~13 concurrent streams spread over the HBM channels -- the same
// iteration takes 0.72 .. 0.87 ms depending on the allocations (see storage_begin). wx_tune_placement times the handle's OWN iteration
// (current parameters / pass mask; the clocks brought to steady state first; four untimed + iters_per_try timed iterations) on the
// allocations it has and on `tries` further sets that each receive a copy of the state; the winner is confirmed against the incumbent
// back to back, becomes the handle's storage and gets the state back from a pristine backup taken at the start. The simulation
// state, iteration counter and every field are exactly what they were before the call.
int wx_tune_placement(wx_sim *s, int tries, int iters_per_try, float *ms_before, float *ms_after)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (tries < 1 || iters_per_try < 1) return fail(s, WX_E_INVALID, "wx_tune_placement: tries >= 1, iters_per_try >= 1");
  if (!s->uploaded || !s->have_params) return fail(s, WX_E_STATE, "wx_tune_placement before wx_upload / wx_set_params");
  if (s->pool_remote) return fail(s, WX_E_STATE, "wx_tune_placement: not on slab handles with particles (the exchange period would advance)");
  if (int rc = materialize_water0(s)) return rc; // (the probes' iterations overwrite the inputs it is made from)
  if (int rc = wx_sync(s)) return rc;
  struct Busy { // (the probes call wx_step: no search inside the search; whoever calls, the handle's one search is this one)
    wx_sim *s;
    explicit Busy(wx_sim *s_) : s(s_) { s->place_busy = true; s->place_done = true; }
    ~Busy() { s->place_busy = false; }
  } busy(s);
  TuneSnap snap;
  if (s->blocks.empty() || !snap_take(s, snap)) return fail(s, WX_E_STATE, "wx_tune_placement: the handle's planes are not in registered blocks");
  hipEvent_t e0, e1;
  HIPCHK(s, hipEventCreate(&e0));
  HIPCHK(s, hipEventCreate(&e1));
  const bool was_profiling = s->profiling; // (the probes' launches are not the caller's)
  s->profiling = false;
  const bool dbg = wx_tune_env("WX_TUNE_DEBUG") != nullptr;
  auto probe = [&](float *ms) -> int { // four untimed iterations (launch shape, terrain scan, caches), then the timed ones
    int rc = wx_step(s, 4);
    if (rc == WX_OK && hipEventRecord(e0, s->stream) != hipSuccess) rc = WX_E_DEVICE;
    if (rc == WX_OK) rc = wx_step(s, iters_per_try);
    if (rc == WX_OK && (hipEventRecord(e1, s->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(ms, e0, e1) != hipSuccess)) rc = WX_E_DEVICE;
    *ms /= (float)iters_per_try;
    return rc;
  };
  const std::vector<Block> original = s->blocks;
  const size_t total = set_bytes(original);
  std::vector<char *> pads;
  auto room_for = [&](size_t bytes) {
    size_t free_b = 0, total_b = 0;
    return hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > bytes + (size_t(2) << 30);
  };
  // A pristine copy of the state (not a candidate): every set, the current one included, is probed in place and the winner gets the
  // state back from here.
  std::vector<Block> backup;
  if (!room_for(total) || !alloc_set(original, backup, 0, pads) || copy_set(s, backup, original) != WX_OK) {
    free_set(backup);
    for (char *p : pads) hipFree(p);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    s->profiling = was_profiling;
    return fail(s, WX_E_NOMEM, "wx_tune_placement: no room for a copy of the state (%zu bytes)", total);
  }
  int rc = WX_OK;
  float first_ms = 0.f, best_ms = 0.f;
  // the clocks follow the load: a probe right after an idle phase or a burst of copies reads several % high. Bring the chip to its
  // steady state first, and time the incumbent twice
  rc = wx_step(s, 6 * iters_per_try);
  snap_put(s, snap);
  if (rc == WX_OK) rc = probe(&first_ms);
  snap_put(s, snap);
  if (rc == WX_OK) rc = probe(&first_ms); // the allocations the handle has
  snap_put(s, snap);
  std::vector<Block> best = original;
  best_ms = first_ms;
  if (dbg) fprintf(stderr, "[wx_tune_placement] current allocations (%zu blocks, %zu MB): %.4f ms / iteration\n", original.size(), total >> 20, first_ms);
  if (dbg && wx_tune_env("WX_TUNE_DEBUG")[0] == '2') {
    fprintf(stderr, "[wx_tune_blocks] 0 %.4f", first_ms);
    for (const Block &b : original) fprintf(stderr, " %llx:%zu", (unsigned long long)(uintptr_t)b.p, b.bytes >> 20);
    fprintf(stderr, "\n");
  }
  // Rejected candidates stay allocated while there is room: freed memory is handed straight back by the next hipMalloc (same
  // memory, nothing learned). They are released at the end, or earlier when device memory runs short.
  std::vector<std::vector<Block>> rejects;
  for (int t = 1; t <= tries && rc == WX_OK; t++) {
    std::vector<Block> cand;
    bool got = false;
    for (;;) {
      if (room_for(total) && alloc_set(original, cand, t, pads)) {
        got = true;
        break;
      }
      // (never the allocations the handle came with: the confirmation below moves back into them)
      auto victim = std::find_if(rejects.begin(), rejects.end(), [&](const std::vector<Block> &r) { return !r.empty() && r[0].p != original[0].p; });
      if (victim == rejects.end()) break;
      free_set(*victim);
      rejects.erase(victim);
    }
    if (!got) break; // out of device memory: keep the best so far
    rc = copy_set(s, cand, backup);
    if (rc == WX_OK) rc = rebase(s, cand);
    snap_put(s, snap);
    float ms = 0.f;
    if (rc == WX_OK) rc = probe(&ms);
    snap_put(s, snap);
    if (dbg) fprintf(stderr, "[wx_tune_placement] candidate %d (first plane at %p): %.4f ms / iteration\n", t, (void *)(cand.size() > 1 ? cand[1].p : cand[0].p), ms);
    if (dbg && wx_tune_env("WX_TUNE_DEBUG")[0] == '2') {
      fprintf(stderr, "[wx_tune_blocks] %d %.4f", t, ms);
      for (const Block &b : cand) fprintf(stderr, " %llx:%zu", (unsigned long long)(uintptr_t)b.p, b.bytes >> 20);
      fprintf(stderr, "\n");
    }
    if (rc == WX_OK && ms < best_ms) {
      rejects.push_back(best);
      best = cand;
      best_ms = ms;
    } else {
      rejects.push_back(cand);
    }
  }
  const bool moved = best[0].p != original[0].p;
  if (moved && rc == WX_OK) { // confirm against the incumbent, back to back (a probe is ~1 % noisy; clocks drift over the call)
    float ms_o = 0.f, ms_b = 0.f;
    rc = rebase(s, original);
    snap_put(s, snap);
    if (rc == WX_OK) rc = probe(&ms_o);
    snap_put(s, snap);
    if (rc == WX_OK) rc = rebase(s, best);
    snap_put(s, snap);
    if (rc == WX_OK) rc = probe(&ms_b);
    snap_put(s, snap);
    if (dbg) fprintf(stderr, "[wx_tune_placement] confirmation: incumbent %.4f, winner %.4f ms / iteration\n", ms_o, ms_b);
    if (rc == WX_OK && !(ms_b < ms_o * 0.99f)) { // not clearly better: stay where we are
      for (auto &r : rejects)
        if (!r.empty() && r[0].p == original[0].p) r = best;
      best = original;
      best_ms = first_ms = ms_o;
    } else {
      best_ms = ms_b;
      first_ms = ms_o;
    }
  }
  // give the winner (possibly the allocations the handle had) the pristine state back, then move in -- in this order: the state holds
  // the FullCtx block with device pointers into the set it was copied from, which rebase() rewrites
  if (copy_set(s, best, backup) != WX_OK || hipStreamSynchronize(s->stream) != hipSuccess) rc = WX_E_DEVICE;
  if (rebase(s, best) != WX_OK) rc = WX_E_DEVICE;
  snap_put(s, snap);
  for (auto &r : rejects) free_set(r);
  for (char *p : pads) hipFree(p);
  free_set(backup);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  s->profiling = was_profiling;
  if (ms_before) *ms_before = first_ms;
  if (ms_after) *ms_after = best_ms;
  if (rc == WX_OK) {
    s->place_first_ms = first_ms;
    s->place_kept_ms = best_ms;
  }
  if (rc != WX_OK && s->err.empty()) s->err = "wx_tune_placement: device error";
  return rc;
}

// the handle's placement search (explicit or the implicit one of the first wx_step): 1 and the two iteration times if one has run, else 0
int wx_placement_info(const wx_sim *s, float *ms_first, float *ms_kept)
{
  if (!s) return WX_E_INVALID;
  if (ms_first) *ms_first = s->place_first_ms;
  if (ms_kept) *ms_kept = s->place_kept_ms;
  return s->place_kept_ms > 0.f ? 1 : 0;
}

// largest |velocity component| that sent a cell to the exact path since the last call (0: none reached 0.9); resets it
int wx_fastest_velocity(wx_sim *s, float *cells_per_iteration)
{
  if (!s || !cells_per_iteration) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  int bits = 0;
  HIPCHK(s, hipMemcpyAsync(&bits, &s->state->fastest_bits, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(s, hipMemsetAsync(&s->state->fastest_bits, 0, 4, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  memcpy(cells_per_iteration, &bits, 4);
  return WX_OK;
}

// the pair kernel's exact path since the last call: second-iteration cells recomputed by k_dry2_fix, pairs repeated whole; resets both
int wx_pair_stats(wx_sim *s, int64_t *cells_recomputed, int64_t *pairs_repeated)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  int ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (s->pair_ctl) {
    HIPCHK(s, hipMemcpyAsync(ctl, s->pair_ctl, 32, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipMemsetAsync(s->pair_ctl + D2_N_REDO, 0, 4, s->stream));
    HIPCHK(s, hipMemsetAsync(s->pair_ctl + D2_FIXED, 0, 8, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
  }
  int64_t fixed;
  memcpy(&fixed, ctl + D2_FIXED, 8);
  if (cells_recomputed) *cells_recomputed = fixed;
  if (pairs_repeated) *pairs_repeated = ctl[D2_N_REDO];
  return WX_OK;
}

int wx_water_free(const wx_sim *s) { return s && s->local_water_free ? 1 : 0; }

int wx_slab_assert_water_free(wx_sim *s, int agreed)
{
  if (!s) return WX_E_INVALID;
  s->slab_dry_agreed = agreed != 0;
  s->water_trivial = s->local_water_free && (s->halo == 0 || s->slab_dry_agreed);
  s->halo_base_only = s->slab_dry_agreed && s->halo > 0 && !s->pool_remote; // (by the host's word alone: equal on every slab even if the word was wrong)
  return WX_OK;
}

// ---- slabs exact at any speed ----
int wx_slab_cone(const wx_sim *s) { return s ? s->cone : 0; }

int wx_slab_period(const wx_sim *s)
{
  if (!s || s->halo <= 0) return 0;
  if (s->pool_remote) { // with particles: `cone` columns for the first iteration, cone + 3 for every further one, and in the LAST one the zone in
    // which droplets are processed must still reach every droplet that ENDS the iteration within a sprite radius of the owned columns: it is
    // tested on the position a droplet starts from and deposits where it arrives, up to floor|vx| + 1 = cone - 5 columns further -- a margin
    // of 5.5 + |vx| < cone + 1 columns (round 6, tools/fuzz_parity.py --mode group: with jets of 2 cells / iteration the margin of 6 left
    // one sprite column at a slab's edge undeposited; for |vx| < 1 and the halos particles allow -- multiples of 64 -- the period is
    // WX_SLAB_PERIOD_PARTICLES(halo) as before)
    if (s->halo < 2 * s->cone + 1) return 0;
    return std::min(15, 1 + (s->halo - 2 * s->cone - 1) / (s->cone + 3)); // (at most 15: the flip history is a 16-bit mask)
  }
  return s->halo / s->cone;
}

int wx_slab_set_vx_bound(wx_sim *s, float v_measured)
{
  if (!s) return WX_E_INVALID;
  if (!(v_measured >= 0.0f)) return fail(s, WX_E_INVALID, "wx_slab_set_vx_bound: %g", v_measured);
  // the margin: a flow measured at v may be at 1.25 v + 0.25 by the time the period it sizes is over (one to two periods later); below
  // 0.6 cells / iteration that still is less than one cell
  const float bound = v_measured * 1.25f + 0.25f;
  const int cone = WX_SLAB_CONE + (bound >= 1.0f ? (int)floorf(bound) : 0);
  if (s->halo > 0 && (cone > s->halo || (s->pool_remote && 2 * cone + 1 > s->halo)))
    return fail(s, WX_E_STATE, "|vx| up to %.2f cells / iteration needs %d ghost columns per iteration; the handle has %d: a wider halo (wx_create_slab) is needed for this flow",
                v_measured, cone, s->halo);
  s->vx_known = v_measured;
  s->vx_bound = bound;
  s->cone = cone;
  return WX_OK;
}

// the state as it lies in base_0, for kernels that do not track while they run and for states that came from outside
// check: the state is the product of iterations of the CURRENT exchange period (kernels that do not track, or velocities a host wrote
// through wx_device_ptr): a |vx| that reaches the period's bound is reported like the marching kernels report theirs (cone_violation ->
// WX_E_STATE at the next blocking call), never silent. Without it: a freshly uploaded state, from which the bound is about to be set.
static void vx_scan_enqueue(wx_sim *s, hipStream_t st, bool check)
{
  const VxTrack t = vx_track(s);
  hipLaunchKernelGGL(k_vx_scan, dim3(1024), dim3(256), 0, st, s->X, s->Y, t.zone_l, t.zone_r, s->base[0], VxTrack{t.max_bits, t.violation, check ? t.limit : 0.0f, 0, 0, check ? t.limit_in : 0.0f});
}

int wx_slab_vx_take(wx_sim *s, float *vmax)
{
  if (!s || !vmax) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (!s->uploaded) return fail(s, WX_E_STATE, "wx_slab_vx_take before wx_upload");
  settle_edges(s);
  wait_unpacked(s);
  if (s->vx_stale || s->vx_untracked) vx_scan_enqueue(s, s->stream, !s->vx_stale);
  int bits = 0;
  HIPCHK(s, hipMemcpyAsync(&bits, &s->state->vx_max_bits, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(s, hipMemsetAsync(&s->state->vx_max_bits, 0, 4, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  s->vx_stale = s->vx_untracked = false;
  memcpy(vmax, &bits, 4);
  return WX_OK;
}

int wx_sync(wx_sim *s)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  settle_edges(s);
  if (s->comm_stream) HIPCHK(s, hipStreamSynchronize(s->comm_stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  return validate_ghost_flag(s);
}

int64_t wx_get_iter(const wx_sim *s) { return s ? s->iter : -1; }
int wx_set_iter(wx_sim *s, int64_t iter)
{
  if (!s || iter < 0) return WX_E_INVALID;
  s->iter = iter;
  return WX_OK;
}

// precipitationFeedbackTexture as the RGBA32F texture the reference holds: three stored channels + alpha
__global__ void k_fb_to_rgba(size_t first, size_t n, const float3 *__restrict__ fb, float4 *__restrict__ out, const DevState *__restrict__ st, int mailbox)
{
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < first + n; i += (size_t)gridDim.x * blockDim.x) {
    const float3 v = fb[i];
    out[i] = make_float4(v.x, v.y, v.z, (mailbox && i == 1) ? st->mailbox_w : 0.0f);
  }
}
static int fb_to_rgba(wx_sim *s, int y0, int h) // rows [y0, y0 + h) (h <= 0: the whole texture)
{
  const size_t n = ncell(s);
  if (h <= 0 || y0 < 0 || y0 + h > s->Y) {
    y0 = 0;
    h = s->Y;
  }
  if (!s->fb_rgba && hipMalloc((void **)&s->fb_rgba, n * sizeof(float4)) != hipSuccess) {
    (void)hipGetLastError();
    s->fb_rgba = nullptr;
    s->err = "WX_FIELD_PRECIP_FB: no device memory for the RGBA copy of the feedback texture";
    return WX_E_NOMEM;
  }
  const size_t cells = (size_t)h * s->X;
  hipLaunchKernelGGL(k_fb_to_rgba, dim3((unsigned)std::min<size_t>(2048, (cells + 255) / 256)), dim3(256), 0, s->stream, (size_t)y0 * s->X, cells, s->fb, s->fb_rgba, s->state,
                     s->pool_remote ? 0 : 1);
  return hipGetLastError() == hipSuccess ? WX_OK : WX_E_DEVICE;
}

static void water0_scratch_free(wx_sim *s)
{
  for (void *q : {(void *)s->w0_b1, (void *)s->w0_b2, (void *)s->w0_light, (void *)s->w0_w1, (void *)s->w0_w2, (void *)s->w0_curl}) hipFree(q);
  s->w0_b1 = s->w0_b2 = s->w0_light = nullptr;
  s->w0_w1 = s->w0_w2 = nullptr;
  s->w0_curl = nullptr;
}
// waterTexture_0 of the last iteration, made now (see wx_sim::water0_pending): the per-pass kernels velocity -> curl -> vorticity -> boundary
// on the inputs that iteration read -- the same cell functions on the same operands as the marching kernel's stages, so the same values
// (tests/test_gpu_parity.py compares the two kernel sets field by field, WATER_0 included). On the compute stream, behind the step.
static int materialize_water0(wx_sim *s)
{
  if (!s->water0_pending) return WX_OK;
  DeviceScope dev_scope(s);
  const size_t n = ncell(s);
  if (!s->w0_b1) { // (first use on this handle: 60 B per cell of scratch)
    bool ok = hipMalloc((void **)&s->w0_b1, n * 16) == hipSuccess && hipMalloc((void **)&s->w0_b2, n * 16) == hipSuccess && hipMalloc((void **)&s->w0_light, n * 16) == hipSuccess &&
              hipMalloc((void **)&s->w0_w1, n * 4) == hipSuccess && hipMalloc((void **)&s->w0_w2, n * 4) == hipSuccess && hipMalloc((void **)&s->w0_curl, n * 4) == hipSuccess;
    if (!ok) {
      water0_scratch_free(s);
      (void)hipGetLastError();
      return fail(s, WX_E_NOMEM, "waterTexture_0 on demand: %zu bytes of scratch", n * 60);
    }
  }
  const dim3 grid = grid2d(s), block(BX, BY);
  // where the ping-pong left that iteration's inputs (iterate_march_wet's swaps): base_0 / wall_0 -> [1], water_1 -> [2]; light_0: the
  // planes the odd iterations retire into [2], untouched by the even ones
  const float4 *base_in = s->base[1], *water_in = s->water[2];
  const char4 *wall_in = s->wall[1];
  const LightPlanes &l0 = s->w0_even ? s->lp[0] : s->lp[2];
  const float4 *light0 = s->w0_light;
  if (s->w0_even && !s->light_planar)
    light0 = s->light[0]; // (a reader switched the light textures to RGBA since: light_0 was not written by that iteration)
  else
    hipLaunchKernelGGL(k_light_from_planes, dim3(2048), dim3(256), 0, s->stream, n, LightPlanesC{l0.x, l0.y, l0.zw}, s->w0_light);
  ProfScope ps(s, K_BOUNDARY);
  hipLaunchKernelGGL(k_velocity, grid, block, 0, s->stream, s->geo, s->w0_uni, base_in, wall_in, s->w0_b1, s->w0_w1);
  hipLaunchKernelGGL(k_curl, grid, block, 0, s->stream, s->geo, s->w0_b1, s->w0_curl); // (not s->curl: a streamed frame may be reading it)
  hipLaunchKernelGGL(k_vorticity, grid, block, 0, s->stream, s->geo, s->w0_curl, s->vort);
  GridPtrs in{s->w0_b1, water_in, s->w0_w1, s->vort, light0, nullptr, nullptr};
  hipLaunchKernelGGL(k_boundary, grid, block, 0, s->stream, s->geo, s->w0_uni, s->w0_initT_saved ? s->w0_initT : s->initial_T, in, s->w0_b2, s->water[0], s->w0_w2);
  LAUNCH_CHECK(s, "waterTexture_0 on demand");
  s->water0_pending = false;
  return WX_OK;
}

// (y0, h: the rows a reader is going to look at -- fields that are made on demand are only made there)
static int field_info(wx_sim *s, int field, const void **ptr, int *channels, int *elem, int y0 = 0, int h = 0)
{
  switch (field) {
  case WX_FIELD_BASE_CUR: *ptr = s->base[0]; *channels = 4; *elem = 4; return 0;
  case WX_FIELD_BASE_DISP:
    if (s->ran_fused && s->disp_lazy) { // the marching wet kernel's display iteration: the rows asked for, assembled now (wx_wet.h, WetOut::p_disp)
      const int r0 = h > 0 ? y0 : 0, rows = h > 0 ? h : s->Y;
      const size_t cnt = (size_t)rows * s->X;
      hipLaunchKernelGGL(k_base_disp_assemble, dim3((unsigned)std::min<size_t>(4096, (cnt + 255) / 256)), dim3(256), 0, s->stream, s->X, s->Y, r0, rows, s->base[0], s->pdisp, s->tdisp,
                         s->wall[0], s->base[2]);
    }
    *ptr = s->ran_fused ? s->base[2] : s->base[1]; *channels = 4; *elem = 4; return 0;
  case WX_FIELD_WATER_0:
    if (int rc = materialize_water0(s)) return rc < -1 ? rc : WX_E_DEVICE; // (not -1: that means "unknown field" to the callers)
    *ptr = s->water[0]; *channels = 4; *elem = 4; return 0;
  case WX_FIELD_WATER_CUR: *ptr = s->water[1]; *channels = 4; *elem = 4; return 0;
  case WX_FIELD_WALL_CUR: *ptr = s->wall[0]; *channels = 4; *elem = 1; return 0;
  // wallTexture_1 (post-advection) == wallTexture_0 after the pressure pass-through; the single-kernel paths keep one copy
  case WX_FIELD_WALL_DISP: *ptr = s->ran_fused ? s->wall[0] : s->wall[1]; *channels = 4; *elem = 1; return 0;
  case WX_FIELD_LIGHT_0:
  case WX_FIELD_LIGHT_1: {
    // the marching wet kernel keeps the light textures as planes: a reader gets the RGBA texels of the rows it looks at, made now in the
    // interleaved buffer -- the planes stay the live copy (switching both textures to RGBA and back cost a display host that samples
    // lightTexture_0 every frame 128 B per cell and frame)
    const int i = field == WX_FIELD_LIGHT_1 ? 1 : 0;
    if (s->light_planar) {
      const int r0 = h > 0 ? y0 : 0, rows = h > 0 ? h : s->Y;
      const size_t off = (size_t)r0 * s->X, cnt = (size_t)rows * s->X;
      const int wgs = (int)std::min<size_t>(2048, (cnt + 255) / 256);
      hipLaunchKernelGGL(k_light_from_planes, dim3(wgs), dim3(256), 0, s->stream, cnt, LightPlanesC{s->lp[i].x + off, s->lp[i].y + off, s->lp[i].zw + off}, s->light[i] + off);
    }
    *ptr = s->light[i]; *channels = 4; *elem = 4; return 0;
  }
  case WX_FIELD_CURL: *ptr = s->curl; *channels = 1; *elem = 4; return 0;
  case WX_FIELD_VORT: *ptr = s->vort; *channels = 2; *elem = 4; return 0;
  case WX_FIELD_PRECIP_FB: // stored with three channels; the RGBA texture (alpha: 0, the lightning request's fourth component at texel (1,0)) is made here
    if (int rc = fb_to_rgba(s, y0, h)) return rc < -1 ? rc : WX_E_DEVICE; // (not -1: that means "unknown field" to the callers)
    *ptr = s->fb_rgba; *channels = 4; *elem = 4; return 0;
  case WX_FIELD_PRECIP_DEP: *ptr = s->dep; *channels = 2; *elem = 4; return 0;
  default: return -1;
  }
}

// WX_FIELD_EMITTED: (re)compute the rectangle into the handle's RGBA16F buffer
static int emitted_rect(wx_sim *s, int x, int y, int w, int h)
{
  if (!s->emitted) {
    HIPCHK(s, hipMalloc((void **)&s->emitted, ncell(s) * sizeof(half4)));
    HIPCHK(s, hipMemsetAsync(s->emitted, 0, ncell(s) * sizeof(half4), s->stream));
  }
  if (s->copy_in_flight) { // a streamed frame may still be reading the buffer
    HIPCHK(s, hipStreamWaitEvent(s->stream, s->ev_copy_done, 0));
    s->copy_in_flight = false;
  }
  const dim3 grid((w + BX - 1) / BX, (h + BY - 1) / BY), block(BX, BY);
  if (!s->emit_lit) { // no lighting pass has drawn into the texture since it was created: zero
    HIPCHK(s, hipMemset2DAsync(s->emitted + ((size_t)y * s->X + x), (size_t)s->X * sizeof(half4), 0, (size_t)w * sizeof(half4), h, s->stream));
    return WX_OK;
  }
  const int src = s->even ? 1 : 0; // the last iteration read light[even_then ? 0 : 1] and flipped `even`
  if (s->light_planar)
    hipLaunchKernelGGL(k_emitted<true>, grid, block, 0, s->stream, s->geo, s->emit_uni, x, y, w, h, s->water[1], s->wall[0], nullptr, s->lp[src].x, s->emitted);
  else
    hipLaunchKernelGGL(k_emitted<false>, grid, block, 0, s->stream, s->geo, s->emit_uni, x, y, w, h, s->water[1], s->wall[0], s->light[src], nullptr, s->emitted);
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}

int wx_read_rect(wx_sim *s, int field, int x, int y, int w, int h, void *dst, int dtype)
{
  if (!s || !dst) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (field == WX_FIELD_LIGHTNING) {
    if (dtype != WX_DTYPE_F32 || x != 0 || y != 0 || w != 1 || h != 1) return fail(s, WX_E_RANGE, "lightning data is a 1x1 f32 texture");
    wait_unpacked(s); // (the per-period pool exchange writes the strike on the side stream: k_pool_lightning_latest)
    HIPCHK(s, hipMemcpyAsync(dst, s->state->lightning, 16, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return WX_OK;
  }
  settle_edges(s);
  wait_unpacked(s); // ghost columns written on the comm stream are part of what a readback sees
  if (int rc = validate_ghost_flag(s)) return rc;
  if (field == WX_FIELD_EMITTED) {
    if (w <= 0 || h <= 0 || x < 0 || y < 0 || (long long)x + w > s->X || (long long)y + h > s->Y)
      return fail(s, WX_E_RANGE, "wx_read_rect: rect (%d,%d %dx%d) outside %dx%d (no wrap)", x, y, w, h, s->X, s->Y);
    if (dtype != WX_DTYPE_F16 && dtype != WX_DTYPE_F32) return fail(s, WX_E_INVALID, "wx_read_rect: emittedLight is RGBA16F: WX_DTYPE_F16 or WX_DTYPE_F32");
    const int rc = emitted_rect(s, x, y, w, h);
    if (rc != WX_OK) return rc;
    const half4 *src = s->emitted + ((size_t)y * s->X + x);
    if (dtype == WX_DTYPE_F16) {
      HIPCHK(s, hipMemcpy2DAsync(dst, (size_t)w * 8, src, (size_t)s->X * 8, (size_t)w * 8, h, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(s, hipStreamSynchronize(s->stream));
      return WX_OK;
    }
    std::vector<__half> tmp((size_t)w * h * 4); // like gl.readPixels(..., gl.FLOAT, ...) of the half-float attachment
    HIPCHK(s, hipMemcpy2DAsync(tmp.data(), (size_t)w * 8, src, (size_t)s->X * 8, (size_t)w * 8, h, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    float *d = (float *)dst;
    for (size_t i = 0; i < tmp.size(); i++) d[i] = __half2float(tmp[i]);
    return WX_OK;
  }
  // (the rectangle first: fields that are made on demand are made for the rows asked for -- rows outside the grid would be written outside it)
  if (w <= 0 || h <= 0 || x < 0 || y < 0 || (long long)x + w > s->X || (long long)y + h > s->Y)
    return fail(s, WX_E_RANGE, "wx_read_rect: rect (%d,%d %dx%d) outside %dx%d (no wrap)", x, y, w, h, s->X, s->Y);
  const void *ptr;
  int ch, el;
  if (int rc = field_info(s, field, &ptr, &ch, &el, y, h)) {
    if (rc != -1) return rc; // (a field that is made on demand could not be made: the message is set)
    return fail(s, WX_E_INVALID, "wx_read_rect: unknown field %d", field);
  }
  const bool is_wall = (el == 1);
  if (is_wall ? (dtype != WX_DTYPE_I8 && dtype != WX_DTYPE_I32) : (dtype != WX_DTYPE_F32))
    return fail(s, WX_E_INVALID, "wx_read_rect: dtype %d does not fit field %d", dtype, field);
  const size_t texel = (size_t)ch * el;
  const char *src = (const char *)ptr + ((size_t)y * s->X + x) * texel;
  if (is_wall && dtype == WX_DTYPE_I32) {
    std::vector<int8_t> tmp((size_t)w * h * 4);
    HIPCHK(s, hipMemcpy2DAsync(tmp.data(), (size_t)w * texel, src, (size_t)s->X * texel, (size_t)w * texel, h, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    int32_t *d = (int32_t *)dst;
    for (size_t i = 0; i < tmp.size(); i++) d[i] = tmp[i];
    return WX_OK;
  }
  HIPCHK(s, hipMemcpy2DAsync(dst, (size_t)w * texel, src, (size_t)s->X * texel, (size_t)w * texel, h, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  return WX_OK;
}

int wx_read_particles(wx_sim *s, int first, int count, float *dst)
{
  if (!s || !dst) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  if (first < 0 || count < 0 || (long long)first + count > s->n_drops) return fail(s, WX_E_RANGE, "wx_read_particles: [%d, %d) outside 0..%d", first, first + count, s->n_drops);
  if (count == 0) return WX_OK;
  wait_unpacked(s); // (a pool exchange still running on the side stream)
  HIPCHK(s, hipMemcpyAsync(dst, s->drops[s->drop_cur] + 5 * (size_t)first, (size_t)count * 20, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  return WX_OK;
}

// ---- field streaming for display consumers ----
namespace {
const int kStreamFields[6] = {WX_FIELD_BASE_DISP, WX_FIELD_WATER_CUR, WX_FIELD_WALL_DISP, WX_FIELD_LIGHT_0, WX_FIELD_CURL, WX_FIELD_PRECIP_FB};
}

size_t wx_stream_bytes(int w, int h) { return (w > 0 && h > 0) ? (size_t)w * h * (16 + 16 + 4 + 16 + 4 + 16 + 8) : 0; }

void *wx_host_alloc(size_t bytes)
{
  void *p = nullptr;
  return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void wx_host_free(void *p)
{
  if (p) hipHostFree(p);
}

int wx_stream_frame(wx_sim *s, int x, int y, int w, int h, void *host_dst)
{
  if (!s || !host_dst) return WX_E_INVALID;
  if (!s->uploaded) return fail(s, WX_E_STATE, "wx_stream_frame before wx_upload");
  if (w <= 0 || h <= 0 || x < 0 || y < 0 || (long long)x + w > s->X || (long long)y + h > s->Y)
    return fail(s, WX_E_RANGE, "wx_stream_frame: rect (%d,%d %dx%d) outside %dx%d (no wrap)", x, y, w, h, s->X, s->Y);
  if (!s->copy_stream) {
    HIPCHK(s, hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    HIPCHK(s, hipEventCreateWithFlags(&s->ev_fields_ready, hipEventDisableTiming));
    HIPCHK(s, hipEventCreateWithFlags(&s->ev_copy_done, hipEventDisableTiming));
  }
  settle_edges(s);
  wait_unpacked(s); // ghost columns written on the comm stream are part of what a frame shows
  const void *ptr[6];
  size_t texel[6];
  for (int f = 0; f < 6; f++) { // (may enqueue a layout conversion of the light texture on the compute stream)
    int ch, el;
    if (int rc = field_info(s, kStreamFields[f], &ptr[f], &ch, &el, y, h)) {
      if (rc != -1) return rc;
      return fail(s, WX_E_INVALID, "wx_stream_frame: field %d", kStreamFields[f]);
    }
    texel[f] = (size_t)ch * el;
  }
  { // the seventh block: emittedLight (RGBA16F) of the rectangle, computed now on the compute stream
    const int rc = emitted_rect(s, x, y, w, h);
    if (rc != WX_OK) return rc;
  }
  // the copies start when everything enqueued so far on the compute stream has produced the fields ...
  HIPCHK(s, hipEventRecord(s->ev_fields_ready, s->stream));
  HIPCHK(s, hipStreamWaitEvent(s->copy_stream, s->ev_fields_ready, 0));
  char *dst = (char *)host_dst;
  for (int f = 0; f < 6; f++) {
    HIPCHK(s, hipMemcpy2DAsync(dst, (size_t)w * texel[f], (const char *)ptr[f] + ((size_t)y * s->X + x) * texel[f], (size_t)s->X * texel[f],
                               (size_t)w * texel[f], h, hipMemcpyDeviceToHost, s->copy_stream));
    dst += (size_t)w * h * texel[f];
  }
  HIPCHK(s, hipMemcpy2DAsync(dst, (size_t)w * 8, s->emitted + ((size_t)y * s->X + x), (size_t)s->X * 8, (size_t)w * 8, h, hipMemcpyDeviceToHost, s->copy_stream));
  // ... and the next wx_step (which overwrites them) waits for the copies on the device, not on the host
  HIPCHK(s, hipEventRecord(s->ev_copy_done, s->copy_stream));
  s->copy_in_flight = true;
  return WX_OK;
}

int wx_stream_wait(wx_sim *s)
{
  if (!s) return WX_E_INVALID;
  if (s->copy_stream) HIPCHK(s, hipStreamSynchronize(s->copy_stream));
  return WX_OK;
}

int wx_set_stream(wx_sim *s, void *hip_stream)
{
  if (!s) return WX_E_INVALID;
  HIPCHK(s, hipStreamSynchronize(s->stream));
  s->stream = (hipStream_t)hip_stream;
  return WX_OK;
}

int wx_set_comm_stream(wx_sim *s, void *hip_stream)
{
  if (!s) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  HIPCHK(s, hipStreamSynchronize(s->stream));
  settle_edges(s);
  HIPCHK(s, hipStreamSynchronize(s->stream));
  if (s->comm_stream) HIPCHK(s, hipStreamSynchronize(s->comm_stream));
  s->comm_stream = (hipStream_t)hip_stream;
  s->edges_recorded = s->unpack_pending = false;
  if (hip_stream && !s->ev_edges) {
    HIPCHK(s, hipEventCreateWithFlags(&s->ev_edges, hipEventDisableTiming));
    HIPCHK(s, hipEventCreateWithFlags(&s->ev_unpacked, hipEventDisableTiming));
  }
  // (the epoch word has to exist before the first unpack on a side stream: an edge strip polls it for exactly that unpack)
  if (hip_stream)
    if (int rc = split_launch_ready(s, false)) return rc;
  return WX_OK;
}

void *wx_device_ptr(wx_sim *s, int field)
{
  if (!s) return nullptr;
  settle_edges(s);
  wait_unpacked(s); // whatever the caller enqueues on the compute stream next sees the unpacked ghost columns (and the exchanged lightning state)
  if (field == WX_FIELD_LIGHTNING) return s->state->lightning;
  if (field == WX_FIELD_EMITTED) return emitted_rect(s, 0, 0, s->X, s->Y) == WX_OK ? s->emitted : nullptr; // (whole grid, computed now)
  const void *ptr;
  int ch, el;
  if (field_info(s, field, &ptr, &ch, &el)) return nullptr;
  // (the caller may write velocities through the pointer: devtools.seed_flow does. Rank-LOCAL knowledge: it must not change which collectives
  // this rank enqueues -- ADVICE round 5 -- so it only makes the next exchange's roll look at the state, with the period's bound checked)
  if (field == WX_FIELD_BASE_CUR) s->vx_untracked = true;
  return const_cast<void *>(ptr);
}

int wx_local_width(const wx_sim *s) { return s ? s->X : 0; }

// base_0, water_1, light_0, light_1 (float4), wall_0 (char4); handles that carry particles also exchange the feedback
// (float4) and deposition (float2) textures, which the boundary pass of the next iteration reads
size_t wx_halo_bytes(const wx_sim *s) { return s ? (size_t)s->halo * s->Y * (4 * 16 + 4 + (s->pool_remote ? 16 + 8 : 0)) : 0; }
// what a halo message of the CURRENT period occupies at the front of such a buffer: the base texture alone between slabs that agreed on
// the water-free dry stencil (wx_slab_assert_water_free), else all of it. Equal on every rank at every exchange (see halo_base_only).
size_t wx_halo_message_bytes(const wx_sim *s) { return s ? (s->halo_base_only ? (size_t)s->halo * s->Y * 16 : wx_halo_bytes(s)) : 0; }

static int halo_xstart(const wx_sim *s, int side, bool pack)
{
  if (pack) return side == 0 ? s->halo : s->X - 2 * s->halo; // outermost OWNED columns
  return side == 0 ? 0 : s->X - s->halo;                     // ghost columns
}

// sides: dev_buf[k] != NULL packs / unpacks side k (0 = left, 1 = right); both in one launch where both are given
static HaloBufs halo_bufs(const wx_sim *s, void *const dev_buf[2], bool pack, int *n_slots)
{
  const int n = s->halo * s->Y;
  const size_t o8 = (size_t)n * (s->pool_remote ? 80 : 64), o4 = o8 + (s->pool_remote ? (size_t)n * 8 : 0);
  HaloBufs hb{};
  int k = 0;
  for (int side = 0; side < 2; side++) {
    if (!dev_buf[side]) continue;
    hb.x_start[k] = halo_xstart(s, side, pack);
    hb.buf16[k] = (float4 *)dev_buf[side];
    hb.buf8[k] = (float2 *)((char *)dev_buf[side] + o8);
    hb.buf4[k] = (char4 *)((char *)dev_buf[side] + o4);
    k++;
  }
  *n_slots = k;
  return hb;
}
static int halo_pack_impl(wx_sim *s, void *const dev_buf[2])
{
  DeviceScope dev_scope(s);
  if (s->halo == 0) return fail(s, WX_E_STATE, "handle has no halo");
  if (s->halo_base_only && !(s->water_trivial && s->wall_veg_ok))
    return fail(s, WX_E_STATE, "the slabs agreed on the water-free dry stencil (wx_slab_assert_water_free(s, 1)) but THIS slab is not water-free -- its upload "
                               "carried water, or it was given new contents since: its neighbours expect base-only halo messages. Combine wx_water_free "
                               "of every slab and call wx_slab_assert_water_free on all of them again");
  const int n = s->halo * s->Y;
  const LightPlanes none{nullptr, nullptr, nullptr};
  HaloPtrs f{s->base[0], s->water[1], s->light[0], s->light[1], s->light_planar ? s->lp[0] : none, s->light_planar ? s->lp[1] : none, s->wall[0],
             s->pool_remote ? s->fb : nullptr, s->pool_remote ? s->dep : nullptr};
  hipStream_t st = exchange_stream(s);
  if (st != s->stream && s->gate_pending && s->gate_wet && !wx_tune_env("WX_SPLIT_FIX_ON_COMM")) {
    // ONE ordered launch of the WET kernel: its edge strips left exact-path cells on their own list. Consuming it HERE, on the comm
    // stream, behind a gate kernel would let the pack start while the interior still marches -- built and measured in round 5, and
    // withdrawn: k_wet_fix needs 652 bytes of scratch per lane, which the runtime hands out per dispatch ("use once") only when the
    // queue can be given it -- seen to stall for SECONDS while the next iteration's edge strips were resident and polling for this very
    // exchange (the poll limit turned the deadlock into an error; profiles/r05_slab_protocol_cost.txt). So the list is consumed on the
    // compute stream behind the whole launch, and the pack waits for that: no overlap inside this iteration, none lost in the next.
    settle_edges(s);
    HIPCHK(s, hipEventRecord(s->ev_edges, s->stream));
    s->edges_recorded = true;
    HIPCHK(s, hipStreamWaitEvent(st, s->ev_edges, 0));
    s->gate_passed = true;
  } else if (st != s->stream && s->gate_pending) {
    // the latest iteration was ONE ordered launch whose edge strips report on a device word: a one-wave gate kernel waits for them (the
    // interior strips are still marching), then the pack may read (dry stencil: no exact-path list, every kernel here is scratch-free)
    hipLaunchKernelGGL(k_strip_gate, dim3(1), dim3(64), 0, st, s->sync_words, s->arrive_target);
    if (s->gate_wet && !wx_tune_env("WX_SPLIT_NOFIX2")) {
      const WetFixList fix2{s->fix_count2, s->fix_cells2, s->fix_cap2, nullptr, &s->state->fastest_bits, nullptr};
      launch_wet_fix(s->gate_iter, s->full_ctx, s->gate_in, s->gate_out, fix2, &s->state->fix_overflow, s->gate_opt_out, st, 64);
    }
    s->gate_pending = false;
    s->gate_passed = true;
  } else if (st != s->stream && s->gate_passed) {
    // (the other side of the same exchange, packed by a call of its own: the gate is earlier in this stream)
  } else if (st != s->stream) { // the packed columns are final behind ev_edges (or, without an edge-first step, behind everything enqueued so far)
    if (!s->edges_recorded) {
      HIPCHK(s, hipEventRecord(s->ev_edges, s->stream));
      s->edges_recorded = true;
    }
    HIPCHK(s, hipStreamWaitEvent(st, s->ev_edges, 0));
  }
  settle_edges(s); // (an in-order pack: the edge strips' exact-path cells on the compute stream)
  ProfScope ps(s, K_HALO);
  int slots = 0;
  const HaloBufs hb = halo_bufs(s, dev_buf, true, &slots);
  hipLaunchKernelGGL(k_halo_pack, dim3((n + 255) / 256, slots), dim3(256), 0, st, f, s->X, s->Y, s->halo, hb, s->halo_base_only ? 1 : 0);
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}
static int halo_unpack_impl(wx_sim *s, void *const dev_buf[2])
{
  DeviceScope dev_scope(s);
  if (s->halo == 0) return fail(s, WX_E_STATE, "handle has no halo");
  const int n = s->halo * s->Y;
  const LightPlanes none{nullptr, nullptr, nullptr};
  HaloPtrs f{s->base[0], s->water[1], s->light[0], s->light[1], s->light_planar ? s->lp[0] : none, s->light_planar ? s->lp[1] : none, s->wall[0],
             s->pool_remote ? s->fb : nullptr, s->pool_remote ? s->dep : nullptr};
  hipStream_t st = exchange_stream(s);
  ProfScope ps(s, K_HALO);
  int slots = 0;
  const HaloBufs hb = halo_bufs(s, dev_buf, false, &slots);
  // (a base-only message carries no water to validate: what the neighbour holds was checked by its own upload)
  const bool check = s->water_trivial && !s->halo_base_only;
  hipLaunchKernelGGL(k_halo_unpack, dim3((n + 255) / 256, slots), dim3(256), 0, st, f, s->X, s->Y, s->halo, hb,
                     check ? &s->state->ghost_nontrivial : nullptr, s->halo_base_only ? 1 : 0);
  if (check) s->ghost_check = true;
  if (st != s->stream) // whoever touches the ghost columns next on the compute stream waits for this
    if (int rc = mark_unpacked(s, st)) return rc;
  if (s->pool_remote) { // the ghost tiles of the feedback texture now hold a neighbour's values: nothing is "known zero" any more
    hipMemsetAsync(s->sg.fb_zero, 0, 2 * (size_t)s->sg.TXn * s->sg.TYn, s->stream);
    s->fb_dirty = true;
  }
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}
int wx_halo_pack(wx_sim *s, int side, void *dev_buf)
{
  if (!s || !dev_buf || (side != 0 && side != 1)) return WX_E_INVALID;
  void *const b[2] = {side == 0 ? dev_buf : nullptr, side == 1 ? dev_buf : nullptr};
  return halo_pack_impl(s, b);
}
int wx_halo_unpack(wx_sim *s, int side, const void *dev_buf)
{
  if (!s || !dev_buf || (side != 0 && side != 1)) return WX_E_INVALID;
  void *const b[2] = {side == 0 ? const_cast<void *>(dev_buf) : nullptr, side == 1 ? const_cast<void *>(dev_buf) : nullptr};
  return halo_unpack_impl(s, b);
}
int wx_halo_pack_both(wx_sim *s, void *dev_left, void *dev_right)
{
  if (!s || !dev_left || !dev_right) return WX_E_INVALID;
  void *const b[2] = {dev_left, dev_right};
  return halo_pack_impl(s, b);
}
int wx_halo_unpack_both(wx_sim *s, const void *dev_left, const void *dev_right)
{
  if (!s || !dev_left || !dev_right) return WX_E_INVALID;
  void *const b[2] = {const_cast<void *>(dev_left), const_cast<void *>(dev_right)};
  return halo_unpack_impl(s, b);
}

// ---- particles on slabs: reconciliation of the replicated droplet pool (protocol in slab.py / SlabP) ----
int wx_slab_set_rank(wx_sim *s, int rank)
{
  if (!s || rank < 0 || rank > 1023) return WX_E_INVALID;
  s->rank = rank;
  return WX_OK;
}

int wx_slab_period_begin(wx_sim *s)
{
  if (!s) return WX_E_INVALID;
  s->period_j = 0;
  return WX_OK;
}

size_t wx_pool_event_bytes(const wx_sim *s) { return s && s->pool_remote ? POOL_HDR + (size_t)s->pool_event_cap * sizeof(PoolEvent) : 0; }
size_t wx_pool_edge_bytes(const wx_sim *s) { return s && s->pool_remote ? POOL_HDR + (size_t)s->pool_edge_cap * sizeof(PoolRec) : 0; }

// The stream the droplet-pool exchange kernels run on: the side stream when the handle has one (they then overlap with the interior
// strips of the next iteration, like the grid halos), the compute stream otherwise and always in the exact mode (whose per-iteration
// rounds need the finished iteration and are needed by the next one).
static hipStream_t pool_stream(const wx_sim *s) { return exchange_stream(s); }
// side stream: the pool kernels read what the iterations enqueued so far leave behind
static int pool_fence(wx_sim *s, hipStream_t st)
{
  if (st == s->stream) return WX_OK;
  if (!s->edges_recorded) {
    HIPCHK(s, hipEventRecord(s->ev_edges, s->stream));
    s->edges_recorded = true;
  }
  HIPCHK(s, hipStreamWaitEvent(st, s->ev_edges, 0));
  return WX_OK;
}
// side stream: whatever touches the pool (or the ghost columns) next on the compute stream waits for what was just enqueued
static int pool_applied(wx_sim *s, hipStream_t st)
{
  if (st == s->stream) return WX_OK;
  return mark_unpacked(s, st);
}

#define POOL_ONLY(s, what) \
  if (!(s)->pool_remote) return fail((s), WX_E_STATE, what ": not a slab handle with particles")

// mode: 0 = status flips only, 1 = + the iteration record of the exact mode (lightning request, deposit at texel (0,0)), 2 = + the
// period record of the library's own transport (this rank's lightning state)
static int pool_events_pack_mode(wx_sim *s, void *dev_buf, int mode);
static int pool_events_apply_mode(wx_sim *s, const void *dev_bufs, int n_ranks, size_t stride_bytes, int mode);
int wx_pool_events_pack(wx_sim *s, void *dev_buf)
{
  if (!s || !dev_buf) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  POOL_ONLY(s, "wx_pool_events_pack");
  return pool_events_pack_mode(s, dev_buf, s->pool_exact ? 1 : 0);
}
static int pool_events_pack_mode(wx_sim *s, void *dev_buf, int mode)
{
  hipStream_t st = pool_stream(s);
  if (int rc = pool_fence(s, st)) return rc;
  // the header: entry counter (+ 3 spare words). With an iteration / period record (mode != 0) entry 0 is the record's and the counter
  // starts at 1 (k_pool_events_pack)
  HIPCHK(s, hipMemsetD32Async((hipDeviceptr_t)dev_buf, mode != 0 ? 1 : 0, POOL_HDR / 4, st));
  // (exact mode: + this rank's iteration record; the deposit at the domain's texel (0,0) comes from the rank that owns global column 0)
  hipLaunchKernelGGL(k_pool_events_pack, dim3((s->n_drops + 255) / 256), dim3(256), 0, st, s->n_drops, s->rank, s->pool_event_cap, s->pool_flips, s->pool_owned,
                     s->drops[0], (int *)dev_buf, (PoolEvent *)((char *)dev_buf + POOL_HDR), s->state, mode,
                     (mode == 1 && s->x0 == 0) ? s->fb + s->halo : nullptr);
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}

int wx_pool_events_apply(wx_sim *s, const void *dev_bufs, int n_ranks, size_t stride_bytes)
{
  if (!s || !dev_bufs || n_ranks < 1) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  POOL_ONLY(s, "wx_pool_events_apply");
  return pool_events_apply_mode(s, dev_bufs, n_ranks, stride_bytes, s->pool_exact ? 1 : 0);
}
static int pool_events_apply_mode(wx_sim *s, const void *dev_bufs, int n_ranks, size_t stride_bytes, int mode)
{
  const size_t stride = stride_bytes ? stride_bytes : wx_pool_event_bytes(s);
  if (stride < (size_t)POOL_HDR || stride > wx_pool_event_bytes(s)) return fail(s, WX_E_INVALID, "wx_pool_events_apply: stride %zu outside 16 .. %zu", stride, wx_pool_event_bytes(s));
  const dim3 grid(64, n_ranks), block(256);
  hipStream_t st = pool_stream(s);
  if (int rc = pool_fence(s, st)) return rc;
  const char *b = (const char *)dev_bufs;
  const int cap = (int)((stride - POOL_HDR) / sizeof(PoolEvent)); // entries a rank's (possibly truncated) buffer holds
  hipLaunchKernelGGL(k_pool_check, dim3(1), dim3(64), 0, st, n_ranks, stride, cap, b, s->state, 1);
  if (s->ev_seen_host) { // the library's transport sizes the coming all-gathers by what this one carried (wx_comm.h: pool_stride_update)
    HIPCHK(s, hipMemcpyAsync(s->ev_seen_host, &s->state->pool_seen_max, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(s, hipEventRecord(s->ev_counted, st));
    s->count_pending = true;
  }
  hipLaunchKernelGGL(k_pool_events_best, grid, block, 0, st, n_ranks, stride, cap, b, s->pool_best);
  if (mode == 1) HIPCHK(s, hipMemsetAsync(&s->state->pool_retired, 0, 4, st));
  hipLaunchKernelGGL(k_pool_events_apply, grid, block, 0, st, n_ranks, stride, cap, b, s->pool_best, s->rank, s->drops[0], s->pool_remote, s->geo, mode == 1 ? 1 : 0,
                     mode == 1 ? &s->state->pool_retired : nullptr);
  if (mode == 2) hipLaunchKernelGGL(k_pool_lightning_latest, dim3(1), dim3(64), 0, st, n_ranks, stride, cap, b, s->state);
  if (mode == 1 && s->exact_pending > 0) { // the iteration that just ran: lightning of the whole domain, the 600-iteration inactive count
    const int64_t it = s->iter - 1;
    const int refresh = it % 600 == 0 ? 1 : 0;
    if (refresh) hipLaunchKernelGGL(k_pool_count_inactive, dim3((s->n_drops + 255) / 256), dim3(256), 0, st, s->n_drops, s->drops[0], s->pool_remote, s->state);
    hipLaunchKernelGGL(k_pool_exact_resolve, dim3(1), dim3(64), 0, st, n_ranks, stride, cap, b, s->state, (float)it, refresh, &s->state->pool_retired);
    s->exact_pending = 0;
  }
  hipLaunchKernelGGL(k_pool_events_reset, grid, block, 0, st, n_ranks, stride, cap, b, s->pool_best);
  s->pool_check = true;
  if (int rc = pool_applied(s, st)) return rc;
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}

int wx_pool_edges_pack(wx_sim *s, void *dev_left, void *dev_right, int refresh_inactive)
{
  if (!s || !dev_left || !dev_right) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  POOL_ONLY(s, "wx_pool_edges_pack");
  hipStream_t st = pool_stream(s);
  if (int rc = pool_fence(s, st)) return rc;
  HIPCHK(s, hipMemsetAsync(dev_left, 0, POOL_HDR, st));
  HIPCHK(s, hipMemsetAsync(dev_right, 0, POOL_HDR, st));
  if (refresh_inactive) HIPCHK(s, hipMemsetAsync(&s->state->px_count, 0, 4, st));
  hipLaunchKernelGGL(k_pool_edges_pack, dim3((s->n_drops + 255) / 256), dim3(256), 0, st, s->geo, s->n_drops, s->halo, s->X - s->halo, s->halo, s->pool_edge_cap,
                     s->drops[0], s->pool_remote, (int *)dev_left, (PoolRec *)((char *)dev_left + POOL_HDR), (int *)dev_right,
                     (PoolRec *)((char *)dev_right + POOL_HDR), s->state, refresh_inactive);
  if (refresh_inactive) hipLaunchKernelGGL(k_inactive_from_count, dim3(1), dim3(1), 0, st, s->state);
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}

int wx_pool_edges_apply(wx_sim *s, const void *dev_buf)
{
  if (!s || !dev_buf) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  POOL_ONLY(s, "wx_pool_edges_apply");
  hipStream_t st = pool_stream(s);
  if (int rc = pool_fence(s, st)) return rc;
  hipLaunchKernelGGL(k_pool_check, dim3(1), dim3(64), 0, st, 1, (size_t)0, s->pool_edge_cap, (const char *)dev_buf, s->state, 0);
  hipLaunchKernelGGL(k_pool_edges_apply, dim3(64), dim3(256), 0, st, s->pool_edge_cap, (const int *)dev_buf, (const PoolRec *)((const char *)dev_buf + POOL_HDR),
                     s->drops[0], s->pool_remote);
  s->pool_check = true;
  if (int rc = pool_applied(s, st)) return rc;
  HIPCHK(s, hipGetLastError());
  return WX_OK;
}

int wx_pool_flags(wx_sim *s, uint8_t *host_dst)
{
  if (!s || !host_dst) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  POOL_ONLY(s, "wx_pool_flags");
  wait_unpacked(s); // (a pool exchange still running on the side stream)
  unsigned char *d = nullptr;
  if (hipMalloc((void **)&d, (size_t)s->n_drops) != hipSuccess) return fail(s, WX_E_NOMEM, "wx_pool_flags");
  hipLaunchKernelGGL(k_pool_flags, dim3((s->n_drops + 255) / 256), dim3(256), 0, s->stream, s->geo, s->n_drops, s->halo, s->X - s->halo, s->drops[0], s->pool_remote, d);
  const hipError_t e1 = hipMemcpyAsync(host_dst, d, (size_t)s->n_drops, hipMemcpyDeviceToHost, s->stream), e2 = hipStreamSynchronize(s->stream);
  hipFree(d);
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(s, WX_E_DEVICE, "wx_pool_flags: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  return WX_OK;
}

int wx_lightning_get(wx_sim *s, float out[4])
{
  if (!s || !out) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  wait_unpacked(s);
  HIPCHK(s, hipMemcpyAsync(out, s->state->lightning, 16, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  return WX_OK;
}

int wx_lightning_set(wx_sim *s, const float in[4])
{
  if (!s || !in) return WX_E_INVALID;
  DeviceScope dev_scope(s);
  wait_unpacked(s);
  HIPCHK(s, hipMemcpyAsync(s->state->lightning, in, 16, hipMemcpyHostToDevice, s->stream));
  HIPCHK(s, hipStreamSynchronize(s->stream));
  return WX_OK;
}

int wx_profile(wx_sim *s, int enable)
{
  if (!s) return WX_E_INVALID;
  collect_profile(s);
  s->profiling = enable != 0;
  if (enable) {
    for (int k = 0; k < K_COUNT; k++) {
      s->prof_ms[k] = 0;
      s->prof_n[k] = 0;
    }
  }
  return WX_OK;
}

int wx_profile_read(wx_sim *s, int cap, float *ms, int *launches)
{
  if (!s || !ms || !launches) return WX_E_INVALID;
  collect_profile(s);
  for (int k = 0; k < K_COUNT && k < cap; k++) {
    ms[k] = (float)s->prof_ms[k];
    launches[k] = s->prof_n[k];
    s->prof_ms[k] = 0;
    s->prof_n[k] = 0;
  }
  return WX_OK;
}

#include "wx_comm.h"

} // extern "C"
