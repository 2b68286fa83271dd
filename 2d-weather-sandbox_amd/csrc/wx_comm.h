// wx_comm.h -- the halo exchange INSIDE the library (included at the end of wxsim.hip, inside its extern "C" block).
//
// BASELINE north_star: "x-slab decomposition ... one-cell halo exchange on RCCL send/recv over xGMI overlapped on a side HIP
// stream; host code stays in JavaScript". Rounds 1-3 packed and unpacked halos in the library but left the transport to the Python
// host (torch.distributed); a Node process could drive one GPU only. Here:
//
//   * one rank per process:  wx_comm_unique_id / wx_comm_init (ncclCommInitRank) -> wx_exchange (pack -> ncclGroupStart; ncclSend x 2;
//     ncclRecv x 2; ncclGroupEnd -> unpack, all on the handle's comm stream) -> wx_slab_step (n iterations, one exchange per
//     halo / 6 iterations, edge strips first / interior first around it: the host never blocks);
//   * one process, N slabs (what a Node host needs): wx_group_create (a handle per slab, on N devices -- ncclCommInitAll -- or, on a
//     box with fewer devices, several slabs per device with device-to-device copies between the slabs' buffers behind the same
//     calls: "local" transport) -> wx_group_step / wx_group_sync.
// RCCL is bound at run time (dlopen: the copy the process already holds -- PyTorch ships its own -- else the ROCm one), so the
// library loads and runs single-GPU without it. Slabs with particles run the partitioned-pool protocol (include/wxsim.h) through the
// same calls: status flips (+ lightning) all-gathered with a fixed stride -- no host round trip --, edge droplets in the halos' batch of
// transfers, in order on the compute stream; with WX_OPT_POOL_EXACT one iteration at a time.
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};
static RcclApi *rccl_api()
{
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names) // the copy this process already holds (one RCCL per HIP runtime)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (!api.lib) api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) api.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) {
      api.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?");
    } else {
      bool ok = true;
#define WX_SYM(field, name) ok = ((api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name))) != nullptr) && ok
      WX_SYM(GetUniqueId, "ncclGetUniqueId");
      WX_SYM(CommInitRank, "ncclCommInitRank");
      WX_SYM(CommInitAll, "ncclCommInitAll");
      WX_SYM(CommDestroy, "ncclCommDestroy");
      WX_SYM(GroupStart, "ncclGroupStart");
      WX_SYM(GroupEnd, "ncclGroupEnd");
      WX_SYM(Send, "ncclSend");
      WX_SYM(Recv, "ncclRecv");
      WX_SYM(AllGather, "ncclAllGather");
      WX_SYM(GetErrorString, "ncclGetErrorString");
#undef WX_SYM
      if (!ok) {
        api.why = "librccl.so lacks an entry point";
        api.lib = nullptr;
      }
    }
  }
  return api.lib ? &api : nullptr;
}
static const char *rccl_missing() { return "RCCL is not available in this process (librccl.so could not be loaded)"; }

#define NCCLCHK(s, expr)                                                                                            \
  do {                                                                                                              \
    ncclResult_t r_ = (expr);                                                                                       \
    if (r_ != ncclSuccess) return fail((s), WX_E_DEVICE, "%s: %s", #expr, rccl_api()->GetErrorString(r_));          \
  } while (0)

// iterations per exchange under the |vx| bound the slabs currently agree on (wx_slab_period: halo / cone, cone = 6 + floor(bound))
static int iters_per_exchange(const wx_sim *s) { return std::max(1, wx_slab_period(s)); }
// the stream the exchange of a slab runs on (wxsim.hip: exchange_stream -- the comm stream unless the exchange must stay in order)
static hipStream_t xstream(const wx_sim *s) { return exchange_stream(s); }
// bytes per rank in the all-gather of status-flip events: a stride every rank knows without a host round trip inside the period
// (pool_stride_update); a rank with more flips than fit is reported by the next blocking call (pool_overflow)
static size_t pool_stride(const wx_sim *s)
{
  // (exact mode sends a round after EVERY iteration: until the first counts are in, 262 144 events per rank and round -- eight times
  // the start-up burst of configs[4] -- instead of the whole buffer; the per-period protocol's first exchange carries a whole period's
  // burst and takes everything)
  const long long ev = (s->pool_exact && s->pool_stride_events > (1 << 18)) ? (1 << 18) : s->pool_stride_events;
  const size_t want = (POOL_HDR + sizeof(PoolEvent) * (size_t)(1 + ev) + 4095) / 4096 * 4096;
  return std::min(want, wx_pool_event_bytes(s));
}

// send / recv buffers (+ the droplet-pool buffers of slabs with particles), a comm stream of the library's own for grid-only slabs
static int transport_prepare(wx_sim *s, int world)
{
  if (s->halo == 0) return fail(s, WX_E_STATE, "the handle has no ghost columns (wx_create_slab with halo > 0)");
  const size_t bytes = wx_halo_bytes(s);
  if (s->xbytes != bytes) {
    for (int i = 0; i < 2; i++) {
      hipFree(s->xsend[i]);
      hipFree(s->xrecv[i]);
      s->xsend[i] = s->xrecv[i] = nullptr;
      if (hipMalloc((void **)&s->xsend[i], bytes) != hipSuccess || hipMalloc((void **)&s->xrecv[i], bytes) != hipSuccess)
        return fail(s, WX_E_NOMEM, "halo buffers: 4 x %zu bytes", bytes);
    }
    s->xbytes = bytes;
  }
  if (s->pool_remote && (!s->ev_mine || s->ev_world != world)) {
    hipFree(s->ev_mine);
    hipFree(s->ev_all);
    s->ev_mine = s->ev_all = nullptr;
    const size_t eb = wx_pool_event_bytes(s), gb = wx_pool_edge_bytes(s);
    if (hipMalloc((void **)&s->ev_mine, eb) != hipSuccess || hipMalloc((void **)&s->ev_all, eb * (size_t)world) != hipSuccess)
      return fail(s, WX_E_NOMEM, "droplet-pool event buffers: %zu bytes", eb * (size_t)(world + 1));
    if (!s->ev_seen_host) {
      HIPCHK(s, hipHostMalloc((void **)&s->ev_seen_host, sizeof(int), hipHostMallocDefault));
      *s->ev_seen_host = 0;
      HIPCHK(s, hipEventCreateWithFlags(&s->ev_counted, hipEventDisableTiming));
    }
    s->pool_stride_events = 1 << 30; // a new ring: everything until the first counts are in
    s->count_pending = false;
    for (int i = 0; i < 2; i++) {
      hipFree(s->psend[i]);
      hipFree(s->precv[i]);
      if (hipMalloc((void **)&s->psend[i], gb) != hipSuccess || hipMalloc((void **)&s->precv[i], gb) != hipSuccess) return fail(s, WX_E_NOMEM, "droplet-pool edge buffers");
    }
    s->ev_world = world;
  }
  if (!s->vx_dev || s->vx_world != world) { // the slabs' measured |vx| maxima: one word each, all-gathered / copied with every exchange
    hipFree(s->vx_dev);
    if (s->vx_host) hipHostFree(s->vx_host);
    s->vx_dev = nullptr;
    s->vx_host = nullptr;
    if (hipMalloc((void **)&s->vx_dev, sizeof(int) * (size_t)(1 + world)) != hipSuccess || hipHostMalloc((void **)&s->vx_host, sizeof(int) * 2 * (size_t)world, hipHostMallocDefault) != hipSuccess)
      return fail(s, WX_E_NOMEM, "the |vx| words of the exchange");
    memset(s->vx_host, 0, sizeof(int) * 2 * (size_t)world);
    for (int i = 0; i < 2; i++)
      if (!s->ev_vx[i]) HIPCHK(s, hipEventCreateWithFlags(&s->ev_vx[i], hipEventDisableTiming));
    s->vx_have[0] = s->vx_have[1] = false;
    s->vx_slot = 0;
    s->vx_world = world;
  }
  // the side stream of the exchange; the exact particle mode runs everything in order on the compute stream (wx_set_option may have
  // switched modes since the last call)
  const bool in_order = (s->pool_remote && s->pool_exact) || s->exchange_in_order;
  if (!in_order && !s->comm_stream) {
    if (!s->own_comm_stream) { // highest priority: its small pack / unpack kernels run next to a marching kernel that holds every wave slot
      int prio_lo = 0, prio_hi = 0;
      if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
      HIPCHK(s, hipStreamCreateWithPriority(&s->own_comm_stream, hipStreamNonBlocking, prio_hi));
    }
    if (int rc = wx_set_comm_stream(s, s->own_comm_stream)) return rc;
  } else if (in_order && s->comm_stream && s->comm_stream == s->own_comm_stream) {
    if (int rc = wx_set_comm_stream(s, nullptr)) return rc;
  }
  return WX_OK;
}

void transport_release(wx_sim *s)
{
  if (s->comm) {
    if (RcclApi *a = rccl_api()) a->CommDestroy((ncclComm_t)s->comm);
    s->comm = nullptr;
  }
  for (int i = 0; i < 2; i++) {
    hipFree(s->xsend[i]);
    hipFree(s->xrecv[i]);
    hipFree(s->psend[i]);
    hipFree(s->precv[i]);
    s->xsend[i] = s->xrecv[i] = s->psend[i] = s->precv[i] = nullptr;
  }
  hipFree(s->ev_mine);
  hipFree(s->ev_all);
  s->ev_mine = s->ev_all = nullptr;
  hipFree(s->vx_dev);
  s->vx_dev = nullptr;
  if (s->vx_host) hipHostFree(s->vx_host);
  s->vx_host = nullptr;
  s->vx_world = 0;
  for (int i = 0; i < 2; i++) {
    if (s->ev_vx[i]) hipEventDestroy(s->ev_vx[i]);
    s->ev_vx[i] = nullptr;
  }
  if (s->ev_seen_host) hipHostFree(s->ev_seen_host);
  s->ev_seen_host = nullptr;
  if (s->ev_counted) hipEventDestroy(s->ev_counted);
  s->ev_counted = nullptr;
  s->count_pending = false;
  s->xbytes = 0;
  for (hipEvent_t *e : {&s->ev_packed, &s->ev_copied, &s->ev_evpacked, &s->ev_evcopied}) {
    if (*e) hipEventDestroy(*e);
    *e = nullptr;
  }
  if (s->own_comm_stream) {
    if (s->comm_stream == s->own_comm_stream) s->comm_stream = nullptr;
    hipStreamDestroy(s->own_comm_stream);
    s->own_comm_stream = nullptr;
  }
  if (s->own_stream) {
    if (s->stream == s->own_stream) s->stream = nullptr;
    hipStreamDestroy(s->own_stream);
    s->own_stream = nullptr;
  }
}

// ---- the exchange, written once for both ways of running it ----
// The slabs this process drives: one (a rank of a job, RCCL) or all of them (a group, RCCL between devices or device-to-device copies).
struct Party {
  wx_sim *s;
  ncclComm_t comm; // RCCL transports
  int rank;        // position in the ring
};
struct Ring {
  std::vector<Party> p;
  int world = 1;
  int transport = WX_TRANSPORT_RCCL;
  std::string *err = nullptr; // where a group wants its messages
};
static int rfail(Ring &R, wx_sim *s, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (s) s->err = buf;
  if (R.err) *R.err = buf;
  return code;
}
static int rpass(Ring &R, wx_sim *s, int rc)
{
  if (rc != WX_OK && R.err) *R.err = s->err;
  return rc;
}
// local transport: every slab's exchange stream waits for event `ev` of every OTHER slab (or of its two ring neighbours only)
static int local_wait(Ring &R, hipEvent_t wx_sim::*ev, bool neighbours_only)
{
  const int n = (int)R.p.size();
  for (int i = 0; i < n; i++) {
    wx_sim *s = R.p[i].s;
    DeviceScope ds(s);
    for (int r = 0; r < n; r++) {
      if (r == i || (neighbours_only && r != (i + 1) % n && r != (i + n - 1) % n)) continue;
      if (hipStreamWaitEvent(xstream(s), R.p[r].s->*ev, 0) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "hipStreamWaitEvent");
    }
  }
  return WX_OK;
}
static int local_record(Ring &R, hipEvent_t wx_sim::*ev)
{
  for (Party &q : R.p) {
    DeviceScope ds(q.s);
    if (hipEventRecord(q.s->*ev, xstream(q.s)) != hipSuccess) return rfail(R, q.s, WX_E_DEVICE, "hipEventRecord");
  }
  return WX_OK;
}

// status flips (+ the iteration / period record, `mode`: 1 = exact mode's lightning request, 2 = this rank's lightning state) of every
// slab -> all-gather -> applied on every slab
static int pool_events_round(Ring &R, int mode)
{
  const bool local = R.transport == WX_TRANSPORT_LOCAL;
  if (local) // nobody may still be copying my previous events
    if (int rc = local_wait(R, &wx_sim::ev_evcopied, false)) return rc;
  for (Party &q : R.p) {
    DeviceScope ds(q.s);
    if (int rc = rpass(R, q.s, pool_events_pack_mode(q.s, q.s->ev_mine, mode))) return rc;
  }
  if (local) {
    if (int rc = local_record(R, &wx_sim::ev_evpacked)) return rc;
    if (int rc = local_wait(R, &wx_sim::ev_evpacked, false)) return rc;
    for (Party &q : R.p) {
      wx_sim *s = q.s;
      DeviceScope ds(s);
      const size_t stride = pool_stride(s);
      for (Party &o : R.p)
        if (hipMemcpyAsync(s->ev_all + (size_t)o.rank * stride, o.s->ev_mine, stride, hipMemcpyDefault, xstream(s)) != hipSuccess)
          return rfail(R, s, WX_E_DEVICE, "device-to-device copy of the status-flip events");
    }
    if (int rc = local_record(R, &wx_sim::ev_evcopied)) return rc;
  } else {
    RcclApi *a = rccl_api();
    if (a->GroupStart() != ncclSuccess) return rfail(R, nullptr, WX_E_DEVICE, "ncclGroupStart");
    for (Party &q : R.p) {
      DeviceScope ds(q.s);
      const ncclResult_t r = a->AllGather(q.s->ev_mine, q.s->ev_all, pool_stride(q.s), ncclUint8, q.comm, xstream(q.s));
      if (r != ncclSuccess) {
        a->GroupEnd();
        return rfail(R, q.s, WX_E_DEVICE, "ncclAllGather of the status-flip events: %s", a->GetErrorString(r));
      }
    }
    if (a->GroupEnd() != ncclSuccess) return rfail(R, nullptr, WX_E_DEVICE, "ncclGroupEnd");
  }
  for (Party &q : R.p) {
    DeviceScope ds(q.s);
    if (int rc = rpass(R, q.s, pool_events_apply_mode(q.s, q.s->ev_all, R.world, pool_stride(q.s), mode))) return rc;
  }
  return WX_OK;
}

// The stride of the coming all-gathers, from the counts the rounds since the last exchange carried: every rank saw the same headers,
// so every rank arrives at the same number. The only host wait of the protocol, and on something that finished a period ago (the
// previous exchange's status-flip round); it bounds how far the host runs ahead of the device to one exchange period.
static int pool_stride_update(Ring &R)
{
  for (Party &q : R.p) {
    wx_sim *s = q.s;
    if (!s->count_pending) continue;
    DeviceScope ds(s);
    if (hipEventSynchronize(s->ev_counted) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "waiting for the status-flip counts: %s", hipGetErrorString(hipGetLastError()));
    const long long seen = *s->ev_seen_host;
    s->count_pending = false;
    if (hipMemsetAsync(&s->state->pool_seen_max, 0, 4, pool_stream(s)) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "hipMemsetAsync");
    s->pool_stride_events = (int)std::min<long long>(std::max<long long>(65536, 4 * seen), 1 << 30);
  }
  return WX_OK;
}

// ---- slabs exact at any speed: the ring agrees on a |vx| bound per exchange period (include/wxsim.h, VxTrack in wx_tile.h) ----
static float bits_to_float(int b)
{
  float v;
  memcpy(&v, &b, 4);
  return v;
}
// After an upload: every slab looks at its state, the ring takes the maximum. The one place where the protocol waits for the device -- once
// per upload, before the first iteration. With one rank per process the decision below is rank-local and the all-gather is a collective:
// wx_upload / wx_setup_* on the slabs of an initialised ring are therefore COLLECTIVE calls (every rank, before the next exchange:
// include/wxsim.h); nothing else sets vx_stale (velocities written through wx_device_ptr only mark the state for the next roll's scan).
static int ring_vx_bootstrap(Ring &R)
{
  bool stale = false;
  for (Party &q : R.p) stale = stale || q.s->vx_stale;
  if (!stale) return WX_OK;
  float v = 0.0f;
  for (Party &q : R.p) {
    DeviceScope ds(q.s);
    float vi = 0.0f;
    if (int rc = rpass(R, q.s, wx_slab_vx_take(q.s, &vi))) return rc;
    v = std::max(v, vi);
  }
  if (R.transport == WX_TRANSPORT_RCCL && (int)R.p.size() < R.world) { // one rank per process: the other ranks' maxima
    wx_sim *s = R.p[0].s;
    DeviceScope ds(s);
    RcclApi *a = rccl_api();
    int bits;
    memcpy(&bits, &v, 4);
    std::vector<int> all((size_t)R.world, 0);
    if (hipMemcpyAsync(s->vx_dev, &bits, 4, hipMemcpyHostToDevice, xstream(s)) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "hipMemcpyAsync");
    const ncclResult_t r = a->AllGather(s->vx_dev, s->vx_dev + 1, 1, ncclInt32, R.p[0].comm, xstream(s));
    if (r != ncclSuccess) return rfail(R, s, WX_E_DEVICE, "ncclAllGather of the slabs' |vx|: %s", a->GetErrorString(r));
    if (hipMemcpyAsync(all.data(), s->vx_dev + 1, 4 * (size_t)R.world, hipMemcpyDeviceToHost, xstream(s)) != hipSuccess || hipStreamSynchronize(xstream(s)) != hipSuccess)
      return rfail(R, s, WX_E_DEVICE, "reading the slabs' |vx| back");
    for (int b : all) v = std::max(v, bits_to_float(b));
  }
  for (Party &q : R.p) {
    if (int rc = rpass(R, q.s, wx_slab_set_vx_bound(q.s, v))) return rc;
    q.s->vx_have[0] = q.s->vx_have[1] = false;
  }
  return WX_OK;
}
// With every exchange: each slab's maximum of the period travels as one word (all-gathered over RCCL, or copied to the hosts' pinned
// words where the slabs share a process), and the period that starts now is sized by the two LATEST COMPLETE measurements -- the rolls of
// the previous exchanges, which finished a period ago: the host never waits for the device here.
static int ring_vx_roll(Ring &R)
{
  wx_sim *s0 = R.p[0].s;
  const bool rccl = R.transport == WX_TRANSPORT_RCCL;
  const int slot = s0->vx_slot, prev = slot ^ 1, W = R.world;
  // 1. the bound of the coming period, from what the earlier rolls left in the pinned words (read BEFORE the new roll overwrites `slot`)
  float v = s0->vx_known;
  if (s0->vx_have[prev]) {
    v = 0.0f;
    for (Party &q : R.p) {
      wx_sim *s = q.s;
      DeviceScope ds(s);
      if (hipEventSynchronize(s->ev_vx[prev]) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "waiting for the slabs' |vx| of the previous exchange");
      for (int sl = 0; sl < 2; sl++) {
        if (!s->vx_have[sl]) continue; // (the older roll too: a flow that slowed down for one period keeps its margin for one more)
        if (rccl)
          for (int r = 0; r < W; r++) v = std::max(v, bits_to_float(s->vx_host[sl * W + r]));
        else
          v = std::max(v, bits_to_float(s->vx_host[sl * W + q.rank]));
      }
    }
  }
  // 2. this exchange's roll
  for (Party &q : R.p) {
    wx_sim *s = q.s;
    DeviceScope ds(s);
    hipStream_t st = xstream(s);
    if (s->vx_untracked) { // iterations of kernels that do not report their |vx|: look at the state they left
      if (st != s->stream) { // (such kernels are never split: the exchange is behind the whole iteration)
        if (!s->edges_recorded) {
          if (hipEventRecord(s->ev_edges, s->stream) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "hipEventRecord");
          s->edges_recorded = true;
        }
        if (hipStreamWaitEvent(st, s->ev_edges, 0) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "hipStreamWaitEvent");
      }
      vx_scan_enqueue(s, st, true);
      s->vx_untracked = false;
    }
    hipLaunchKernelGGL(k_vx_roll, dim3(1), dim3(1), 0, st, s->state, s->vx_dev);
  }
  if (rccl) {
    RcclApi *a = rccl_api();
    if (a->GroupStart() != ncclSuccess) return rfail(R, nullptr, WX_E_DEVICE, "ncclGroupStart");
    for (Party &q : R.p) {
      DeviceScope ds(q.s);
      const ncclResult_t r = a->AllGather(q.s->vx_dev, q.s->vx_dev + 1, 1, ncclInt32, q.comm, xstream(q.s));
      if (r != ncclSuccess) {
        a->GroupEnd();
        return rfail(R, q.s, WX_E_DEVICE, "ncclAllGather of the slabs' |vx|: %s", a->GetErrorString(r));
      }
    }
    if (a->GroupEnd() != ncclSuccess) return rfail(R, nullptr, WX_E_DEVICE, "ncclGroupEnd");
  }
  for (Party &q : R.p) {
    wx_sim *s = q.s;
    DeviceScope ds(s);
    hipStream_t st = xstream(s);
    const hipError_t e = rccl ? hipMemcpyAsync(s->vx_host + slot * W, s->vx_dev + 1, 4 * (size_t)W, hipMemcpyDeviceToHost, st)
                              : hipMemcpyAsync(s->vx_host + slot * W + q.rank, s->vx_dev, 4, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess || hipEventRecord(s->ev_vx[slot], st) != hipSuccess) return rfail(R, s, WX_E_DEVICE, "copying the slabs' |vx| to the host");
    s->vx_have[slot] = true;
    s->vx_slot = prev;
    if (int rc = rpass(R, s, wx_slab_set_vx_bound(s, v))) return rc;
  }
  return WX_OK;
}

// One exchange of every slab of the ring: my left edge -> the left neighbour's right ghosts, my right edge -> the right neighbour's left
// ghosts; with particles also the droplet pool (status flips unless the exact mode already sent them, edge droplets in the same batch of
// transfers as the grid halos).
static int ring_exchange(Ring &R)
{
  const int n = (int)R.p.size();
  if (R.world < 2) return WX_OK;
  const bool local = R.transport == WX_TRANSPORT_LOCAL;
  const bool particles = R.p[0].s->pool_remote != nullptr;
  for (Party &q : R.p) q.s->xchg_inline = dry_runs_in_order(q.s); // (decided once per exchange: xstream() below is one stream throughout)
  if (particles)
    if (int rc = pool_stride_update(R)) return rc;
  if (local) // my send buffers are free once both neighbours have copied the previous exchange out of them
    if (int rc = local_wait(R, &wx_sim::ev_copied, true)) return rc;
  for (Party &q : R.p) { // (grid-only slabs: on the comm stream, behind the edge strips only)
    DeviceScope ds(q.s);
    if (int rc = rpass(R, q.s, wx_halo_pack_both(q.s, q.s->xsend[0], q.s->xsend[1]))) return rc; // (one launch: see HaloBufs)
  }
  if (int rc = ring_vx_roll(R)) return rc; // the period that starts behind this exchange: how many ghost columns per iteration?
  if (particles && !R.p[0].s->pool_exact)
    if (int rc = pool_events_round(R, 2)) return rc;
  if (particles)
    for (Party &q : R.p) { // ownership by position; the droplets near my edges become the neighbours' ghost copies
      wx_sim *s = q.s;
      DeviceScope ds(s);
      const int refresh = !s->pool_exact && (s->iter / 600) != ((s->iter - s->since_exchange) / 600); // app.js:5957-5966: every 600 iterations
      if (int rc = rpass(R, s, wx_pool_edges_pack(s, s->psend[0], s->psend[1], refresh))) return rc;
    }
  if (local) {
    if (int rc = local_record(R, &wx_sim::ev_packed)) return rc;
    if (int rc = local_wait(R, &wx_sim::ev_packed, true)) return rc;
    for (int i = 0; i < n; i++) {
      wx_sim *s = R.p[i].s, *L = R.p[(i + n - 1) % n].s, *Rt = R.p[(i + 1) % n].s;
      DeviceScope ds(s);
      const size_t mb = wx_halo_message_bytes(s); // (the base texture alone between slabs of the agreed water-free dry stencil)
      bool ok = hipMemcpyAsync(s->xrecv[0], L->xsend[1], mb, hipMemcpyDefault, xstream(s)) == hipSuccess && // left ghosts <- left neighbour's right edge
                hipMemcpyAsync(s->xrecv[1], Rt->xsend[0], mb, hipMemcpyDefault, xstream(s)) == hipSuccess; // right ghosts <- right neighbour's left edge
      if (ok && particles) {
        const size_t gb = wx_pool_edge_bytes(s);
        ok = hipMemcpyAsync(s->precv[0], L->psend[1], gb, hipMemcpyDefault, xstream(s)) == hipSuccess &&
             hipMemcpyAsync(s->precv[1], Rt->psend[0], gb, hipMemcpyDefault, xstream(s)) == hipSuccess;
      }
      if (!ok) return rfail(R, s, WX_E_DEVICE, "device-to-device halo copy of slab %d: %s", i, hipGetErrorString(hipGetLastError()));
    }
    if (int rc = local_record(R, &wx_sim::ev_copied)) return rc;
  } else {
    // (two ranks: both neighbours are the same peer -- messages between a pair match in order, so what I receive first is the peer's
    // first send, its LEFT edge, which belongs into my RIGHT ghosts)
    RcclApi *a = rccl_api();
    if (a->GroupStart() != ncclSuccess) return rfail(R, nullptr, WX_E_DEVICE, "ncclGroupStart");
    for (Party &q : R.p) {
      wx_sim *s = q.s;
      DeviceScope ds(s);
      const int left = (q.rank + R.world - 1) % R.world, right = (q.rank + 1) % R.world;
      hipStream_t st = xstream(s);
      const size_t mb = wx_halo_message_bytes(s); // (equal on every rank: the format changes only through collective calls)
      ncclResult_t r = a->Send(s->xsend[0], mb, ncclUint8, left, q.comm, st);
      if (r == ncclSuccess) r = a->Send(s->xsend[1], mb, ncclUint8, right, q.comm, st);
      if (r == ncclSuccess) r = a->Recv(s->xrecv[1], mb, ncclUint8, right, q.comm, st);
      if (r == ncclSuccess) r = a->Recv(s->xrecv[0], mb, ncclUint8, left, q.comm, st);
      if (particles) {
        const size_t gb = wx_pool_edge_bytes(s);
        if (r == ncclSuccess) r = a->Send(s->psend[0], gb, ncclUint8, left, q.comm, st);
        if (r == ncclSuccess) r = a->Send(s->psend[1], gb, ncclUint8, right, q.comm, st);
        if (r == ncclSuccess) r = a->Recv(s->precv[1], gb, ncclUint8, right, q.comm, st);
        if (r == ncclSuccess) r = a->Recv(s->precv[0], gb, ncclUint8, left, q.comm, st);
      }
      if (r != ncclSuccess) {
        a->GroupEnd();
        return rfail(R, s, WX_E_DEVICE, "ncclSend / ncclRecv of slab %d: %s", q.rank, a->GetErrorString(r));
      }
    }
    if (a->GroupEnd() != ncclSuccess) return rfail(R, nullptr, WX_E_DEVICE, "ncclGroupEnd");
  }
  for (Party &q : R.p) { // into the ghost columns (grid-only: records the event the next edge strips wait for)
    wx_sim *s = q.s;
    DeviceScope ds(s);
    if (int rc = rpass(R, s, wx_halo_unpack_both(s, s->xrecv[0], s->xrecv[1]))) return rc;
    if (particles) {
      if (int rc = rpass(R, s, wx_pool_edges_apply(s, s->precv[0]))) return rc;
      if (int rc = rpass(R, s, wx_pool_edges_apply(s, s->precv[1]))) return rc;
      if (int rc = rpass(R, s, wx_slab_period_begin(s))) return rc;
    }
    s->since_exchange = 0;
    s->exchanged = true;
  }
  return WX_OK;
}

// n iterations of every slab of the ring with the exchanges that fall into them
static int ring_step(Ring &R, int n_iter)
{
  wx_sim *s0 = R.p[0].s;
  if (R.world < 2) {
    for (Party &q : R.p) {
      DeviceScope ds(q.s);
      if (int rc = rpass(R, q.s, wx_step(q.s, n_iter))) return rc;
    }
    return WX_OK;
  }
  if (n_iter > 0)
    if (int rc = ring_vx_bootstrap(R)) return rc;
  const bool particles = s0->pool_remote != nullptr, exact = particles && s0->pool_exact;
  for (int done = 0; done < n_iter;) {
    const int ipe = iters_per_exchange(s0); // (the bound -- and with it the period -- is settled at every exchange)
    if (s0->since_exchange >= ipe) { // (a bound that rose inside a period, through wx_slab_set_vx_bound: exchange first)
      if (int rc = ring_exchange(R)) return rc;
      continue;
    }
    const int k = std::min(ipe - s0->since_exchange, n_iter - done);
    if (exact) { // one iteration at a time, each followed by the status flips / lightning requests of all slabs
      for (int it = 0; it < k; it++) {
        for (Party &q : R.p) {
          DeviceScope ds(q.s);
          if (int rc = rpass(R, q.s, wx_step_overlap(q.s, 1, done + it + 1 < n_iter ? WX_OVERLAP_MORE_TO_COME : 0u))) return rc;
          q.s->since_exchange += 1;
        }
        if (int rc = pool_events_round(R, 1)) return rc;
      }
    } else {
      for (Party &q : R.p) {
        wx_sim *s = q.s;
        DeviceScope ds(s);
        // the iteration before an exchange launches its edge strips first, the one after it its interior strips first (wx_step_overlap;
        // with particles only the latter: the exchange -- grid, feedback texture and droplet pool -- hides behind the next interior)
        const unsigned flags = ((s->since_exchange == 0 && s->exchanged) ? WX_OVERLAP_EDGES_LAST : 0u) | (s->since_exchange + k >= ipe ? WX_OVERLAP_EDGES_FIRST : 0u) |
                               (done + k < n_iter ? WX_OVERLAP_MORE_TO_COME : 0u); // (only the call's last piece stores the display-side fields)
        if (int rc = rpass(R, s, wx_step_overlap(s, k, flags))) return rc;
        s->since_exchange += k;
      }
    }
    done += k;
    if (s0->since_exchange >= ipe)
      if (int rc = ring_exchange(R)) return rc;
  }
  return WX_OK;
}

int wx_comm_unique_id(void *id128)
{
  if (!id128) return WX_E_INVALID;
  RcclApi *a = rccl_api();
  if (!a) return fail(nullptr, WX_E_DEVICE, "%s", rccl_missing());
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == WX_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (a->GetUniqueId(&id) != ncclSuccess) return fail(nullptr, WX_E_DEVICE, "ncclGetUniqueId failed");
  memcpy(id128, &id, sizeof(id));
  return WX_OK;
}

int wx_comm_init(wx_sim *s, const void *id128, int rank, int world)
{
  if (!s || !id128 || world < 1 || rank < 0 || rank >= world) return WX_E_INVALID;
  RcclApi *a = rccl_api();
  if (!a) return fail(s, WX_E_DEVICE, "%s", rccl_missing());
  DeviceScope ds(s);
  // (a lone slab needs no ghost columns -- the kernels wrap in x themselves -- and cannot have any: with X_owned + 2 * halo > X_global a
  // global column would lie in the local array twice)
  if (world == 1 && s->halo != 0) return fail(s, WX_E_INVALID, "wx_comm_init: a job of one rank takes a handle without ghost columns (halo 0)");
  if (world > 1)
    if (int rc = transport_prepare(s, world)) return rc;
  if (s->comm) {
    a->CommDestroy((ncclComm_t)s->comm);
    s->comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  NCCLCHK(s, a->CommInitRank(&c, world, id, rank));
  s->comm = c;
  s->comm_rank = rank;
  s->comm_world = world;
  s->rank = rank;
  return WX_OK;
}

static Ring ring_of(wx_sim *s)
{
  Ring R;
  R.p.push_back(Party{s, (ncclComm_t)s->comm, s->comm_rank});
  R.world = s->comm_world;
  R.transport = WX_TRANSPORT_RCCL;
  return R;
}

// ring exchange of one rank (all of it enqueued; nothing waits on the host)
int wx_exchange(wx_sim *s)
{
  if (!s) return WX_E_INVALID;
  if (!s->comm) return fail(s, WX_E_STATE, "wx_exchange before wx_comm_init");
  if (s->comm_world > 1)
    if (int rc = transport_prepare(s, s->comm_world)) return rc;
  Ring R = ring_of(s);
  return ring_exchange(R);
}

int wx_slab_step(wx_sim *s, int n_iter)
{
  if (!s || n_iter < 0) return WX_E_INVALID;
  if (!s->comm) return fail(s, WX_E_STATE, "wx_slab_step before wx_comm_init");
  if (s->comm_world > 1)
    if (int rc = transport_prepare(s, s->comm_world)) return rc;
  Ring R = ring_of(s);
  return ring_step(R, n_iter);
}

// ---- N slabs in one process ----
struct wx_group {
  std::vector<wx_sim *> slab;
  std::vector<ncclComm_t> comms; // transport RCCL
  int transport = WX_TRANSPORT_LOCAL;
  std::string err;
};

static int gfail(wx_group *g, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (g) g->err = buf;
  else g_create_error = buf;
  return code;
}
static int gpass(wx_group *g, wx_sim *s, int rc)
{
  if (rc != WX_OK) g->err = s->err;
  return rc;
}
static Ring ring_of(wx_group *g)
{
  Ring R;
  for (size_t i = 0; i < g->slab.size(); i++) R.p.push_back(Party{g->slab[i], g->comms.empty() ? nullptr : g->comms[i], (int)i});
  R.world = (int)g->slab.size();
  R.transport = g->transport;
  R.err = &g->err;
  return R;
}

const char *wx_group_last_error(const wx_group *g) { return g ? g->err.c_str() : g_create_error.c_str(); }
int wx_group_count(const wx_group *g) { return g ? (int)g->slab.size() : 0; }
int wx_group_transport(const wx_group *g) { return g ? g->transport : 0; }
wx_sim *wx_group_slab(wx_group *g, int i) { return g && i >= 0 && i < (int)g->slab.size() ? g->slab[i] : nullptr; }

void wx_group_destroy(wx_group *g)
{
  if (!g) return;
  for (wx_sim *s : g->slab) {
    if (!s) continue;
    DeviceScope ds(s);
    if (s->stream) hipStreamSynchronize(s->stream);
    if (s->comm_stream) hipStreamSynchronize(s->comm_stream);
  }
  if (RcclApi *a = rccl_api())
    for (ncclComm_t c : g->comms)
      if (c) a->CommDestroy(c);
  for (wx_sim *s : g->slab) {
    if (!s) continue;
    DeviceScope ds(s);
    wx_destroy(s);
  }
  delete g;
}

int wx_group_create(int n_slabs, const int *devices, int X_global, int Y, int halo, int n_droplets, int transport, wx_group **out)
{
  if (!out) return WX_E_INVALID;
  *out = nullptr;
  if (n_slabs < 1 || X_global % n_slabs) return gfail(nullptr, WX_E_INVALID, "wx_group_create: X_global = %d is not divisible by %d slabs", X_global, n_slabs);
  if (n_droplets < 0) return gfail(nullptr, WX_E_INVALID, "wx_group_create: n_droplets < 0");
  if (n_slabs > 1 && halo < WX_SLAB_CONE) return gfail(nullptr, WX_E_INVALID, "wx_group_create: halo >= %d", WX_SLAB_CONE);
  if (n_slabs > 1 && n_droplets > 0 && WX_SLAB_PERIOD_PARTICLES(halo) < 1) return gfail(nullptr, WX_E_INVALID, "wx_group_create: slabs with particles need halo >= 12");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return gfail(nullptr, WX_E_DEVICE, "no HIP device available: libwxsim has no CPU fallback");
  std::vector<int> dev(n_slabs);
  bool distinct = true;
  for (int i = 0; i < n_slabs; i++) {
    dev[i] = devices ? devices[i] : i % ndev;
    if (dev[i] < 0 || dev[i] >= ndev) return gfail(nullptr, WX_E_INVALID, "wx_group_create: device %d of %d", dev[i], ndev);
    for (int j = 0; j < i; j++) distinct = distinct && dev[j] != dev[i];
  }
  if (transport == WX_TRANSPORT_AUTO) transport = (distinct && n_slabs > 1 && rccl_api()) ? WX_TRANSPORT_RCCL : WX_TRANSPORT_LOCAL;
  if (transport == WX_TRANSPORT_RCCL && !distinct) return gfail(nullptr, WX_E_INVALID, "wx_group_create: RCCL needs one device per slab (it refuses two ranks on one device); use WX_TRANSPORT_LOCAL");
  if (transport == WX_TRANSPORT_RCCL && !rccl_api()) return gfail(nullptr, WX_E_DEVICE, "%s", rccl_missing());
  if (transport != WX_TRANSPORT_RCCL && transport != WX_TRANSPORT_LOCAL) return gfail(nullptr, WX_E_INVALID, "wx_group_create: transport %d", transport);
  int prev = 0;
  (void)hipGetDevice(&prev);
  wx_group *g = new wx_group();
  g->transport = transport;
  g->slab.assign(n_slabs, nullptr);
  const int xo = X_global / n_slabs;
  int rc = WX_OK;
  for (int i = 0; i < n_slabs && rc == WX_OK; i++) {
    if (hipSetDevice(dev[i]) != hipSuccess) {
      rc = gfail(nullptr, WX_E_DEVICE, "hipSetDevice(%d)", dev[i]);
      break;
    }
    wx_sim *s = nullptr;
    rc = wx_create_slab(X_global, Y, i * xo, xo, n_slabs > 1 ? halo : 0, n_droplets, &s);
    if (rc != WX_OK) break;
    g->slab[i] = s;
    s->rank = i;
    // streams of the slab's own: the slabs of a group run concurrently (also when several share a device)
    if (hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking) != hipSuccess) rc = gfail(nullptr, WX_E_DEVICE, "hipStreamCreate");
    else s->stream = s->own_stream;
    if (rc == WX_OK && n_slabs > 1) {
      rc = transport_prepare(s, n_slabs);
      for (hipEvent_t *e : {&s->ev_packed, &s->ev_copied, &s->ev_evpacked, &s->ev_evcopied})
        if (rc == WX_OK && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = gfail(nullptr, WX_E_DEVICE, "hipEventCreate");
      if (rc != WX_OK && g_create_error.empty()) g_create_error = s->err;
    }
  }
  if (rc == WX_OK && transport == WX_TRANSPORT_RCCL && n_slabs > 1) {
    g->comms.assign(n_slabs, nullptr);
    if (rccl_api()->CommInitAll(g->comms.data(), n_slabs, dev.data()) != ncclSuccess) rc = gfail(nullptr, WX_E_DEVICE, "ncclCommInitAll over %d devices failed", n_slabs);
  }
  if (rc == WX_OK && transport == WX_TRANSPORT_LOCAL && n_slabs > 1) // copies between slabs on different devices go directly where the devices allow it
    for (int i = 0; i < n_slabs; i++)
      for (int nb = 0; nb < n_slabs; nb++)
        if (dev[nb] != dev[i] && hipSetDevice(dev[i]) == hipSuccess && hipDeviceEnablePeerAccess(dev[nb], 0) != hipSuccess) (void)hipGetLastError();
  (void)hipSetDevice(prev);
  if (rc != WX_OK) {
    const std::string keep = g_create_error;
    wx_group_destroy(g);
    g_create_error = keep;
    return rc;
  }
  *out = g;
  return WX_OK;
}

// after the slabs were uploaded: what has to be agreed between them (the water-free dry kernel is only valid if NO slab carries water)
int wx_group_agree(wx_group *g)
{
  if (!g) return WX_E_INVALID;
  bool all = true;
  for (wx_sim *s : g->slab) all = all && wx_water_free(s);
  for (wx_sim *s : g->slab)
    if (int rc = gpass(g, s, wx_slab_assert_water_free(s, all ? 1 : 0))) return rc;
  return WX_OK;
}

// an option on every slab (WX_OPT_SPLAT_ORDER, WX_OPT_POOL_EXACT ...)
int wx_group_set_option(wx_group *g, int option, int value)
{
  if (!g) return WX_E_INVALID;
  for (wx_sim *s : g->slab) {
    DeviceScope ds(s);
    if (int rc = gpass(g, s, wx_set_option(s, option, value))) return rc;
  }
  return WX_OK;
}

// (stream roles follow the particle mode, which wx_group_set_option may have changed)
static int group_prepare(wx_group *g)
{
  if (g->slab.size() < 2) return WX_OK;
  for (wx_sim *s : g->slab) {
    DeviceScope ds(s);
    if (int rc = gpass(g, s, transport_prepare(s, (int)g->slab.size()))) return rc;
  }
  return WX_OK;
}

int wx_group_step(wx_group *g, int n_iter)
{
  if (!g || n_iter < 0) return WX_E_INVALID;
  if (int rc = group_prepare(g)) return rc;
  Ring R = ring_of(g);
  return ring_step(R, n_iter);
}

// brings the droplet pool to the state right after an exchange (every active droplet owned by exactly one slab): what a host
// calls before it reads the pool of a group inside a period
int wx_group_exchange(wx_group *g)
{
  if (!g) return WX_E_INVALID;
  if (int rc = group_prepare(g)) return rc;
  Ring R = ring_of(g);
  return ring_exchange(R);
}

int wx_group_sync(wx_group *g)
{
  if (!g) return WX_E_INVALID;
  for (wx_sim *s : g->slab) {
    DeviceScope ds(s);
    if (int rc = gpass(g, s, wx_sync(s))) return rc;
  }
  return WX_OK;
}
