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
// library loads and runs single-GPU without it. Slabs with particles keep the host-driven exchange of slab.py for now.
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

static int iters_per_exchange(const wx_sim *s) { return std::max(1, s->halo / WX_SLAB_CONE); }

// send / recv buffers, a comm stream of the library's own if the host has not set one
static int transport_prepare(wx_sim *s)
{
  if (s->halo == 0) return fail(s, WX_E_STATE, "the handle has no ghost columns (wx_create_slab with halo > 0)");
  if (s->pool_remote) return fail(s, WX_E_STATE, "slabs with particles exchange through the host (slab.py): the droplet-pool protocol is not in wx_exchange yet");
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
  if (!s->comm_stream) {
    if (!s->own_comm_stream) HIPCHK(s, hipStreamCreateWithFlags(&s->own_comm_stream, hipStreamNonBlocking));
    if (int rc = wx_set_comm_stream(s, s->own_comm_stream)) return rc;
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
    s->xsend[i] = s->xrecv[i] = nullptr;
  }
  s->xbytes = 0;
  if (s->ev_packed) hipEventDestroy(s->ev_packed);
  if (s->ev_copied) hipEventDestroy(s->ev_copied);
  s->ev_packed = s->ev_copied = nullptr;
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
    if (int rc = transport_prepare(s)) return rc;
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

// ring exchange of one rank: my left edge -> the left neighbour's right ghosts, my right edge -> the right neighbour's left ghosts
int wx_exchange(wx_sim *s)
{
  if (!s) return WX_E_INVALID;
  if (!s->comm) return fail(s, WX_E_STATE, "wx_exchange before wx_comm_init");
  DeviceScope ds(s);
  if (s->comm_world > 1) {
    if (int rc = transport_prepare(s)) return rc;
    if (int rc = wx_halo_pack(s, 0, s->xsend[0])) return rc;
    if (int rc = wx_halo_pack(s, 1, s->xsend[1])) return rc;
    RcclApi *a = rccl_api();
    const int left = (s->comm_rank + s->comm_world - 1) % s->comm_world, right = (s->comm_rank + 1) % s->comm_world;
    ncclComm_t c = (ncclComm_t)s->comm;
    // (two ranks: both neighbours are the same peer -- messages between a pair match in order, so what I receive first is the peer's
    // first send, its LEFT edge, which belongs into my RIGHT ghosts)
    NCCLCHK(s, a->GroupStart());
    NCCLCHK(s, a->Send(s->xsend[0], s->xbytes, ncclUint8, left, c, s->comm_stream));
    NCCLCHK(s, a->Send(s->xsend[1], s->xbytes, ncclUint8, right, c, s->comm_stream));
    NCCLCHK(s, a->Recv(s->xrecv[1], s->xbytes, ncclUint8, right, c, s->comm_stream));
    NCCLCHK(s, a->Recv(s->xrecv[0], s->xbytes, ncclUint8, left, c, s->comm_stream));
    NCCLCHK(s, a->GroupEnd());
    if (int rc = wx_halo_unpack(s, 0, s->xrecv[0])) return rc;
    if (int rc = wx_halo_unpack(s, 1, s->xrecv[1])) return rc;
  }
  s->since_exchange = 0;
  s->exchanged = true;
  return WX_OK;
}

// the iterations of one call up to the next exchange, with the launch order that overlaps it (see wx_step_overlap)
static int slab_advance(wx_sim *s, int k, bool lone)
{
  const int ipe = iters_per_exchange(s);
  unsigned flags = 0;
  if (!lone) flags = ((s->since_exchange == 0 && s->exchanged) ? WX_OVERLAP_EDGES_LAST : 0u) | (s->since_exchange + k >= ipe ? WX_OVERLAP_EDGES_FIRST : 0u);
  if (int rc = wx_step_overlap(s, k, flags)) return rc;
  s->since_exchange += k;
  return WX_OK;
}

int wx_slab_step(wx_sim *s, int n_iter)
{
  if (!s || n_iter < 0) return WX_E_INVALID;
  if (!s->comm) return fail(s, WX_E_STATE, "wx_slab_step before wx_comm_init");
  DeviceScope ds(s);
  if (s->comm_world == 1) return wx_step(s, n_iter);
  const int ipe = iters_per_exchange(s);
  for (int done = 0; done < n_iter;) {
    const int k = std::min(ipe - s->since_exchange, n_iter - done);
    if (int rc = slab_advance(s, k, false)) return rc;
    done += k;
    if (s->since_exchange >= ipe)
      if (int rc = wx_exchange(s)) return rc;
  }
  return WX_OK;
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
  if (n_droplets != 0) return gfail(nullptr, WX_E_INVALID, "wx_group_create: slabs with particles exchange through the host (slab.py) for now");
  if (n_slabs > 1 && halo < WX_SLAB_CONE) return gfail(nullptr, WX_E_INVALID, "wx_group_create: halo >= %d", WX_SLAB_CONE);
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
    rc = wx_create_slab(X_global, Y, i * xo, xo, n_slabs > 1 ? halo : 0, 0, &s);
    if (rc != WX_OK) break;
    g->slab[i] = s;
    s->rank = i;
    // streams of the slab's own: the slabs of a group run concurrently (also when several share a device)
    if (hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking) != hipSuccess) rc = gfail(nullptr, WX_E_DEVICE, "hipStreamCreate");
    else s->stream = s->own_stream;
    if (rc == WX_OK && n_slabs > 1) {
      rc = transport_prepare(s);
      if (rc == WX_OK && (hipEventCreateWithFlags(&s->ev_packed, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->ev_copied, hipEventDisableTiming) != hipSuccess))
        rc = gfail(nullptr, WX_E_DEVICE, "hipEventCreate");
      if (rc != WX_OK && g_create_error.empty()) g_create_error = s->err;
    }
  }
  if (rc == WX_OK && transport == WX_TRANSPORT_RCCL && n_slabs > 1) {
    g->comms.assign(n_slabs, nullptr);
    if (rccl_api()->CommInitAll(g->comms.data(), n_slabs, dev.data()) != ncclSuccess) rc = gfail(nullptr, WX_E_DEVICE, "ncclCommInitAll over %d devices failed", n_slabs);
  }
  if (rc == WX_OK && transport == WX_TRANSPORT_LOCAL && n_slabs > 1) // copies between slabs on different devices go directly where the devices allow it
    for (int i = 0; i < n_slabs; i++)
      for (int nb : {(i + 1) % n_slabs, (i + n_slabs - 1) % n_slabs})
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

static int group_exchange(wx_group *g)
{
  const int n = (int)g->slab.size();
  if (n == 1) return WX_OK;
  // 1. every slab packs its two edges on its comm stream (behind its edge strips)
  for (int i = 0; i < n; i++) {
    wx_sim *s = g->slab[i];
    DeviceScope ds(s);
    if (g->transport == WX_TRANSPORT_LOCAL) { // my send buffers are free once both neighbours have copied the previous exchange out of them
      for (int nb : {(i + n - 1) % n, (i + 1) % n})
        if (hipStreamWaitEvent(s->comm_stream, g->slab[nb]->ev_copied, 0) != hipSuccess) return gfail(g, WX_E_DEVICE, "hipStreamWaitEvent");
    }
    if (int rc = gpass(g, s, wx_halo_pack(s, 0, s->xsend[0]))) return rc;
    if (int rc = gpass(g, s, wx_halo_pack(s, 1, s->xsend[1]))) return rc;
    if (g->transport == WX_TRANSPORT_LOCAL && hipEventRecord(s->ev_packed, s->comm_stream) != hipSuccess) return gfail(g, WX_E_DEVICE, "hipEventRecord");
  }
  // 2. the transfers
  if (g->transport == WX_TRANSPORT_RCCL) {
    RcclApi *a = rccl_api();
    if (a->GroupStart() != ncclSuccess) return gfail(g, WX_E_DEVICE, "ncclGroupStart");
    for (int i = 0; i < n; i++) {
      wx_sim *s = g->slab[i];
      DeviceScope ds(s);
      const int left = (i + n - 1) % n, right = (i + 1) % n;
      ncclResult_t r = a->Send(s->xsend[0], s->xbytes, ncclUint8, left, g->comms[i], s->comm_stream);
      if (r == ncclSuccess) r = a->Send(s->xsend[1], s->xbytes, ncclUint8, right, g->comms[i], s->comm_stream);
      if (r == ncclSuccess) r = a->Recv(s->xrecv[1], s->xbytes, ncclUint8, right, g->comms[i], s->comm_stream);
      if (r == ncclSuccess) r = a->Recv(s->xrecv[0], s->xbytes, ncclUint8, left, g->comms[i], s->comm_stream);
      if (r != ncclSuccess) {
        a->GroupEnd();
        return gfail(g, WX_E_DEVICE, "ncclSend / ncclRecv of slab %d: %s", i, a->GetErrorString(r));
      }
    }
    if (a->GroupEnd() != ncclSuccess) return gfail(g, WX_E_DEVICE, "ncclGroupEnd");
  } else {
    for (int i = 0; i < n; i++) {
      wx_sim *s = g->slab[i];
      DeviceScope ds(s);
      wx_sim *L = g->slab[(i + n - 1) % n], *R = g->slab[(i + 1) % n];
      if (hipStreamWaitEvent(s->comm_stream, L->ev_packed, 0) != hipSuccess || hipStreamWaitEvent(s->comm_stream, R->ev_packed, 0) != hipSuccess ||
          hipMemcpyAsync(s->xrecv[0], L->xsend[1], s->xbytes, hipMemcpyDefault, s->comm_stream) != hipSuccess || // left ghosts <- left neighbour's right edge
          hipMemcpyAsync(s->xrecv[1], R->xsend[0], s->xbytes, hipMemcpyDefault, s->comm_stream) != hipSuccess || // right ghosts <- right neighbour's left edge
          hipEventRecord(s->ev_copied, s->comm_stream) != hipSuccess)
        return gfail(g, WX_E_DEVICE, "device-to-device halo copy of slab %d: %s", i, hipGetErrorString(hipGetLastError()));
    }
  }
  // 3. unpack into the ghost columns (records the event the next edge strips wait for)
  for (int i = 0; i < n; i++) {
    wx_sim *s = g->slab[i];
    DeviceScope ds(s);
    if (int rc = gpass(g, s, wx_halo_unpack(s, 0, s->xrecv[0]))) return rc;
    if (int rc = gpass(g, s, wx_halo_unpack(s, 1, s->xrecv[1]))) return rc;
    s->since_exchange = 0;
    s->exchanged = true;
  }
  return WX_OK;
}

int wx_group_step(wx_group *g, int n_iter)
{
  if (!g || n_iter < 0) return WX_E_INVALID;
  const int n = (int)g->slab.size();
  wx_sim *s0 = g->slab[0];
  const int ipe = iters_per_exchange(s0);
  for (int done = 0; done < n_iter;) {
    const int k = n == 1 ? n_iter - done : std::min(ipe - s0->since_exchange, n_iter - done);
    for (wx_sim *s : g->slab) {
      DeviceScope ds(s);
      if (int rc = gpass(g, s, n == 1 ? wx_step(s, k) : slab_advance(s, k, false))) return rc;
    }
    done += k;
    if (n > 1 && s0->since_exchange >= ipe)
      if (int rc = group_exchange(g)) return rc;
  }
  return WX_OK;
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
