"""ctypes binding of libwxsim.so (the C ABI of include/wxsim.h).

This is the ONLY compute path of the package: if the HIP library is missing or no GPU is present the
calls fail loudly (RuntimeError) -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from .params import WxParams

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# WXSIM_LIB: another build of the same sources -- the tolerance build FAST_LIB_PATH (`make -C csrc fast`, SURVEY Appendix A's opt-in fast
# arithmetic; lib().wx_arith() says which one is loaded) or a tuning variant. The default is the parity build.
FAST_LIB_PATH = os.path.join(CSRC, "libwxsim_fast.so")
LIB_PATH = os.environ.get("WXSIM_LIB") or os.path.join(CSRC, "libwxsim.so")

FIELD_IDS = {
    "BASE_CUR": 0, "BASE_DISP": 1, "WATER_0": 2, "WATER_CUR": 3, "WALL_CUR": 4, "WALL_DISP": 5,
    "LIGHT_0": 6, "LIGHT_1": 7, "CURL": 8, "VORT": 9, "PRECIP_FB": 10, "PRECIP_DEP": 11, "LIGHTNING": 12, "EMITTED": 13,
}
FIELD_CHANNELS = {"CURL": 1, "VORT": 2, "PRECIP_DEP": 2}
DTYPE_F32, DTYPE_I8, DTYPE_I32, DTYPE_F16 = 0, 1, 2, 3

# every symbol include/wxsim.h declares
EXPORTS = [
    "wx_create", "wx_create_slab", "wx_destroy", "wx_last_error", "wx_abi_version", "wx_upload", "wx_set_params",
    "wx_step", "wx_sync", "wx_get_iter", "wx_set_iter", "wx_read_rect", "wx_read_particles", "wx_set_stream",
    "wx_device_ptr", "wx_local_width", "wx_halo_bytes", "wx_halo_message_bytes", "wx_halo_pack", "wx_halo_unpack", "wx_halo_pack_both", "wx_halo_unpack_both", "wx_profile",
    "wx_profile_read", "wx_kernel_count", "wx_kernel_name", "wx_slab_set_rank", "wx_slab_period_begin", "wx_pool_event_bytes",
    "wx_pool_edge_bytes", "wx_pool_events_pack", "wx_pool_events_apply", "wx_pool_edges_pack", "wx_pool_edges_apply", "wx_pool_flags", "wx_lightning_get", "wx_lightning_set", "wx_setup_columns", "wx_setup_terrain", "wx_init_droplets", "wx_fastest_velocity",
    "wx_stream_bytes", "wx_host_alloc", "wx_host_free", "wx_stream_frame", "wx_stream_wait", "wx_set_comm_stream", "wx_step_overlap",
    "wx_set_option", "wx_water_free", "wx_slab_assert_water_free", "wx_tune_placement",
    "wx_comm_unique_id", "wx_comm_init", "wx_exchange", "wx_slab_step", "wx_group_create", "wx_group_destroy", "wx_group_last_error",
    "wx_group_count", "wx_group_transport", "wx_group_slab", "wx_group_agree", "wx_group_step", "wx_group_sync", "wx_group_set_option",
    "wx_group_exchange", "wx_slab_vx_take", "wx_slab_set_vx_bound", "wx_slab_cone", "wx_slab_period", "wx_pair_stats", "wx_placement_info", "wx_arith",
]


def build(force: bool = False, fast: bool = False) -> str:
    """Compile libwxsim.so (``fast``: the tolerance build libwxsim_fast.so) for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "wxsim.h"))
    fast = fast or os.path.abspath(LIB_PATH) == os.path.abspath(FAST_LIB_PATH)
    target = FAST_LIB_PATH if fast else os.path.join(CSRC, "libwxsim.so")
    if os.path.abspath(LIB_PATH) not in (os.path.abspath(target),) and not fast:
        return LIB_PATH  # (a tuning variant named by WXSIM_LIB: built by its own make target)
    stale = (not os.path.exists(target)) or any(os.path.getmtime(f) > os.path.getmtime(target) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", os.path.basename(target)])
    return target


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the simulation step.")
    try:
        # PyTorch ships its own libamdhip64.so.7: import it FIRST so that the process holds a single HIP runtime
        # (libwxsim.so's NEEDED libamdhip64.so.7 then resolves to the already loaded one). Loading ours first and
        # torch second gives two runtimes, and the second one finds "no ROCm-capable device".
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.wx_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.wx_create_slab.argtypes = [i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.wx_destroy.argtypes = [vp]
    L.wx_destroy.restype = None
    L.wx_last_error.argtypes = [vp]
    L.wx_last_error.restype = C.c_char_p
    L.wx_abi_version.restype = i32
    L.wx_arith.restype = i32
    L.wx_upload.argtypes = [vp, vp, vp, vp, vp]
    L.wx_set_params.argtypes = [vp, C.POINTER(WxParams), vp, vp, vp, vp]
    L.wx_step.argtypes = [vp, i32]
    L.wx_step_overlap.argtypes = [vp, i32, C.c_uint]
    L.wx_set_comm_stream.argtypes = [vp, vp]
    L.wx_sync.argtypes = [vp]
    L.wx_set_option.argtypes = [vp, i32, i32]
    L.wx_water_free.argtypes = [vp]
    L.wx_tune_placement.argtypes = [vp, i32, i32, vp, vp]
    L.wx_slab_assert_water_free.argtypes = [vp, i32]
    L.wx_get_iter.argtypes = [vp]
    L.wx_get_iter.restype = i64
    L.wx_set_iter.argtypes = [vp, i64]
    L.wx_read_rect.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32]
    L.wx_read_particles.argtypes = [vp, i32, i32, vp]
    L.wx_set_stream.argtypes = [vp, vp]
    L.wx_device_ptr.argtypes = [vp, i32]
    L.wx_device_ptr.restype = vp
    L.wx_local_width.argtypes = [vp]
    L.wx_halo_bytes.argtypes = [vp]
    L.wx_halo_bytes.restype = C.c_size_t
    L.wx_halo_message_bytes.argtypes = [vp]
    L.wx_halo_message_bytes.restype = C.c_size_t
    L.wx_halo_pack.argtypes = [vp, i32, vp]
    L.wx_halo_unpack.argtypes = [vp, i32, vp]
    L.wx_halo_pack_both.argtypes = [vp, vp, vp]
    L.wx_halo_unpack_both.argtypes = [vp, vp, vp]
    L.wx_profile.argtypes = [vp, i32]
    L.wx_profile_read.argtypes = [vp, i32, vp, vp]
    L.wx_kernel_count.restype = i32
    L.wx_kernel_name.argtypes = [i32]
    L.wx_kernel_name.restype = C.c_char_p
    L.wx_slab_set_rank.argtypes = [vp, i32]
    L.wx_slab_period_begin.argtypes = [vp]
    L.wx_pool_event_bytes.argtypes = [vp]
    L.wx_pool_event_bytes.restype = C.c_size_t
    L.wx_pool_edge_bytes.argtypes = [vp]
    L.wx_pool_edge_bytes.restype = C.c_size_t
    L.wx_pool_events_pack.argtypes = [vp, vp]
    L.wx_pool_events_apply.argtypes = [vp, vp, i32, C.c_size_t]
    L.wx_pool_edges_pack.argtypes = [vp, vp, vp, i32]
    L.wx_pool_edges_apply.argtypes = [vp, vp]
    L.wx_pool_flags.argtypes = [vp, vp]
    L.wx_lightning_get.argtypes = [vp, vp]
    L.wx_lightning_set.argtypes = [vp, vp]
    L.wx_setup_columns.argtypes = [vp] + [vp] * 8
    L.wx_init_droplets.argtypes = [vp, C.c_uint32]
    L.wx_fastest_velocity.argtypes = [vp, C.POINTER(C.c_float)]
    L.wx_placement_info.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.wx_pair_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.wx_slab_vx_take.argtypes = [vp, C.POINTER(C.c_float)]
    L.wx_slab_set_vx_bound.argtypes = [vp, C.c_float]
    L.wx_slab_cone.argtypes = [vp]
    L.wx_slab_period.argtypes = [vp]
    L.wx_setup_terrain.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_double] + [vp] * 4
    L.wx_stream_bytes.argtypes = [i32, i32]
    L.wx_stream_bytes.restype = C.c_size_t
    L.wx_host_alloc.argtypes = [C.c_size_t]
    L.wx_host_alloc.restype = vp
    L.wx_host_free.argtypes = [vp]
    L.wx_host_free.restype = None
    L.wx_stream_frame.argtypes = [vp, i32, i32, i32, i32, vp]
    L.wx_stream_wait.argtypes = [vp]
    L.wx_comm_unique_id.argtypes = [vp]
    L.wx_comm_init.argtypes = [vp, vp, i32, i32]
    L.wx_exchange.argtypes = [vp]
    L.wx_slab_step.argtypes = [vp, i32]
    L.wx_group_create.argtypes = [i32, vp, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.wx_group_destroy.argtypes = [vp]
    L.wx_group_destroy.restype = None
    L.wx_group_last_error.argtypes = [vp]
    L.wx_group_last_error.restype = C.c_char_p
    L.wx_group_count.argtypes = [vp]
    L.wx_group_transport.argtypes = [vp]
    L.wx_group_slab.argtypes = [vp, i32]
    L.wx_group_slab.restype = vp
    L.wx_group_agree.argtypes = [vp]
    L.wx_group_step.argtypes = [vp, i32]
    L.wx_group_sync.argtypes = [vp]
    L.wx_group_set_option.argtypes = [vp, i32, i32]
    L.wx_group_exchange.argtypes = [vp]
    _lib = L
    return L


def set_default_option(option: int, value: int):
    """wx_set_option(NULL, ...): the default of handles created afterwards (kernel set, dry kernel, row bands, exact-path capacity)."""
    rc = lib().wx_set_option(None, int(option), int(value))
    if rc != 0:
        raise ValueError(f"wx_set_option(NULL, {option}, {value}) -> {rc}")


# The C library reads no environment variable. The test-suite and the experiment scripts select the kernel set etc. per process
# through these variables, which THIS wrapper turns into wx_set_option defaults before it creates a handle.
_ENV_OPTIONS = (("WX_FUSED", 3, 1), ("WX_DRY_MARCH", 4, 1), ("WX_WET_BANDS", 5, 1), ("WX_WET_FIX_CAP", 6, 0))


def _apply_env_defaults():
    for name, opt, dflt in _ENV_OPTIONS:
        v = os.environ.get(name)
        set_default_option(opt, dflt if v is None else (min(int(v), 1) if opt in (3, 4) else int(v)))


class WxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libwxsim error {code}: {msg}")
        self.code = code


class Handle:
    """Thin RAII wrapper of a ``wx_sim*``."""

    def __init__(self, X: int, Y: int, n_droplets: int = 0, *, X_global: Optional[int] = None, x0: int = 0, halo: int = 0):
        L = lib()
        h = C.c_void_p()
        _apply_env_defaults()
        if X_global is None:
            rc = L.wx_create(X, Y, n_droplets, C.byref(h))
        else:
            rc = L.wx_create_slab(X_global, Y, x0, X, halo, n_droplets, C.byref(h))
        if rc != 0:
            raise WxError(rc, (L.wx_last_error(None) or b"").decode())
        self._h = h
        self.X_owned, self.Y, self.n_droplets, self.halo = X, Y, n_droplets, halo
        self.X = L.wx_local_width(h)
        self.generation = 0  # bumped whenever device pointers obtained earlier become invalid (tune_placement)
        self._stepped = False

    @classmethod
    def _borrowed(cls, ptr, X_owned: int, Y: int, halo: int, owner, n_droplets: int = 0) -> "Handle":
        """A slab of a Group: the group destroys it (``owner`` is kept alive as long as this wrapper is)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(ptr)
        self._owner = owner
        self.X_owned, self.Y, self.n_droplets, self.halo = X_owned, Y, n_droplets, halo
        self.X = lib().wx_local_width(self._h)
        self.generation = 0
        self._stepped = False
        return self

    def _chk(self, rc: int):
        if rc != 0:
            raise WxError(rc, (lib().wx_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_owner", None) is not None:  # a group's slab
            self._h = None
            return
        if getattr(self, "_h", None):
            lib().wx_destroy(self._h)  # (waits for a streamed frame in flight)
            self._h = None
            if getattr(self, "_pin", None) is not None:
                lib().wx_host_free(self._pin[0])
                self._pin = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ----
    def upload(self, base, water, wall, drops=None):
        n = self.X * self.Y * 4
        base = np.ascontiguousarray(base, np.float32).reshape(-1)
        water = np.ascontiguousarray(water, np.float32).reshape(-1)
        wall = np.ascontiguousarray(wall, np.int8).reshape(-1)
        if not (base.size == water.size == wall.size == n):
            raise ValueError(f"grid arrays must hold {self.Y}x{self.X}x4 values")
        d = None
        if drops is not None and self.n_droplets > 0:
            drops = np.ascontiguousarray(drops, np.float32).reshape(-1)
            if drops.size != self.n_droplets * 5:
                raise ValueError("drops must hold n_droplets x 5 values")
            d = drops.ctypes.data
        self._chk(lib().wx_upload(self._h, base.ctypes.data, water.ctypes.data, wall.ctypes.data, d))

    def set_params(self, p: WxParams, initial_T=None, snd_T=None, snd_W=None, snd_Vel=None):
        arrs = []
        for a in (initial_T, snd_T, snd_W, snd_Vel):
            if a is None:
                arrs.append(None)
            else:
                a = np.ascontiguousarray(a, np.float32)
                if a.size < self.Y + 1:
                    raise ValueError("profile arrays need Y+1 entries")
                arrs.append(a)
        ptr = [None if a is None else a.ctypes.data for a in arrs]
        self._chk(lib().wx_set_params(self._h, C.byref(p), *ptr))

    OVERLAP_EDGES_FIRST, OVERLAP_EDGES_LAST = 1, 2

    def step(self, n: int = 1, overlap: int = 0):
        """n iterations; ``overlap`` (slab handles with a comm stream): OVERLAP_EDGES_FIRST -- edge strips of the last iteration
        first, so that the halo can be packed and sent while the interior computes; OVERLAP_EDGES_LAST -- interior strips of the
        first iteration first, the edge strips once the ghost columns have arrived (wx_step_overlap)."""
        if not self._stepped and n > 0:  # (the first step of a big whole-domain handle may move the planes: WX_OPT_PLACEMENT_SEARCH)
            self._stepped = True
            self.generation += 1
        if overlap:
            self._chk(lib().wx_step_overlap(self._h, int(n), int(overlap)))
        else:
            self._chk(lib().wx_step(self._h, int(n)))

    def sync(self):
        self._chk(lib().wx_sync(self._h))

    def tune_placement(self, tries: int = 6, iters_per_try: int = 30):
        """wx_tune_placement: try ``tries`` further device allocations for the handle's planes, keep the fastest; returns
        (ms per iteration of the first candidate, of the winner). The state is unchanged."""
        a, b = C.c_float(0), C.c_float(0)
        self.generation += 1  # (the planes move: every cached view of device memory -- device_ptr, devtools.field_tensor -- dangles)
        self._chk(lib().wx_tune_placement(self._h, int(tries), int(iters_per_try), C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def water_free(self) -> bool:
        """Did the last upload find this handle's cells water-free (wx_water_free)?"""
        return bool(lib().wx_water_free(self._h))

    def slab_assert_water_free(self, agreed: bool):
        """The host's assertion that EVERY slab of the domain was uploaded water-free (wx_slab_assert_water_free)."""
        self._chk(lib().wx_slab_assert_water_free(self._h, 1 if agreed else 0))

    OPT_SPLAT_ORDER, OPT_CHECK_LAUNCHES, OPT_KERNEL_SET, OPT_DRY_KERNEL, OPT_ROW_BANDS, OPT_FIX_CAP, OPT_POOL_EXACT, OPT_EXCHANGE_OVERLAP, OPT_SPLIT_LAUNCH, OPT_DRY_PAIRS = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
    OPT_PLACEMENT_SEARCH = 12  # tries of the handle's one placement search, run inside the first step of big whole-domain handles (0: never)
    OPT_WATER0_ON_DEMAND = 11  # waterTexture_0 made when asked for (default 1) instead of stored by every frame's last iteration

    def set_option(self, option: int, value: int):
        """wx_set_option: OPT_SPLAT_ORDER 1 = deterministic particle splats (sorted, droplet-index order); OPT_CHECK_LAUNCHES 1 =
        synchronise and check after every kernel launch."""
        self._chk(lib().wx_set_option(self._h, int(option), int(value)))

    @property
    def iter(self) -> int:
        return lib().wx_get_iter(self._h)

    @iter.setter
    def iter(self, v: int):
        self._chk(lib().wx_set_iter(self._h, int(v)))

    # ---- readback ----
    def read_rect(self, field: str, x: int = 0, y: int = 0, w: Optional[int] = None, h: Optional[int] = None, *, int32: bool = False):
        w = self.X - x if w is None else w
        h = self.Y - y if h is None else h
        if field == "LIGHTNING":
            out = np.zeros(4, np.float32)
            self._chk(lib().wx_read_rect(self._h, 12, 0, 0, 1, 1, out.ctypes.data, DTYPE_F32))
            return out
        ch = FIELD_CHANNELS.get(field, 4)
        if field.startswith("WALL"):
            out = np.zeros((max(h, 0), max(w, 0), ch), np.int32 if int32 else np.int8)
            dt = DTYPE_I32 if int32 else DTYPE_I8
        elif field == "EMITTED":  # RGBA16F emittedLight, in its own format
            out = np.zeros((max(h, 0), max(w, 0), 4), np.float16)
            dt = DTYPE_F16
        else:
            out = np.zeros((max(h, 0), max(w, 0), ch), np.float32)
            dt = DTYPE_F32
        self._chk(lib().wx_read_rect(self._h, FIELD_IDS[field], x, y, w, h, out.ctypes.data, dt))
        return out

    def read_particles(self, first: int = 0, count: Optional[int] = None):
        count = self.n_droplets - first if count is None else count
        out = np.zeros((max(count, 0), 5), np.float32)
        self._chk(lib().wx_read_particles(self._h, first, count, out.ctypes.data))
        return out

    # ---- plumbing ----
    def set_stream(self, stream_ptr: int):
        self._chk(lib().wx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_comm_stream(self, stream_ptr: int):
        """Run wx_halo_pack / wx_halo_unpack on this stream (the one the host issues send / recv on), event-fenced against compute."""
        self._chk(lib().wx_set_comm_stream(self._h, C.c_void_p(stream_ptr)))

    def device_ptr(self, field: str) -> int:
        return lib().wx_device_ptr(self._h, FIELD_IDS[field])

    def halo_bytes(self) -> int:
        return lib().wx_halo_bytes(self._h)

    def halo_message_bytes(self) -> int:
        """What a halo message of the current period occupies at the front of a wx_halo_bytes buffer (the base texture alone between slabs
        of the agreed water-free dry stencil)."""
        return lib().wx_halo_message_bytes(self._h)

    def halo_pack(self, side: int, dev_ptr: int):
        self._chk(lib().wx_halo_pack(self._h, side, C.c_void_p(dev_ptr)))

    def halo_unpack(self, side: int, dev_ptr: int):
        self._chk(lib().wx_halo_unpack(self._h, side, C.c_void_p(dev_ptr)))

    def halo_pack_both(self, dev_left: int, dev_right: int):
        """Both sides in one launch (a second small launch would queue behind the marching kernel's thousands of workgroups)."""
        self._chk(lib().wx_halo_pack_both(self._h, C.c_void_p(dev_left), C.c_void_p(dev_right)))

    def halo_unpack_both(self, dev_left: int, dev_right: int):
        self._chk(lib().wx_halo_unpack_both(self._h, C.c_void_p(dev_left), C.c_void_p(dev_right)))

    # ---- display streaming (wx_stream_frame): one pinned buffer per Handle, re-used while the viewport size is unchanged
    STREAM_FIELDS = (("BASE_DISP", np.float32, 4), ("WATER_CUR", np.float32, 4), ("WALL_DISP", np.int8, 4), ("LIGHT_0", np.float32, 4),
                     ("CURL", np.float32, 1), ("PRECIP_FB", np.float32, 4), ("EMITTED", np.float16, 4))

    def stream_frame(self, x: int = 0, y: int = 0, w: Optional[int] = None, h: Optional[int] = None):
        """Start the asynchronous copy of the display fields of a viewport; returns immediately."""
        w = self.X - x if w is None else w
        h = self.Y - y if h is None else h
        L = lib()
        nbytes = L.wx_stream_bytes(w, h)
        if getattr(self, "_pin", None) is None or self._pin[1] != nbytes:
            if getattr(self, "_pin", None) is not None:
                L.wx_host_free(self._pin[0])
            p = L.wx_host_alloc(nbytes)
            if not p:
                raise MemoryError(f"wx_host_alloc({nbytes})")
            self._pin = (p, nbytes)
        self._chk(L.wx_stream_frame(self._h, x, y, w, h, C.c_void_p(self._pin[0])))
        self._frame = (w, h)

    def stream_wait(self):
        """Wait for the frame started by stream_frame(); returns {field: array} views of the pinned buffer."""
        self._chk(lib().wx_stream_wait(self._h))
        w, h = self._frame
        raw = (C.c_char * self._pin[1]).from_address(self._pin[0])
        out, off = {}, 0
        for name, dt, ch in self.STREAM_FIELDS:
            n = w * h * ch
            out[name] = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(h, w, ch)
            off += n * np.dtype(dt).itemsize
        return out

    def setup_columns(self, desc, drops=None):
        """Device-side initialiser from the 1-D descriptors of ``synth.terrain_columns`` (wx_setup_columns)."""
        want = {"wall_rows": (np.int32, self.X), "sea": (np.uint8, self.X), "veg_noise": (np.float64, self.X), "snow": (np.float32, self.X),
                "T_air": (np.float32, self.Y), "total_water": (np.float32, self.Y), "cloud_water": (np.float32, self.Y)}
        arrs = []
        for k, (dt, n) in want.items():
            a = np.ascontiguousarray(desc[k], dt)
            if a.shape != (n,):
                raise ValueError(f"setup_columns: {k} has shape {a.shape}, expected ({n},)")
            arrs.append(a)
        dp = None
        if drops is not None:
            drops = np.ascontiguousarray(drops, np.float32)
            if drops.shape != (self.n_droplets, 5):
                raise ValueError("setup_columns: drops must be (n_droplets, 5) float32")
            dp = drops.ctypes.data_as(C.c_void_p)
        self._chk(lib().wx_setup_columns(self._h, *[a.ctypes.data_as(C.c_void_p) for a in arrs], dp))

    def setup_terrain(self, sounding, seed: float = 0.5, height_mult: float = 0.3, snap: int = 2, sim_height: float = 12000.0, drops=None):
        """New simulation generated entirely on the device (wx_setup_terrain): ``sounding`` = the per-row arrays of
        ``synth.sounding_rows`` (or any dict with T_air / total_water / cloud_water), terrain from the setup shader's noise."""
        arrs = []
        for k in ("T_air", "total_water", "cloud_water"):
            a = np.ascontiguousarray(sounding[k], np.float32)
            if a.shape != (self.Y,):
                raise ValueError(f"setup_terrain: {k} has shape {a.shape}, expected ({self.Y},)")
            arrs.append(a)
        dp = None
        if drops is not None:
            drops = np.ascontiguousarray(drops, np.float32)
            if drops.shape != (self.n_droplets, 5):
                raise ValueError("setup_terrain: drops must be (n_droplets, 5) float32")
            dp = drops.ctypes.data_as(C.c_void_p)
        self._chk(lib().wx_setup_terrain(self._h, float(seed), float(height_mult), int(snap), float(sim_height), *[a.ctypes.data_as(C.c_void_p) for a in arrs], dp))

    def fastest_velocity(self) -> float:
        """Largest |velocity component| [cells / iteration] the exact path of the marching wet kernel saw since the last call (0: none >= 0.9)."""
        v = C.c_float(0)
        self._chk(lib().wx_fastest_velocity(self._h, C.byref(v)))
        return float(v.value)

    def placement_info(self):
        """(ms per iteration on the first allocations, on the kept ones) of the handle's placement search, or None if none has run."""
        a, b = C.c_float(0), C.c_float(0)
        return (float(a.value), float(b.value)) if lib().wx_placement_info(self._h, C.byref(a), C.byref(b)) == 1 else None

    def pair_stats(self):
        """(cells recomputed by the pair kernel's exact path, pairs repeated whole) since the last call; resets both; synchronises."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(lib().wx_pair_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- slabs exact at any speed (include/wxsim.h): the hosts of all slabs agree on a |vx| bound per exchange period ----
    def slab_vx_take(self) -> float:
        """Largest |vx| this slab has seen since the last take (0 below 0.5 cells / iteration); synchronises."""
        v = C.c_float(0)
        self._chk(lib().wx_slab_vx_take(self._h, C.byref(v)))
        return float(v.value)

    def slab_set_vx_bound(self, v_measured: float):
        """v_measured = the maximum over ALL slabs: sizes the coming exchange period (cone = 6 + floor(1.25 v + 0.25) once that reaches 1)."""
        self._chk(lib().wx_slab_set_vx_bound(self._h, C.c_float(float(v_measured))))

    @property
    def slab_cone(self) -> int:
        return int(lib().wx_slab_cone(self._h))

    @property
    def slab_period(self) -> int:
        return int(lib().wx_slab_period(self._h))

    def init_droplets(self, seed: int = 1):
        """initRainDrops() on the device (wx_init_droplets): a fresh all-inactive pool, a pure function of the seed."""
        self._chk(lib().wx_init_droplets(self._h, C.c_uint32(int(seed) & 0xFFFFFFFF)))

    # ---- particles on slabs (device pointers; see include/wxsim.h) ----
    def slab_set_rank(self, rank: int):
        self._chk(lib().wx_slab_set_rank(self._h, rank))

    def slab_period_begin(self):
        self._chk(lib().wx_slab_period_begin(self._h))

    def pool_event_bytes(self) -> int:
        return lib().wx_pool_event_bytes(self._h)

    def pool_edge_bytes(self) -> int:
        return lib().wx_pool_edge_bytes(self._h)

    def pool_events_pack(self, dev_buf: int):
        self._chk(lib().wx_pool_events_pack(self._h, C.c_void_p(dev_buf)))

    def pool_events_apply(self, dev_bufs: int, n_ranks: int, stride_bytes: int = 0):
        self._chk(lib().wx_pool_events_apply(self._h, C.c_void_p(dev_bufs), int(n_ranks), int(stride_bytes)))

    def pool_edges_pack(self, dev_left: int, dev_right: int, refresh_inactive: bool = False):
        self._chk(lib().wx_pool_edges_pack(self._h, C.c_void_p(dev_left), C.c_void_p(dev_right), 1 if refresh_inactive else 0))

    def pool_edges_apply(self, dev_buf: int):
        self._chk(lib().wx_pool_edges_apply(self._h, C.c_void_p(dev_buf)))

    def pool_flags(self) -> np.ndarray:
        """Per droplet: 0 tracked by another rank, 1 inactive, 2 active in this rank's owned columns, 3 ghost copy."""
        out = np.zeros(self.n_droplets, np.uint8)
        self._chk(lib().wx_pool_flags(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def lightning(self) -> np.ndarray:
        out = np.zeros(4, np.float32)
        self._chk(lib().wx_lightning_get(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_lightning(self, v):
        v = np.ascontiguousarray(v, np.float32)
        self._chk(lib().wx_lightning_set(self._h, v.ctypes.data_as(C.c_void_p)))

    # ---- the halo exchange inside the library (one rank per process; see include/wxsim.h) ----
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: 128 bytes that rank 0 hands to every other rank (any side channel: torch.distributed, a file, MPI)."""
        buf = C.create_string_buffer(128)
        rc = lib().wx_comm_unique_id(buf)
        if rc != 0:
            raise WxError(rc, (lib().wx_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != 128:
            raise ValueError("the unique id has 128 bytes")
        self._chk(lib().wx_comm_init(self._h, C.c_char_p(unique_id), int(rank), int(world)))

    def exchange(self):
        """Ring halo exchange over RCCL (pack, send / recv, unpack), enqueued on the handle's comm stream."""
        self._chk(lib().wx_exchange(self._h))

    def slab_step(self, n: int):
        """n iterations with an exchange every halo / 6 iterations, overlapped with compute (wx_slab_step)."""
        self._chk(lib().wx_slab_step(self._h, int(n)))

    def profile(self, enable: bool):
        self._chk(lib().wx_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        L = lib()
        n = L.wx_kernel_count()
        ms = (C.c_float * n)()
        cnt = (C.c_int * n)()
        self._chk(L.wx_profile_read(self._h, n, ms, cnt))
        return {L.wx_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(n) if cnt[k] > 0}


TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_LOCAL = 0, 1, 2


class Group:
    """N column slabs of one periodic domain in THIS process (wx_group_*): one handle per slab -- on N devices with RCCL between them,
    or several per device with device-to-device copies -- stepped together with the halo exchange inside the library."""

    def __init__(self, n_slabs: int, X_global: int, Y: int, halo: int = 42, devices=None, transport: int = TRANSPORT_AUTO, n_droplets: int = 0):
        L = lib()
        g = C.c_void_p()
        _apply_env_defaults()
        dev = None
        if devices is not None:
            dev = (C.c_int * n_slabs)(*[int(d) for d in devices])
        rc = L.wx_group_create(int(n_slabs), dev, int(X_global), int(Y), int(halo), int(n_droplets), int(transport), C.byref(g))
        if rc != 0:
            raise WxError(rc, (L.wx_group_last_error(None) or b"").decode())
        self._g = g
        self.n, self.X, self.Y = n_slabs, X_global, Y
        self.halo = halo if n_slabs > 1 else 0
        self.xo = X_global // n_slabs
        self.transport = L.wx_group_transport(g)
        self.n_droplets = n_droplets
        self.slabs = [Handle._borrowed(L.wx_group_slab(g, i), self.xo, Y, self.halo, self, n_droplets) for i in range(n_slabs)]

    def _chk(self, rc: int):
        if rc != 0:
            raise WxError(rc, (lib().wx_group_last_error(self._g) or b"").decode())

    def columns(self, i: int) -> np.ndarray:
        """Global column of every local column of slab i (owned + ghost columns, periodic)."""
        return (i * self.xo - self.halo + np.arange(self.xo + 2 * self.halo)) % self.X

    def upload(self, base, water, wall, drops=None):
        """Whole-domain arrays (Y, X, 4) cut into the slabs' local arrays; ``drops`` = the WHOLE droplet pool, handed to every slab."""
        for i, h in enumerate(self.slabs):
            idx = self.columns(i)
            h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]), drops)
        self._chk(lib().wx_group_agree(self._g))

    def set_option(self, option: int, value: int):
        self._chk(lib().wx_group_set_option(self._g, int(option), int(value)))

    def exchange(self):
        """An exchange now (wx_group_exchange): afterwards every active droplet is owned by exactly one slab."""
        self._chk(lib().wx_group_exchange(self._g))

    def particles(self) -> np.ndarray:
        """The whole droplet pool assembled from the slabs (call right after an exchange: inside a period a droplet near an edge is held
        by two slabs): an active droplet's record comes from the slab that owns it, an inactive one's is the same everywhere."""
        d = [h.read_particles() for h in self.slabs]
        if self.n == 1:
            return d[0]
        f = np.stack([h.pool_flags() for h in self.slabs])
        if ((f == 2).sum(0) > 1).any():
            raise RuntimeError("Group.particles: a droplet is owned by two slabs (call exchange() first)")
        out = d[0].copy()
        for k in range(self.n):
            out[f[k] == 2] = d[k][f[k] == 2]
        return out

    def set_params(self, p: WxParams, initial_T=None, snd_T=None, snd_W=None, snd_Vel=None):
        for h in self.slabs:
            h.set_params(p, initial_T, snd_T, snd_W, snd_Vel)

    def step(self, n: int = 1):
        self._chk(lib().wx_group_step(self._g, int(n)))

    def sync(self):
        self._chk(lib().wx_group_sync(self._g))

    def read(self, field: str) -> np.ndarray:
        """The whole domain's field assembled from the slabs' owned columns."""
        return np.concatenate([h.read_rect(field, self.halo, 0, self.xo, self.Y) for h in self.slabs], axis=1)

    def close(self):
        if getattr(self, "_g", None):
            for h in self.slabs:
                h._h = None
            lib().wx_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
