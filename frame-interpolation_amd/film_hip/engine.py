"""ctypes binding of libfilm_hip.so (C-ABI: include/film_hip.h).

This is the whole Python <-> native boundary of the product path: plain pointers and sizes.
If the shared library is missing or no MI355X is visible, construction FAILS LOUDLY - there is no
CPU or PyTorch fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import Dict, Optional, Tuple

import numpy as np

from .options import Options, PUBLISHED

_HERE = os.path.dirname(os.path.abspath(__file__))
# FILM_HIP_LIB=<path> names the library explicitly; FILM_EXTRA_FAMILIES=1 selects the flavour that also holds the opt-in kernel
# families (bf16 precision modes, F(2,3) / halo kernels: film_hip/build.py), built next to the default one
LIB_PATH = os.environ.get('FILM_HIP_LIB') or os.path.join(
    _HERE, 'libfilm_hip_extra.so' if os.environ.get('FILM_EXTRA_FAMILIES', '0') not in ('', '0') else 'libfilm_hip.so')

FILM_MEM_HOST = 0
FILM_MEM_DEVICE = 1
FILM_ERR_INVALID = -1
FILM_ERR_NO_DEVICE = -3
_MAXS = 8


class FilmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f'libfilm_hip error {code}: {msg}')
        self.code = code
        self.msg = msg


class _Config(ctypes.Structure):
    _fields_ = [('pyramid_levels', ctypes.c_int32), ('fusion_pyramid_levels', ctypes.c_int32),
                ('specialized_levels', ctypes.c_int32), ('sub_levels', ctypes.c_int32),
                ('filters', ctypes.c_int32), ('flow_convs', ctypes.c_int32 * _MAXS),
                ('flow_filters', ctypes.c_int32 * _MAXS)]


# every symbol include/film_hip.h declares; tests assert the library exports all of them
EXPORTED_SYMBOLS = (
    'film_default_config', 'film_create', 'film_destroy', 'film_last_error', 'film_set_weight',
    'film_finalize', 'film_packed_size', 'film_export_packed', 'film_import_packed', 'film_export_layouts', 'film_forward',
    'film_interpolate',
    'film_set_option', 'film_profile_json', 'film_plan_json', 'film_get_tap', 'film_crc32c', 'film_version',
    'film_export_tune', 'film_import_tune', 'film_to_uint8', 'film_load_bundle', 'film_bcast_weights')

_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen()s the engine; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.isfile(p):
        raise FileNotFoundError(
            f'{p} not found: build the HIP engine first (python -c "import __graft_entry__ as g; g.build()" '
            'or make -C frame-interpolation_amd/csrc). There is no CPU fallback.')
    if os.environ.get('FILM_NO_TORCH', '0') in ('', '0'):
        try:
            # PyTorch-ROCm bundles its own HIP runtime; importing it first makes libfilm_hip.so (whose
            # DT_NEEDED is the unversioned "libamdhip64.so", see film_hip/build.py) bind to that same
            # runtime, so device pointers and streams can be exchanged with torch.  Without torch (or with
            # FILM_NO_TORCH=1: a torch-free host, tools/graph_race_check.py on the system runtime) the
            # runtime under /opt/rocm/lib is used.
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
    lib = ctypes.CDLL(p)
    vp, cp, i64p, fp = ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p
    lib.film_default_config.argtypes = [ctypes.POINTER(_Config)]
    lib.film_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(_Config)]
    lib.film_destroy.argtypes = [vp]
    lib.film_destroy.restype = None
    lib.film_last_error.argtypes = [vp]
    lib.film_last_error.restype = cp
    lib.film_set_weight.argtypes = [vp, cp, fp, i64p, ctypes.c_int]
    lib.film_finalize.argtypes = [vp]
    lib.film_packed_size.argtypes = [vp, i64p]
    lib.film_export_packed.argtypes = [vp, fp, ctypes.c_int64, ctypes.c_int]
    lib.film_import_packed.argtypes = [vp, fp, ctypes.c_int64, ctypes.c_int]
    lib.film_export_layouts.argtypes = [vp, fp, ctypes.c_int64, i64p]
    lib.film_forward.argtypes = [vp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, vp]
    lib.film_interpolate.argtypes = [vp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, fp, ctypes.c_int, vp]
    lib.film_set_option.argtypes = [vp, cp, ctypes.c_int64]
    lib.film_profile_json.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64, i64p]
    lib.film_plan_json.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, i64p]
    lib.film_get_tap.argtypes = [vp, cp, fp, ctypes.c_int64, i64p]
    lib.film_crc32c.argtypes = [ctypes.c_uint32, vp, ctypes.c_int64]
    lib.film_crc32c.restype = ctypes.c_uint32
    lib.film_version.argtypes = []
    lib.film_version.restype = cp
    lib.film_export_tune.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64, i64p]
    lib.film_import_tune.argtypes = [vp, cp]
    lib.film_to_uint8.argtypes = [vp, vp, ctypes.c_int64, vp]
    lib.film_load_bundle.argtypes = [vp, cp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, i64p]
    lib.film_bcast_weights.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is ctypes.c_int and name not in ('film_destroy',):
            fn.restype = ctypes.c_int
    if path is None:
        _lib = lib
    return lib


def hip_runtime_info() -> Tuple[str, int, str]:
    """(path, hipRuntimeGetVersion(), 'major.minor.patch') of the HIP runtime libfilm_hip.so is bound to in THIS process: PyTorch's
    bundled libamdhip64.so when torch was imported first, /opt/rocm/lib's otherwise (film_hip/build.py).  HIP_VERSION is
    major * 10^7 + minor * 10^5 + patch: 70226015 is HIP 7.2.26015, not 7.0.2."""
    load_library()
    path = ''
    with open('/proc/self/maps') as f:
        for line in f:
            if 'libamdhip64' in line:
                path = line.split()[-1]
                break
    if not path:
        return '', 0, 'not loaded'
    v = ctypes.c_int(0)
    ctypes.CDLL(path).hipRuntimeGetVersion(ctypes.byref(v))
    return path, v.value, f'{v.value // 10000000}.{v.value // 100000 % 100}.{v.value % 100000}'


def _cfg_struct(opt: Options) -> _Config:
    opt.validate()
    c = _Config()
    c.pyramid_levels, c.fusion_pyramid_levels = opt.pyramid_levels, opt.fusion_pyramid_levels
    c.specialized_levels, c.sub_levels, c.filters = opt.specialized_levels, opt.sub_levels, opt.filters
    for i, (a, b) in enumerate(zip(opt.flow_convs, opt.flow_filters)):
        c.flow_convs[i] = a
        c.flow_filters[i] = b
    return c


class FilmEngine:
    """One engine = one GPU + one stream + packed weights + cached per-shape plans.

    ``device=-1`` gives a plan-only handle (no GPU needed): it can pack weights and describe plans
    but every compute call raises FilmError(FILM_ERR_NO_DEVICE)."""

    def __init__(self, opt: Options = PUBLISHED, device: int = 0):
        self._lib = load_library()
        self._opt = opt
        self._h = ctypes.c_void_p()
        cfg = _cfg_struct(opt)
        rc = self._lib.film_create(ctypes.byref(self._h), device, ctypes.byref(cfg))
        if rc != 0:
            msg = self._lib.film_last_error(None).decode('utf-8', 'replace')
            self._h = ctypes.c_void_p()
            raise FilmError(rc, msg)
        self.device = device

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.film_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise FilmError(rc, self._lib.film_last_error(self._h).decode('utf-8', 'replace'))

    @property
    def options(self) -> Options:
        return self._opt

    # -- weights ----------------------------------------------------------------------------
    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        """weights: canonical name -> HWIO kernel / bias (film_hip.weights)."""
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            dims = (ctypes.c_int64 * a.ndim)(*a.shape)
            self._check(self._lib.film_set_weight(self._h, name.encode(), a.ctypes.data, dims, a.ndim))
        self._check(self._lib.film_finalize(self._h))
        self._load_tune_cache()

    def load_bundle(self, path: str, verify: bool = True) -> Dict[str, Tuple[str, str]]:
        """film_load_bundle: reads the variables bundle of a Keras SavedModel directory (or a bundle prefix) natively - the
        replacement of tf.saved_model.load's variable restore, eval/interpolator.py:148 - places every tensor and finalizes.
        Returns {canonical name: (rule, checkpoint key)}, rule = 'path' | 'shape'; tensors placed by their (unique) shape are
        logged as a warning (a name match they are not)."""
        need = ctypes.c_int64()
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self._lib.film_load_bundle(self._h, os.fsencode(path), 1 if verify else 0, buf, len(buf), ctypes.byref(need)))
        if need.value > len(buf):    # the report did not fit (long checkpoint keys): read the bundle once more with a buffer that holds it
            buf = ctypes.create_string_buffer(int(need.value) + 1)
            self._check(self._lib.film_load_bundle(self._h, os.fsencode(path), 1 if verify else 0, buf, len(buf), ctypes.byref(need)))
        rep = {}
        # (checkpoint keys are bytes of the file: with verify=False a damaged index can hand back anything)
        for line in buf.value.decode('utf-8', 'replace').splitlines():
            parts = line.split('\t')
            if len(parts) == 3:
                rep[parts[0]] = (parts[1], parts[2])
        by_shape = sorted(n for n, (rule, _) in rep.items() if rule != 'path')
        if by_shape:
            import logging
            logging.getLogger('film_hip.tf_bundle').warning(
                '%s: %d of %d tensors were not found under a known object-graph path and were placed by their (unique) shape: %s',
                path, len(by_shape), len(rep), ', '.join(f'{n} <- {rep[n][1] if len(rep[n][1]) <= 160 else rep[n][1][:157] + "..."}' for n in by_shape))
        self._load_tune_cache()
        return rep

    # -- autotune choices across processes ($FILM_TUNE_CACHE = a file path) --------------------------------
    def export_tune(self) -> str:
        need = ctypes.c_int64()
        self._check(self._lib.film_export_tune(self._h, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        self._check(self._lib.film_export_tune(self._h, buf, need.value, ctypes.byref(need)))
        return buf.value.decode()

    def import_tune(self, text: str) -> None:
        self._check(self._lib.film_import_tune(self._h, text.encode()))

    def _load_tune_cache(self) -> None:
        path = os.environ.get('FILM_TUNE_CACHE')
        self._tune_saved = None
        if path and os.path.isfile(path):
            try:
                with open(path) as f:
                    self.import_tune(f.read())
                self._tune_saved = self.export_tune()
            except (OSError, FilmError):
                pass      # an unreadable / malformed cache only costs the measurements again

    def save_tune_cache(self) -> None:
        """Writes the autotune choices to $FILM_TUNE_CACHE if they changed (atomic rename; called after every
        host-buffer forward, call it yourself after warming up a device-resident pipeline)."""
        path = os.environ.get('FILM_TUNE_CACHE')
        if not path or self.device < 0:
            return
        text = self.export_tune()
        if text == getattr(self, '_tune_saved', None):
            return
        tmp = f'{path}.{os.getpid()}.tmp'
        try:
            with open(tmp, 'w') as f:
                f.write(text)
            os.replace(tmp, path)
            self._tune_saved = text
        except OSError:
            pass

    def packed_size(self) -> int:
        n = ctypes.c_int64()
        self._check(self._lib.film_packed_size(self._h, ctypes.byref(n)))
        return n.value

    def export_packed(self) -> np.ndarray:
        out = np.empty(self.packed_size(), dtype=np.float32)
        self._check(self._lib.film_export_packed(self._h, out.ctypes.data, out.size, FILM_MEM_HOST))
        return out

    def export_layouts(self) -> np.ndarray:
        """The kernel-layout blob packed so far (tests / debug; offsets as in plan())."""
        n = ctypes.c_int64()
        self._check(self._lib.film_export_layouts(self._h, None, 0, ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        self._check(self._lib.film_export_layouts(self._h, out.ctypes.data, out.size, ctypes.byref(n)))
        return out

    def import_packed(self, blob: np.ndarray) -> None:
        b = np.ascontiguousarray(blob, dtype=np.float32)
        self._check(self._lib.film_import_packed(self._h, b.ctypes.data, b.size, FILM_MEM_HOST))
        self._load_tune_cache()

    def import_packed_device(self, ptr: int, n_floats: int) -> None:
        self._check(self._lib.film_import_packed(self._h, ctypes.c_void_p(ptr), n_floats, FILM_MEM_DEVICE))
        self._load_tune_cache()

    def export_packed_device(self, ptr: int, n_floats: int) -> None:
        self._check(self._lib.film_export_packed(self._h, ctypes.c_void_p(ptr), n_floats, FILM_MEM_DEVICE))

    def bcast_weights(self, nccl_comm: int, root: int, rank: int, stream: int = 0) -> None:
        """film_bcast_weights: the RCCL broadcast of the parameter blob behind the C-ABI, over the CALLER's ncclComm_t (an integer
        handle here) - for hosts without torch.distributed; film_hip/sharding.py::broadcast_weights is the torch.distributed route."""
        self._check(self._lib.film_bcast_weights(self._h, ctypes.c_void_p(nccl_comm), root, rank, ctypes.c_void_p(stream)))
        if rank != root:
            self._load_tune_cache()

    # -- compute ----------------------------------------------------------------------------
    def forward(self, x0: np.ndarray, x1: np.ndarray) -> np.ndarray:
        """x0, x1: float32 [B,H,W,3] host arrays -> [B,H,W,3] (un-clipped), t = 0.5."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32)
        x1 = np.ascontiguousarray(x1, dtype=np.float32)
        if x0.ndim != 4 or x0.shape[3] != 3 or x0.shape != x1.shape:
            raise ValueError(f'expected two [B,H,W,3] arrays of equal shape, got {x0.shape} and {x1.shape}')
        b, h, w, _ = x0.shape
        out = np.empty_like(x0)
        self._check(self._lib.film_forward(self._h, x0.ctypes.data, x1.ctypes.data, b, h, w,
                                           out.ctypes.data, FILM_MEM_HOST, None))
        self.save_tune_cache()
        return out

    def forward_device(self, x0_ptr: int, x1_ptr: int, b: int, h: int, w: int, out_ptr: int,
                       stream: Optional[int] = None) -> None:
        """Device-resident variant: raw device pointers (e.g. torch.Tensor.data_ptr()); asynchronous
        on `stream` (a hipStream_t as int, e.g. torch.cuda.current_stream().cuda_stream)."""
        self._check(self._lib.film_forward(self._h, ctypes.c_void_p(x0_ptr), ctypes.c_void_p(x1_ptr), b, h, w,
                                           ctypes.c_void_p(out_ptr), FILM_MEM_DEVICE,
                                           ctypes.c_void_p(stream) if stream else None))

    def interpolate_frames(self, x0: np.ndarray, x1: np.ndarray, align: Optional[int] = None,
                           block_shape=None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Interpolator.__call__ semantics in one C-ABI call (film_interpolate): pad to `align`, optional
        block_shape = (bh, bw) tiling with per-patch padding, crop, stitch - all on the device.
        `out`: an optional C-contiguous float32 array of x0's shape to receive the frame (a caller that reuses one - pinned, see
        torch_io.pinned_frame - saves the page faults of a fresh 25 MB array per call); default: a new array, like the reference."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32)
        x1 = np.ascontiguousarray(x1, dtype=np.float32)
        if x0.ndim != 4 or x0.shape[3] != 3 or x0.shape != x1.shape:
            raise ValueError(f'expected two [B,H,W,3] arrays of equal shape, got {x0.shape} and {x1.shape}')
        b, h, w, _ = x0.shape
        bh, bw = (int(block_shape[0]), int(block_shape[1])) if block_shape else (1, 1)
        if out is None:
            out = np.empty_like(x0)
        elif out.dtype != np.float32 or out.shape != x0.shape or not out.flags['C_CONTIGUOUS'] or not out.flags['WRITEABLE']:
            raise ValueError(f'out must be a writable C-contiguous float32 array of shape {x0.shape}')
        self._check(self._lib.film_interpolate(self._h, x0.ctypes.data, x1.ctypes.data, b, h, w, int(align or 0),
                                               bh, bw, out.ctypes.data, FILM_MEM_HOST, None))
        self.save_tune_cache()
        return out

    def interpolate_frames_device(self, x0_ptr: int, x1_ptr: int, b: int, h: int, w: int, out_ptr: int,
                                  align: Optional[int] = None, block_shape=None, stream: Optional[int] = None) -> None:
        """Device-resident film_interpolate: raw device pointers, asynchronous on `stream`."""
        bh, bw = (int(block_shape[0]), int(block_shape[1])) if block_shape else (1, 1)
        self._check(self._lib.film_interpolate(self._h, ctypes.c_void_p(x0_ptr), ctypes.c_void_p(x1_ptr), b, h, w,
                                               int(align or 0), bh, bw, ctypes.c_void_p(out_ptr), FILM_MEM_DEVICE,
                                               ctypes.c_void_p(stream) if stream else None))

    def to_uint8_device(self, src_ptr: int, dst_ptr: int, n: int, stream: Optional[int] = None) -> None:
        """film_to_uint8: write_image's quantisation (eval/util.py:51-52) of n device floats into n device bytes, asynchronous."""
        rc = self._lib.film_to_uint8(ctypes.c_void_p(src_ptr), ctypes.c_void_p(dst_ptr), int(n), ctypes.c_void_p(stream) if stream else None)
        if rc != 0:
            raise FilmError(rc, 'film_to_uint8 failed')

    def set_option(self, key: str, value: int) -> None:
        self._check(self._lib.film_set_option(self._h, key.encode(), int(value)))

    # -- introspection ------------------------------------------------------------------------
    def _json_call(self, fn, *args) -> dict:
        need = ctypes.c_int64()
        self._check(fn(self._h, *args, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        self._check(fn(self._h, *args, buf, need.value, ctypes.byref(need)))
        return json.loads(buf.value.decode())

    def plan(self, b: int, h: int, w: int) -> dict:
        return self._json_call(self._lib.film_plan_json, b, h, w)

    def profile(self) -> dict:
        return self._json_call(self._lib.film_profile_json)

    def tap(self, name: str) -> np.ndarray:
        dims = (ctypes.c_int64 * 4)()
        self._check(self._lib.film_get_tap(self._h, name.encode(), None, 0, dims))
        shape = tuple(int(d) for d in dims)
        out = np.empty(shape, dtype=np.float32)
        self._check(self._lib.film_get_tap(self._h, name.encode(), out.ctypes.data, out.size, dims))
        return out

    def forward_with_aux(self, x0: np.ndarray, x1: np.ndarray) -> Dict[str, object]:
        """The reference model's output dictionary with `use_aux_outputs` on (models/film_net/interpolator.py:
        188-199): 'image', 'x0_warped', 'x1_warped' and the four flow pyramids (lists, finest level first), read
        back from the workspace of this forward.  H, W must already be divisible by 2^(pyramid_levels-1)."""
        from . import weights as W
        image = self.forward(x0, x1)
        b = image.shape[0]
        opt = self.options
        c0 = W.feature_channels(opt)[0]
        aligned0 = self.tap('aligned0')
        res = [self.tap(f'res{l}') for l in range(opt.pyramid_levels)]
        flow = [self.tap(f'v{l}') if l < opt.pyramid_levels - 1 else res[l] for l in range(opt.fusion_pyramid_levels)]
        return {
            'image': image,
            'x0_warped': np.ascontiguousarray(aligned0[..., 2 * c0:2 * c0 + 3]),
            'x1_warped': np.ascontiguousarray(aligned0[..., 2 * c0 + 3:2 * c0 + 6]),
            'forward_residual_flow_pyramid': [r[:b] for r in res],
            'backward_residual_flow_pyramid': [r[b:] for r in res],
            'forward_flow_pyramid': [f[:b] for f in flow],
            'backward_flow_pyramid': [f[b:] for f in flow],
        }

    @staticmethod
    def version() -> str:
        return load_library().film_version().decode()

    @staticmethod
    def has_extra_families() -> bool:
        """True for a FILM_EXTRA_FAMILIES=1 build: the precision modes and the winograd = 2 / halo_all options exist."""
        return '+extra' in FilmEngine.version()
