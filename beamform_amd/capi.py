"""ctypes binding of libbfcore.so (include/bfcore.h).

Thin plumbing only: device memory comes from the caller (torch tensors / raw
pointers); all compute happens in the HIP kernels behind the C ABI.  Loading
fails loudly when the library is missing -- there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .params import ALGO_ID

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BFCORE_LIB") or os.path.join(_HERE, "lib", "libbfcore.so")  # BFCORE_LIB: A/B builds of the same ABI

BF_MAX_MICS = 32
BF_MAX_INTERF = 15
BF_PLANAR, BF_INTERLEAVED = 0, 1
BF_PRECISION_REFERENCE, BF_PRECISION_MIXED = 0, 1   # bf_precision: doubles between the transforms (default) / z48 spectra + fp32 backward transform
BF_DAS_FUSED_F32, BF_DAS_F64 = 0, 1   # bf_das_impl: fp32 opt-in / double like the reference (the default, as in bf_config_init)

#: every symbol include/bfcore.h declares
EXPORTS = (
    "bf_version", "bf_strerror", "bf_last_error", "bf_device_count", "bf_config_init", "bf_config_load_yaml",
    "bf_config_parse_yaml", "bf_create", "bf_destroy", "bf_set_theta", "bf_set_interference", "bf_process_hop",
    "bf_process_batch", "bf_process_batch_device", "bf_get_weights", "bf_state_size", "bf_get_state", "bf_set_state",
    "bf_reset", "bf_reset_async", "bf_shard_halo", "bf_shard_plan", "bf_shard_run", "bf_shard_first_feed", "bf_shard_n_feed", "bf_shard_n_drop", "bf_process_batch_device_strided", "bf_kernel_timing_begin", "bf_kernel_timing_end", "bf_trace_begin", "bf_trace_end", "bf_time_batch_device", "bf_n_interferers", "bf_set_theta_dir", "bf_set_thetas", "bf_stream_rms", "bf_host_alloc", "bf_host_free",
    "bf_wav_writer_open", "bf_wav_writer_write", "bf_wav_writer_write_pcm16", "bf_wav_writer_close", "bf_float_to_pcm16",
    "bf_float_to_pcm16_device", "bf_wav_read", "bf_planar_f32_read", "bf_wav_free",
    "bf_resampler_create", "bf_resampler_set_table", "bf_resampler_reset", "bf_resampler_out_count", "bf_resampler_latency",
    "bf_resampler_process_device", "bf_resampler_process", "bf_resampler_destroy", "bf_resampler_default_table",
    "bf_resampler_set_mode", "bf_resampler_callback", "bf_resampler_callback_device",
)


class BfConfig(C.Structure):
    _fields_ = [
        ("algo", C.c_int), ("n_mics", C.c_int), ("hop", C.c_int), ("sample_rate", C.c_double),
        ("mic_x", C.c_double * BF_MAX_MICS), ("mic_y", C.c_double * BF_MAX_MICS), ("theta", C.c_double),
        ("n_interf", C.c_int), ("interf_angle", C.c_double * BF_MAX_INTERF), ("verbose", C.c_int),
        ("past_windows", C.c_int), ("freq_mag_threshold", C.c_double), ("freq_max", C.c_double),
        ("freq_min", C.c_double), ("out_amp", C.c_double), ("interf_angle_threshold", C.c_double),
        ("mu", C.c_double), ("lambda_", C.c_double),
        ("min_phase", C.c_double), ("mag_mult", C.c_double), ("mag_threshold", C.c_double),
        ("min_mag", C.c_double), ("smooth_size", C.c_int),
        ("mcra_alphaS", C.c_double), ("mcra_alphaD", C.c_double), ("mcra_alphaD2", C.c_double),
        ("mcra_delta", C.c_double), ("mcra_L", C.c_int),
        ("mpf_alphaS", C.c_double), ("mpf_eta", C.c_double), ("mpf_rev_gamma", C.c_double),
        ("mpf_rev_delta", C.c_double), ("noise_floor", C.c_double),
        ("out_only_noise", C.c_int), ("out_only_mcra", C.c_int),
        ("device", C.c_int), ("n_streams", C.c_int), ("layout", C.c_int), ("das_impl", C.c_int), ("precision", C.c_int), ("n_dirs", C.c_int),
        ("gsc_use_vad", C.c_int), ("gsc_vad_threshold", C.c_double), ("gsc_mu0", C.c_double), ("gsc_mu_max", C.c_double),
        ("gsc_filter_size", C.c_int),
    ]


#: defaults of parameters added after the first golden fixtures were written (launch/gsc.launch:6-11)
_LATER_KEYS = dict(gsc_use_vad=0, gsc_vad_threshold=0.1, gsc_mu0=0.0001, gsc_mu_max=0.1, gsc_filter_size=128)


class BfError(RuntimeError):
    def __init__(self, code, what, detail=""):
        super().__init__(f"{what}: [{code}] {detail}")
        self.code = code


_lib = None


#: which HIP runtime libbfcore.so ended up bound to: "torch:<path>" (the wheel's copy was loaded first), "system" or "unknown"
HIP_RUNTIME_BOUND = "unknown"


def _share_torch_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel ships its own libamdhip64.so (same soname as /opt/rocm's); if libbfcore.so
    is loaded first it brings in /opt/rocm's copy, torch then loads its own beside it, and the second runtime to initialise reports
    "no ROCm-capable device".  When torch is installed but not imported yet, load ITS runtime first (without importing torch) so that
    libbfcore.so's DT_NEEDED resolves to the copy torch will use.  BF_NO_TORCH_HIP=1 skips this (a process that never imports torch
    keeps the /opt/rocm runtime libbfcore.so was built against).  What happened is recorded in HIP_RUNTIME_BOUND; BF_HIP_DEBUG=1 also
    prints the runtime's version (that call initialises HIP, so it is made in debug runs only: load() itself stays a plain dlopen, safe
    in front of a fork and for the host-only helpers -- wav, config, pcm16)."""
    global HIP_RUNTIME_BOUND
    import sys
    if os.environ.get("BF_NO_TORCH_HIP") == "1":
        HIP_RUNTIME_BOUND = "system"
        return
    if "torch" in sys.modules:
        HIP_RUNTIME_BOUND = "torch:already imported"
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        HIP_RUNTIME_BOUND = "system"
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            rt = C.CDLL(path, mode=C.RTLD_GLOBAL)
            HIP_RUNTIME_BOUND = "torch:" + path
            if os.environ.get("BF_HIP_DEBUG") == "1":
                v = C.c_int(0)
                if rt.hipRuntimeGetVersion(C.byref(v)) == 0:
                    print(f"[bfcore] bound to the torch wheel's HIP runtime {path} (version {v.value})", file=sys.stderr)
        except OSError:
            HIP_RUNTIME_BOUND = "system"  # not loadable on its own: libbfcore.so falls back to its RUNPATH copy
    else:
        HIP_RUNTIME_BOUND = "system"


def load():
    """dlopen libbfcore.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()); "
                          "beamform_amd has no CPU fallback")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.bf_version.restype = C.c_char_p
    L.bf_strerror.restype = C.c_char_p
    L.bf_strerror.argtypes = [C.c_int]
    L.bf_last_error.restype = C.c_char_p
    L.bf_last_error.argtypes = [C.c_void_p]
    L.bf_config_init.argtypes = [C.POINTER(BfConfig), C.c_int]
    L.bf_config_load_yaml.argtypes = [C.POINTER(BfConfig), C.c_char_p]
    L.bf_config_parse_yaml.argtypes = [C.POINTER(BfConfig), C.c_char_p]
    L.bf_create.argtypes = [C.POINTER(BfConfig), C.POINTER(C.c_void_p)]
    L.bf_destroy.argtypes = [C.c_void_p]
    L.bf_destroy.restype = None
    L.bf_set_theta.argtypes = [C.c_void_p, C.c_double]
    L.bf_set_theta_dir.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.bf_set_thetas.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.bf_stream_rms.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.c_void_p]
    L.bf_set_interference.argtypes = [C.c_void_p, C.c_uint, C.c_double]
    L.bf_n_interferers.argtypes = [C.c_void_p]
    L.bf_process_hop.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32]
    L.bf_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bf_host_alloc.restype = C.c_void_p
    L.bf_host_alloc.argtypes = [C.c_size_t]
    L.bf_host_free.argtypes = [C.c_void_p]
    L.bf_host_free.restype = None
    L.bf_process_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bf_get_weights.argtypes = [C.c_void_p, C.c_void_p]
    L.bf_state_size.restype = C.c_size_t
    L.bf_state_size.argtypes = [C.c_void_p]
    L.bf_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.bf_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.bf_reset.argtypes = [C.c_void_p]
    L.bf_reset_async.argtypes = [C.c_void_p, C.c_void_p]
    L.bf_process_batch_device_strided.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_long]
    L.bf_kernel_timing_begin.argtypes = [C.c_void_p]
    L.bf_kernel_timing_end.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.bf_trace_begin.argtypes = []
    L.bf_trace_end.argtypes = [C.c_char_p, C.c_size_t]
    L.bf_trace_end.restype = C.c_long
    L.bf_time_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bf_wav_writer_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.bf_wav_writer_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.bf_wav_writer_write_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.bf_wav_writer_close.argtypes = [C.c_void_p]
    L.bf_float_to_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.bf_float_to_pcm16.restype = None
    L.bf_float_to_pcm16_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bf_wav_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                              C.POINTER(C.c_int)]
    L.bf_planar_f32_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t)]
    L.bf_wav_free.argtypes = [C.POINTER(C.c_float)]
    L.bf_wav_free.restype = None
    L.bf_resampler_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.bf_resampler_set_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.bf_resampler_reset.argtypes = [C.c_void_p]
    L.bf_resampler_out_count.argtypes = [C.c_void_p, C.c_size_t]
    L.bf_resampler_out_count.restype = C.c_size_t
    L.bf_resampler_latency.argtypes = [C.c_void_p]
    L.bf_resampler_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    L.bf_resampler_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bf_resampler_destroy.argtypes = [C.c_void_p]
    L.bf_resampler_destroy.restype = None
    L.bf_resampler_default_table.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.bf_resampler_set_mode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.bf_resampler_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bf_resampler_callback_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    _lib = L
    return L


# ---- rosjack output stage, file half (rosjack.cpp:189-210, 404-409) and the batch front-end ---------------------------------
class launch_trace:
    """with launch_trace() as t: ... ; t.kernels = the kernels this thread launched through the library inside the block, in launch
    order, named as rocprofv3 names them (bf_trace_begin / bf_trace_end)."""

    def __enter__(self):
        L = load()
        rc = L.bf_trace_begin()
        if rc != 0:
            raise BfError(rc, "bf_trace_begin", "a trace is already open on this thread")
        self.kernels = []
        return self

    def __exit__(self, *exc):
        L = load()
        buf = C.create_string_buffer(1 << 16)
        n = L.bf_trace_end(buf, len(buf))
        if n < 0:
            raise BfError(int(n), "bf_trace_end")
        self.kernels = [k for k in buf.value.decode().split("\n") if k]
        return False


class WavWriter:
    """sf_open(..., SFM_WRITE, WAV | PCM_16, mono) / sf_write_float per callback / sf_close."""

    def __init__(self, path: str, sample_rate: int = 48000):
        self._L = load()
        self._w = C.c_void_p()
        rc = self._L.bf_wav_writer_open(os.fsencode(path), int(sample_rate), C.byref(self._w))
        if rc:
            raise BfError(rc, "bf_wav_writer_open", self._L.bf_strerror(rc).decode())

    def write(self, samples: np.ndarray):
        a = np.ascontiguousarray(samples, np.float32)
        rc = self._L.bf_wav_writer_write(self._w, a.ctypes.data, a.size)
        if rc:
            raise BfError(rc, "bf_wav_writer_write", self._L.bf_strerror(rc).decode())

    def write_pcm16(self, pcm: np.ndarray):
        a = np.ascontiguousarray(pcm, np.int16)
        rc = self._L.bf_wav_writer_write_pcm16(self._w, a.ctypes.data, a.size)
        if rc:
            raise BfError(rc, "bf_wav_writer_write_pcm16", self._L.bf_strerror(rc).decode())

    def close(self):
        if self._w:
            rc = self._L.bf_wav_writer_close(self._w)
            self._w = None
            if rc:
                raise BfError(rc, "bf_wav_writer_close", self._L.bf_strerror(rc).decode())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def float_to_pcm16(x: np.ndarray) -> np.ndarray:
    """The sample rule of sf_write_float on a PCM_16 file (host)."""
    a = np.ascontiguousarray(x, np.float32)
    out = np.empty(a.shape, np.int16)
    load().bf_float_to_pcm16(a.ctypes.data, out.ctypes.data, a.size)
    return out


def float_to_pcm16_device(src_ptr: int, dst_ptr: int, n: int, stream: int = 0):
    rc = load().bf_float_to_pcm16_device(src_ptr, dst_ptr, n, stream or None)
    if rc:
        raise BfError(rc, "bf_float_to_pcm16_device", load().bf_strerror(rc).decode())


def _take_planar(L, p, ch, n):
    arr = np.ctypeslib.as_array(p, shape=(ch * n,)).reshape(ch, n).copy() if ch * n else np.zeros((ch, 0), np.float32)
    L.bf_wav_free(p)
    return arr


def read_wav(path: str):
    """WAV file -> (planar float32 [channels, samples], sample_rate)."""
    L = load()
    p, ch, n, sr = C.POINTER(C.c_float)(), C.c_int(), C.c_size_t(), C.c_int()
    rc = L.bf_wav_read(os.fsencode(path), C.byref(p), C.byref(ch), C.byref(n), C.byref(sr))
    if rc:
        raise BfError(rc, "bf_wav_read", L.bf_strerror(rc).decode())
    return _take_planar(L, p, ch.value, n.value), sr.value


def read_planar_f32(path: str, n_channels: int) -> np.ndarray:
    L = load()
    p, n = C.POINTER(C.c_float)(), C.c_size_t()
    rc = L.bf_planar_f32_read(os.fsencode(path), n_channels, C.byref(p), C.byref(n))
    if rc:
        raise BfError(rc, "bf_planar_f32_read", L.bf_strerror(rc).decode())
    return _take_planar(L, p, n_channels, n.value)


# ---- rosjack output stage, sample-rate half (rosjack.cpp:159-184, 311-338) ----------------------------------------------------
class Resampler:
    """src_new(SRC_SINC_FASTEST, 1) with src_ratio = out_rate / in_rate; process() is src_process with end_of_input = 0."""

    def __init__(self, in_rate: int, out_rate: int):
        self._L = load()
        self._r = C.c_void_p()
        rc = self._L.bf_resampler_create(int(in_rate), int(out_rate), C.byref(self._r))
        if rc:
            raise BfError(rc, "bf_resampler_create", self._L.bf_strerror(rc).decode())

    def _check(self, rc, what):
        if rc:
            raise BfError(rc, what, self._L.bf_strerror(rc).decode())

    def set_table(self, coeffs: np.ndarray, index_inc: int):
        a = np.ascontiguousarray(coeffs, np.float32)
        self._check(self._L.bf_resampler_set_table(self._r, a.ctypes.data, a.size, int(index_inc)), "bf_resampler_set_table")

    def reset(self):
        self._check(self._L.bf_resampler_reset(self._r), "bf_resampler_reset")

    @property
    def latency(self) -> int:
        return self._L.bf_resampler_latency(self._r)

    def out_count(self, n_in: int) -> int:
        return self._L.bf_resampler_out_count(self._r, n_in)

    def process(self, x: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(x, np.float32)
        n = C.c_size_t()
        out = np.empty(self.out_count(a.size), np.float32)
        self._check(self._L.bf_resampler_process(self._r, a.ctypes.data, a.size, out.ctypes.data, out.size, C.byref(n)),
                    "bf_resampler_process")
        return out[:n.value]

    def process_device(self, in_ptr: int, n_in: int, out_ptr: int, out_cap: int, stream: int = 0) -> int:
        n = C.c_size_t()
        self._check(self._L.bf_resampler_process_device(self._r, in_ptr, n_in, out_ptr, out_cap, C.byref(n), stream or None),
                    "bf_resampler_process_device")
        return n.value

    def set_mode_rosjack(self, period: int):
        """BF_RS_ROSJACK: the stage as rosjack drives it (period drops on upsampling, one block per callback); see bfcore.h."""
        self._period = int(period)
        self._check(self._L.bf_resampler_set_mode(self._r, 1, int(period)), "bf_resampler_set_mode")

    def set_mode_stream(self):
        self._check(self._L.bf_resampler_set_mode(self._r, 0, 0), "bf_resampler_set_mode")

    def callback(self, period: np.ndarray):
        """One output_to_rosjack: returns (published block or None, accepted flag)."""
        a = np.ascontiguousarray(period, np.float32)
        assert a.size == self._period
        out = np.empty(self._period, np.float32)
        em, acc = C.c_int(), C.c_int()
        self._check(self._L.bf_resampler_callback(self._r, a.ctypes.data, out.ctypes.data, C.byref(em), C.byref(acc)), "bf_resampler_callback")
        return (out if em.value else None), bool(acc.value)

    def callback_device(self, in_ptr: int, out_ptr: int, stream: int = 0):
        em, acc = C.c_int(), C.c_int()
        self._check(self._L.bf_resampler_callback_device(self._r, in_ptr, out_ptr, C.byref(em), C.byref(acc), stream or None),
                    "bf_resampler_callback_device")
        return bool(em.value), bool(acc.value)

    def close(self):
        if self._r:
            self._L.bf_resampler_destroy(self._r)
            self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resampler_default_table():
    """(built-in coefficient table, index_inc)."""
    L = load()
    inc = C.c_int()
    n = L.bf_resampler_default_table(None, 0, C.byref(inc))
    t = np.empty(n, np.float32)
    L.bf_resampler_default_table(t.ctypes.data, n, C.byref(inc))
    return t, inc.value


def config_from_params(p: dict, device: int = 0, n_streams: int = 1, layout: int = BF_PLANAR,
                       das_impl: int = BF_DAS_F64, n_dirs: int = 1, precision: int = BF_PRECISION_REFERENCE) -> BfConfig:
    """bf_config from a beamform_amd.params dict (launch defaults first, then overrides)."""
    L = load()
    c = BfConfig()
    rc = L.bf_config_init(C.byref(c), ALGO_ID[p["algo"]])
    if rc:
        raise BfError(rc, "bf_config_init", L.bf_strerror(rc).decode())
    c.n_mics = p["n_mics"]
    c.hop = p["hop"]
    c.sample_rate = p["sample_rate"]
    for i, (x, y) in enumerate(p["mics"]):
        c.mic_x[i], c.mic_y[i] = x, y
    c.theta = p["theta"]
    c.n_interf = len(p["interf"])
    for i, a in enumerate(p["interf"]):
        c.interf_angle[i] = a
    for k in ("past_windows", "freq_mag_threshold", "freq_max", "freq_min", "out_amp", "mu", "lambda_", "min_phase",
              "mag_mult", "mag_threshold", "min_mag", "smooth_size", "mcra_alphaS", "mcra_alphaD", "mcra_alphaD2",
              "mcra_delta", "mcra_L", "mpf_alphaS", "mpf_eta", "mpf_rev_gamma", "mpf_rev_delta", "noise_floor",
              "out_only_noise", "out_only_mcra", "gsc_use_vad", "gsc_vad_threshold", "gsc_mu0", "gsc_mu_max",
              "gsc_filter_size"):
        setattr(c, k, p[k] if k in p else _LATER_KEYS[k])  # fixtures written before a key existed
    c.device, c.n_streams, c.layout, c.das_impl, c.n_dirs = device, n_streams, layout, das_impl, n_dirs
    c.precision = precision
    return c


def host_array(shape, dtype=np.float32):
    """numpy array over page-locked memory from bf_host_alloc (freed when the array is collected)."""
    L = load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = L.bf_host_alloc(n)
    if not ptr:
        raise MemoryError("bf_host_alloc failed")
    buf = (C.c_char * n).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    import weakref
    weakref.finalize(buf, L.bf_host_free, ptr)
    return arr


class Beamformer:
    """One beamformer node behind the C ABI (das|mvdr|lcmv|gss|phase|phasempf|mcra|gsc)."""

    def __init__(self, params: dict, device: int = 0, n_streams: int = 1, layout: int = BF_PLANAR,
                 das_impl: int = BF_DAS_F64, n_dirs: int = 1, precision: int = BF_PRECISION_REFERENCE):
        self._L = load()
        self.cfg = config_from_params(params, device, n_streams, layout, das_impl, n_dirs, precision)
        self.n_dirs = max(1, n_dirs)
        self.n_out = n_streams * self.n_dirs  # output streams: [stream][dir]
        self.M, self.H, self.N = params["n_mics"], params["hop"], 2 * params["hop"]
        self.S = len(params["interf"]) + 1 if params["algo"] in ("lcmv", "gss") else 1
        self.n_streams = n_streams
        self._h = C.c_void_p()
        rc = self._L.bf_create(C.byref(self.cfg), C.byref(self._h))
        if rc:
            raise BfError(rc, "bf_create", self._L.bf_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.bf_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _chk(self, rc, what):
        if rc:
            raise BfError(rc, what, (self._L.bf_last_error(self._h) or b"").decode() or self._L.bf_strerror(rc).decode())

    def set_theta(self, deg: float):
        self._chk(self._L.bf_set_theta(self._h, float(deg)), "bf_set_theta")

    def set_theta_dir(self, d: int, deg: float):
        self._chk(self._L.bf_set_theta_dir(self._h, int(d), float(deg)), "bf_set_theta_dir")

    def set_thetas(self, degs):
        """One /theta per look direction (n_dirs of them), one table rebuild."""
        a = (C.c_double * len(degs))(*[float(v) for v in degs])
        self._chk(self._L.bf_set_thetas(self._h, a, len(degs)), "bf_set_thetas")

    def stream_rms(self, y_ptr: int, n_frames: int, stream: int = 0) -> np.ndarray:
        """RMS of every output stream of a device-resident batch -> [n_streams, n_dirs] float64."""
        out = (C.c_double * self.n_out)()
        self._chk(self._L.bf_stream_rms(self._h, y_ptr, n_frames, out, stream or None), "bf_stream_rms")
        return np.array(out[:], np.float64).reshape(self.n_streams, self.n_dirs)

    def set_interference(self, idx: int, deg: float) -> int:
        """interf_theta_roscallback; returns the interferer count afterwards."""
        self._chk(self._L.bf_set_interference(self._h, int(idx), float(deg)), "bf_set_interference")
        k = self._L.bf_n_interferers(self._h)
        self.S = k + 1
        return k

    def reset(self):
        self._chk(self._L.bf_reset(self._h), "bf_reset")

    def reset_async(self, stream: int = 0):
        """Cold start enqueued on `stream` (no host synchronisation)."""
        self._chk(self._L.bf_reset_async(self._h, stream or None), "bf_reset_async")

    def kernel_timing_begin(self):
        self._chk(self._L.bf_kernel_timing_begin(self._h), "bf_kernel_timing_begin")

    def kernel_timing_end(self):
        """(mean ms per launch of the dominant kernel since begin, launches)"""
        ms, n = C.c_float(), C.c_int()
        self._chk(self._L.bf_kernel_timing_end(self._h, C.byref(ms), C.byref(n)), "bf_kernel_timing_end")
        return float(ms.value), int(n.value)

    def weights(self) -> np.ndarray:
        w = np.empty((self.N, self.M, self.S), np.complex128)
        self._chk(self._L.bf_get_weights(self._h, w.ctypes.data), "bf_get_weights")
        return w

    def process_hop(self, x: np.ndarray) -> np.ndarray:
        """x [M, H] float32 (host) -> [H] float32: one jack_callback."""
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.M, self.H)
        ptrs = (C.c_void_p * self.M)(*[x[m].ctypes.data for m in range(self.M)])
        out = np.empty((self.n_dirs, self.H), np.float32)
        self._chk(self._L.bf_process_hop(self._h, ptrs, out.ctypes.data, self.H), "bf_process_hop")
        return out[0] if self.n_dirs == 1 else out

    def process(self, x: np.ndarray, out: np.ndarray = None) -> np.ndarray:
        """Host batch. planar: x [S, M, F*H] (or [M, F*H] when S == 1); interleaved: [S, F*H, M]."""
        x = np.ascontiguousarray(x, np.float32)
        n = x.size // (self.n_streams * self.M * self.H)
        y = np.empty((self.n_out, n * self.H), np.float32) if out is None else out
        assert y.dtype == np.float32 and y.size == self.n_out * n * self.H and y.flags.c_contiguous
        self._chk(self._L.bf_process_batch(self._h, x.ctypes.data, n, y.ctypes.data), "bf_process_batch")
        y = y.reshape(self.n_out, n * self.H)
        return y[0] if self.n_out == 1 else y

    def process_device(self, x_ptr: int, n_frames: int, y_ptr: int, spectrum_ptr: int = 0, stream: int = 0):
        self._chk(self._L.bf_process_batch_device(self._h, x_ptr, n_frames, y_ptr, spectrum_ptr or None, stream or None),
                  "bf_process_batch_device")

    def process_device_strided(self, x_ptr: int, n_frames: int, y_ptr: int, mic_stride: int, stream: int = 0):
        """process_device on a column range of a longer planar buffer (microphone m at x_ptr + m * mic_stride samples)."""
        self._chk(self._L.bf_process_batch_device_strided(self._h, x_ptr, n_frames, y_ptr, stream or None, mic_stride),
                  "bf_process_batch_device_strided")

    def time_device(self, x_ptr: int, n_frames: int, y_ptr: int, iters: int, stream: int = 0):
        """(mean ms per call, mean ms per launch of the dominant kernel), HIP events on `stream`."""
        ms, msk = C.c_float(), C.c_float()
        self._chk(self._L.bf_time_batch_device(self._h, x_ptr, n_frames, y_ptr, stream or None, iters, C.byref(ms),
                                               C.byref(msk)), "bf_time_batch_device")
        return float(ms.value), float(msk.value)

    def get_state(self) -> bytes:
        n = self._L.bf_state_size(self._h)
        buf = C.create_string_buffer(n)
        self._chk(self._L.bf_get_state(self._h, buf, n), "bf_get_state")
        return buf.raw

    def set_state(self, blob: bytes):
        self._chk(self._L.bf_set_state(self._h, blob, len(blob)), "bf_set_state")
