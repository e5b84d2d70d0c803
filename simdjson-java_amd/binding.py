"""ctypes binding of libsjmi.so (C ABI: include/sjmi.h).  No compute happens in Python and nothing
here falls back to a CPU path: a missing library or GPU raises SjmiError."""
import atexit
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.path.join(_HERE, "libsjmi.so")
SOURCES = ["stage1.hip", "strings.hip", "batch.hip", "walk.hip", "coop_walk.hip", "masks.hip", "sjmi_api.hip", "host/simdjson_parser.cpp"]

ST_UTF8, ST_UNCLOSED, ST_UNESCAPED, ST_CAPACITY, ST_INTERNAL = 1, 2, 4, 0x100, 0x200
PADDING = 64

_MESSAGES = {  # exact reference messages (Utf8Validator.java:166, StructuralIndexer.java:298,301)
    ST_UTF8: "The input is not valid UTF-8",
    ST_UNCLOSED: "Unclosed string. A string is opened, but never closed.",
    ST_UNESCAPED: "Unescaped characters. Within strings, there are characters that should be escaped.",
}


class SjmiError(RuntimeError):
    pass


def status_message(status):
    """Message of the JsonParsingException SimdJsonParser.stage1 would throw (first check wins)."""
    for bit in (ST_UTF8, ST_UNCLOSED, ST_UNESCAPED):
        if status & bit:
            return _MESSAGES[bit]
    return None


def lib_path():
    return _LIB


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> simdjson-java_amd/libsjmi.so (in-tree; cross-compiles without a GPU).  Under a file lock
    with the staleness check repeated behind it (N ranks of a node that find the library stale at the same time: one builds, the
    others wait and load its result); every source compiled to an object of its own (in parallel, only what changed), then linked
    into a temporary file that is renamed into place -- a process that loads the library never sees half of it."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    headers = [os.path.join(_CSRC, h) for h in os.listdir(_CSRC) if h.endswith(".h")] + \
        [os.path.join(_CSRC, "host", h) for h in os.listdir(os.path.join(_CSRC, "host")) if h.endswith(".h")] + \
        [os.path.join(_ROOT, "include", "sjmi.h")]
    newest = max(os.path.getmtime(d) for d in srcs + headers)
    fresh = lambda: os.path.exists(_LIB) and os.path.getmtime(_LIB) >= newest
    if not force and fresh():
        return _LIB
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():
                return _LIB  # (another process built it while this one waited for the lock)
            newest_h = max(os.path.getmtime(h) for h in headers)
            flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-I", os.path.join(_ROOT, "include")]

            def compile_one(src):
                obj = os.path.join(objdir, os.path.basename(src) + ".o")
                if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_h):
                    cmd = ["hipcc"] + flags + ["-c", src, "-o", obj]
                    if verbose:
                        print(" ".join(cmd))
                    subprocess.check_call(cmd)
                return obj

            with ThreadPoolExecutor(min(len(srcs), os.cpu_count() or 1)) as ex:
                objs = list(ex.map(compile_one, srcs))
            tmp = "%s.tmp.%d" % (_LIB, os.getpid())
            cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread"] + objs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, _LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB


_lib = None

EXPORTS = ["sjmi_create", "sjmi_destroy", "sjmi_last_error", "sjmi_version", "sjmi_stage1", "sjmi_stage1_device",
           "sjmi_selftest", "sjmi_set_tile_steps", "sjmi_set_profiling", "sjmi_kernel_time", "sjmi_debug_set_flags", "sjmi_set_tile_mode",
           "sjmi_unescape", "sjmi_unescape_device",
           "sjmi_parser_create", "sjmi_parser_destroy", "sjmi_parser_parse", "sjmi_parser_last_message",
           "sjmi_stage1_batch", "sjmi_stage1_batch_device", "sjmi_parser_parse_batch",
           "sjmi_stage1_batch_isolated", "sjmi_stage1_batch_isolated_device", "sjmi_host_register", "sjmi_set_input_staging",
           "sjmi_host_unregister", "sjmi_stage1_unescape", "sjmi_unescape_batch",
           "sjmi_unescape_batch_device", "sjmi_walk_batch_device", "sjmi_stage1_masks", "sjmi_stage1_masks_device",
           "sjmi_parser_root", "sjmi_parser_batch_root", "sjmi_value_type", "sjmi_value_as_long", "sjmi_value_as_double",
           "sjmi_value_as_boolean", "sjmi_value_as_string", "sjmi_value_get", "sjmi_value_size", "sjmi_value_first",
           "sjmi_value_next", "sjmi_parse_batch_device", "sjmi_parse_batch_device_optimistic", "sjmi_parse_batch_device_rejected", "sjmi_parse_document",
           "sjmi_parser_set_gpu_walk", "sjmi_set_auto_safe", "sjmi_stage1_shard_device", "sjmi_stage1_shard_device2",
           "sjmi_stream_open", "sjmi_stream_push", "sjmi_stream_close", "sjmi_split_open", "sjmi_split_scan", "sjmi_split_resolve", "sjmi_split_close",
           "sjmi_parser_ondemand_init", "sjmi_od_skip_child", "sjmi_od_get_boolean", "sjmi_od_get_long", "sjmi_od_get_integral", "sjmi_od_get_double", "sjmi_od_get_float", "sjmi_od_get_char",
           "sjmi_od_get_string", "sjmi_od_get_field_name", "sjmi_od_start_array", "sjmi_od_next_array_element",
           "sjmi_od_start_object", "sjmi_od_next_object_field", "sjmi_od_move_to_field_value", "sjmi_od_assert_no_more_values",
           "sjmi_od_depth", "sjmi_od_peek"]


# Handles that are still open when the interpreter exits are closed HERE, in an atexit hook -- i.e. while the HIP runtime
# (PyTorch's instance) is still up.  Left to __del__ during interpreter finalisation they would call hipFree /
# hipStreamDestroy in an arbitrary order relative to the runtime's own teardown, which can abort the process after the
# work is done (seen once as a core dump at the end of a GPU test run).
_live = weakref.WeakSet()
_exiting = False


def _close_all_at_exit():
    global _exiting
    for obj in list(_live):
        try:
            obj.close()
        except Exception:
            pass
    _exiting = True


atexit.register(_close_all_at_exit)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise SjmiError("libsjmi.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        try:
            # PyTorch-ROCm bundles its own libamdhip64; load it first so that libsjmi.so binds to the same
            # HIP runtime instance (two runtimes in one process cannot both own the GPU).
            import torch  # noqa: F401
        except ImportError:
            pass
        # (SJMI_LIB: an experiment build of the same sources, tools/build_variant.sh -- A/B measurements only)
        L = C.CDLL(os.environ.get("SJMI_LIB") or _LIB)
        L.sjmi_create.restype = C.c_int
        L.sjmi_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint64]
        L.sjmi_destroy.restype = None
        L.sjmi_destroy.argtypes = [C.c_void_p]
        L.sjmi_last_error.restype = C.c_char_p
        L.sjmi_last_error.argtypes = [C.c_void_p]
        L.sjmi_version.restype = C.c_char_p
        L.sjmi_stage1.restype = C.c_int
        L.sjmi_stage1.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.sjmi_stage1_device.restype = C.c_int
        L.sjmi_stage1_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                         C.c_void_p]
        L.sjmi_selftest.restype = C.c_int
        L.sjmi_selftest.argtypes = [C.c_void_p, C.c_void_p]
        L.sjmi_set_tile_steps.restype = C.c_int
        L.sjmi_set_tile_steps.argtypes = [C.c_void_p, C.c_int]
        L.sjmi_set_profiling.restype = C.c_int
        L.sjmi_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.sjmi_debug_set_flags.restype = C.c_int
        L.sjmi_debug_set_flags.argtypes = [C.c_void_p, C.c_uint32]
        L.sjmi_unescape_batch.restype = C.c_int
        L.sjmi_unescape_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_unescape.restype = C.c_int
        L.sjmi_unescape.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_unescape_device.restype = C.c_int
        L.sjmi_unescape_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_void_p]
        L.sjmi_unescape_batch_device.restype = C.c_int
        L.sjmi_unescape_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                 C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_walk_batch_device.restype = C.c_int
        L.sjmi_walk_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_parser_create.restype = C.c_int
        L.sjmi_parser_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        L.sjmi_parser_destroy.restype = None
        L.sjmi_parser_destroy.argtypes = [C.c_void_p]
        L.sjmi_parser_parse.restype = C.c_int
        L.sjmi_parser_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.sjmi_parser_last_message.restype = C.c_char_p
        L.sjmi_parser_last_message.argtypes = [C.c_void_p]
        L.sjmi_stage1_batch.restype = C.c_int
        L.sjmi_stage1_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_stage1_batch_device.restype = C.c_int
        L.sjmi_stage1_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                               C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_stage1_batch_isolated.restype = C.c_int
        L.sjmi_stage1_batch_isolated.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                                 C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_stage1_batch_isolated_device.restype = C.c_int
        L.sjmi_stage1_batch_isolated_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                                        C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_stage1_unescape.restype = C.c_int
        L.sjmi_stage1_unescape.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_parser_parse_batch.restype = C.c_int
        L.sjmi_parser_parse_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_stage1_shard_device.restype = C.c_int
        L.sjmi_stage1_shard_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_uint64,
                                               C.c_void_p, C.c_void_p]
        L.sjmi_stage1_shard_device2.restype = C.c_int
        L.sjmi_stage1_shard_device2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_uint64, C.c_void_p, C.c_void_p]
        L.sjmi_set_auto_safe.restype = C.c_int
        L.sjmi_set_auto_safe.argtypes = [C.c_void_p, C.c_int]
        L.sjmi_set_tile_mode.restype = C.c_int
        L.sjmi_set_tile_mode.argtypes = [C.c_void_p, C.c_int]
        L.sjmi_stage1_masks.restype = C.c_int
        L.sjmi_stage1_masks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        L.sjmi_stage1_masks_device.restype = C.c_int
        L.sjmi_stage1_masks_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        for name, extra in (("sjmi_parser_root", [C.c_void_p]), ("sjmi_parser_batch_root", [C.c_uint64, C.c_void_p]),
                            ("sjmi_value_type", [C.c_void_p]), ("sjmi_value_as_long", [C.c_void_p, C.c_void_p]),
                            ("sjmi_value_as_double", [C.c_void_p, C.c_void_p]), ("sjmi_value_as_boolean", [C.c_void_p, C.c_void_p]),
                            ("sjmi_value_as_string", [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
                            ("sjmi_value_get", [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
                            ("sjmi_value_size", [C.c_void_p]), ("sjmi_value_first", [C.c_void_p, C.c_void_p]),
                            ("sjmi_value_next", [C.c_void_p, C.c_void_p, C.c_void_p])):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p] + extra
        L.sjmi_parser_set_gpu_walk.restype = C.c_int
        L.sjmi_parser_set_gpu_walk.argtypes = [C.c_void_p, C.c_int]
        P = C.c_void_p
        for name, args in (("sjmi_parser_ondemand_init", [P, P, C.c_uint64, C.c_int]), ("sjmi_od_skip_child", [P, C.c_int]),
                           ("sjmi_od_get_boolean", [P, C.c_int, C.c_int, P, P]), ("sjmi_od_get_long", [P, C.c_int, C.c_int, P, P]),
                           ("sjmi_od_get_double", [P, C.c_int, C.c_int, P, P]), ("sjmi_od_get_float", [P, C.c_int, C.c_int, P, P]), ("sjmi_od_get_char", [P, C.c_int, C.c_int, P, P]), ("sjmi_od_get_integral", [P, C.c_int, C.c_int, C.c_int, P, P]), ("sjmi_od_get_string", [P, C.c_int, P, P, P]),
                           ("sjmi_od_get_field_name", [P, P, P]), ("sjmi_od_start_array", [P, C.c_int, P]),
                           ("sjmi_od_next_array_element", [P, P]), ("sjmi_od_start_object", [P, C.c_int, P]),
                           ("sjmi_od_next_object_field", [P, P]), ("sjmi_od_move_to_field_value", [P]),
                           ("sjmi_od_assert_no_more_values", [P]), ("sjmi_od_depth", [P]), ("sjmi_od_peek", [P])):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = args
        L.sjmi_parse_document.restype = C.c_int
        L.sjmi_parse_document.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                          C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_parse_batch_device.restype = C.c_int
        L.sjmi_parse_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_parse_batch_device_optimistic.restype = C.c_int
        L.sjmi_parse_batch_device_optimistic.argtypes = L.sjmi_parse_batch_device.argtypes
        L.sjmi_parse_batch_device_rejected.restype = C.c_int
        L.sjmi_parse_batch_device_rejected.argtypes = L.sjmi_parse_batch_device.argtypes
        L.sjmi_kernel_time.restype = C.c_int
        L.sjmi_kernel_time.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class Context:
    """One sjmi_ctx: the engine-side state of one SimdJsonParser (SimdJsonParser.java:19-26)."""

    def __init__(self, device=0, capacity=34 * 1024 * 1024):
        self._h = C.c_void_p()
        rc = lib().sjmi_create(C.byref(self._h), device, capacity)
        if rc != 0:
            self._h = C.c_void_p()
            raise SjmiError("sjmi_create failed (rc=%d): no usable MI355X / HIP device; there is no CPU fallback" % rc)
        self.capacity = capacity
        _live.add(self)
        if os.environ.get("SJMI_TICKET_MODE") == "1":  # several processes on ONE GPU (bench.py's oversubscribed test mode): the FAST
            self.set_tile_mode(True)                   # kernel's whole grid cannot be resident beside another process's -- SAFE mode from the start
        if os.environ.get("SJMI_TILE_STEPS"):  # (experiments: the granule size of every stage-1 launch, sjmi_set_tile_steps)
            self.set_tile_steps(int(os.environ["SJMI_TILE_STEPS"]))

    def close(self):
        if self._h:
            lib().sjmi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if _exiting:
            return
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SjmiError("%s failed (rc=%d): %s" % (what, rc, lib().sjmi_last_error(self._h).decode()))

    def selftest(self):
        mm = C.c_uint32(1)
        self._check(lib().sjmi_selftest(self._h, C.addressof(mm)), "sjmi_selftest")
        return mm.value

    def set_tile_steps(self, steps):
        self._check(lib().sjmi_set_tile_steps(self._h, steps), "sjmi_set_tile_steps")

    def stage1(self, data, length=None, index_capacity=None, idx=None):
        """Host-buffer path (SimdJsonParser.stage1 + padIfNeeded): -> (indexes incl. sentinel stripped, status).
        idx: the caller's own np.uint32 index array (e.g. page-locked: the zero-copy path)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
        n = a.size if length is None else length
        cap = ((n + 2) if index_capacity is None else index_capacity) if idx is None else idx.size
        if idx is None:
            idx = np.empty(max(cap, 1), dtype=np.uint32)
        cnt = C.c_uint64(0)
        st = C.c_uint32(0)
        ptr = a.ctypes.data if a.size else None
        self._check(lib().sjmi_stage1(self._h, ptr, n, idx.ctypes.data, cap, C.addressof(cnt), C.addressof(st)),
                    "sjmi_stage1")
        assert idx[cnt.value] == 0, "sentinel missing"
        return idx[:cnt.value].copy(), st.value

    def parse_document(self, data, length=None, max_depth=1024):
        """sjmi_parse_document: all three stages on the GPU -> (tape np.uint64 or None, string buffer bytes, error code,
        stage-1 status)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        n = a.size if length is None else length
        tape = np.zeros(2 * n + 16, dtype=np.uint64)
        sb = np.zeros(n + 4 * (n // 2 + 2) + 64, dtype=np.uint8)
        tl, sl = C.c_uint64(0), C.c_uint64(0)
        err = C.c_int32(0)
        st = C.c_uint32(0)
        self._check(lib().sjmi_parse_document(self._h, a.ctypes.data if a.size else None, n, max_depth, tape.ctypes.data, tape.size,
                                              C.addressof(tl), sb.ctypes.data, sb.size, C.addressof(sl), C.addressof(err),
                                              C.addressof(st)), "sjmi_parse_document")
        return (tape[:tl.value].copy() if err.value == 0 else None), bytes(sb[:sl.value]), err.value, st.value

    def stage1_masks(self, data, length=None):
        """The reference's per-block masks (sjmi_stage1_masks): -> np.uint64 [len // 64 + 1, 6] =
        {escaped, quote, inString, op, whitespace, structurals} per 64-byte block."""
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
        n = a.size if length is None else length
        nb = n // 64 + 1
        masks = np.zeros((nb, 6), dtype=np.uint64)
        got = C.c_uint64(0)
        self._check(lib().sjmi_stage1_masks(self._h, a.ctypes.data if a.size else None, n, masks.ctypes.data, nb,
                                            C.addressof(got)), "sjmi_stage1_masks")
        assert got.value == nb
        return masks

    def stage1_masks_device(self, d_buf, length, d_masks, capacity_blocks, stream=0):
        self._check(lib().sjmi_stage1_masks_device(self._h, d_buf, length, d_masks, capacity_blocks, stream),
                    "sjmi_stage1_masks_device")

    def stage1_unescape(self, data, idx=None, sb=None):
        """Fused host path (sjmi_stage1_unescape): -> (indexes, status, string_buffer bytes, first_error_index, code)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = a.size
        if idx is None:
            idx = np.empty(n + 2, dtype=np.uint32)
        if sb is None:
            sb = np.empty(n + 4 * (n // 2 + 2) + 64, dtype=np.uint8)
        cnt, total, fei = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        st, fec = C.c_uint32(0), C.c_uint32(0)
        self._check(lib().sjmi_stage1_unescape(self._h, a.ctypes.data if n else None, n, idx.ctypes.data, idx.size,
                                               C.addressof(cnt), C.addressof(st), sb.ctypes.data, sb.size, C.addressof(total),
                                               C.addressof(fei), C.addressof(fec)), "sjmi_stage1_unescape")
        return idx[:cnt.value], st.value, sb[:total.value], (None if fei.value == 2**64 - 1 else fei.value), fec.value

    def stage1_batch(self, data, doc_offsets):
        """Batched host path: -> (indexes, index_offsets[np.uint64 n+1], status)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        offs = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        n = offs.size - 1
        cap = a.size + 2
        idx = np.empty(max(cap, 1), dtype=np.uint32)
        io = np.zeros(n + 1, dtype=np.uint64)
        cnt = C.c_uint64(0)
        st = C.c_uint32(0)
        self._check(lib().sjmi_stage1_batch(self._h, a.ctypes.data if a.size else None, a.size, offs.ctypes.data, n,
                                            idx.ctypes.data, cap, io.ctypes.data, C.addressof(cnt), C.addressof(st)),
                    "sjmi_stage1_batch")
        return idx[:cnt.value].copy(), io, st.value

    def stage1_batch_isolated(self, data, doc_offsets):
        """Isolated batched host path: -> (indexes, index_offsets[np.uint64 n+1], doc_status[np.uint32 n], status)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        offs = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        n = offs.size - 1
        cap = a.size + 2
        idx = np.empty(max(cap, 1), dtype=np.uint32)
        io = np.zeros(n + 1, dtype=np.uint64)
        ds = np.zeros(max(n, 1), dtype=np.uint32)
        cnt = C.c_uint64(0)
        st = C.c_uint32(0)
        self._check(lib().sjmi_stage1_batch_isolated(self._h, a.ctypes.data if a.size else None, a.size, offs.ctypes.data, n,
                                                     idx.ctypes.data, cap, io.ctypes.data, ds.ctypes.data, C.addressof(cnt),
                                                     C.addressof(st)), "sjmi_stage1_batch_isolated")
        assert idx[cnt.value] == 0, "sentinel missing"
        return idx[:cnt.value].copy(), io, ds[:n], st.value

    def unescape(self, string_capacity):
        """Unescape every string of the document of the last stage1() call.
        -> (string_buffer bytes, first_error_index or None, first_error_code)."""
        sb = np.empty(max(string_capacity, 1), dtype=np.uint8)
        total = C.c_uint64(0)
        fei = C.c_uint64(0)
        fec = C.c_uint32(0)
        self._check(lib().sjmi_unescape(self._h, sb.ctypes.data, string_capacity, C.addressof(total), C.addressof(fei),
                                        C.addressof(fec)), "sjmi_unescape")
        idx = None if fei.value == 0xFFFFFFFFFFFFFFFF else fei.value
        return sb[:total.value].tobytes(), idx, fec.value

    def unescape_batch(self, string_capacity, n_docs):
        """Unescape the strings of the batch of the last stage1_batch*() call.
        -> (string_buffer bytes, doc_string_offsets[np.uint64 n_docs+1], first_error_index or None, first_error_code)."""
        sb = np.empty(max(string_capacity, 1), dtype=np.uint8)
        dso = np.zeros(n_docs + 1, dtype=np.uint64)
        total = C.c_uint64(0)
        fei = C.c_uint64(0)
        fec = C.c_uint32(0)
        self._check(lib().sjmi_unescape_batch(self._h, sb.ctypes.data, string_capacity, dso.ctypes.data, C.addressof(total),
                                              C.addressof(fei), C.addressof(fec)), "sjmi_unescape_batch")
        idx = None if fei.value == 0xFFFFFFFFFFFFFFFF else fei.value
        return sb[:total.value].tobytes(), dso, idx, fec.value

    def unescape_device(self, d_buf, length, d_indexes, count, d_sb, sb_capacity, d_result, stream=0):
        self._check(lib().sjmi_unescape_device(self._h, d_buf, length, d_indexes, count, d_sb, sb_capacity, d_result,
                                               stream), "sjmi_unescape_device")

    def unescape_batch_device(self, d_buf, total_len, d_indexes, count, d_doc_offsets, d_index_offsets, n_docs, d_sb,
                              sb_capacity, d_doc_string_offsets, d_result, stream=0):
        self._check(lib().sjmi_unescape_batch_device(self._h, d_buf, total_len, d_indexes, count, d_doc_offsets, d_index_offsets,
                                                     n_docs, d_sb, sb_capacity, d_doc_string_offsets, d_result, stream),
                    "sjmi_unescape_batch_device")

    def walk_batch_device(self, d_buf, d_doc_offsets, n_docs, d_indexes, count, d_index_offsets, d_doc_status, d_sb,
                          d_doc_string_offsets, string_base, max_depth, d_tape, tape_capacity, d_tape_offsets, d_doc_errors,
                          d_result, stream=0):
        self._check(lib().sjmi_walk_batch_device(self._h, d_buf, d_doc_offsets, n_docs, d_indexes, count, d_index_offsets,
                                                 d_doc_status, d_sb, d_doc_string_offsets, string_base, max_depth, d_tape,
                                                 tape_capacity, d_tape_offsets, d_doc_errors, d_result, stream),
                    "sjmi_walk_batch_device")

    def parse_batch_device(self, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets,
                           d_doc_status, d_sb, sb_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity,
                           d_tape_offsets, d_doc_errors, d_result, stream=0):
        """sjmi_parse_batch_device: isolated stage 1 -> string records -> GPU walk, queued without a host round trip;
        d_result = device sjmi_batch_result (9 x int64: count, status | strings total, first_error_inv, flags |
        tape words, host documents, failed documents, flags)."""
        self._check(lib().sjmi_parse_batch_device(self._h, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity,
                                                  d_index_offsets, d_doc_status, d_sb, sb_capacity, d_doc_string_offsets,
                                                  max_depth, d_tape, tape_capacity, d_tape_offsets, d_doc_errors, d_result,
                                                  stream), "sjmi_parse_batch_device")

    def parse_batch_device_optimistic(self, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets,
                                      d_doc_status, d_sb, sb_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity,
                                      d_tape_offsets, d_doc_errors, d_result, stream=0):
        """sjmi_parse_batch_device_optimistic: ONLY the optimistic pipeline is queued (eight entries); stage1.status & 0x800
        (SJMI_ST_REJECTED) in d_result = nothing is valid, make the exact call (parse_batch_device)."""
        self._check(lib().sjmi_parse_batch_device_optimistic(self._h, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity,
                                                             d_index_offsets, d_doc_status, d_sb, sb_capacity, d_doc_string_offsets,
                                                             max_depth, d_tape, tape_capacity, d_tape_offsets, d_doc_errors, d_result,
                                                             stream), "sjmi_parse_batch_device_optimistic")

    def parse_batch_device_rejected(self, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets,
                                    d_doc_status, d_sb, sb_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity,
                                    d_tape_offsets, d_doc_errors, d_result, stream=0):
        """sjmi_parse_batch_device_rejected: the call behind SJMI_ST_REJECTED -- per-document verdicts, the failing documents
        blanked in a copy, the optimistic pipeline over the copy (repair); the per-document passes only for a batch that cannot
        take either.  Outputs as parse_batch_device."""
        self._check(lib().sjmi_parse_batch_device_rejected(self._h, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity,
                                                           d_index_offsets, d_doc_status, d_sb, sb_capacity, d_doc_string_offsets,
                                                           max_depth, d_tape, tape_capacity, d_tape_offsets, d_doc_errors, d_result,
                                                           stream), "sjmi_parse_batch_device_rejected")

    def stage1_batch_device(self, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets,
                            d_result, stream=0):
        self._check(lib().sjmi_stage1_batch_device(self._h, d_buf, total_len, d_doc_offsets, n_docs, d_indexes,
                                                   index_capacity, d_index_offsets, d_result, stream),
                    "sjmi_stage1_batch_device")

    def stage1_batch_isolated_device(self, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity,
                                     d_index_offsets, d_doc_status, d_result, stream=0):
        self._check(lib().sjmi_stage1_batch_isolated_device(self._h, d_buf, total_len, d_doc_offsets, n_docs, d_indexes,
                                                            index_capacity, d_index_offsets, d_doc_status, d_result,
                                                            stream), "sjmi_stage1_batch_isolated_device")

    def stage1_device(self, d_buf, length, d_indexes, index_capacity, d_result, stream=0):
        """Device-resident path; arguments are raw device pointers (ints) and a hipStream_t handle."""
        self._check(lib().sjmi_stage1_device(self._h, d_buf, length, d_indexes, index_capacity, d_result, stream),
                    "sjmi_stage1_device")

    def stage1_shard_device(self, d_buf, length, halo_bytes, is_last, entry_parity, d_indexes, index_capacity, d_result, stream=0,
                            halo_from_start=False):
        """One shard / stream chunk of a longer document (sjmi_stage1_shard_device2)."""
        self._check(lib().sjmi_stage1_shard_device2(self._h, d_buf, length, halo_bytes, 1 if halo_from_start else 0,
                                                    1 if is_last else 0, 1 if entry_parity else 0,
                                                    d_indexes, index_capacity, d_result, stream), "sjmi_stage1_shard_device2")

    def stream(self, max_chunk_bytes, halo_bytes=0):
        """sjmi_stream_open: one document as a stream of chunks -> a Stream (push(chunk, is_last), close())."""
        return Stream(self, max_chunk_bytes, halo_bytes)

    def split(self, d_shard, length, halo_bytes, halo_from_start, is_last, d_indexes, index_capacity):
        """sjmi_split_open: one rank's shard of a document split over GPUs -> a Split (scan(), resolve(entry_parity), close())."""
        return Split(self, d_shard, length, halo_bytes, halo_from_start, is_last, d_indexes, index_capacity)

    def set_auto_safe(self, on):
        self._check(lib().sjmi_set_auto_safe(self._h, 1 if on else 0), "sjmi_set_auto_safe")

    def set_tile_mode(self, ticket):
        self._check(lib().sjmi_set_tile_mode(self._h, 1 if ticket else 0), "sjmi_set_tile_mode")

    def debug_set_flags(self, flags):
        self._check(lib().sjmi_debug_set_flags(self._h, flags), "sjmi_debug_set_flags")

    def set_profiling(self, on):
        self._check(lib().sjmi_set_profiling(self._h, 1 if on else 0), "sjmi_set_profiling")

    def kernel_time(self):
        """-> (sum of stage-1 kernel durations in ms since profiling was enabled, number of launches)."""
        ms = C.c_double(0)
        n = C.c_uint32(0)
        self._check(lib().sjmi_kernel_time(self._h, C.addressof(ms), C.addressof(n)), "sjmi_kernel_time")
        return ms.value, n.value


class JsonParsingException(Exception):
    """org.simdjson.JsonParsingException (JsonParsingException.java:3-12); .code = SJMI_E_*."""

    def __init__(self, code, message, position=0):
        super().__init__(message)
        self.code = code
        self.position = position


class _Value(C.Structure):
    _fields_ = [("doc", C.c_uint64), ("tape_idx", C.c_uint64)]


class JsonValue:
    """org.simdjson.JsonValue (JsonValue.java:18-221) through the sjmi_value_* C ABI (which runs the C++ mirror class
    org_simdjson::JsonValue); valid until the next parse on its parser, like the reference's."""

    def __init__(self, parser, handle):
        self._p = parser
        self._v = handle

    def _call(self, fn, *args):
        rc = fn(self._p._h, C.byref(self._v), *args)
        if rc < 0:
            raise SjmiError("%s failed (rc=%d): wrong type or stale handle" % (fn.__name__, rc))
        return rc

    def type(self):
        return chr(self._call(lib().sjmi_value_type))

    def isArray(self):
        return self.type() == "["

    def isObject(self):
        return self.type() == "{"

    def isString(self):
        return self.type() == '"'

    def isLong(self):
        return self.type() == "l"

    def isDouble(self):
        return self.type() == "d"

    def isBoolean(self):
        return self.type() in "tf"

    def isNull(self):
        return self.type() == "n"

    def asLong(self):
        out = C.c_int64(0)
        self._call(lib().sjmi_value_as_long, C.byref(out))
        return out.value

    def asDouble(self):
        out = C.c_double(0)
        self._call(lib().sjmi_value_as_double, C.byref(out))
        return out.value

    def asBoolean(self):
        out = C.c_int(0)
        self._call(lib().sjmi_value_as_boolean, C.byref(out))
        return bool(out.value)

    def asString(self):
        n = C.c_uint64(0)
        buf = (C.c_uint8 * 256)()
        rc = lib().sjmi_value_as_string(self._p._h, C.byref(self._v), buf, 256, C.byref(n))
        if rc == -3:  # SJMI_ERR_CAPACITY: n holds the length
            buf = (C.c_uint8 * n.value)()
            rc = lib().sjmi_value_as_string(self._p._h, C.byref(self._v), buf, n.value, C.byref(n))
        if rc != 0:
            raise SjmiError("sjmi_value_as_string failed (rc=%d)" % rc)
        return bytes(buf[:n.value]).decode("utf-8")

    def get(self, name):
        """JsonValue.get(String): the field's value or None."""
        key = name.encode("utf-8")
        out = _Value()
        rc = self._call(lib().sjmi_value_get, key, len(key), C.byref(out))
        return None if rc == 1 else JsonValue(self._p, out)

    def getSize(self):
        return self._call(lib().sjmi_value_size)

    def _children(self):
        cur = _Value()
        rc = self._call(lib().sjmi_value_first, C.byref(cur))
        while rc == 0:
            yield JsonValue(self._p, cur)
            nxt = _Value()
            rc = lib().sjmi_value_next(self._p._h, C.byref(self._v), C.byref(cur), C.byref(nxt))
            if rc < 0:
                raise SjmiError("sjmi_value_next failed (rc=%d)" % rc)
            cur = nxt

    def arrayIterator(self):
        assert self.isArray()
        return self._children()

    def objectIterator(self):
        """-> (key str, JsonValue) pairs in document order."""
        assert self.isObject()
        it = self._children()
        for k in it:
            yield k.asString(), next(it)

    def to_python(self):
        """The whole subtree in the oracle's Parsed.to_python() notation (type tags, raw IEEE bits, UTF-8 bytes)."""
        t = self.type()
        if t == '"':
            return ("s", self.asString().encode("utf-8"))
        if t == "l":
            return ("l", self.asLong())
        if t == "d":
            import struct
            return ("d", struct.unpack("<Q", struct.pack("<d", self.asDouble()))[0])
        if t in "tfn":
            return (t,)
        if t == "[":
            return ("a", self.getSize(), [c.to_python() for c in self.arrayIterator()])
        return ("o", self.getSize(), [(k.encode("utf-8"), v.to_python()) for k, v in self.objectIterator()])


class ParsedDocument:
    """Tape + string buffer of one parse (views copied out of the parser)."""

    def __init__(self, tape, strings):
        self.tape = tape
        self.strings = strings


class SimdJsonParser:
    """Python handle on the C++ host mirror org_simdjson::SimdJsonParser (csrc/host/simdjson_parser.h):
    parse(buffer, len) = GPU stage 1 + GPU string unescape + host stage 2 (SimdJsonParser.java:35-40)."""

    DEFAULT_CAPACITY = 34 * 1024 * 1024
    DEFAULT_MAX_DEPTH = 1024

    def __init__(self, capacity=DEFAULT_CAPACITY, max_depth=DEFAULT_MAX_DEPTH, device=0, gpu_walk=None):
        self._h = C.c_void_p()
        rc = lib().sjmi_parser_create(C.byref(self._h), capacity, max_depth, device)
        if rc != 0:
            self._h = C.c_void_p()
            raise SjmiError("sjmi_parser_create failed (rc=%d): no usable MI355X; there is no CPU fallback" % rc)
        if gpu_walk is not None:  # stage 2 on the GPU (True) / on the host (False); None: the library's default, by size
            lib().sjmi_parser_set_gpu_walk(self._h, 1 if gpu_walk else 0)
        _live.add(self)

    def close(self):
        if self._h:
            lib().sjmi_parser_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if _exiting:
            return
        try:
            self.close()
        except Exception:
            pass

    def parse(self, buffer, length=None):
        a = np.frombuffer(bytes(buffer), dtype=np.uint8)
        n = a.size if length is None else length
        tape_p = C.POINTER(C.c_uint64)()
        sb_p = C.POINTER(C.c_uint8)()
        tape_len = C.c_uint64(0)
        sb_len = C.c_uint64(0)
        err_pos = C.c_uint64(0)
        rc = lib().sjmi_parser_parse(self._h, a.ctypes.data if a.size else None, n, C.byref(tape_p), C.byref(tape_len),
                                     C.byref(sb_p), C.byref(sb_len), C.byref(err_pos))
        if rc > 0:
            raise JsonParsingException(rc, lib().sjmi_parser_last_message(self._h).decode("utf-8"), err_pos.value)
        if rc < 0:
            raise SjmiError("sjmi_parser_parse failed (rc=%d): %s" % (rc, lib().sjmi_parser_last_message(self._h).decode()))
        tape = np.ctypeslib.as_array(tape_p, shape=(tape_len.value,)).copy()
        strings = bytes(np.ctypeslib.as_array(sb_p, shape=(max(sb_len.value, 1),))[:sb_len.value])
        return ParsedDocument(tape, strings)

    def ondemand(self, buffer, length=None):
        """The on-demand front end (OnDemandJsonIterator.java; csrc/host/ondemand.h): pad + GPU stage 1 + iterator.init -> the
        cursor.  One cursor per parser at a time, invalidated by the next parse / ondemand."""
        a = np.frombuffer(bytes(buffer), dtype=np.uint8)
        n = a.size if length is None else length
        rc = lib().sjmi_parser_ondemand_init(self._h, a.ctypes.data if a.size else None, n, 0)
        if rc > 0:
            raise JsonParsingException(rc, lib().sjmi_parser_last_message(self._h).decode("utf-8"))
        if rc < 0:
            raise SjmiError("sjmi_parser_ondemand_init failed (rc=%d): %s" % (rc, lib().sjmi_parser_last_message(self._h).decode()))
        return OnDemandIterator(self)

    def root(self):
        """JsonValue of the last parse() (what SimdJsonParser.parse returns in the reference)."""
        v = _Value()
        rc = lib().sjmi_parser_root(self._h, C.byref(v))
        if rc != 0:
            raise SjmiError("sjmi_parser_root failed (rc=%d): no successful parse on this parser" % rc)
        return JsonValue(self, v)

    def batch_root(self, doc):
        """JsonValue of document `doc` of the last parse_batch(); raises JsonParsingException with that document's code."""
        v = _Value()
        rc = lib().sjmi_parser_batch_root(self._h, doc, C.byref(v))
        if rc > 0:
            raise JsonParsingException(rc, "document %d of the batch failed with code %d" % (doc, rc))
        if rc < 0:
            raise SjmiError("sjmi_parser_batch_root failed (rc=%d)" % rc)
        return JsonValue(self, v)

    def parse_batch(self, buffer, doc_offsets):
        """-> (list of per-document tapes (np.uint64) or None where errors[k] != 0, shared strings bytes, errors)."""
        a = np.frombuffer(bytes(buffer), dtype=np.uint8)
        offs = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        n = offs.size - 1
        tape_p = C.POINTER(C.c_uint64)()
        to_p = C.POINTER(C.c_uint64)()
        sb_p = C.POINTER(C.c_uint8)()
        err_p = C.POINTER(C.c_int32)()
        sb_len = C.c_uint64(0)
        rc = lib().sjmi_parser_parse_batch(self._h, a.ctypes.data if a.size else None, a.size, offs.ctypes.data, n,
                                           C.byref(tape_p), C.byref(to_p), C.byref(sb_p), C.byref(sb_len), C.byref(err_p))
        if rc > 0:
            raise JsonParsingException(rc, lib().sjmi_parser_last_message(self._h).decode("utf-8"))
        if rc < 0:
            raise SjmiError("sjmi_parser_parse_batch failed (rc=%d): %s" % (rc, lib().sjmi_parser_last_message(self._h).decode()))
        def view(ptr, count, dtype):  # (an empty result may come back as a NULL pointer)
            return np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count and bool(ptr) else np.zeros(0, dtype=dtype)
        to = np.ctypeslib.as_array(to_p, shape=(n + 1,)).copy()
        errors = view(err_p, n, np.int32)
        alltape = view(tape_p, int(to[-1]), np.uint64)
        strings = bytes(view(sb_p, sb_len.value, np.uint8))
        tapes = [alltape[int(to[k]):int(to[k + 1])] if errors[k] == 0 else None for k in range(n)]
        return tapes, strings, errors


class Stream:
    """sjmi_stream_*: chunks of one document, scanned once each; push -> (base offset, indexes relative to the chunk, status so far)"""

    def __init__(self, ctx, max_chunk_bytes, halo_bytes=0):
        L = lib()
        L.sjmi_stream_open.restype = C.c_int
        L.sjmi_stream_open.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
        L.sjmi_stream_push.restype = C.c_int
        L.sjmi_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_stream_close.restype = None
        L.sjmi_stream_close.argtypes = [C.c_void_p]
        self._ctx, self._max = ctx, int(max_chunk_bytes)
        h = C.c_void_p()
        ctx._check(L.sjmi_stream_open(ctx._h, self._max, halo_bytes, C.byref(h)), "sjmi_stream_open")
        self._h = h
        self._idx = np.empty(self._max + 70, dtype=np.uint32)

    def push(self, chunk, is_last):
        a = np.frombuffer(bytes(chunk), dtype=np.uint8)
        count, base, st = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        self._ctx._check(lib().sjmi_stream_push(self._h, a.ctypes.data if a.size else None, a.size, 1 if is_last else 0, self._idx.ctypes.data,
                                                self._idx.size, C.byref(count), C.byref(base), C.byref(st)), "sjmi_stream_push")
        return base.value, self._idx[:count.value].copy(), st.value

    def close(self):
        if self._h:
            lib().sjmi_stream_close(self._h)
            self._h = None


class Split:
    """sjmi_split_*: scan() -> (flips the parity?, status bits); resolve(entry_parity) -> (count, status bits, parity behind the shard)"""

    def __init__(self, ctx, d_shard, length, halo_bytes, halo_from_start, is_last, d_indexes, index_capacity):
        L = lib()
        L.sjmi_split_open.restype = C.c_int
        L.sjmi_split_open.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.sjmi_split_scan.restype = C.c_int
        L.sjmi_split_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_split_resolve.restype = C.c_int
        L.sjmi_split_resolve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sjmi_split_close.restype = None
        L.sjmi_split_close.argtypes = [C.c_void_p]
        self._ctx = ctx
        h = C.c_void_p()
        ctx._check(L.sjmi_split_open(ctx._h, d_shard, length, halo_bytes, 1 if halo_from_start else 0, 1 if is_last else 0, d_indexes,
                                     index_capacity, C.byref(h)), "sjmi_split_open")
        self._h = h

    def scan(self, stream=0):
        flips, st = C.c_int(0), C.c_uint32(0)
        self._ctx._check(lib().sjmi_split_scan(self._h, stream, C.byref(flips), C.byref(st)), "sjmi_split_scan")
        return flips.value, st.value

    def resolve(self, entry_parity, stream=0):
        count, st, after = C.c_uint64(0), C.c_uint32(0), C.c_int(0)
        self._ctx._check(lib().sjmi_split_resolve(self._h, 1 if entry_parity else 0, stream, C.byref(count), C.byref(st), C.byref(after)),
                         "sjmi_split_resolve")
        return count.value, st.value, after.value

    def close(self):
        if self._h:
            lib().sjmi_split_close(self._h)
            self._h = None


class OnDemandIterator:
    """Python face of the sjmi_od_* entry points: the methods of the reference's OnDemandJsonIterator (names as in
    oracle/ondemand.py, the checker).  Nullable getters return None where the reference returns null."""
    EMPTY, NULL, NOT_EMPTY = 0, 1, 2

    def __init__(self, parser):
        self._p = parser

    def _check(self, rc):
        if rc > 0:
            raise JsonParsingException(rc, lib().sjmi_parser_last_message(self._p._h).decode("utf-8"))
        if rc < 0:
            raise SjmiError("on-demand call failed (rc=%d): %s" % (rc, lib().sjmi_parser_last_message(self._p._h).decode()))

    def depth_value(self):
        return lib().sjmi_od_depth(self._p._h)

    def peek_byte(self):
        return lib().sjmi_od_peek(self._p._h)

    def skip_child(self, parent_depth=None):
        self._check(lib().sjmi_od_skip_child(self._p._h, -1 if parent_depth is None else parent_depth))

    def get_boolean(self, root=False, nullable=True):
        n, v = C.c_int(0), C.c_int(0)
        self._check(lib().sjmi_od_get_boolean(self._p._h, int(root), int(nullable), C.addressof(n), C.addressof(v)))
        return None if n.value else bool(v.value)

    def get_long(self, root=False, nullable=True, bits=64):
        n, v = C.c_int(0), C.c_int64(0)
        if bits == 64:
            self._check(lib().sjmi_od_get_long(self._p._h, int(root), int(nullable), C.addressof(n), C.addressof(v)))
        else:
            self._check(lib().sjmi_od_get_integral(self._p._h, bits, int(root), int(nullable), C.addressof(n), C.addressof(v)))
        return None if n.value else v.value

    def get_double(self, root=False, nullable=True):
        n, v = C.c_int(0), C.c_double(0)
        self._check(lib().sjmi_od_get_double(self._p._h, int(root), int(nullable), C.addressof(n), C.addressof(v)))
        return None if n.value else v.value

    def get_float(self, root=False, nullable=True):
        n, v = C.c_int(0), C.c_float(0)
        self._check(lib().sjmi_od_get_float(self._p._h, int(root), int(nullable), C.addressof(n), C.addressof(v)))
        return None if n.value else v.value

    def get_char(self, root=False, nullable=True):
        n, v = C.c_int(0), C.c_uint16(0)
        self._check(lib().sjmi_od_get_char(self._p._h, int(root), int(nullable), C.addressof(n), C.addressof(v)))
        return None if n.value else v.value

    def _bytes(self, ptr, ln):
        return C.string_at(ptr.value, ln.value) if ln.value else b""

    def get_string(self, root=False):
        n, ptr, ln = C.c_int(0), C.c_void_p(), C.c_uint64(0)
        self._check(lib().sjmi_od_get_string(self._p._h, int(root), C.addressof(n), C.addressof(ptr), C.addressof(ln)))
        return None if n.value else self._bytes(ptr, ln)

    def get_field_name(self):
        ptr, ln = C.c_void_p(), C.c_uint64(0)
        self._check(lib().sjmi_od_get_field_name(self._p._h, C.addressof(ptr), C.addressof(ln)))
        return self._bytes(ptr, ln)

    def start_iterating_array(self, root=False):
        r = C.c_int(0)
        self._check(lib().sjmi_od_start_array(self._p._h, int(root), C.addressof(r)))
        return r.value

    def next_array_element(self):
        r = C.c_int(0)
        self._check(lib().sjmi_od_next_array_element(self._p._h, C.addressof(r)))
        return bool(r.value)

    def start_iterating_object(self, root=False):
        r = C.c_int(0)
        self._check(lib().sjmi_od_start_object(self._p._h, int(root), C.addressof(r)))
        return r.value

    def next_object_field(self):
        r = C.c_int(0)
        self._check(lib().sjmi_od_next_object_field(self._p._h, C.addressof(r)))
        return bool(r.value)

    def move_to_field_value(self):
        self._check(lib().sjmi_od_move_to_field_value(self._p._h))

    def assert_no_more_json_values(self):
        self._check(lib().sjmi_od_assert_no_more_values(self._p._h))
