"""ctypes binding of the product library rawspeed_amd/librsx.so (include/rsx.h).

The library is the product; this module only marshals arguments.  Loading fails
loudly if the library is missing, and every decode entry point fails with
RSX_ERR_DEVICE when there is no GPU -- there is no CPU fallback anywhere.
"""
import ctypes as C
import os

import numpy as np

from . import abi, build

_lib = None

# every symbol include/rsx.h declares (tests check that all of them resolve)
EXPORTS = [
    "rsx_abi_version", "rsx_status_string", "rsx_device_count", "rsx_ctx_create",
    "rsx_ctx_destroy", "rsx_ctx_last_error", "rsx_ctx_host_calls", "rsx_ctx_chunked_calls",
    "rsx_host_alloc", "rsx_host_free", "rsx_host_register", "rsx_host_unregister",
    "rsx_unpack_validate", "rsx_unpack_u16",
    "rsx_unpack_f32_validate", "rsx_unpack_f32", "rsx_unpack_f32_plan_create",
    "rsx_unpack_variant_validate", "rsx_unpack_variant_u16", "rsx_unpack_variant_plan_create",
    "rsx_ljpeg_validate", "rsx_ljpeg_decode",
    "rsx_cr2_validate", "rsx_cr2_decode",
    "rsx_sraw_validate", "rsx_sraw_interpolate", "rsx_sraw_plan_create",
    "rsx_nikon_validate", "rsx_nikon_decompress", "rsx_nikon_plan_create",
    "rsx_pentax_validate", "rsx_pentax_decompress", "rsx_pentax_plan_create",
    "rsx_hasselblad_validate", "rsx_hasselblad_decompress", "rsx_hasselblad_plan_create",
    "rsx_samsung_v1_validate", "rsx_samsung_v1_decompress", "rsx_samsung_v1_plan_create",
    "rsx_samsung_v2_validate", "rsx_samsung_v2_decompress", "rsx_samsung_v2_plan_create",
    "rsx_sony_arw1_validate", "rsx_sony_arw1_decompress", "rsx_sony_arw1_plan_create",
    "rsx_dng_decompress_ljpeg", "rsx_dng_decompress_uncompressed",
    "rsx_unpack_plan_create", "rsx_ljpeg_plan_create", "rsx_cr2_plan_create",
    "rsx_plan_run", "rsx_plan_results", "rsx_plan_set_timing",
    "rsx_plan_kernel_time", "rsx_plan_kernel_table", "rsx_plan_destroy", "rsx_probe_stream_copy",
]


class RsxError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        super().__init__("%s (%d) %s" % (abi.STATUS_NAMES.get(status, "?"), status, what))


def lib():
    global _lib
    if _lib is None:
        # RSX_LIB: A/B experiments with alternative builds of the same library
        path = os.environ.get("RSX_LIB") or build.LIB_CORE
        if not os.path.exists(path):
            raise RuntimeError(
                "rawspeed_amd/librsx.so is not built: run `python -m rawspeed_amd.build` "
                "(or __graft_entry__.build()); there is no fallback implementation")
        # PyTorch bundles its own libamdhip64; two HIP runtimes in one process do
        # not coexist ("No HIP GPUs are available").  Loading torch's first makes
        # librsx.so's DT_NEEDED libamdhip64 resolve to the same, single runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        L.rsx_status_string.restype = C.c_char_p
        L.rsx_ctx_last_error.restype = C.c_char_p
        L.rsx_ctx_last_error.argtypes = [C.c_void_p]
        L.rsx_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rsx_ctx_destroy.argtypes = [C.c_void_p]
        L.rsx_unpack_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rsx_ljpeg_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rsx_cr2_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rsx_unpack_u16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]
        L.rsx_unpack_f32_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rsx_unpack_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]
        L.rsx_unpack_variant_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rsx_unpack_variant_u16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p]
        L.rsx_ljpeg_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_void_p]
        L.rsx_cr2_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p]
        L.rsx_sraw_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rsx_sraw_interpolate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rsx_nikon_validate.argtypes = [C.c_void_p, C.c_void_p]
        L.rsx_pentax_validate.argtypes = [C.c_void_p, C.c_void_p]
        L.rsx_hasselblad_validate.argtypes = [C.c_void_p, C.c_void_p]
        L.rsx_hasselblad_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p, C.c_void_p]
        L.rsx_sony_arw1_validate.argtypes = [C.c_void_p]
        L.rsx_sony_arw1_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_void_p]
        L.rsx_samsung_v1_validate.argtypes = [C.c_void_p, C.c_void_p]
        L.rsx_samsung_v2_validate.argtypes = [C.c_void_p, C.c_void_p]
        L.rsx_samsung_v2_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p]
        L.rsx_samsung_v1_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p]
        L.rsx_pentax_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_size_t, C.c_void_p]
        L.rsx_nikon_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]
        L.rsx_dng_decompress_ljpeg.argtypes = [C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p]
        L.rsx_dng_decompress_uncompressed.argtypes = [C.c_void_p, C.c_int, C.c_void_p,
                                                      C.c_void_p, C.c_void_p]
        for name in ("rsx_unpack_plan_create", "rsx_ljpeg_plan_create",
                     "rsx_cr2_plan_create", "rsx_unpack_variant_plan_create",
                     "rsx_nikon_plan_create", "rsx_unpack_f32_plan_create",
                     "rsx_pentax_plan_create", "rsx_samsung_v1_plan_create",
                     "rsx_samsung_v2_plan_create",
                     "rsx_sraw_plan_create", "rsx_hasselblad_plan_create",
                     "rsx_sony_arw1_plan_create"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p,
                                         C.POINTER(C.c_void_p)]
        L.rsx_plan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rsx_plan_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rsx_plan_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.rsx_plan_kernel_table.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p),
                                            C.POINTER(C.c_double), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int)]
        L.rsx_plan_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.rsx_plan_destroy.argtypes = [C.c_void_p]
        L.rsx_probe_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                            C.c_size_t, C.c_void_p, C.c_int,
                                            C.POINTER(C.c_double)]
        _lib = L
    return _lib


def status_string(st):
    return lib().rsx_status_string(st).decode()


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


class Context:
    """rsx_ctx: one per device."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        st = lib().rsx_ctx_create(device, C.byref(self._h))
        if st != abi.RSX_OK:
            raise RsxError(st, "rsx_ctx_create(device=%d): no usable GPU" % device)
        self.device = device

    def last_error(self):
        return lib().rsx_ctx_last_error(self._h).decode(errors="replace")

    def close(self):
        if self._h:
            lib().rsx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def host_calls(self):
        """host-pointer entry points this context has served (rsx_ctx_host_calls)"""
        f = lib().rsx_ctx_host_calls
        f.restype = C.c_uint64
        return int(f(self._h))

    def chunked_calls(self):
        """... of which ran one large stream in chunks (rsx_ctx_chunked_calls)"""
        f = lib().rsx_ctx_chunked_calls
        f.restype = C.c_uint64
        return int(f(self._h))

    # page-locked host memory (rsx.h: optional, ABI 4)
    def host_alloc(self, nbytes):
        """rsx_host_alloc: a page-locked block as a numpy uint8 array (host_free() it)"""
        import numpy as np
        p = C.c_void_p()
        st = lib().rsx_host_alloc(self._h, C.c_size_t(nbytes), C.byref(p))
        if st != 0:
            raise RsxError(st, "rsx_host_alloc(%d)" % nbytes)
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8)
        arr.flags.writeable = True
        return arr

    def host_free(self, arr):
        return lib().rsx_host_free(self._h, C.c_void_p(arr.ctypes.data))

    def host_register(self, arr):
        """rsx_host_register: page-lock an existing numpy array in place (status)"""
        return lib().rsx_host_register(self._h, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes))

    def host_unregister(self, arr):
        return lib().rsx_host_unregister(self._h, C.c_void_p(arr.ctypes.data))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def probe_stream_copy(self, in_ptr, in_bytes, out_ptr, out_bytes, stream=None, reps=20):
        """Average ms of a plain streaming kernel moving in_bytes in / out_bytes out
        (device pointers): the copy ceiling bench.py quotes next to the vendor peak."""
        ms = C.c_double(0)
        st = lib().rsx_probe_stream_copy(self._h, C.c_void_p(in_ptr), in_bytes,
                                         C.c_void_p(out_ptr), out_bytes,
                                         C.c_void_p(stream or 0), reps, C.byref(ms))
        if st != abi.RSX_OK:
            raise RsxError(st, self.last_error())
        return ms.value

    # ---- host-pointer calls (what the patched reference methods call) ------
    def unpack_u16(self, desc, data, img_view):
        a = _u8(data)
        return lib().rsx_unpack_u16(self._h, C.byref(desc), a.ctypes.data, a.size,
                                    C.byref(img_view))

    def unpack_f32(self, desc, data, img_view):
        a = _u8(data)
        return lib().rsx_unpack_f32(self._h, C.byref(desc), a.ctypes.data, a.size,
                                    C.byref(img_view))

    def unpack_variant_u16(self, desc, data, img_view):
        a = _u8(data)
        return lib().rsx_unpack_variant_u16(self._h, C.byref(desc), a.ctypes.data, a.size,
                                            C.byref(img_view))

    def ljpeg_decode(self, desc, data, img_view):
        a = _u8(data)
        consumed = C.c_uint32(0)
        st = lib().rsx_ljpeg_decode(self._h, C.byref(desc), a.ctypes.data, a.size,
                                    C.byref(img_view), C.byref(consumed))
        return st, consumed.value

    def cr2_decode(self, desc, data, img_view):
        a = _u8(data)
        consumed = C.c_uint32(0)
        st = lib().rsx_cr2_decode(self._h, C.byref(desc), a.ctypes.data, a.size,
                                  C.byref(img_view), C.byref(consumed))
        return st, consumed.value

    def sraw_interpolate(self, desc, in_view, out_view):
        return lib().rsx_sraw_interpolate(self._h, C.byref(desc), C.byref(in_view),
                                          C.byref(out_view))

    def nikon_decompress(self, desc, data, img_view):
        a = _u8(data)
        return lib().rsx_nikon_decompress(self._h, C.byref(desc), a.ctypes.data, a.size,
                                          C.byref(img_view))

    def pentax_decompress(self, desc, data, img_view):
        a = _u8(data)
        return lib().rsx_pentax_decompress(self._h, C.byref(desc), a.ctypes.data, a.size,
                                           C.byref(img_view))

    def hasselblad_decompress(self, desc, data, img_view):
        a = _u8(data)
        consumed = C.c_uint32(0)
        st = lib().rsx_hasselblad_decompress(self._h, C.byref(desc), a.ctypes.data, a.size,
                                             C.byref(img_view), C.byref(consumed))
        return st, consumed.value

    def samsung_v1_decompress(self, desc, data, img_view):
        a = _u8(data)
        return lib().rsx_samsung_v1_decompress(self._h, C.byref(desc), a.ctypes.data,
                                               a.size, C.byref(img_view))

    def samsung_v2_decompress(self, desc, data, img_view):
        a = _u8(data)
        return lib().rsx_samsung_v2_decompress(self._h, C.byref(desc), a.ctypes.data,
                                               a.size, C.byref(img_view))

    def sony_arw1_decompress(self, data, img_view):
        a = _u8(data)
        return lib().rsx_sony_arw1_decompress(self._h, a.ctypes.data, a.size,
                                              C.byref(img_view))

    def dng_decompress_ljpeg(self, descs, datas, img_view):
        n = len(descs)
        arrs = [_u8(d) for d in datas]
        tiles = (abi.DngLJpegTile * n)()
        for i in range(n):
            tiles[i].desc = descs[i]
            tiles[i].in_ = arrs[i].ctypes.data
            tiles[i].in_bytes = arrs[i].size
        st = (C.c_int32 * n)()
        cons = (C.c_uint32 * n)()
        rc = lib().rsx_dng_decompress_ljpeg(self._h, n, tiles, C.byref(img_view), st,
                                            cons)
        return rc, list(st), list(cons)

    def dng_decompress_uncompressed(self, descs, datas, img_view):
        n = len(descs)
        arrs = [_u8(d) for d in datas]
        tiles = (abi.DngUnpackTile * n)()
        for i in range(n):
            tiles[i].desc = descs[i]
            tiles[i].in_ = arrs[i].ctypes.data
            tiles[i].in_bytes = arrs[i].size
        st = (C.c_int32 * n)()
        rc = lib().rsx_dng_decompress_uncompressed(self._h, n, tiles,
                                                   C.byref(img_view), st)
        return rc, list(st)

    # ---- device-resident plans ---------------------------------------------
    def unpack_plan(self, jobs):
        return Plan(self, "rsx_unpack_plan_create", abi.UnpackJob, jobs)

    def unpack_f32_plan(self, jobs):
        return Plan(self, "rsx_unpack_f32_plan_create", abi.UnpackJob, jobs)

    def unpack_variant_plan(self, jobs):
        return Plan(self, "rsx_unpack_variant_plan_create", abi.UnpackVariantJob, jobs)

    def ljpeg_plan(self, jobs):
        return Plan(self, "rsx_ljpeg_plan_create", abi.LJpegJob, jobs)

    def cr2_plan(self, jobs):
        return Plan(self, "rsx_cr2_plan_create", abi.Cr2Job, jobs)

    def hasselblad_plan(self, jobs):
        return Plan(self, "rsx_hasselblad_plan_create", abi.HasselbladJob, jobs)

    def samsung_v2_plan(self, jobs):
        return Plan(self, "rsx_samsung_v2_plan_create", abi.SamsungV2Job, jobs)

    def samsung_v1_plan(self, jobs):
        return Plan(self, "rsx_samsung_v1_plan_create", abi.SamsungV1Job, jobs)

    def sony_arw1_plan(self, jobs):
        return Plan(self, "rsx_sony_arw1_plan_create", abi.SonyArw1Job, jobs)

    def pentax_plan(self, jobs):
        return Plan(self, "rsx_pentax_plan_create", abi.PentaxJob, jobs)

    def sraw_plan(self, jobs):
        return Plan(self, "rsx_sraw_plan_create", abi.SrawJob, jobs)

    def nikon_plan(self, jobs):
        return Plan(self, "rsx_nikon_plan_create", abi.NikonJob, jobs)


class Plan:
    def __init__(self, ctx, create_fn, job_type, jobs):
        self.ctx = ctx
        self.n = len(jobs)
        arr = (job_type * self.n)(*jobs)
        self._h = C.c_void_p()
        st = getattr(lib(), create_fn)(ctx._h, self.n, arr, C.byref(self._h))
        if st != abi.RSX_OK:
            raise RsxError(st, ctx.last_error())

    def run(self, in_ptr, out_ptr, stream=None):
        st = lib().rsx_plan_run(self._h, C.c_void_p(in_ptr), C.c_void_p(out_ptr),
                                C.c_void_p(stream or 0))
        if st != abi.RSX_OK:
            raise RsxError(st, self.ctx.last_error())

    def results(self):
        st = (C.c_int32 * self.n)()
        cons = (C.c_uint32 * self.n)()
        rc = lib().rsx_plan_results(self._h, st, cons)
        return rc, list(st), list(cons)

    def set_timing(self, on=True):
        lib().rsx_plan_set_timing(self._h, 1 if on else 0)

    def kernel_time(self):
        name = C.c_char_p()
        ms = C.c_double(0)
        n = C.c_int(0)
        st = lib().rsx_plan_kernel_time(self._h, C.byref(name), C.byref(ms),
                                        C.byref(n))
        if st != abi.RSX_OK:
            return None
        return name.value.decode(), ms.value, n.value

    def kernel_table(self, cap=64):
        """[(kernel name, average ms per run)] of the timed runs so far, and the run count
        (LJPEG-family plans; call before kernel_time(), which resets the totals)."""
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        n = C.c_int(0)
        runs = C.c_int(0)
        st = lib().rsx_plan_kernel_table(self._h, cap, names, ms, C.byref(n), C.byref(runs))
        if st != abi.RSX_OK:
            return None
        return [(names[i].decode(), ms[i]) for i in range(min(cap, n.value))], runs.value

    def close(self):
        if self._h:
            lib().rsx_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
