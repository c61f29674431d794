"""Whole synthetic raw FILES through the reference's own front door --
RawParser::getDecoder() -> DngDecoder / ArwDecoder / Cr2Decoder -> RawDecoder::decodeRaw()
(parsers/RawParser.cpp:45-98, decoders/RawDecoder.cpp:320-340) -- the callers on the
host side of the hot path ("the callers ... either side of the path").

Here (no GPU): the UNMODIFIED reference build decodes every file to the image it was
made from -- which pins the file writer (tests/rawfiles.py) and the full-library
build of oracle/_ref -- and the PATCHED build, which finds no device, produces
the same bytes through its fall-through to the original loops ("plumbing, no GPU":
BASELINE configs[0]).  tests/test_gpu_raw_files.py runs the same files with the GPU."""
import numpy as np
import pytest

import raw_file_cases as F
from oracle_lib import REF_RSX_SO, Ref


@pytest.fixture(scope="module")
def ref():
    if not Ref.available():
        pytest.skip("oracle/_ref/librawspeed_ref.so absent")
    r = Ref()
    if not hasattr(r.lib, "ref_decode_file"):
        pytest.skip("oracle/_ref predates the whole-file entry point")
    return r


@pytest.mark.parametrize("name", sorted(F.CASES))
def test_reference_decodes_synthetic_file(ref, name):
    blob, want = F.CASES[name]()
    st, img = ref.decode_file(blob, uncorrected=name in F.UNCORRECTED, threads=2)
    assert st == 0, ref.last_error()
    assert img.errors() == ""
    if want is not None:
        assert (img.full_w * img.cpp, img.full_h) == (want.shape[1], want.shape[0])
        assert np.array_equal(img.u16(), want)


@pytest.mark.parametrize("name", sorted(F.CASES))
def test_patched_build_without_device_falls_through(ref, name):
    """The GPU-backed build on a machine without a GPU: rsx_shim::context() is null,
    every hunk falls through to the original body."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu tests")
    if not Ref.available(REF_RSX_SO):
        pytest.skip("oracle/_ref/librawspeed_rsx.so absent")
    rsx = Ref(REF_RSX_SO)
    blob, want = F.CASES[name]()
    st, img = rsx.decode_file(blob, uncorrected=name in F.UNCORRECTED, threads=2)
    assert st == 0, rsx.last_error()
    if want is None:  # (the unmodified build is the expectation)
        st0, img0 = ref.decode_file(blob, uncorrected=name in F.UNCORRECTED, threads=2)
        assert st0 == 0
        want = img0.u16().copy()
    assert np.array_equal(img.u16(), want)
    assert rsx.lib.ref_rsx_host_calls() == 0


def test_corrupt_file_reports_the_reference_error(ref):
    blob, _ = F.CASES["dng_ljpeg_tiles"]()
    blob = blob.copy()
    blob[4:8] = 0xFF  # IFD offset outside the file
    st, img = ref.decode_file(blob)
    assert st != 0 and img is None
