"""SamsungV2Decompressor on the device (rsx_samsung_v2_*, rawspeed_amd/csrc/rsx_samsung_v2.hip)
through the C-ABI against the oracle's restatement -- which tests/test_oracle_samsung_v2.py
pins against the reference build: streams of the writer in samsung_v2_cases.py (every block
mode, scale changes, all eight optimisation-flag combinations, both bit depths), damaged
ones (same status), batches, and a frame at the constructor's size limit."""
import numpy as np
import pytest
import torch

from rawspeed_amd import abi

import samsung_v2_cases as V2
from oracle_lib import HostImage
from test_oracle_samsung_v2 import _target

pytestmark = pytest.mark.gpu

INVALID_ARG = 1


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def open_stream(data, bits, w, h):
    """What SamsungV2Decompressor's constructor does with the strip (cpp:87-141): status of
    its checks, the descriptor, the member `data`."""
    if data.size < 16:
        return 2, None, None  # bs.check(headerSize)
    d, flags = abi.SamsungV2Desc.from_header(data[:16])
    if d.bit_depth != bits:
        return INVALID_ARG, None, None
    if flags > 7:
        return INVALID_ARG, None, None
    return 0, d, data[16:]


def decode(gpu, data, bits, w, h):
    img = HostImage(w, h)
    st, d, payload = open_stream(np.asarray(data, np.uint8), bits, w, h)
    if st:
        return st, img
    return gpu.samsung_v2_decompress(d, payload, img.view()), img


@pytest.mark.parametrize("optflags", range(8))
@pytest.mark.parametrize("bits", [12, 14])
def test_writer_streams(gpu, oracle, optflags, bits):
    rng = np.random.default_rng([190, optflags, bits])
    for trial in range(3):
        h, w = int(rng.integers(2, 70)), 16 * int(rng.integers(1, 30))
        data, want = V2.encode(rng, _target(rng, h, w, bits), bits, optflags)
        host = HostImage(w, h)
        assert oracle.samsung_v2(bits, data, host) == 0
        st, img = decode(gpu, data, bits, w, h)
        assert st == 0
        assert np.array_equal(img.pixels(), host.pixels()), (trial, h, w)
        assert np.array_equal(img.pixels(), want)


@pytest.mark.parametrize("seed", range(60))
def test_damaged_streams_same_verdict(gpu, oracle, seed):
    rng = np.random.default_rng([191, seed])
    bits = int(rng.choice([12, 14]))
    h, w = int(rng.integers(2, 40)), 16 * int(rng.integers(1, 12))
    data, _ = V2.encode(rng, _target(rng, h, w, bits), bits, int(rng.integers(0, 8)))
    data = data.copy()
    kind = seed % 4
    if kind == 0:
        for _ in range(int(rng.integers(1, 6))):
            data[int(rng.integers(16, data.size))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        data = data[:int(rng.integers(16, data.size))]
    elif kind == 2:
        data[int(rng.integers(0, 16))] ^= 1 << int(rng.integers(0, 8))
    else:
        data[16:] = rng.integers(0, 256, size=data.size - 16, dtype=np.uint8)
    host = HostImage(w, h)
    s_or = oracle.samsung_v2(bits, data, host)
    st, img = decode(gpu, data, bits, w, h)
    assert st == s_or, (st, s_or)
    if s_or == 0:
        assert np.array_equal(img.pixels(), host.pixels())


def test_validate_is_the_constructor(gpu):
    from rawspeed_amd import capi
    import ctypes as C
    L = capi.lib()
    img = HostImage(64, 8)
    d = abi.SamsungV2Desc()
    d.bit_depth, d.width, d.height, d.optflags, d.init_val = 12, 64, 8, 0, 5
    v = img.view()
    assert L.rsx_samsung_v2_validate(C.byref(d), C.byref(v)) == 0
    for field, bad in (("bit_depth", 13), ("width", 48), ("width", 6512), ("height", 4337),
                       ("height", 9), ("optflags", 8), ("width", 0)):
        e = abi.SamsungV2Desc.from_buffer_copy(d)
        setattr(e, field, bad)
        assert L.rsx_samsung_v2_validate(C.byref(e), C.byref(v)) == INVALID_ARG, (field, bad)


def _job(d, off, n, w, h, img_off):
    j = abi.SamsungV2Job()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = off, n, img_off
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = w * 2, w, h, 1, 1
    return j


def test_batch_of_frames_in_one_plan(gpu, oracle):
    """Several frames of different sizes, one of them damaged, one launch sequence."""
    rng = np.random.default_rng(192)
    frames, jobs, parts, off, img_off = [], [], [], 0, 0
    for k in range(5):
        bits = (12, 14)[k & 1]
        h, w = int(rng.integers(20, 120)), 16 * int(rng.integers(4, 40))
        data, want = V2.encode(rng, _target(rng, h, w, bits), bits, int(rng.integers(0, 8)))
        if k == 3:
            data = data[:data.size // 2]
        host = HostImage(w, h)
        s_or = oracle.samsung_v2(bits, data, host)
        st, d, payload = open_stream(data, bits, w, h)
        assert st == 0
        pad = (-payload.size) % 16
        parts.append(np.concatenate([payload, np.zeros(pad, np.uint8)]))
        jobs.append(_job(d, off, payload.size, w, h, img_off))
        frames.append((w, h, img_off, s_or, host.pixels().copy()))
        off += payload.size + pad
        img_off += w * h * 2
    plan = gpu.samsung_v2_plan(jobs)
    inp = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.zeros(img_off, dtype=torch.uint8, device="cuda")
    for run in range(2):
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        rc, st, _ = plan.results()
        got = out.cpu().numpy()
        for k, (w, h, io, s_or, px) in enumerate(frames):
            assert st[k] == s_or, (k, st, s_or)
            if s_or == 0:
                assert np.array_equal(got[io:io + w * h * 2].view(np.uint16).reshape(h, w), px), k
        assert (rc == 0) == all(f[3] == 0 for f in frames)


@pytest.mark.parametrize("w,h", [(6496, 100), (1024, 700)])
def test_frames_at_the_size_limits(gpu, oracle, w, h):
    """The widest row the constructor accepts (406 blocks: the anti-diagonals of the
    reconstruction are longest), and a tall frame (the row-start table over many hops)."""
    rng = np.random.default_rng([193, w])
    bits = 14
    data, want = V2.encode(rng, _target(rng, h, w, bits), bits, 0)
    host = HostImage(w, h)
    assert oracle.samsung_v2(bits, data, host) == 0
    st, img = decode(gpu, data, bits, w, h)
    assert st == 0
    assert np.array_equal(img.pixels(), host.pixels())
