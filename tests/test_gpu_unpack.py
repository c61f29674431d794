"""GPU parity: the HIP unpack path (through the C-ABI) vs the oracle, the
golden hashes of the reference, and size-independent properties at the
BASELINE sizes."""
import json
import os

import numpy as np
import pytest
import torch

from rawspeed_amd import abi, synth

import golden_cases as G
from oracle_lib import HostImage, out_pitch

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "golden_hashes.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


@pytest.mark.parametrize("i", range(len(G.UNPACK_CASES)))
def test_unpack_golden_host_api(gpu, oracle, i):
    """rsx_unpack_u16 (host pointers, what the patched reference method calls)."""
    d, data, (w, h, cpp) = G.build_unpack(G.UNPACK_CASES[i])
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    v = img.view()
    st = gpu.unpack_u16(d, data, v)
    assert st == oracle.unpack(d, data, want) == GOLD["unpack"][str(i)]["status"]
    assert np.array_equal(img.u16(), want.u16())      # incl. untouched padding/rows
    assert G.image_hash(img.pixels()) == GOLD["unpack"][str(i)]["hash"]


def test_unpack_sweep_vs_oracle(gpu, oracle):
    """All orders x bps 1..16 x odd widths / paddings / row offsets, one plan
    per order mix (device-resident API)."""
    import gpu_util
    rng = np.random.default_rng(21)
    jobs, wants, metas = [], [], []
    in_chunks, in_off, out_off = [], 0, 0
    for order in range(4):
        for bps in range(1, 17):
            for w in (8, 24, 56, 1000, 8200):
                if (w * bps) % 8:
                    continue
                pad = int(rng.integers(0, 5))
                oy = int(rng.integers(0, 3))
                h = int(rng.integers(1, 5))
                pitch = w * bps // 8 + pad
                data = rng.integers(0, 256, size=h * pitch, dtype=np.uint8)
                d = abi.UnpackDesc(0, oy, w, h, pitch, bps, order)
                dim_y = h + oy - int(rng.integers(0, 2))  # sometimes clamps the last row
                dim_y = max(dim_y, oy)
                if dim_y == 0:
                    dim_y = 1
                want = HostImage(w, dim_y, 1)
                st = oracle.unpack(d, data, want)
                if st != 0:
                    # < 4 bytes of stream: the bit streamer refuses (BitStreamer.h:58-59)
                    assert st == abi.RSX_ERR_IO and h * pitch < 4
                    got = HostImage(w, dim_y, 1)
                    assert gpu.unpack_u16(d, data, got.view()) == st
                    continue
                j = abi.UnpackJob()
                j.desc = d
                j.in_offset, j.in_bytes = in_off, data.size
                j.img_offset = out_off
                j.img = gpu_util.image_job_view(w, dim_y, 1, want.pitch)
                jobs.append(j)
                wants.append(want)
                metas.append((order, bps, w, pad, oy, h, dim_y))
                in_chunks.append(data)
                # deliberately unaligned strip starts for a third of the jobs
                in_off += data.size + int(rng.integers(0, 3)) * 5
                in_chunks.append(np.zeros(in_off - sum(c.size for c in in_chunks), np.uint8))
                out_off += want.buf.size
    inp = gpu_util.to_dev(np.concatenate(in_chunks + [np.zeros(64, np.uint8)]))
    out = torch.full((out_off,), 0xA5, dtype=torch.uint8, device="cuda")
    plan = gpu.unpack_plan(jobs)
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, _ = plan.results()
    assert rc == 0 and not any(st)
    got = out.cpu().numpy()
    for j, want, m in zip(jobs, wants, metas):
        g = got[j.img_offset:j.img_offset + want.buf.size]
        assert np.array_equal(g, want.buf), m


def test_unpack_16bit_lsb_honours_crop_x(gpu, oracle):
    # copyPixels fast path writes at offset.x (UncompressedDecompressor.cpp:257-264)
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, size=6 * 40, dtype=np.uint8)
    d = abi.UnpackDesc(3, 1, 20, 6, 40, 16, abi.ORDER_LSB)
    img, want = HostImage(24, 8), HostImage(24, 8)
    assert gpu.unpack_u16(d, data, img.view()) == oracle.unpack(d, data, want) == 0
    assert np.array_equal(img.u16(), want.u16())
    # ... while the packed paths ignore it (UncompressedDecompressor.cpp:196)
    data = rng.integers(0, 256, size=6 * 30, dtype=np.uint8)
    d = abi.UnpackDesc(3, 1, 20, 6, 30, 12, abi.ORDER_MSB)
    img, want = HostImage(24, 8), HostImage(24, 8)
    assert gpu.unpack_u16(d, data, img.view()) == oracle.unpack(d, data, want) == 0
    assert np.array_equal(img.u16(), want.u16())


def test_unpack_errors_match_oracle(gpu, oracle):
    data = np.zeros(64, np.uint8)
    for d in [abi.UnpackDesc(0, 0, 8, 40, 12, 12, 0), abi.UnpackDesc(0, 0, 0, 4, 12, 12, 0),
              abi.UnpackDesc(0, 0, 8, 4, 11, 12, 0), abi.UnpackDesc(0, 0, 8, 4, 12, 12, 4),
              abi.UnpackDesc(0, 0, 8, 4, 12, 17, 0), abi.UnpackDesc(0, 0, 7, 4, 12, 12, 0),
              abi.UnpackDesc(0, 5, 8, 4, 12, 12, 0), abi.UnpackDesc(1, 0, 8, 4, 12, 12, 0),
              abi.UnpackDesc(0, 0, 8, 1, 2, 2, 1)]:
        img, want = HostImage(8, 4), HostImage(8, 4)
        st = gpu.unpack_u16(d, data, img.view())
        assert st == oracle.unpack(d, data, want) != 0
        assert np.array_equal(img.buf, want.buf)  # nothing written


def test_dng_uncompressed_tiles(gpu, oracle):
    """AbstractDngDecompressor::decompressThread<1>: 2x2 tiles, 12-bit MSB (non
    8/16/32-bit integer => big-endian, AbstractDngDecompressor.cpp:64-77)."""
    rng = np.random.default_rng(4)
    W, H, tw, th, bps = 40, 13, 24, 8, 12
    img, want = HostImage(W, H), HostImage(W, H)
    descs, datas = [], []
    for ty in range(2):
        for tx in range(2):
            w = min(tw, W - tx * tw)
            h = min(th, H - ty * th)
            pitch = tw * bps // 8
            data = rng.integers(0, 256, size=th * pitch, dtype=np.uint8)
            d = abi.UnpackDesc(tx * tw, ty * th, w, h, pitch, bps, abi.ORDER_MSB)
            descs.append(d)
            datas.append(data)
            assert oracle.unpack(d, data, want) == 0
    rc, st = gpu.dng_decompress_uncompressed(descs, datas, img.view())
    assert rc == 0 and not any(st)
    assert np.array_equal(img.u16(), want.u16())


@pytest.mark.parametrize("cfg", ["cfg1_12bit_lsb_4096x3072", "cfg2_14bit_msb_8192x5464"])
def test_unpack_baseline_sizes_roundtrip(gpu, cfg):
    """BASELINE configs 1 and 2 at full size: pack(v) -> GPU unpack == v
    (size-independent round trip), plus a sampled oracle comparison."""
    import gpu_util
    from oracle_lib import Oracle
    if cfg.startswith("cfg1"):
        w, h, bps, order = 4096, 3072, 12, abi.ORDER_LSB
    else:
        w, h, bps, order = 8192, 5464, 14, abi.ORDER_MSB
    px = synth.uniform(w * h, bps, 42).reshape(h, w)
    packed = synth.pack_rows(px, bps, order)
    pitch = w * bps // 8
    opitch = out_pitch(w, 1)
    j = abi.UnpackJob()
    j.desc = abi.UnpackDesc(0, 0, w, h, pitch, bps, order)
    j.in_offset, j.in_bytes, j.img_offset = 0, packed.size, 0
    j.img = gpu_util.image_job_view(w, h, 1, opitch)
    inp = gpu_util.to_dev(packed)
    out = torch.zeros(opitch * h, dtype=torch.uint8, device="cuda")
    plan = gpu.unpack_plan([j])
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, _ = plan.results()
    assert rc == 0
    got = out.cpu().numpy().view(np.uint16).reshape(h, opitch // 2)[:, :w]
    assert np.array_equal(got, px)
    # oracle on a 64-row strip from the middle
    r0 = h // 2
    d = abi.UnpackDesc(0, 0, w, 64, pitch, bps, order)
    want = HostImage(w, 64)
    assert Oracle().unpack(d, packed[r0 * pitch:(r0 + 64) * pitch], want) == 0
    assert np.array_equal(want.pixels(), got[r0:r0 + 64])
