#!/usr/bin/env python3
"""Regenerates tests/golden/golden_hashes.json from the REFERENCE build.

Run in the build container (needs /root/reference -> oracle/_ref):
    make -C oracle ref && python tests/golden/make_golden.py
Every entry is {status, consumed, hash} as produced by the unmodified
reference (oracle/ref_shim.cpp); hash = rstest-style md5 of per-line md5s.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import golden_cases as G  # noqa: E402
from oracle_lib import Ref  # noqa: E402


def main():
    ref = Ref()
    out = {"unpack": {}, "f32": {}, "variant": {}, "ljpeg": {}, "cr2": {}, "nikon": {},
           "pentax": {}, "samsung_v1": {}, "sraw": {},
           "hasselblad": {}, "sony_arw1": {}}
    for i, c in enumerate(G.UNPACK_CASES):
        d, data, (w, h, cpp) = G.build_unpack(c)
        img = ref.image(w, h, cpp)
        st = ref.unpack(d, data, img)
        out["unpack"][str(i)] = {"status": st, "hash": G.image_hash(img.pixels())}
    for i, c in enumerate(G.F32_CASES):
        d, data, (w, h, cpp) = G.build_f32(c)
        img = ref.image(w, h, cpp, f32=True)
        st = ref.unpack(d, data, img)
        out["f32"][str(i)] = {"status": st, "hash": G.image_hash(img.u32()[:, :w * cpp])}
    for i, c in enumerate(G.VARIANT_CASES):
        d, data, (w, h, cpp) = G.build_variant(c)
        img = ref.image(w, h, cpp)
        st = ref.unpack_variant(d, data, img)
        out["variant"][str(i)] = {"status": st, "hash": G.image_hash(img.pixels())}
    for c in G.LJPEG_CASES:
        d, data, (w, h, cpp), _ = G.build_ljpeg(c)
        img = ref.image(w, h, cpp)
        st, consumed = ref.ljpeg(d, data, img)
        out["ljpeg"][c["name"]] = {"status": st, "consumed": consumed,
                                   "hash": G.image_hash(img.pixels())}
    for c in G.CR2_CASES:
        d, data, (w, h, cpp), _ = G.build_cr2(c)
        img = ref.image(w, h, cpp, is_cfa="sraw" not in c)
        st, consumed = ref.cr2(d, data, img)
        out["cr2"][c["name"]] = {"status": st, "consumed": consumed,
                                 "hash": G.image_hash(img.pixels())}
    for c in G.NIKON_CASES:
        meta, d, data, (w, h, cpp), _ = G.build_nikon(c)
        img = ref.image(w, h, cpp)
        st = ref.nikon(meta, c["bits"], data, img, bool(c["unc"]))
        out["nikon"][c["name"]] = {"status": st, "hash": G.image_hash(img.pixels())}
    for c in G.PENTAX_CASES:
        meta, d, data, (w, h, cpp), _ = G.build_pentax(c)
        img = ref.image(w, h, cpp)
        st = ref.pentax(meta, data, img)
        out["pentax"][c["name"]] = {"status": st, "hash": G.image_hash(img.pixels())}
    for c in G.SAMSUNG_V1_CASES:
        d, data, (w, h, cpp), _ = G.build_samsung_v1(c)
        img = ref.image(w, h, cpp)
        st = ref.samsung_v1(12, data, img)
        out["samsung_v1"][c["name"]] = {"status": st, "hash": G.image_hash(img.pixels())}
    for c in G.SRAW_CASES:
        d, px, (iw, ih), (ow, oh) = G.build_sraw(c)
        src, dst = ref.image(iw, ih, 1, False), ref.image(ow, oh, 3, False)
        src.set_pixels(px)
        st = ref.sraw(d, src, dst)
        out["sraw"][c["name"]] = {"status": st, "hash": G.image_hash(dst.pixels())}
    for c in G.HASSELBLAD_CASES:
        d, data, (w, h, cpp), _ = G.build_hasselblad(c)
        img = ref.image(w, h, cpp)
        st, consumed = ref.hasselblad(d, data, img)
        out["hasselblad"][c["name"]] = {"status": st, "consumed": consumed,
                                        "hash": G.image_hash(img.pixels())}
    for c in G.SONY_ARW1_CASES:
        data, (w, h, cpp), _ = G.build_sony_arw1(c)
        img = ref.image(w, h, cpp)
        st = ref.sony_arw1(data, img)
        out["sony_arw1"][c["name"]] = {"status": st, "hash": G.image_hash(img.pixels())}
    path = os.path.join(HERE, "golden_hashes.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
