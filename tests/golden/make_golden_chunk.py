#!/usr/bin/env python3
"""Known-answer chunk bytes for the fixture genotype matrix (tests/golden/fixture_G.npy):
each sample row (1000 variants x 2 = 2000 bytes) is one block, typesize 2.  The Blosc1-format
chunk is additionally verified with the image's real c-blosc 1.21 decoder before being recorded."""
import hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle
from tests import extlibs

G = np.load(os.path.join(HERE, "fixture_G.npy"))
raw = G.reshape(-1).view(np.uint8)
out = {}
for key, fmt in (("blosc1", oracle.BLOSC1), ("blosc2", oracle.BLOSC2)):
    chunk = oracle.blosc_compress(raw, 2, 2000, fmt)
    assert np.array_equal(oracle.blosc_decompress(chunk), raw)
    if fmt == oracle.BLOSC1 and extlibs.have_blosc():
        assert np.array_equal(extlibs.blosc1_decompress(chunk, raw.size), raw)
        out["verified_with_cblosc_1_21"] = True
    out[key] = dict(sha256=hashlib.sha256(chunk.tobytes()).hexdigest(), cbytes=int(chunk.size), nbytes=int(raw.size))
json.dump(out, open(os.path.join(HERE, "fixture_chunk_golden.json"), "w"), indent=1)
print(out)
