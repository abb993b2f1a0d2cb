"""Which independent decoders ran beside the oracle's own.  The parity tests always decode with the oracle
(oracle/codec_oracle.c) and, where the base image offers them, ALSO with liblz4, c-blosc and libhdf5 — base-image
libraries, neither reference code nor part of this repository.  Those legs are `if present:` branches inside the tests; this
module makes their presence a visible test outcome in both suites (a skip with its reason instead of a silent pass), and
tests/conftest.py prints presence and call counts in the header and the summary of every run."""
import pytest

from tests import extlibs

LEGS = ["liblz4", "c-blosc", "libhdf5"]


def _check(leg):
    ok, what = extlibs.legs()[leg]
    if not ok:
        pytest.skip(f"{leg} is not loadable on this box: the {what} legs of the parity tests did NOT run here "
                    f"(the oracle's decoder did)")


@pytest.mark.parametrize("leg", LEGS)
def test_independent_decoder_present(leg):
    _check(leg)


@pytest.mark.gpu
@pytest.mark.parametrize("leg", LEGS)
def test_independent_decoder_present_on_the_gpu_box(leg):
    _check(leg)


@pytest.mark.gpu
def test_independent_decoders_agree_with_the_device_codec(ctx):
    """one small chunk per format through every decoder that is present, in one place: GPU chunk -> oracle, c-blosc;
    every LZ4 stream of it -> liblz4"""
    import numpy as np
    from oracle import oracle
    from tests.gpu_util import split_chunks, to_dev
    from haplohyped_varawareml_amd import device as dev
    rng = np.random.default_rng(3)
    data = (rng.random(4 * 8192) < 0.06).astype(np.uint8)
    dst, off, total = ctx.compress(to_dev(data), 4 * 8192, typesize=2, blocksize=8192, fmt=dev.BLOSC1)
    ck = split_chunks(dst, off, total)[0]
    assert np.array_equal(oracle.blosc_decompress(ck), data)
    ran = ["oracle"]
    if extlibs.have_blosc():
        assert np.array_equal(extlibs.blosc1_decompress(ck, data.size), data)
        ran.append("c-blosc")
    if extlibs.have_lz4():
        bstarts = ck[16:16 + 16].view("<i4")
        for b in range(4):
            q = int(bstarts[b])
            for plane in range(2):
                cs = int(ck[q:q + 4].view("<i4")[0])
                stream = ck[q + 4:q + 4 + cs]
                want = data[b * 8192:(b + 1) * 8192][plane::2]
                got = stream if cs == 4096 else extlibs.lz4_decompress(stream, 4096)
                assert np.array_equal(got, want)
                q += 4 + cs
        ran.append("liblz4")
    print("decoders that checked the device codec here:", ", ".join(ran))
