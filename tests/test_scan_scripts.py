"""Progressive files with scan scripts libjpeg's default script does not produce -- DC scans of a single component of a colour picture, DC
scans of some of the components, AC bands cut anywhere -- written from chosen coefficients by tests/helpers/jpeg_from_coefficients.py
(spectral selection, and successive approximation with refinement passes over parts of a band).  The oracle's decoder (an independent restatement of jdphuff.c) must give back the coefficients; the host
entropy decoder and the host emulation of the GPU walk + replay must agree with it; on the GPU the pixels must be the oracle's."""
import numpy as np
import pytest

import oracle
from helpers import jpeg_from_coefficients as jc

S444 = [(1, 1), (1, 1), (1, 1)]
S420 = [(2, 2), (1, 1), (1, 1)]
S422 = [(2, 1), (1, 1), (1, 1)]
S411 = [(4, 1), (1, 1), (1, 1)]
GRAY = [(1, 1)]

SCRIPTS = {
    "dc_each_alone": [("dc", [0]), ("dc", [1]), ("dc", [2]), ("ac", 0, 1, 63), ("ac", 1, 1, 63), ("ac", 2, 1, 63)],
    "dc_luma_then_chroma_pair": [("dc", [0]), ("dc", [1, 2]), ("ac", 1, 1, 63), ("ac", 0, 1, 5), ("ac", 2, 1, 63), ("ac", 0, 6, 63)],
    "dc_pair_then_last": [("dc", [0, 1]), ("dc", [2]), ("ac", 2, 1, 1), ("ac", 2, 2, 63), ("ac", 0, 1, 63), ("ac", 1, 1, 30), ("ac", 1, 31, 63)],
    "dc_reversed": [("dc", [2]), ("dc", [0]), ("dc", [1]), ("ac", 0, 1, 63), ("ac", 1, 1, 63), ("ac", 2, 1, 63)],
}
# successive approximation with refinement passes over PARTS of a band, DC refined late, chains of two to five scans per component
SCRIPTS["refine_parts_of_bands"] = [
    ("dc", [0, 1, 2], 0, 2), ("ac", 0, 1, 5, 0, 2), ("ac", 0, 6, 63, 0, 2), ("ac", 1, 1, 63, 0, 1), ("ac", 2, 1, 63, 0, 1),
    ("ac", 0, 1, 5, 2, 1), ("dc", [0, 1, 2], 2, 1), ("ac", 0, 6, 63, 2, 1), ("ac", 0, 1, 20, 1, 0), ("ac", 0, 21, 63, 1, 0),
    ("ac", 1, 1, 9, 1, 0), ("ac", 1, 10, 63, 1, 0), ("dc", [1], 1, 0), ("dc", [0, 2], 1, 0), ("ac", 2, 1, 63, 1, 0)]
SCRIPTS["three_bit_planes"] = [
    ("dc", [0], 0, 0), ("dc", [1, 2], 0, 0), ("ac", 0, 1, 63, 0, 3), ("ac", 1, 1, 63, 0, 3), ("ac", 2, 1, 63, 0, 3),
    ("ac", 0, 1, 63, 3, 2), ("ac", 0, 1, 63, 2, 1), ("ac", 1, 1, 63, 3, 2), ("ac", 2, 1, 63, 3, 2), ("ac", 0, 1, 63, 1, 0),
    ("ac", 1, 1, 63, 2, 1), ("ac", 1, 1, 63, 1, 0), ("ac", 2, 1, 63, 2, 1), ("ac", 2, 1, 63, 1, 0)]
CASES = [(name, samp, w, h) for name in SCRIPTS for samp, w, h in ((S444, 83, 61), (S420, 83, 61), (S422, 50, 37), (S411, 130, 20), (S420, 16, 16))]
CASES.append(("gray_bands", GRAY, 70, 45))
SCRIPTS["gray_bands"] = [("dc", [0]), ("ac", 0, 1, 2), ("ac", 0, 3, 20), ("ac", 0, 21, 63)]


def _ids(case):
    name, samp, w, h = case
    return "%s_%s_%dx%d" % (name, "".join("%d%d" % s for s in samp[:1]), w, h)


def make(case):
    name, samp, w, h = case
    rng = np.random.default_rng(sum(map(ord, _ids(case))))
    coefs = jc.random_coefficients(rng, w, h, samp, 200, dense=10, small=4, dc=40)
    qt = [np.full(64, 3 + c, dtype=np.int32) for c in range(len(samp))]
    return jc.write_progressive(w, h, samp, coefs, qt, SCRIPTS[name]), coefs


def _real(case, c):
    _, samp, w, h = case
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    cw, ch = -(-w * samp[c][0] // hmax), -(-h * samp[c][1] // vmax)
    return -(-ch // 8), -(-cw // 8)


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_oracle_and_host_decoders_give_back_the_coefficients(case):
    from nvimagecodec_amd import lowlevel
    jpeg, coefs = make(case)
    got, _ = oracle.decode_coefficients(jpeg)
    host, _ = lowlevel.entropy_decode_host(jpeg)
    emu, _ = lowlevel.entropy_decode_gpu_algorithm_host(jpeg)  # (raises UNSUPPORTED if the GPU walk would not take the file)
    for c in range(len(coefs)):
        rows, cols = _real(case, c)
        want = np.asarray(coefs[c])[:rows, :cols].astype(np.int16)
        assert np.array_equal(np.asarray(got[c])[:rows, :cols], want), ("oracle", c)
        assert np.array_equal(np.asarray(host[c])[:rows, :cols], want), ("host decoder", c)
        assert np.array_equal(np.asarray(emu[c])[:rows, :cols], want), ("GPU algorithm on the host", c)


@pytest.mark.gpu
def test_gpu_decodes_the_scan_scripts():
    import torch
    from nvimagecodec_amd.lowlevel import BatchDecoder
    dec = BatchDecoder(0, 4)
    jpegs = [make(case)[0] for case in CASES]
    for gh in (True, False):
        outs, st = dec.decode(jpegs, fmt="rgb", gpu_huffman=gh)
        torch.cuda.synchronize()
        assert all(s == 0 for s in st)
        if gh:
            assert dec.stats()["gpu_entropy_images"] == len(jpegs)
        for case, j, o in zip(CASES, jpegs, outs):
            assert np.array_equal(o.cpu().numpy(), oracle.decode(j)), (_ids(case), gh)
    dec.close()
