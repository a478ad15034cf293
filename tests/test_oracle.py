"""CPU tests: the oracle's C restatement against the golden vectors made by the REAL reference
(tests/golden/make_vectors.py -> oracle/_ref/ref_driver), and against the reference itself when its
build is present in this checkout."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, MODELS, REF_DRIVER, VECTORS, oracle_ppmd, oracle_qvz, oracle_rc

import sys
sys.path.insert(0, GOLDEN)
import qvz_inputs

PPMD_VECTORS = sorted(f[:-3] for f in os.listdir(VECTORS) if f.startswith("ppmd_") and f.endswith(".in"))
RC_VECTORS = sorted(f[:-3] for f in os.listdir(VECTORS) if f.startswith("rc_") and f.endswith(".in") and f != "rc_empty.in")


@pytest.mark.parametrize("name", PPMD_VECTORS)
def test_ppmd_oracle_matches_reference_vector(oracle, name):
    data = open(os.path.join(VECTORS, name + ".in"), "rb").read()
    want = open(os.path.join(VECTORS, name + ".out"), "rb").read()
    assert oracle_ppmd(oracle, data) == want


@pytest.mark.parametrize("name", RC_VECTORS)
def test_rc_oracle_matches_reference_vector(oracle, name):
    pairs = open(os.path.join(VECTORS, name + ".in"), "rb").read()
    want = open(os.path.join(VECTORS, name + ".out"), "rb").read()
    assert oracle_rc(oracle, name[3:], pairs) == want


@pytest.mark.parametrize("name", qvz_inputs.CASES)
def test_qvz_oracle_matches_reference_vector(oracle, name):
    # vectors: the reference's ReadCodebook + choose_quantizer + WELL + QVZEncoder on these seeded reads (ref_driver qvz)
    lens, quals = qvz_inputs.reads_case(name)
    want = open(os.path.join(VECTORS, name + ".out"), "rb").read()
    assert oracle_qvz(oracle, qvz_inputs.footer_for(name), lens, quals) == want


def test_qvz_oracle_rejects_truncated_codebook(oracle):
    import ctypes
    oracle.fso_qvz_footer_size.restype = ctypes.c_long
    f = qvz_inputs.qvz_footer()
    size = oracle.fso_qvz_footer_size(f, len(f))
    assert 132 < size <= len(f)
    assert oracle.fso_qvz_footer_size(f, size - 1) == -1


def test_rc_empty_stream_is_eight_flush_bytes(oracle):
    want = open(os.path.join(VECTORS, "rc_empty.out"), "rb").read()
    assert len(want) == 8
    for m in MODELS:
        assert oracle_rc(oracle, m, b"") == want


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="reference build (oracle/_ref) not present")
def test_ppmd_oracle_matches_live_reference_incl_model_restart(oracle, tmp_path):
    # 2.2 MB of 41-symbol noise overruns the 2 MiB text area: the model restarts (Model.cpp:367)
    rng = np.random.default_rng(9)
    data = rng.integers(0, 41, 2_200_000, dtype=np.uint8).tobytes()
    (tmp_path / "a").write_bytes(data)
    subprocess.check_call([REF_DRIVER, "ppmd", str(tmp_path / "a"), str(tmp_path / "b")])
    assert oracle_ppmd(oracle, data) == (tmp_path / "b").read_bytes()


def test_rle_binary_known_answers(oracle):
    import ctypes
    def enc(bits):
        buf = ctypes.create_string_buffer(len(bits) + 16)
        n = oracle.fso_rle_binary(bytes(bits), len(bits), buf, len(buf))
        return list(buf.raw[:n])
    # RleEncoder.h:21-79: run of k matches then a mismatch -> k+2 ; lone mismatch -> 0 ; 253 matches -> 255 (no mismatch implied)
    assert enc([1, 1, 1, 0]) == [5]
    assert enc([0]) == [0]
    assert enc([0, 0]) == [0, 0]
    assert enc([1] * 253) == [255]
    assert enc([1] * 253 + [0]) == [255, 0]
    assert enc([1] * 254) == [255, 3]
    assert enc([1, 1]) == [4]
    assert enc([]) == []


def test_rle0_known_answers(oracle):
    import ctypes
    def enc(vals):
        arr = (ctypes.c_uint32 * len(vals))(*vals)
        buf = ctypes.create_string_buffer(5 * len(vals) + 16)
        n = oracle.fso_rle0(arr, len(vals), buf, len(buf))
        return list(buf.raw[:n])
    # RleEncoder.h:140-212
    assert enc([5]) == [6]
    assert enc([0]) == [1]
    assert enc([0, 0]) == [0]
    assert enc([0, 0, 0]) == [0, 1]
    assert enc([0, 7]) == [1, 8]
    assert enc([251]) == [252]
    assert enc([252]) == [0xFE, 0, 253]
    assert enc([70000]) == [0xFF, 0, 1, 0x11, 0x71]
