"""f2 presentation (src/pipeline.rs:56-60, Bgra8UnormSrgb): the threshold table behind the exact byte encode.
CPU only: the table files are what tools/gen_srgb_table.py writes, the table is pinned to the sRGB OETF evaluated in
binary64 on both sides of every threshold, and the oracle's encode (which counts thresholds) agrees with the formula on
a dense sweep."""
import os
import sys

import numpy as np

import oracle_py as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_srgb_table as gen   # noqa: E402


def _byte64(x):
    """round(255 * srgb_oetf(x)) in binary64 (x: float32 array in (0, 1))."""
    x = x.astype(np.float64)
    y = np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
    return 255.0 * y + 0.5


def test_table_files_are_the_generators_output():
    for path, text in gen.outputs().items():
        assert open(path).read() == text, f"{path} is stale: run tools/gen_srgb_table.py"


def test_thresholds_are_where_the_formula_steps():
    t = gen.thresholds()
    k = np.arange(1, 256)
    at, below = _byte64(t), _byte64(np.nextafter(t, np.float32(-np.inf)))
    assert np.all(np.floor(at) == k) and np.all(np.floor(below) == k - 1)
    # the binary64 evaluation is nowhere near ambiguous at the decisive values (pow is good to ~1e-16)
    assert np.min(at - k) > 1e-10 and np.min(k - below) > 1e-10


def test_oracle_encode_counts_thresholds():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(1 << 20, dtype=np.float32), np.linspace(0, 1, 1 << 16, dtype=np.float32),
                        gen.thresholds(), np.nextafter(gen.thresholds(), np.float32(-np.inf)),
                        np.array([0.0, -0.0, -1.0, 1.0, 1.5, np.inf, -np.inf, np.nan, 0.0031308, 1e-45, 1e-30], np.float32)])
    rgba = np.zeros((x.size, 4), np.float32)
    rgba[:, 0] = x
    got = orc.encode_bgra8_srgb(rgba)[:, 2]                  # R lands in byte 2 of BGRA
    inside = (x > 0) & (x < 1)
    want = np.zeros(x.size, np.int64)
    want[inside] = np.floor(_byte64(x[inside])).astype(np.int64)
    want[x >= 1] = 255                                       # NaN, negatives, zeros stay 0
    assert np.array_equal(got.astype(np.int64), want)
