"""The byte / flop accounting bench.py's roofline blocks are computed from (no GPU): the SURVEY 8(d) whole-frame figure,
the per-kernel algorithmic bytes, and the stale-entry refusal of the PMC traffic file."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench


def _args(**kw):
    d = dict(height=384, width=1280, levels=6, dscv_range=4, sncv_range=3, batch=1)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_hotpath_bytes_match_survey_8d():
    """sum_l 4 h w (3C + F_in + 13) + 6 h w [l < L] = 106.72 MB per full frame at 384x1280 / 6 levels / r = 4, 3
    (level 1: 62.18 MB), 5.23 MB at config 1 (128x256 / 3 levels / r = 2, 2), 844.0 MB at config 5; final upsample 2.46 MB."""
    hp, rs = bench.hotpath_bytes_per_frame(_args(), 1)
    assert abs(hp / 1e6 - 106.72) < 0.01 and abs(rs / 1e6 - 2.46) < 0.01
    hp32, _ = bench.hotpath_bytes_per_frame(_args(), 32)
    assert hp32 == 32 * hp
    hp1, _ = bench.hotpath_bytes_per_frame(_args(height=128, width=256, levels=3, dscv_range=2, sncv_range=2), 1)
    assert abs(hp1 / 1e6 - 5.23) < 0.01
    hp5, _ = bench.hotpath_bytes_per_frame(_args(height=768, width=2560, dscv_range=6, sncv_range=6), 1)
    assert abs(hp5 / 1e6 - 844.0) < 0.1
    h, w, C, k, f_in = bench.level_geometry(_args(), 1)
    assert (h, w, C, k, f_in) == (192, 640, 16, 1, 64)
    assert abs((4 * h * w * (3 * C + f_in + 13) + 6 * h * w) / 1e6 - 62.18) < 0.7          # SURVEY's level-1 share


def test_level_kernel_bytes():
    by = bench.level_bytes(_args(), 1, 1)
    px = 192 * 640
    assert by["dscv"] == 4 * px * (2 * 16 + 2 + 9 + 1) == 21626880
    assert by["sncv"] == 4 * px * (16 + 49) == 31948800
    # fused front: raw + previous features + depth memory + 5/4 coarse floats in, normalised features + the 64-float row out
    assert by["front"] == px * (4 * (3 * 16 + 1 + 64) + 5) == 56156160
    assert bench.level_bytes(_args(), 4, 2)["front"] == 4 * 96 * 320 * (4 * (3 * 32 + 1 + 122) + 5)


def test_stale_traffic_entries_are_refused(tmp_path, monkeypatch):
    src = tmp_path / "kernel.hip"
    src.write_text("v1")
    rel = os.path.relpath(str(src), ROOT)
    doc = {"collected": "now", "batch1": {
        "front": {"bytes": 123, "kernel": "level_front_kernel<16, 1, 32, 8, 2, false>", "sources": [rel], "sources_sha": bench._sha([rel])},
        "dscv": {"bytes": 456, "kernel": "dscv_wave_kernel<4, 4, 9>", "sources": [rel], "sources_sha": "0000000000000000"}}}
    tj = tmp_path / "pmc_traffic.json"
    tj.write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "TRAFFIC_JSON", str(tj))
    ok, note = bench.load_traffic(1)
    assert set(ok) == {"front"} and ok["front"]["bytes"] == 123 and "REFUSED as stale" in note and "dscv" in note
    src.write_text("v2")                                              # the kernel source changes -> the entry goes stale
    ok, note = bench.load_traffic(1)
    assert ok == {} and "front" in note
    assert bench.load_traffic(7)[0] == {}                             # no measurement for that batch size


def test_committed_traffic_file_is_well_formed():
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, entries in doc.items():
        if not key.startswith("batch"):
            continue
        for name, ent in entries.items():
            assert ent["bytes"] > 0 and ent["kernel"] and all(os.path.isfile(os.path.join(ROOT, p)) for p in ent["sources"]), (key, name)
