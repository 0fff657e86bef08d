"""Host logic of the lazy binning (gftorf_amd/api.py): when the depth cut is used, widened, paused.  CPU only."""
from gftorf_amd import api


def test_clean_frames_keep_the_cut():
    s = api.new_slab_state()
    for _ in range(300):
        api.slab_report(s, 0, 4800)
        assert api.slab_next_cut(s, 1.7) == 1.7
    assert s["per_tile"] == api._SLAB_DEFAULT and s["backoff"] == 0


def test_a_few_flagged_quadrants_widen_the_slab_and_clean_frames_narrow_it():
    s = api.new_slab_state()
    api.slab_report(s, 3, 4800)                       # <= 1 % of the quadrants
    assert s["per_tile"] == int(896 * 1.25) + 1 and s["off"] == 0
    assert api.slab_next_cut(s, 1.7) == 1.7
    api.slab_report(s, 48, 4800)
    wide = s["per_tile"]
    assert wide > int(896 * 1.25) + 1
    for _ in range(200):
        api.slab_report(s, 0, 4800)
    assert api._SLAB_DEFAULT <= s["per_tile"] < wide


def test_many_flagged_quadrants_pause_the_cut_with_growing_pauses():
    s = api.new_slab_state()
    pauses = []
    for _ in range(5):
        api.slab_report(s, 500, 4800)                 # > 1 %
        assert s["per_tile"] == api._SLAB_DEFAULT
        n = 0
        while api.slab_next_cut(s, 1.7) == 0.0:
            n += 1
            assert n < 1000
        pauses.append(n)
    assert pauses == [4, 12, 28, 60, 64]
    # clean frames with the cut shorten the next pause again
    for _ in range(64):
        api.slab_report(s, 0, 4800)
    api.slab_report(s, 500, 4800)
    assert s["off"] < 64


def test_small_frames_tolerate_four_quadrants():
    s = api.new_slab_state()
    api.slab_report(s, 4, 64)                         # 64 quadrants: 1 % would be none
    assert s["off"] == 0 and s["per_tile"] > api._SLAB_DEFAULT
    api.slab_report(s, 5, 64)
    assert s["off"] == 4


def test_a_widened_slab_without_a_cut_starts_over():
    s = api.new_slab_state()
    s["per_tile"] = 3000                              # the device finds no cut that leaves out half of the frame
    for i in range(49):
        assert api.slab_next_cut(s, 0.0) == 0.0
        assert s["per_tile"] == 3000
    api.slab_next_cut(s, 0.0)
    assert s["per_tile"] == api._SLAB_DEFAULT
