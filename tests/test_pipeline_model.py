"""dashing_amd.multigpu.pipeline_model: the N-rank step predicted from per-rank timelines, following the rounds of
dsh_exchange_collect_async (CPU only: arithmetic on made-up timelines)."""
import pytest

from dashing_amd.multigpu import pipeline_model

MB = 1e6


def rank(r, wall, parts, rowsorted=True, **kw):
    return dict(rank=r, wall_ms=wall, rowsorted=rowsorted, part_info=parts, **kw)


def test_compute_bound_when_the_links_are_fast():
    rows = [rank(0, 3.0, [(3.0, 10 * MB)], rowsorted=False), rank(1, 2.0, [(1.0, 5 * MB), (2.0, 5 * MB)])]
    ms, worst = pipeline_model(rows, 1e12, 1e6, round_ms=0.0, place_launch_ms=0.0, nmsg=2, dst_gate=False)
    assert ms == pytest.approx(3.0) and worst == 0


def test_a_round_waits_for_its_slowest_message_and_lasts_as_long_as_its_largest():
    # two sources, two rounds of 5 MB + 5 MB and 20 MB + 20 MB shares: at 10 GB/s a round of 10 MB shares takes 1 ms
    a = rank(1, 2.0, [(1.0, 10 * MB), (2.0, 10 * MB)])
    b = rank(2, 2.5, [(1.5, 5 * MB), (2.5, 5 * MB)])
    rows = [rank(0, 0.5, [(0.5, MB)], rowsorted=False), a, b]
    ms, worst = pipeline_model(rows, 0, 10.0, round_ms=0.0, place_launch_ms=0.0, nmsg=2, dst_gate=False)
    # round 0 starts at 1.5 (b's first half), lasts 1 ms (a's 10 MB); round 1 starts at max(2.5, 2.5) and lasts 1 ms
    assert ms == pytest.approx(3.5) and worst in (1, 2)
    # per-round overhead and the placement of the last round add to the end
    ms2, _ = pipeline_model(rows, 15 * MB * 1e3, 10.0, round_ms=0.1, place_launch_ms=0.05, nmsg=2, dst_gate=False)
    assert ms2 == pytest.approx(1.5 + 1.1 + 1.1 + 1.0 + 0.05)


def test_a_message_waits_for_every_part_it_overlaps():
    # four parts of 5 MB, the third one final LAST: the first message (10 MB) can go at 1.0, the second only at 3.0
    src = rank(1, 3.0, [(0.5, 5 * MB), (1.0, 5 * MB), (3.0, 5 * MB), (2.0, 5 * MB)])
    rows = [rank(0, 0.1, [(0.1, MB)], rowsorted=False), src]
    ms, _ = pipeline_model(rows, 0, 10.0, round_ms=0.0, place_launch_ms=0.0, nmsg=2, dst_gate=False)
    assert ms == pytest.approx(3.0 + 1.0)


def test_more_bandwidth_never_hurts_and_the_gate_delays_the_first_round():
    rows = [rank(0, 2.0, [(2.0, 10 * MB)], rowsorted=False, finalize_ms=0.5, bands=1, rounds_of_512=7),
            rank(1, 1.9, [(0.8, 8 * MB), (1.2, 8 * MB), (1.9, 8 * MB)])]
    steps = [pipeline_model(rows, 1e12, g, nmsg=3, dst_gate=False)[0] for g in (10.0, 20.0, 40.0, 80.0)]
    assert all(x >= y for x, y in zip(steps, steps[1:]))
    free, _ = pipeline_model(rows, 1e12, 10.0, nmsg=3, dst_gate=False)
    gated, _ = pipeline_model(rows, 1e12, 10.0, nmsg=3)  # as the library decides: one launch of 7 rounds -> behind the tile kernel (1.5 ms)
    assert gated >= free and gated == pytest.approx(1.5 + 3 * (0.8 + 0.02) + 8 * MB / 1e12 * 1e3 + 0.005)


def test_measured_interference_lengthens_the_destinations_own_kernels():
    """dst_interference (round 6: what waiting receive kernels cost the destination's kernels, measured on one GPU): k_finalize
    always, the tile kernel only when the receives are not gated behind it; nothing changes for a single rank"""
    from dashing_amd.multigpu import MEASURED_RECV_INTERFERENCE

    dst = rank(0, 2.0, [(2.0, 10 * MB)], rowsorted=False, finalize_ms=0.5, pair_ms=1.2, bands=1, rounds_of_512=7)
    src = rank(1, 1.0, [(0.5, 4 * MB), (1.0, 4 * MB)])
    f = {"pair": 1.10, "finalize": 1.20}
    base, _ = pipeline_model([dst, src], 1e12, 1e6, nmsg=2, dst_gate=False)
    free, _ = pipeline_model([dst, src], 1e12, 1e6, nmsg=2, dst_gate=False, dst_interference=f)
    gated, _ = pipeline_model([dst, src], 1e12, 1e6, nmsg=2, dst_gate=True, dst_interference=f)
    assert base == pytest.approx(2.0)
    assert free == pytest.approx(2.0 + 0.5 * 0.20 + 1.2 * 0.10)
    assert gated == pytest.approx(max(2.0 + 0.5 * 0.20, 1.5 + 2 * 0.02 + 2 * 4 * MB / 1e15 * 1e3 + 0.005), abs=1e-3)
    alone, _ = pipeline_model([dst], 1e12, 1e6, nmsg=2, dst_interference=f)
    assert alone == pytest.approx(2.0)
    assert set(MEASURED_RECV_INTERFERENCE) == {"lds_over_32k", "lds_4k"}
