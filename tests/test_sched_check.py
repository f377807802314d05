"""The happens-before checker itself (tests/_sched_check.py), on synthetic traces in the format csrc/sched_trace.hip emits.  The real launch
lists come from the engines on the GPU (tests/test_gpu_schedules.py); round 2's hand-transcribed launch lists are gone."""
import _sched_check as SC


def _trace(join=True, fork=True, overlap_rows=False):
    """two streams fill the image rows (stream a) and the text rows (stream b) of one [4 blocks][8 rows][16 B] buffer, then stream a reads it all"""
    q = 0x1000
    img = f"W:{q:x}:96:128:4"                                       # rows 0-5 of every block
    txt = f"W:{q + (80 if overlap_rows else 96):x}:32:128:4"        # rows 6-7 (or 5-6: overlapping row 5)
    lines = ["L a init W:9000:64:0:1"]
    if fork:
        lines += ["E a ev_fork", "T b ev_fork"]
    lines += [f"L b text_proj R:9000:64:0:1 {txt}", f"L a img_proj R:9000:64:0:1 {img}"]
    if join:
        lines += ["E b ev_join", "T a ev_join"]
    lines += [f"L a attention R:{q:x}:512:0:1 W:a000:64:0:1"]
    return "\n".join(lines)


def test_strided_regions_of_one_buffer_do_not_conflict_and_a_join_orders_the_reader():
    assert SC.parse(_trace()).races() == []
    assert SC.parse(_trace()).streams() == ["s0", "s1"]


def test_missing_edges_are_reported():
    r = SC.parse(_trace(join=False)).races()
    assert r and {"text_proj@s1", "attention@s0"} == set(r[0][:2])
    r = SC.parse(_trace(fork=False)).races()                           # the side stream reads what `init` wrote without waiting for it
    assert r and "init@s0" in r[0][:2]
    assert SC.n_waits(_trace()) == 2
    assert SC.parse(_trace(), drop_waits=[1]).races() and SC.parse(_trace(), drop_waits=[0]).races()


def test_overlapping_strided_blocks_are_a_conflict():
    r = SC.parse(_trace(overlap_rows=True)).races()
    assert r and {"text_proj@s1", "img_proj@s0"} == set(r[0][:2])


def test_event_reuse_takes_the_latest_record():
    t = "\n".join(["L a k1 W:100:16:0:1", "E a ev", "L a k2 W:200:16:0:1", "E a ev", "T b ev", "L b k3 R:200:16:0:1 R:100:16:0:1"])
    assert SC.parse(t).races() == []
    t2 = "\n".join(["L a k1 W:100:16:0:1", "E a ev", "L a k2 W:200:16:0:1", "T b ev", "L b k3 R:200:16:0:1"])
    assert SC.parse(t2).races()                                         # recorded BEFORE k2: the wait does not cover it
