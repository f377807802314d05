"""CPU re-check of launch lists the engines emitted on an MI355X (csrc/sched_trace.hip), recorded by tests/test_gpu_schedules.py under
MI355_DUMP_TRACES and committed under tests/golden/sched_traces/: the happens-before checker (tests/_sched_check.py) must find every recorded
multi-stream schedule race-free, and must find a race in most of them when any single stream wait is removed (i.e. it sees the edges).
The live traces are checked on the GPU; this keeps a schedule check in the CPU suite (a checker regression, or a trace-format change that
silently empties the region lists, fails here)."""
import glob
import os

import pytest

import _sched_check as SC

HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = sorted(glob.glob(os.path.join(HERE, "golden", "sched_traces", "*.txt")))


@pytest.mark.skipif(not TRACES, reason="no recorded traces committed yet")
@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[:-4] for p in TRACES])
def test_recorded_schedule_is_race_free(path):
    text = open(path).read()
    s = SC.parse(text)
    n_launch = sum(1 for o in s.ops if o.regions)
    assert n_launch >= 20, n_launch
    races = s.races()
    assert races == [], races[:5]
    nw = SC.n_waits(text)
    if len(s.streams()) == 1:          # (the Wan training step since round 6b: its weight-gradient GEMMs left the side stream -- nothing to order)
        assert nw == 0, nw
        return
    assert nw > 0
    needed = [k for k in range(nw) if SC.parse(text, drop_waits=[k]).races(limit=1)]
    assert len(needed) >= 0.6 * nw, (nw, len(needed))
