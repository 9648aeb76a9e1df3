"""Which code runs on a stream that was itself forked from the caller's stream (a seed shard of ``PipelinedLBFGS``).

A rollout that forks a side stream of its own from there (``TrajOptRollout`` with torque limits: the joint-space chain next to the
task-space chain) makes a TWO-level fork; inside one hipGraph capture that crashes ``hipStreamEndCapture`` on this ROCm
(tools/r04/c4_shards.py, docs/NOTEBOOK.md round 4).  The shards already run side by side, so such a rollout keeps its chains
on the shard's stream instead.  Thread-local: captures run on the thread that launches."""

import contextlib
import threading

_state = threading.local()


def inside_forked_stream() -> bool:
    return getattr(_state, "depth", 0) > 0


@contextlib.contextmanager
def forked_stream():
    _state.depth = getattr(_state, "depth", 0) + 1
    try:
        yield
    finally:
        _state.depth -= 1
