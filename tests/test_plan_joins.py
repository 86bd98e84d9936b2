"""CPU: Workspace._assert_joined, the plan-build check behind the forward plan's single main-stream join (the waits for `out_mask`
and `opt_gen_done` and the gradient zeroing are covered by `img_fwd` only as long as they stay on streams that `img_fwd`'s stream has joined)."""
import pytest


def _plan(*entries):
    # (fn, args, name, keepalive, stream id): fn None = a sync entry of the executor
    return [((None if n in ("EV_RECORD", "EV_WAIT", "BG_FLUSH2") else object()), a, n, None, sid) for n, a, sid in entries]


def _check(P, need, ev="img_fwd"):
    from variational_mmt_amd.engine.workspace import Workspace
    Workspace._assert_joined(P, need, ev)


def test_join_through_a_second_stream_is_seen():
    P = _plan(("EV_RECORD", "fwd_begin", 0), ("EV_WAIT", "fwd_begin", 1), ("EV_RECORD", "side_fwd", 1), ("EV_WAIT", "side_fwd", 2),
              ("vmmt_zero_multi", None, 2), ("vmmt_dropout_mask", None, 2), ("BG_FLUSH2", None, 2), ("EV_RECORD", "aux_fwd", 2),
              ("gemm", None, 1), ("EV_WAIT", "aux_fwd", 1), ("EV_RECORD", "img_fwd", 1), ("EV_WAIT", "img_fwd", 0))
    _check(P, ["vmmt_zero_multi", "vmmt_dropout_mask", "BG_FLUSH2"])


def test_a_mask_moved_to_an_unjoined_stream_fails_at_plan_build():
    P = _plan(("EV_RECORD", "fwd_begin", 0), ("EV_WAIT", "fwd_begin", 1), ("vmmt_zero_multi", None, 1), ("vmmt_dropout_mask", None, 2),
              ("EV_RECORD", "img_fwd", 1), ("EV_WAIT", "img_fwd", 0))
    with pytest.raises(AssertionError, match="vmmt_dropout_mask"):
        _check(P, ["vmmt_zero_multi", "vmmt_dropout_mask"])


def test_record_before_the_work_or_no_wait_on_main_fails():
    early = _plan(("EV_RECORD", "img_fwd", 1), ("vmmt_zero_multi", None, 1), ("EV_WAIT", "img_fwd", 0))
    with pytest.raises(AssertionError):
        _check(early, ["vmmt_zero_multi"])
    nowait = _plan(("vmmt_zero_multi", None, 1), ("EV_RECORD", "img_fwd", 1))
    with pytest.raises(AssertionError):
        _check(nowait, ["vmmt_zero_multi"])
