"""bench.py explains its own line: the per-step summary and the `diagnosis` sentence on synthetic regions -- a steady
one, round 4's signature (one step of twenty stalled for 27 ms with the kernels at their usual speed), a descheduled host, a host-bound
loop, lost stream overlap, an untypical first region."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import bench
    return bench


def test_step_summary_and_steady_region():
    b = _bench()
    dev = [1.72 + 0.01 * (i % 3) for i in range(20)]
    host = [0.5] * 20
    summ, slow, med = b.step_summary(dev, host)
    assert slow == [] and abs(med - 1.73) < 1e-9 and summ["min"] == 1.72 and summ["max"] == 1.74 and len(summ["list"]) == 20
    d = b.diagnose(1.735, 20, med, slow, 10.0, {"region_ms": 34.7, "device_span_ms": 34.6}, [1.735, 1.73, 1.74], {"ratio": 1.25, "verdict": "streams overlap"})
    assert d.startswith("steady")


def test_one_stalled_step_is_named():
    b = _bench()
    dev = [1.74] * 20
    dev[7] = 28.9                                              # 27 ms more than its neighbours: 62 ms for the region = 3.10 ms per step
    host = [0.5] * 20
    summ, slow, med = b.step_summary(dev, host)
    assert [x["step"] for x in slow] == [7] and med == 1.74
    ms = sum(dev) / 20
    d = b.diagnose(ms, 20, med, slow, 10.0, {"region_ms": ms * 20, "device_span_ms": ms * 20, "runq_wait_ms": 0.0}, [ms, 1.74, 1.74], None)
    assert d.startswith("STALL: 1 of 20 steps") and "the device itself stalled" in d and "NOT typical" in d
    host[7] = 27.5                                             # ... or the host was late with that step's launches
    summ, slow, med = b.step_summary(dev, host)
    d = b.diagnose(ms, 20, med, slow, 37.0, {"region_ms": ms * 20, "device_span_ms": ms * 20, "runq_wait_ms": 1.0}, [ms], None)
    assert "the host was late there" in d


def test_descheduled_host_host_bound_loop_and_lost_overlap():
    b = _bench()
    summ, slow, med = b.step_summary([1.75] * 20, [0.5] * 20)
    d = b.diagnose(3.84, 20, med, slow, 10.0, {"region_ms": 76.9, "device_span_ms": 35.7, "runq_wait_ms": 66.7, "preempted": 1,
                                               "cgroup_nr_throttled": 1, "cgroup_throttled_ms": 3371.7}, [3.84], None)
    assert d.startswith("HOST DESCHEDULED") and "66.7 ms" in d
    d = b.diagnose(2.25, 50, 2.0, [], 108.0, {"region_ms": 112.5, "device_span_ms": 112.4}, [2.25, 2.25], None)
    assert d.startswith("HOST-BOUND")
    d = b.diagnose(3.1, 20, 3.05, [], 10.0, {"region_ms": 62.0, "device_span_ms": 61.9}, [3.1, 3.1],
                   {"ratio": 1.02, "verdict": "NO OVERLAP: the side / aux streams' work does not run beside the main stream's on this box"})
    assert d.startswith("steady") and "NO OVERLAP" in d


def test_host_probes_do_not_need_a_gpu():
    b = _bench()
    s0 = b.host_sched()
    assert "on_cpu_ms" in s0 or s0 == {}                       # /proc is there on Linux; the function never raises
    assert isinstance(b.gpu_state(), list) and isinstance(b.sched_delta(s0, b.host_sched()), dict)
    q = b.cpu_quota()
    assert q is None or q > 0
