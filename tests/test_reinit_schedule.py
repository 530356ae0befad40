"""The host-side schedule of domain_rand.reinit_epis_rand (pbhc_amd.envs.motion_tracking.ReinitSchedule) against the reference's own lines,
replayed verbatim in spirit (legged_robot_base.py:142 `self.reinit_epis_rand_counter = 0`, :390-395 in `_update_tasks_callback`, which runs
after `_update_counters_each_step` incremented `common_step_counter`): same firing steps, same consumption of the np.random stream."""
import numpy as np

from pbhc_amd.envs.motion_tracking import ReinitSchedule


def _reference_firing_steps(mean, steps, seed):
    np.random.seed(seed)
    counter, common, fired = 0, 0, []
    for _ in range(steps):
        common += 1                                             # _update_counters_each_step
        if mean > 0 and common >= counter:                      # legged_robot_base.py:390-391
            fired.append(common)
            counter = common + (-np.log(np.random.rand(1)) * mean)   # :394-395
    return fired, np.random.rand()


def test_schedule_fires_in_the_first_step_and_consumes_the_same_stream():
    for mean, seed in ((40.0, 0), (3.0, 7), (250.0, 11)):
        want, tail = _reference_firing_steps(mean, 2000, seed)
        np.random.seed(seed)
        s = ReinitSchedule(mean)
        got = [k for k in range(1, 2001) if s.due(k)]
        assert got == want and got[0] == 1
        assert np.random.rand() == tail                          # the global stream is where the reference's would be


def test_schedule_is_off_without_the_config_key():
    s = ReinitSchedule(-1)
    np.random.seed(0)
    before = np.random.get_state()[1].copy()
    assert not any(s.due(k) for k in range(1, 100))
    assert (np.random.get_state()[1] == before).all()            # and draws nothing
