"""ppo._LazyLog: model.logger as a mapping whose latest values are resolved when it is READ (the update leaves them as asynchronous
copies in pinned host words; nothing waits for them unless somebody looks).  The reference's callback reads such scalars with
.get() (simulation/train_hover.py:20-23); ** / dict() / iteration must see resolved values too, and what cannot (json.dumps walks
a dict's storage in C) must fail loudly rather than see stale ones."""
import json

import pytest

pytest.importorskip("torch")


def _log():
    from fpv_drone_rl_agent_amd.ppo import _LazyLog
    return _LazyLog()


def test_reads_resolve_writes_do_not():
    log, calls = _log(), []
    log["rollout/ep_len_mean"] = 32.0                       # a write: nothing pending, nothing resolved
    log.defer(lambda: calls.append(1) or {"train/loss": 0.5})
    log.record("rollout/ep_rew_mean", -1.0)                 # how PPO.learn records episode statistics: still no resolve
    assert calls == []
    assert log.get("train/loss") == 0.5 and calls == [1]    # the reference's access pattern (train_hover.py:20-23)
    assert log["train/loss"] == 0.5 and calls == [1]        # resolved once
    assert set(log) == {"rollout/ep_len_mean", "rollout/ep_rew_mean", "train/loss"}


@pytest.mark.parametrize("reader", [lambda d: dict(d.items()), lambda d: {**d}, lambda d: json.loads(json.dumps(dict(d))), lambda d: d.copy(),
                                    lambda d: dict(zip(d.keys(), d.values())), lambda d: {k: d[k] for k in d},
                                    lambda d: {"train/loss": 1.5} if "train/loss" in d and len(d) == 1 and "1.5" in repr(d) else None])
def test_every_way_of_reading_sees_the_resolved_values(reader):
    log = _log()
    log.defer(lambda: {"train/loss": 1.5})
    assert reader(log) == {"train/loss": 1.5}


def test_a_second_update_resolves_the_first_ones_numbers_before_its_own_are_registered():
    log, order = _log(), []
    log.defer(lambda: order.append("a") or {"train/loss": 1.0, "train/std": 0.1})
    log.defer(lambda: order.append("b") or {"train/loss": 2.0})     # (each resolver reads its own pinned words: a first)
    assert order == ["a"]
    assert log["train/loss"] == 2.0 and log["train/std"] == 0.1 and order == ["a", "b"]


def test_json_refuses_the_mapping_itself():
    log = _log()
    log.defer(lambda: {"train/loss": 1.5})
    with pytest.raises(TypeError):
        json.dumps(log)
