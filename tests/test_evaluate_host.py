"""Host logic of evaluate.run_episode without a GPU: which creatures go where when the engine flags them.

REM2D_ERR_HANDOVER (the step train could not vouch for a hand-over) is a launch-form failure: the creatures are evaluated again on
per-step launches of the SAME build and never reach the wide build or the penalty for it; the capacity bits go to the wide build;
what stays flagged after that is reported with its real bits.  (The GPU tier runs the same paths for real:
tests/test_handover_gpu.py, tests/test_env_gpu.py.)"""
import numpy as np
import pytest
import torch

from gym_rem2d_amd import _lib, evaluate


class FakeEnv:
    """What run_episode needs of a BatchedModular2D: fitness / frozen / errors / handover counter, all finished after one chunk."""

    def __init__(self, codes, failures=0):
        self.codes = torch.tensor(codes, dtype=torch.int32)
        self.n = len(codes)
        self.flags, self.options, self.wide, self.on_handover = 0, {}, False, "raise"
        self._failures = failures
        self.stepped = 0

    def step(self, n):
        self.stepped += n

    @property
    def frozen(self):
        return torch.ones(self.n, dtype=torch.int32)

    @property
    def fitness(self):
        if self.on_handover == "raise" and self._failures:
            raise _lib.HandoverError(self._failures)
        return torch.arange(self.n, dtype=torch.float64)

    def errors(self):
        return self.codes.clone()

    def handover_failures(self, clear=False):
        n = self._failures
        if clear:
            self._failures = 0
        return n


def _patch_reevaluate(monkeypatch, second):
    """`second`: {(wide, per_step): codes after that re-run}; records the calls."""
    calls = []

    def fake(env, mask, fit, wide=None, options=None, max_steps=0, chunk=0):
        key = (bool(wide), bool(options and options.get("fuse_velpost") == 1))
        calls.append((key, torch.nonzero(mask).flatten().tolist()))
        fit[mask] = 100.0 + (10.0 if key[0] else 0.0) + (1.0 if key[1] else 0.0)   # who produced this creature's fitness
        out = torch.zeros(mask.shape, dtype=torch.int32)
        out[mask] = torch.tensor(second.get(key, 0), dtype=torch.int32)
        return out
    monkeypatch.setattr(evaluate, "reevaluate", fake)
    return calls


def test_handover_goes_to_per_step_launches_capacity_to_the_wide_build(monkeypatch):
    H, S, P = _lib.ERR_HANDOVER, _lib.ERR_SOLVER_OVERFLOW, _lib.ERR_PAIR_OVERFLOW
    env = FakeEnv([0, H, S, H | P, 0], failures=2)
    calls = _patch_reevaluate(monkeypatch, {(False, True): 0, (True, False): 0})
    with pytest.warns(UserWarning, match="REM2D_ERR_HANDOVER"):
        fit = evaluate.run_episode(env, max_steps=10, chunk=10)
    # creatures 1 and 3 (hand-over) again on per-step launches of the env's own build; creature 2 (capacity, no hand-over) in the wide
    # build; creature 3's capacity bit came from a state that was not to be trusted and is gone with the re-run
    assert calls == [((False, True), [1, 3]), ((True, False), [2])]
    assert fit.tolist() == [0.0, 101.0, 110.0, 101.0, 4.0]
    assert env.last_handover == [1, 3] and env.last_overflow == [2] and env.last_unresolved == []
    assert env.on_handover == "raise" and env.handover_failures() == 0          # policy restored, counter consumed


def test_capacity_found_by_the_per_step_rerun_still_reaches_the_wide_build(monkeypatch):
    H, S = _lib.ERR_HANDOVER, _lib.ERR_SOLVER_OVERFLOW
    env = FakeEnv([H, 0], failures=1)
    calls = _patch_reevaluate(monkeypatch, {(False, True): S, (True, False): S})
    with pytest.warns(UserWarning):
        with pytest.raises(evaluate.SolverOverflow) as ei:
            evaluate.run_episode(env, max_steps=10, chunk=10)
    assert [c[0] for c in calls] == [(False, True), (True, False)]
    assert ei.value.indices == [0] and ei.value.codes == [S] and not isinstance(ei.value, _lib.HandoverError)


def test_penalty_mode_never_scores_a_handover(monkeypatch):
    H = _lib.ERR_HANDOVER
    env = FakeEnv([H, H, 0], failures=5)
    _patch_reevaluate(monkeypatch, {(False, True): 0})
    with pytest.warns(UserWarning, match="per-step launches"):
        fit = evaluate.run_episode(env, max_steps=10, chunk=10, on_error="penalty")
    assert fit.tolist() == [101.0, 101.0, 2.0] and env.last_unresolved == []
    assert evaluate.UNRESOLVED_FITNESS not in fit.tolist()[:2]


def test_handover_that_survives_per_step_launches_raises(monkeypatch):
    H = _lib.ERR_HANDOVER
    env = FakeEnv([H], failures=1)
    _patch_reevaluate(monkeypatch, {(False, True): H})
    with pytest.warns(UserWarning):
        with pytest.raises(_lib.HandoverError, match="even on per-step launches"):
            evaluate.run_episode(env, max_steps=10, chunk=10)


def test_strict_modes_name_the_handover():
    H, S = _lib.ERR_HANDOVER, _lib.ERR_SOLVER_OVERFLOW
    env = FakeEnv([0, H, S])
    env.on_handover = "flag"
    for mode in ("raise", "warn"):
        with pytest.raises(_lib.HandoverError, match="hand-over"):
            evaluate.check_errors(env, mode)
    assert evaluate.check_errors(env, "ignore").tolist() == [False, True, True]
    with pytest.raises(evaluate.SolverOverflow) as ei:
        evaluate.check_errors(FakeEnv([0, 0, S]), "raise")
    assert ei.value.indices == [2] and ei.value.codes == [S]
    assert "solver / pair slots" not in str(_lib.HandoverError(3))


def test_train_fault_encoding():
    assert _lib.train_fault(3) == 3 | (1 << 16) and _lib.train_fault(2, 4, drop=True) == 2 | (4 << 16) | (1 << 30)
    assert _lib.OPTIONS[-1] == "train_fault" and _lib.ERR_CAPACITY == 3
