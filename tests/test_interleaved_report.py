"""The report of tests/test_other_shapes.py::test_two_handles_interleaved_equal_the_same_handles_run_alone only runs when that test FAILS, on a device, behind the whole suite --
so its code is exercised here on the CPU with stand-ins for the runs: a handle whose second update reports other loss rows when it has neighbours, a raw buffer that differs one stage
earlier, a slab that does not add up, a section that raises.  What is checked is that every finding comes out as a line and that a failing section does not stop the rest."""
import os

import numpy as np
import pytest

import tests.test_other_shapes as t


def _fake_run(members, iterations=3, debug=False, snap=False):
    together = len(members) > 1
    out = {i: [] for i in members}
    dbg = {i: {} for i in members}
    for it in range(iterations):
        for i in members:
            out[i] += [np.zeros(3, np.float32) for _ in range(6)]
            if debug:
                dbg[i]["iteration %d, after the collect" % it] = {k: np.zeros(8, np.uint32) for k in t._IL_STATE}
        for i in members:
            w = np.arange(1, 6, dtype=np.float32)
            rows = np.arange(40, dtype=np.float32).reshape(8, 5) + 100 * it
            if together and i == 1 and it == 1:
                rows[1:] = np.arange(40, dtype=np.float32).reshape(8, 5)[1:]           # the rows the previous update left behind
            out[i] += [rows, w, w.copy(), w.copy(), np.zeros(2, np.float32)]
            if snap:
                sd = {k: np.zeros(8, np.uint32) for k in t._IL_WORK + t._IL_STATE}
                if together and i == 1 and it == 1:
                    sd["slabs"][2] = 5                                                    # the first train step's slabs differ: the weight-gradient kernel
                dbg[i]["iteration %d, behind the first train step of the update" % it] = sd
            if not debug:
                continue
            d = {k: np.zeros(8, np.uint32) for k in t._IL_STATE + t._IL_WORK}
            pad = np.zeros(16, np.float32); pad[:5] = w
            if together and i == 1:
                pad[9] = 3.0                                                             # a padding word that is not zero
            for k in ("theta", "adam_m", "adam_v"):
                d[k] = pad.view(np.uint32).copy()
            if together and i == 1 and it == 0:
                d["thetaT"][3] = 7                                                       # the mirror differs one update before the loss rows do
            sl = np.random.RandomState(it).normal(size=(8, 16)).astype(np.float32); sl[4:] = 0; sl[:, 12:] = 0
            g = np.zeros(16 + 256, np.float32); g[:16] = ((sl[0] + sl[1]) + sl[2]) + sl[3]; g[12:16] = [9, 8, 7, 6]
            if together and it == 2:
                g[0] += 1                                                                # a tile that is not the sum of its slabs
            d["slabs"] = sl.reshape(-1).view(np.uint32).copy() if i == 1 else np.zeros(0, np.uint32)
            d["grad"] = g.view(np.uint32).copy()
            if i == 2 and not together:
                del d["par"]                                                             # makes this handle's buffer section raise
            dbg[i]["iteration %d, after the update" % it] = d
    return out, dbg


def test_report_names_every_finding_and_survives_a_failing_section(monkeypatch, tmp_path):
    monkeypatch.setattr(t, "_il_run", _fake_run)
    monkeypatch.setattr(os.path, "abspath", lambda p: str(tmp_path / "tests" / "x.py") if p.endswith("test_other_shapes.py") else p)
    fn = t.test_two_handles_interleaved_equal_the_same_handles_run_alone
    with pytest.raises(AssertionError) as e:
        fn(monkeypatch)
    text = str(e.value)
    assert "asserted run, handle 1 ((256, 256), 64, 16, 4): public outputs differ first at iteration 1: loss rows" in text
    assert "asserted run, handle 0 ((64, 64), 1, 512, 8): public outputs equal" in text
    assert "iteration 1, together: loss rows [1, 2, 3, 4, 5, 6, 7] differ from the other run's; rows equal to the previous update's at the same place: [1, 2, 3, 4, 5, 6, 7]" in text
    for title in ("the same three again, against alone", "order 1, 0, 2", "handles 1 and 2 only", "eager launches (PPO_HIP_NO_GRAPH=1)", "alone again, against alone"):
        assert title in text, title
    assert "iteration 0, after the update: thetaT differs in 1 of 8 words, first at 3" in text
    assert "theta holds 6 non-zero words, its dense part 5" in text
    assert "iteration 2, after the update, together: gradient == sum of the slabs in place on 11 of 12 covered words; the other 1 (words 0 .. 0) hold OTHER values" in text
    assert "iteration 2, after the update, alone: gradient == sum of the slabs in place on 12 of 12 covered words" in text
    assert "(this part of the report failed: KeyError: 'par')" in text
    assert "iteration 1, behind the first train step of the update: slabs differs in 1 of 8 words, first at 2" in text
    written = (tmp_path / "gpurun_out" / "interleaved_report.txt").read_text()
    assert written.strip() == text.split("\n", 1)[1].strip()                            # the file holds the same lines as the assertion's message


def test_equal_runs_pass_without_a_report(monkeypatch, tmp_path):
    monkeypatch.setattr(t, "_il_run", lambda members, iterations=3, debug=False, snap=False: _fake_run((members[0],), iterations, debug) if len(members) == 1 else
                        ({i: _fake_run((i,), iterations, debug)[0][i] for i in members}, {}))
    monkeypatch.setattr(os.path, "abspath", lambda p: str(tmp_path / "tests" / "x.py") if p.endswith("test_other_shapes.py") else p)
    t.test_two_handles_interleaved_equal_the_same_handles_run_alone(monkeypatch)
    assert not (tmp_path / "gpurun_out").exists()
