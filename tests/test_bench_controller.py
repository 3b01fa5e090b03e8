"""bench.py's N > 1 controller without GPUs: the ladder rccl -> peer -> error line with the rungs replaced by stand-ins (what a rung
really does needs devices: tests/test_gpu_multi.py runs the real ones, including an injected crash and an injected hang)."""
import argparse
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def line(ms, transport):
    return json.dumps({"metric": "gf2_matmul_n3_equiv_bitops_per_sec", "value": 2.8e17 / ms, "ms_per_step": ms, "n_gpus": 8, "host_issue_ms_per_step": 0.3,
                       "config": {"variant": "strassen", "made_by": transport}})


def run(bench, capsys, monkeypatch, outcomes, **kw):
    """outcomes: transport -> (json line or None, reason or None); returns (exit code, the JSON lines printed, the rungs tried)."""
    tried = []

    def fake_rung(transport, n_ranks, argv, watchdog_s):
        tried.append(transport)
        ln, why = outcomes[transport]
        return ln, why, [f"[check] from {transport}"]
    monkeypatch.setattr(bench, "run_rung", fake_rung)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    args = argparse.Namespace(gpus=8, transport=kw.get("transport", "auto"), variant=kw.get("variant", "auto"), layout=kw.get("layout", "distributed"),
                              backend=kw.get("backend", "nccl"), watchdog=5.0)
    rc = bench.controller(args)
    out = capsys.readouterr().out.splitlines()
    return rc, [json.loads(ln) for ln in out if ln.startswith("{")], tried, out


def test_both_transports_are_measured_and_the_faster_one_is_the_line(bench, capsys, monkeypatch):
    rc, lines, tried, out = run(bench, capsys, monkeypatch, {"rccl": (line(5.2, "rccl"), None), "peer": (line(4.9, "peer"), None)})
    assert rc == 0 and tried == ["rccl", "peer"] and len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["transport"] == "peer" and cfg["made_by"] == "peer" and cfg["transport_fallback"] == []
    assert set(cfg["transports_measured"]) == {"rccl", "peer"} and cfg["transports_measured"]["rccl"]["ms_per_step"] == 5.2
    assert "[check] from rccl" in out and "[check] from peer" in out          # the rungs' other output is passed through


def test_a_failed_first_rung_is_a_fallback_and_still_one_line(bench, capsys, monkeypatch):
    rc, lines, tried, _ = run(bench, capsys, monkeypatch, {"rccl": (None, "no result within the 5 s watchdog: ranks killed"), "peer": (line(4.9, "peer"), None)})
    assert rc == 0 and tried == ["rccl", "peer"] and len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["transport"] == "peer" and cfg["transport_fallback"] == [{"transport": "rccl", "reason": "no result within the 5 s watchdog: ranks killed"}]
    assert "transports_measured" not in cfg


def test_a_failed_second_rung_is_only_noted(bench, capsys, monkeypatch):
    rc, lines, tried, _ = run(bench, capsys, monkeypatch, {"rccl": (line(5.2, "rccl"), None), "peer": (None, "exit code 2: only 1 device(s) visible")})
    assert rc == 0 and len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["transport"] == "rccl" and cfg["transport_fallback"] == [] and cfg["transports_unavailable"][0]["transport"] == "peer"


def test_nothing_left_is_one_error_line_and_a_failure(bench, capsys, monkeypatch):
    rc, lines, tried, _ = run(bench, capsys, monkeypatch, {"rccl": (None, "exit code 9"), "peer": (None, "exit code 2")})
    assert rc == 1 and len(lines) == 1 and "error" in lines[0] and "metric" not in lines[0] and "n_gpus" not in lines[0]
    assert [f["transport"] for f in lines[0]["transport_fallback"]] == ["rccl", "peer"]


def test_forced_transports_and_the_development_backend_stop_at_the_first_success(bench, capsys, monkeypatch):
    ok = {"rccl": (line(5.2, "rccl"), None), "peer": (line(4.9, "peer"), None)}
    assert run(bench, capsys, monkeypatch, ok, transport="rccl")[2] == ["rccl"]
    assert run(bench, capsys, monkeypatch, ok, transport="peer")[2] == ["peer"]
    assert run(bench, capsys, monkeypatch, ok, backend="gloo")[2] == ["rccl"]          # gloo is a test aid: no second measurement
    assert run(bench, capsys, monkeypatch, ok, variant="blocks")[2] == ["rccl"]        # scatter / gather layouts exist over torch.distributed only


def test_a_rung_gets_the_commands_arguments_minus_the_controllers(bench, monkeypatch):
    seen = {}

    class FakeProc:
        pid = 0

        def __init__(self, cmd, **kw):
            seen["cmd"], seen["env"] = cmd, kw["env"]
            kw["stdout"].write(line(5.0, "x") + "\n")

        def wait(self, timeout=None):
            return 0
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(bench.os, "killpg", lambda *a: (_ for _ in ()).throw(ProcessLookupError()))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "outer")
    got, why, rest = bench.run_rung("rccl", 8, ["--gpus", "8", "--steps", "7", "--transport", "auto", "--watchdog=99", "--inner"], 30.0)
    assert why is None and json.loads(got)["ms_per_step"] == 5.0
    cmd = seen["cmd"]
    assert "--standalone" in cmd and "--nproc-per-node=8" in cmd and cmd[-3:] == ["--inner", "--transport", "rccl"]
    assert "--steps" in cmd and "auto" not in cmd and "--watchdog=99" not in cmd and cmd.count("--inner") == 1
    assert not any(k in seen["env"] for k in ("RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID"))   # the outer launcher's environment stays outside
