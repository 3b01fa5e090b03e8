"""bench.py's N > 1 controller without GPUs: link probe -> the same product on one GPU -> the ladder peer -> rccl -> error line ->
the reference's multi-core baseline, with every child process replaced by a stand-in (what the children really do needs devices:
tests/test_gpu_multi.py runs the real ones, including an injected crash and an injected hang)."""
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


def line(ms, transport, pipelined_ms=None, **extra):
    d = {"metric": "gf2_matmul_n3_equiv_bitops_per_sec", "value": 2.8e17 / ms, "ms_per_step": ms, "n_gpus": 8, "steps": 5, "host_issue_ms_per_step": 0.3,
         "config": {"variant": "strassen", "made_by": transport, "m": 65536, "l": 65536, "n": 65536}, **extra}
    if pipelined_ms:
        d.update({"pipelined_ms_per_step": pipelined_ms, "pipelined_value": 2.8e17 / pipelined_ms})
    return json.dumps(d)


LINKS = {"links": {"pairs": 56, "gbs_per_direction_min": 48.0, "gbs_per_direction_median": 50.0, "all_at_once_gbs": 2500.0, "peer_access": [[1] * 8] * 8}}


def run(bench, capsys, monkeypatch, outcomes, **kw):
    """outcomes: transport -> (json line or None, reason or None); returns (exit code, the JSON lines printed, the rungs tried, stdout)."""
    tried, children = [], []

    def fake_rung(transport, n_ranks, argv, watchdog_s, extra=()):
        tried.append((transport, list(extra)))
        ln, why = outcomes[transport]
        return ln, why, [f"[check] from {transport}"]

    def fake_child(cmd, watchdog_s, key='"metric"'):
        children.append(cmd)
        if "--links-probe" in cmd:
            return (json.dumps(LINKS), None, []) if kw.get("links_ok", True) else (None, "exit code 1: no devices", [])
        return (line(27.0, "n1"), None, []) if kw.get("n1_ok", True) else (None, "exit code 1", [])
    monkeypatch.setattr(bench, "run_rung", fake_rung)
    monkeypatch.setattr(bench, "run_child", fake_child)
    monkeypatch.setattr(bench, "cpu_baseline_or_note", lambda n: {"value": 1.8e13, "unit": "bit-op/s", "cores": 256, "kind": "reference", "sample": f"stand-in {n}"})
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5"])
    args = argparse.Namespace(gpus=8, size=65536, transport=kw.get("transport", "auto"), variant="auto", backend=kw.get("backend", "nccl"), watchdog=5.0,
                              no_links=kw.get("no_links", False), no_n1=kw.get("no_n1", False), no_cpu_baseline=kw.get("no_cpu_baseline", False), virtual_ranks=False)
    rc = bench.controller(args)
    out = capsys.readouterr().out.splitlines()
    return rc, [json.loads(ln) for ln in out if ln.startswith("{")], tried, out, children


def test_the_first_rung_that_completes_is_the_line_and_the_other_is_measured_beside_it(bench, capsys, monkeypatch):
    """peer first on the real backend; the line is NEVER the better of two: a faster second rung only shows in transports_measured."""
    rc, lines, tried, out, children = run(bench, capsys, monkeypatch, {"peer": (line(5.2, "peer", 4.4), None), "rccl": (line(4.9, "rccl", 4.0), None)})
    assert rc == 0 and [t for t, _ in tried] == ["peer", "rccl"] and len(lines) == 1
    assert tried[1][1] == ["--compare-ms", "5.2000"]          # the second rung is told what it is compared with (it stops after 3 steps when > 1.3 x)
    o, cfg = lines[0], lines[0]["config"]
    assert cfg["transport"] == "peer" and cfg["made_by"] == "peer" and cfg["transport_fallback"] == [] and o["ms_per_step"] == 5.2
    assert set(cfg["transports_measured"]) == {"rccl", "peer"} and cfg["transports_measured"]["rccl"]["ms_per_step"] == 4.9
    assert "[check] from rccl" in out and "[check] from peer" in out          # the rungs' other output is passed through
    # every N > 1 line: the CPU baseline, the links, one product AND the stream, the speed-up over the same binary on one GPU, the wall time
    assert o["cpu_baseline"]["value"] == 1.8e13 and o["cpu_baseline"]["cores"] == 256
    assert cfg["links"]["gbs_per_direction_min"] == 48.0 and cfg["controller_wall_s"] >= 0
    assert o["pipelined_value"] > o["value"] > 0
    assert o["speedup_vs_n1"]["n1_ms_per_step"] == 27.0 and abs(o["speedup_vs_n1"]["one_product"] - 27.0 / 5.2) < 1e-9
    assert abs(o["speedup_vs_n1"]["pipelined"] - 27.0 / 4.4) < 1e-9
    assert sum("--links-probe" in c for c in children) == 1 and sum("--links-probe" not in c for c in children) == 1
    n1_cmd = [c for c in children if "--links-probe" not in c][0]
    assert n1_cmd[n1_cmd.index("--gpus") + 1] == "1" and n1_cmd.count("--gpus") == 1 and "--no-cpu-baseline" in n1_cmd and "--steps" in n1_cmd


def test_a_failed_first_rung_is_a_fallback_and_still_one_line(bench, capsys, monkeypatch):
    rc, lines, tried, _, _ = run(bench, capsys, monkeypatch, {"peer": (None, "no result within the 5 s watchdog: ranks killed"), "rccl": (line(4.9, "rccl"), None)})
    assert rc == 0 and [t for t, _ in tried] == ["peer", "rccl"] and tried[1][1] == [] and len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["transport"] == "rccl" and cfg["transport_fallback"] == [{"transport": "peer", "reason": "no result within the 5 s watchdog: ranks killed"}]
    assert "transports_measured" not in cfg and lines[0]["cpu_baseline"]["kind"] == "reference"


def test_a_failed_second_rung_is_only_noted(bench, capsys, monkeypatch):
    rc, lines, tried, _, _ = run(bench, capsys, monkeypatch, {"peer": (line(5.2, "peer"), None), "rccl": (None, "exit code 2: RCCL refused")})
    assert rc == 0 and len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["transport"] == "peer" and cfg["transport_fallback"] == [] and cfg["transports_unavailable"][0]["transport"] == "rccl"


def test_nothing_left_is_one_error_line_and_a_failure(bench, capsys, monkeypatch):
    rc, lines, tried, _, _ = run(bench, capsys, monkeypatch, {"rccl": (None, "exit code 9"), "peer": (None, "exit code 2")})
    assert rc == 1 and len(lines) == 1 and "error" in lines[0] and "metric" not in lines[0] and "n_gpus" not in lines[0]
    assert [f["transport"] for f in lines[0]["transport_fallback"]] == ["peer", "rccl"] and lines[0]["links"]["pairs"] == 56


def test_forced_transports_and_the_development_backend_stop_at_the_first_success(bench, capsys, monkeypatch):
    ok = {"rccl": (line(5.2, "rccl"), None), "peer": (line(4.9, "peer"), None)}
    assert [t for t, _ in run(bench, capsys, monkeypatch, ok, transport="rccl")[2]] == ["rccl"]
    assert [t for t, _ in run(bench, capsys, monkeypatch, ok, transport="peer")[2]] == ["peer"]
    assert [t for t, _ in run(bench, capsys, monkeypatch, ok, backend="gloo")[2]] == ["rccl"]   # gloo = the torch.distributed rung as a test aid: first, and alone


def test_a_failed_probe_or_one_gpu_run_is_noted_not_fatal(bench, capsys, monkeypatch):
    rc, lines, _, _, _ = run(bench, capsys, monkeypatch, {"peer": (line(5.2, "peer"), None), "rccl": (line(5.5, "rccl"), None)}, links_ok=False, n1_ok=False)
    assert rc == 0 and "error" in lines[0]["config"]["links"] and "error" in lines[0]["speedup_vs_n1"] and lines[0]["value"] > 0
    rc, lines, _, _, children = run(bench, capsys, monkeypatch, {"peer": (line(5.2, "peer"), None), "rccl": (line(5.5, "rccl"), None)},
                                    no_links=True, no_n1=True, no_cpu_baseline=True)
    assert rc == 0 and children == [] and "links" not in lines[0]["config"] and "speedup_vs_n1" not in lines[0] and "cpu_baseline" not in lines[0]


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
    got, why, rest = bench.run_rung("rccl", 8, ["--gpus", "8", "--steps", "7", "--transport", "auto", "--watchdog=99", "--inner", "--no-cpu-baseline", "--no-links"],
                                    30.0, ["--compare-ms", "5.0"])
    assert why is None and json.loads(got)["ms_per_step"] == 5.0
    cmd = seen["cmd"]
    assert "--standalone" in cmd and "--nproc-per-node=8" in cmd and cmd[-3:] == ["--inner", "--transport", "rccl"]
    assert "--steps" in cmd and "auto" not in cmd and "--watchdog=99" not in cmd and cmd.count("--inner") == 1 and "--compare-ms" in cmd
    assert "--no-cpu-baseline" not in cmd and "--no-links" not in cmd      # the controller's own business: the ranks never run the CPU baseline
    assert not any(k in seen["env"] for k in ("RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID"))   # the outer launcher's environment stays outside
    assert bench.own_args(["--gpus", "8", "--variant", "slabs", "--check", "--size", "8192", "--backend=gloo"],
                          {"--gpus": 1, "--variant": 1, "--check": 0, "--backend": 1}) == ["--size", "8192"]
