"""bench.py's guard for the legs after the timed region of a multi-process run: a stalled leg must not cost the headline line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time
sys.path.insert(0, {root!r})
import bench
res = {{"metric": "m", "value": 1.0}}
t = bench.extras_watchdog(res, {rank}, 0.3)
if {cancel}:
    t.cancel()
    print("finished normally", flush=True)
    sys.exit(0)
time.sleep(30)            # stands in for a collective that never returns
print("not reached", flush=True)
"""


def run(rank, cancel):
    return subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, rank=rank, cancel=cancel)], capture_output=True, text=True, timeout=240)


def test_watchdog_prints_headline_and_exits_on_rank0():
    r = run(0, False)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "not reached" not in r.stdout
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and "watchdog" in d["notes"][0] and d["incomplete"] is True


def test_watchdog_other_ranks_exit_quietly_and_cancel_works():
    r = run(1, False)
    assert r.returncode == 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = run(0, True)
    assert r.returncode == 0 and "finished normally" in r.stdout and "{" not in r.stdout
