"""Measured parity numbers of the -m gpu tests: every test that asserts a tolerance also records what it measured, so the margin
of each bound is known (profiles/r3_parity_measured.json is a copy of one run's file).  Appends one JSON line per measurement to
$HH_PARITY_LOG (default gpurun_out/parity_measured.jsonl under the repo root); never fails a test."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(test, name, measured, bound):
    print("[parity] %s %s: measured %.3e (bound %.3e)" % (test, name, measured, bound))
    path = os.environ.get("HH_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, "name": name, "measured": float(measured), "bound": float(bound)}) + "\n")
    except OSError:
        pass


def check(test, name, measured, bound):
    """record + assert measured <= bound."""
    record(test, name, measured, bound)
    assert measured <= bound, (test, name, measured, bound)
