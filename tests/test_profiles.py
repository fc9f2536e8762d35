"""Consistency of the committed bench records under profiles/ (CPU tier: reads JSON only).

A kernel's algorithmic bytes are a share of the step's: a per_kernel[].alg_bytes above
step.alg_bytes_per_launch is an accounting bug (round 4's C4 record carried an uninitialised
count for the ranges sent ahead -- nxs_gpu_search.hip launch_t.postings).  Records from round 5
on are checked; the round-4 files stay as they were measured.
"""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_records():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_*bench*.json"))):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if not m or int(m.group(1)) < 5:
            continue
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            out.append((os.path.basename(f), json.loads(lines[-1])))
    return out


def test_kernel_bytes_within_step_bytes():
    recs = bench_records()
    if not recs:
        pytest.skip("no round-5 bench records under profiles/ yet")
    for name, d in recs:
        rf = d.get("roofline") or {}
        step = (rf.get("step") or {}).get("alg_bytes_per_launch")
        if step is None:
            continue
        assert 0 < step < 1 << 40, (name, step)
        tot = 0
        for k in rf.get("per_kernel") or []:
            assert 0 <= k["alg_bytes"] <= step, (name, k["kernel"], k["alg_bytes"], step)
            tot += k["alg_bytes"]
        # the classes partition the batch's postings -- but for the ranges a retry list scans AGAIN (k_scan8 after an
        # overflow: charged to both kernels), a fraction of a per cent
        assert tot <= step * 1.01, (name, tot, step)


def test_record_shape():
    for name, d in bench_records():
        assert d["metric"] and d["unit"] and d["n_gpus"] >= 1, name
        assert d["value"] > 0 and d["ms_per_step"] > 0, name
        rf = d.get("roofline")
        if rf and rf.get("frac") is not None:
            assert rf["bound"] in ("hbm", "mfma") and 0 <= rf["frac"] < 1, (name, rf.get("frac"))
            assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3, name
        if "repeat_ms_per_step" in d:
            # the median loop is the one reported
            r = sorted(d["repeat_ms_per_step"])
            assert r[0] <= d["ms_per_step"] <= r[-1] * 1.0001, name
