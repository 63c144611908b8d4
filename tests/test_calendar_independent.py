"""TimeDimension's calendar (olap-in-memory_amd/js/calendar.js) against an INDEPENDENT computation with Python's datetime.

The reference delegates its calendar to the third-party `timeslot-dag@2.2.0`, which is absent from /root/reference, so
the restatement is formally "parity unpinned" beyond the literals of the reference's tests.  This
test removes implementation error from that: every day of 2009-12-20 .. 2020-01-10 (ten years, leap years, every
year boundary) rolled up to month / quarter / semester / year, to weeks starting on Monday (= ISO 8601, checked with
date.isocalendar()), Saturday and Sunday (week 1 = the week holding January 4th, restated here from scratch) and to
month-weeks (weeks cut at month ends), compared slot for slot."""
import datetime
import json
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
NODE = shutil.which("node")
START = {"sat": 5, "sun": 6, "mon": 0}  # date.weekday() numbering of the day a week starts on


def week_start(d, first):
    return d - datetime.timedelta(days=(d.weekday() - START[first]) % 7)


def week_slot(d, first):
    """week 1 of a year is the week (starting on `first`) that holds January 4th"""
    ws = week_start(d, first)
    year = ws.year + (1 if week_start(datetime.date(ws.year + 1, 1, 4), first) <= ws else 0)
    w1 = week_start(datetime.date(year, 1, 4), first)
    if ws < w1:
        year -= 1
        w1 = week_start(datetime.date(year, 1, 4), first)
    return "%04d-W%02d-%s" % (year, (ws - w1).days // 7 + 1, first)


def month_week_slot(d, first):
    """weeks cut at month ends: week 1 of a month starts on its first day, a new one starts on every `first`"""
    first_of_month = d.replace(day=1)
    n = 1 + sum(1 for k in range(1, d.day) if (first_of_month + datetime.timedelta(days=k)).weekday() == START[first])
    return "%04d-%02d-W%d-%s" % (d.year, d.month, n, first)


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_calendar_against_python_datetime():
    r = subprocess.run([NODE, os.path.join(HERE, "js", "dump_calendar.js"), "2009-12-20", "2020-01-10"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    d, end = datetime.date(2009, 12, 20), datetime.date(2020, 1, 10)
    n = 0
    while d <= end:
        row = got[d.isoformat()]
        assert row["month"] == "%04d-%02d" % (d.year, d.month)
        assert row["quarter"] == "%04d-Q%d" % (d.year, (d.month - 1) // 3 + 1)
        assert row["semester"] == "%04d-S%d" % (d.year, (d.month - 1) // 6 + 1)
        assert row["year"] == "%04d" % d.year
        iso = d.isocalendar()
        assert row["week_mon"] == "%04d-W%02d-mon" % (iso[0], iso[1]), (d, row["week_mon"])  # ISO 8601
        for first in ("sat", "sun", "mon"):
            assert row["week_" + first] == week_slot(d, first), (d, first, row["week_" + first], week_slot(d, first))
            assert row["month_week_" + first] == month_week_slot(d, first), (d, first, row["month_week_" + first], month_week_slot(d, first))
        d += datetime.timedelta(days=1)
        n += 1
    assert n == 3674
