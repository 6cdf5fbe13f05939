"""tests/bounded_run.py itself (CPU): a child that answers, a child that never does."""
import os
import sys
import warnings

import pytest

import bounded_run


def test_a_child_that_answers():
    res = bounded_run.run([sys.executable, "-c", "import sys; print('out'); print('err', file=sys.stderr); sys.exit(3)"],
                          timeout=60, label="selftest_ok")
    assert (res.returncode, res.stdout.strip(), res.stderr.strip()) == (3, "out", "err")


def test_a_silent_child_is_asked_where_it_sits_and_started_once_more(tmp_path):
    flag = tmp_path / "second"
    # first start: sleeps for ever inside a named function; second start (the flag exists): answers
    code = ("import os, sys, time\n"
            "def stuck_here():\n"
            "    time.sleep(1000)\n"
            "if os.path.exists(%r):\n"
            "    print('second start'); sys.exit(0)\n"
            "open(%r, 'w').close()\n"
            "stuck_here()\n") % (str(flag), str(flag))
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        res = bounded_run.run([sys.executable, "-c", code], timeout=3, label="selftest_hang", retry=True)
    assert res.returncode == 0 and res.stdout.strip() == "second start"
    assert len(seen) == 1 and "hung" in str(seen[0].message)
    log = str(seen[0].message).split("stacks in ")[1].split(";")[0]
    text = open(log).read()
    assert "stuck_here" in text and "no answer after 3 s" in text       # faulthandler's dump of the sleeping thread
    os.remove(log)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        with pytest.raises(AssertionError, match="hung twice"):
            bounded_run.run([sys.executable, "-c", "import time\ntime.sleep(1000)"], timeout=2, label="selftest_twice",
                            retry=True)
    os.remove(str(seen[0].message).split("stacks in ")[1].split(";")[0])
    # the default: the first hang fails the test, with the stacks
    with pytest.raises(AssertionError, match="child process hung") as exc:
        bounded_run.run([sys.executable, "-c", "import time\ndef here():\n    time.sleep(1000)\nhere()"], timeout=2,
                        label="selftest_once")
    assert "here" in str(exc.value)
    os.remove(str(exc.value).split("stacks in ")[1].split("\n")[0])
