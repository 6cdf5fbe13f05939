"""Child processes of the GPU suite (torchrun launches, CLIs, bench.py) with a deadline that SAYS
where the child sits instead of only killing it: the child and its descendants run in their own
process group under PYTHONFAULTHANDLER; at the deadline the group gets SIGABRT (every Python in it
prints the stack of each of its threads), then SIGKILL, and what they printed is kept in
gpurun_out/hang_<label>.log.  A hang FAILS the test.  One run of the suite in round 5 lost a rank
under torch.distributed.run + RCCL without a trace (HISTORY.md section 5): `retry=True` -- the
multi-rank launches only -- or THR_TEST_RETRY_HANGS=1 repeats such a launch ONCE, with a warning; a
deadlock in the library's own threads (input window, text thread) must never pass on a second start."""
import os
import signal
import subprocess
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Result(object):
    def __init__(self, returncode, stdout, stderr):
        self.returncode, self.stdout, self.stderr = returncode, stdout, stderr


def _once(cmd, env, cwd, timeout):
    env = dict(os.environ if env is None else env, PYTHONFAULTHANDLER="1")
    p = subprocess.Popen(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return Result(p.returncode, out, err), None
    except subprocess.TimeoutExpired:
        for sig, wait in ((signal.SIGABRT, 15), (signal.SIGKILL, 15)):
            try:
                os.killpg(p.pid, sig)
            except OSError:
                pass
            try:
                out, err = p.communicate(timeout=wait)
                break
            except subprocess.TimeoutExpired:
                out, err = "", ""
        return None, "no answer after %d s: %s\n---- stdout\n%s\n---- stderr\n%s" % (
            timeout, " ".join(cmd), (out or "")[-4000:], (err or "")[-12000:])


def run(cmd, env=None, cwd=ROOT, timeout=300, label="child", retry=False):
    """-> Result(returncode, stdout, stderr).  Raises AssertionError if the child is silent (with
    `retry` or THR_TEST_RETRY_HANGS=1: silent twice)."""
    res, hang = _once(cmd, env, cwd, timeout)
    if res is not None:
        return res
    log = os.path.join(ROOT, "gpurun_out", "hang_%s_%d.log" % (label, int(time.time())))
    try:
        os.makedirs(os.path.dirname(log), exist_ok=True)
        with open(log, "w") as f:
            f.write(hang)
    except OSError:
        log = "(not written)"
    if not (retry or os.environ.get("THR_TEST_RETRY_HANGS") == "1"):
        raise AssertionError("%s: child process hung, stacks in %s\n%s" % (label, log, hang[-6000:]))
    warnings.warn("%s: child process hung, stacks in %s; started once more" % (label, log))
    res, again = _once(cmd, env, cwd, timeout)
    assert res is not None, "hung twice:\n" + hang[-6000:] + "\n==== second start\n" + again[-3000:]
    return res
