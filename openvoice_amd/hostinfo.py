"""How many host threads this process may actually use (affinity mask and cgroup CPU quota, not the
machine's core count): oversubscribing OpenMP inside a CPU-limited container makes the CPU legs of
tests/bench orders of magnitude slower."""
import os


def usable_cpus(cap=None):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:   # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                quota = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                period = int(fh.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    if cap:
        n = min(n, cap)
    return max(1, n)
