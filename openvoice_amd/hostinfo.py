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


def cpu_model():
    """Model name of the host CPU (``/proc/cpuinfo``), for the CPU-baseline line of bench.py."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_digest():
    """sha256 (16 hex digits) over the kernel sources, the C ABI header and the launch sequence -- what decides the
    HBM traffic of a conversion.  Measurement records (profiles/pmc_traffic_latest.json) carry it, and bench.py only
    reports a PMC traffic figure whose digest equals the running tree's."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) +
                   glob.glob(os.path.join(here, "..", "include", "*.h")) + [os.path.join(here, "engine.py")])
    # the opt-in paths' kernels (bf16 generator: conv1d_bf16*; split-precision MRF: conv1d_split3*) launch nothing on
    # the fp32 path the record is about
    files = [f for f in files if not os.path.basename(f).startswith(("conv1d_bf16", "conv1d_split3"))]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def launch_config_digest(batch, frames, fuse_pairs, pair_policy, chain_streams=1):
    """``kernel_source_digest()`` extended by what else decides the bytes a launch moves and how many launches a step
    has: the workload shape (utterances per GPU, frames) and the engine's fusion policy.  A PMC traffic record
    (tools/pmc_traffic.py) carries it; bench.py reports the record's figure only when it equals the running
    configuration's -- a change of ``PAIR_POLICY`` / ``fuse_pairs`` keeps the source digest and changes the traffic."""
    import hashlib
    policy = sorted(f"{c}x{k}" for c, k in pair_policy) if fuse_pairs else []
    text = f"{kernel_source_digest()}|B={int(batch)}|T={int(frames)}|fuse={int(bool(fuse_pairs))}|{','.join(policy)}|cs={int(chain_streams)}"
    return hashlib.sha256(text.encode()).hexdigest()[:16]


def bf16_source_digest():
    """sha256 (16 hex digits) over what decides the opt-in bf16 generator's launches: its kernels and bf16.py.  The
    counter record profiles/bf16_counters_latest.json carries it (tools/bf16_counters.py)."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "conv1d_bf16*")) + [os.path.join(here, "bf16.py")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def split3_source_digest():
    """sha256 (16 hex digits) over the split-precision conv kernels and their host helpers.  The counter record
    profiles/split3_traffic_latest.json carries it (tools/pmc_traffic_split.py)."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "conv1d_split3*")) + [os.path.join(here, "split3.py")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
