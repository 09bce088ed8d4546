"""Process-environment settings of the data-parallel step (one process per GPU, RCCL over xGMI) -- ONE place, shared by
``train.py`` and ``bench.py``, so that what the benchmark measures is what training runs with.

Both knobs are read by the HIP runtime / RCCL when they start, so ``data_parallel_env()`` has to run BEFORE the first
``torch.cuda`` call of the process (``torch.cuda.is_available()`` already initialises the runtime).  Both are
``setdefault``: an operator's own value wins.

  GPU_MAX_HW_QUEUES = 3    with a process group the step uses four streams (main, weight-gradient, all-reduce, RCCL's
                           own).  Measured on one rank (bench.py DS2_BENCH_FORCE_DIST=1): 308 k frames/s with the
                           runtime's default four hardware queues, 350 k -- the rate without a process group -- with two
                           or three.
  NCCL_MAX_NCHANNELS = 32  RCCL's kernels take one workgroup per channel; capped at 32 they fit on the CUs every backward
                           recurrence launch of a data-parallel step leaves free, so a collective can delay a recurrence
                           launch (the bounded spins cover that) but never keep one from becoming resident.  The launches
                           (B = 9 .. 12, H = 800; csrc/gru_persist.hip, ds2_gru_bidir_bwd_persistent_ex): the top layer's
                           backward recurrence runs 204 workgroups (``top_layer_spare_cus()`` = channels + 8 = 40 CUs asked
                           for, 52 left) when the gradient all-reduce is overlapped with backward -- 240, nothing spare,
                           only without a process group --, the layers below 174 (82 CUs left, shared with the
                           low-priority weight-gradient GEMMs of the layer above).  The forward recurrence (240 CUs) never
                           meets a collective: the step waits for its all-reduce before the optimizer, i.e. before the
                           next forward pass.

Neither value has been measured with more than one rank (no multi-GPU box was in reach of rounds 1-5): they are the
settings DESIGN.md section 5's predictions assume, and ``assert_no_fallbacks`` is the check that fails loudly when the
co-residency assumption behind them does not hold on a real node.

CPU placement (``pin_to_gpu_numa_node``): one process per GPU, each with a poller thread and four loader workers; on a
two-socket node half of the GPUs hang off each socket, and a rank whose threads run on the other socket pays the
inter-socket link on every pinned-memory access of the step (the statistics slot the device writes, the offsets it reads)
and on every wave file it decodes.  ``data_parallel_env()`` therefore restricts the process -- before any GPU call, so that
the runtime's own threads and every DataLoader worker forked later inherit it -- to the CPUs of its GPU's NUMA node,
split evenly among the ranks of that node.  The GPU is found without touching the runtime: the LOCAL_RANK-th entry of
HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (if set, integers) among the KFD topology's GPU nodes
(/sys/class/kfd/kfd/topology/nodes/*/properties: simd_count > 0, in node order = the runtime's enumeration order), its PCI
address from ``domain`` / ``location_id``, the NUMA node from /sys/bus/pci/devices/<address>/numa_node.  Anything missing
(no KFD topology, numa_node -1, one NUMA node, an affinity mask already narrower than the node, DS2_CPU_AFFINITY=0)
leaves the affinity alone; the plan that was applied is logged by train.py / reported by bench.py.
"""
import os

DEFAULTS = {'GPU_MAX_HW_QUEUES': '3', 'NCCL_MAX_NCHANNELS': '32'}


def data_parallel_env(environ=None, pin=True):
    """Apply the defaults above to ``environ`` (default: this process), pin the process to its GPU's NUMA node (only when
    acting on this process, LOCAL_RANK is set and ``pin``), and return the values now in force (+ ``cpu_affinity``: the plan)."""
    env = os.environ if environ is None else environ
    for k, v in DEFAULTS.items():
        env.setdefault(k, v)
    out = {k: env[k] for k in DEFAULTS}
    if pin and environ is None and 'LOCAL_RANK' in env:
        out['cpu_affinity'] = pin_to_gpu_numa_node(int(env['LOCAL_RANK']), int(env.get('LOCAL_WORLD_SIZE', env.get('WORLD_SIZE', '1'))))
    return out


def top_layer_spare_cus(environ=None):
    """CUs the TOP layer's backward recurrence launch leaves free when the gradient all-reduce runs beside it: one per RCCL
    channel (a channel's kernel is one workgroup) plus a margin of 8."""
    env = os.environ if environ is None else environ
    try:
        return int(env.get('NCCL_MAX_NCHANNELS', DEFAULTS['NCCL_MAX_NCHANNELS'])) + 8
    except ValueError:
        return int(DEFAULTS['NCCL_MAX_NCHANNELS']) + 8


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs='/', environ=None):
    """NUMA node of every GPU the runtime will enumerate, in its order (see the module docstring); None where unknown."""
    env = os.environ if environ is None else environ
    base = os.path.join(sysfs, 'sys/class/kfd/kfd/topology/nodes')
    try:
        ids = sorted(int(d) for d in os.listdir(base) if d.isdigit())
    except OSError:
        return []
    gpus = []
    for i in ids:
        text = _read(os.path.join(base, str(i), 'properties'))
        if text is None:
            continue
        props = dict(line.split(None, 1) for line in text.splitlines() if len(line.split(None, 1)) == 2)
        if int(props.get('simd_count', '0')) <= 0:
            continue                                          # a CPU node
        loc, dom = int(props.get('location_id', '0')), int(props.get('domain', '0'))
        bdf = '%04x:%02x:%02x.%x' % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
        node = _read(os.path.join(sysfs, 'sys/bus/pci/devices', bdf, 'numa_node'))
        gpus.append(int(node) if node is not None and node.strip().lstrip('-').isdigit() else None)
    # ROCR's filter first (the runtime below HIP), then ONE HIP-level filter: on ROCm HIP_VISIBLE_DEVICES and
    # CUDA_VISIBLE_DEVICES are two names of the same list, not a composition -- HIP's wins when both are set
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES' if env.get('HIP_VISIBLE_DEVICES') else 'CUDA_VISIBLE_DEVICES'):
        v = env.get(var)
        if v:
            try:
                gpus = [gpus[int(x)] for x in v.split(',') if x.strip() != '']
            except (ValueError, IndexError):
                return []                                     # UUIDs or out of range: not ours to guess
    return gpus


def numa_cpu_plan(local_rank, local_world, sysfs='/', environ=None, current=None):
    """The CPUs rank ``local_rank`` (of ``local_world`` on this node) should run on: its GPU's NUMA node's CPUs (within the
    current affinity mask ``current``), split evenly among the ranks whose GPUs share that node.  None = leave it alone."""
    nodes = gpu_numa_nodes(sysfs, environ)
    if local_rank >= len(nodes) or nodes[local_rank] is None or nodes[local_rank] < 0:
        return None
    mine = nodes[local_rank]
    text = _read(os.path.join(sysfs, 'sys/devices/system/node/node%d/cpulist' % mine))
    if text is None:
        return None
    cpus = _parse_cpulist(text)
    if current is not None:
        if not cpus & set(current) or set(current) <= cpus and len(set(current)) < len(cpus):
            return None                                       # an operator's (or a container's) narrower mask wins
        cpus &= set(current)
    peers = [r for r in range(min(local_world, len(nodes))) if nodes[r] == mine]
    if local_rank not in peers or not cpus:
        return None
    # split by PHYSICAL core: the SMT siblings of a core (thread_siblings_list) go to the same rank -- an even split of the
    # sorted logical ids can hand two ranks the two hardware threads of the same cores
    cores = {}
    for c in sorted(cpus):
        sib = _read(os.path.join(sysfs, 'sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c))
        key = min(_parse_cpulist(sib)) if sib and sib.strip() else c
        cores.setdefault(key, []).append(c)
    order = [cores[k] for k in sorted(cores)]
    share = len(order) // len(peers)
    if share < 2:
        return set(cpus)                                      # fewer than two cores per rank: share the node
    k = peers.index(local_rank)
    return {c for core in order[k * share:(k + 1) * share] for c in core}


def pin_to_gpu_numa_node(local_rank, local_world):
    """Apply ``numa_cpu_plan`` to this process (call BEFORE the first GPU call and before any worker is forked).  Returns a
    description of what was done, for the log."""
    if os.environ.get('DS2_CPU_AFFINITY') == '0' or not hasattr(os, 'sched_setaffinity'):
        return 'unchanged (disabled)'
    try:
        current = os.sched_getaffinity(0)
        plan = numa_cpu_plan(local_rank, local_world, current=current)
        if not plan:
            return 'unchanged (no NUMA placement found for local rank %d)' % local_rank
        os.sched_setaffinity(0, plan)
        return 'local rank %d -> %d CPUs %d..%d' % (local_rank, len(plan), min(plan), max(plan))
    except OSError as e:
        return 'unchanged (%s)' % e


def assert_no_fallbacks(count, where='this run'):
    """A persistent recurrence launch that could not run (grid not co-resident, hand-off time-out) switches the device to
    the launch-per-step kernels, 3x slower.  ``train.py`` logs that and goes on -- a training run should survive; a
    benchmark or a scaling test must FAIL instead of reporting the fall-back's rate."""
    if int(count) > 0:
        raise RuntimeError('%d persistent recurrence launch(es) fell back to the launch-per-step kernels in %s: the '
                           'co-residency settings (%s) do not hold on this node -- see codes/utils/dist_utils.py'
                           % (int(count), where, ', '.join('%s=%s' % (k, os.environ.get(k)) for k in DEFAULTS)))
